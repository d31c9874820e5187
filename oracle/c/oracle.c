/* TEST INFRASTRUCTURE ONLY -- C restatement of the prover path (the parity oracle at scale and
 * the "cpu_baseline" timed by bench.py).  Not product code: only tests/, smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so.
 *
 * PARITY UNPINNED against the reference: OwshenNetwork/owshen @ 2024_10_08 contains no prover
 * (SURVEY.md 0.1).  Fr restates `Fp`
 * (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11); everything else
 * follows the same public standards as oracle/py (which this file is cross-checked against in
 * tests/test_oracle_c.py): EIP-196/197 BN254, circomlib MiMC7, Groth16 with arkworks naming.
 * All byte interfaces: 32-byte little-endian canonical field elements. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- Fq ---- */
#define FNAME(x) fq_##x
#define FIELD_N {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}
#define FIELD_R2 {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}
#define FIELD_ONE {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}
#define FIELD_INV 0x87d20782e4866389ULL
#include "mont_tmpl.h"
#undef FNAME
#undef FIELD_N
#undef FIELD_R2
#undef FIELD_ONE
#undef FIELD_INV

/* ---- Fr (the reference's Fp) ---- */
#define FNAME(x) fr_##x
#define FIELD_N {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}
#define FIELD_R2 {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}
#define FIELD_ONE {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}
#define FIELD_INV 0xc2e1f593efffffffULL
#include "mont_tmpl.h"
#undef FNAME
#undef FIELD_N
#undef FIELD_R2
#undef FIELD_ONE
#undef FIELD_INV

/* ---- Fq2 = Fq[u]/(u^2+1) ---- */
typedef struct { fq_t c0, c1; } fq2_t;
static inline int fq2_is_zero(const fq2_t* a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fq2_t* a, const fq2_t* b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static inline void fq2_set_zero(fq2_t* a) { fq_set_zero(&a->c0); fq_set_zero(&a->c1); }
static inline void fq2_set_one(fq2_t* a) { fq_set_one(&a->c0); fq_set_zero(&a->c1); }
static inline void fq2_add(fq2_t* r, const fq2_t* a, const fq2_t* b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fq2_t* r, const fq2_t* a, const fq2_t* b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void fq2_neg(fq2_t* r, const fq2_t* a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static inline void fq2_dbl(fq2_t* r, const fq2_t* a) { fq_dbl(&r->c0, &a->c0); fq_dbl(&r->c1, &a->c1); }
static inline void fq2_mul(fq2_t* r, const fq2_t* a, const fq2_t* b) {
  fq_t t0, t1, s0, s1, m;
  fq_mul(&t0, &a->c0, &b->c0);
  fq_mul(&t1, &a->c1, &b->c1);
  fq_add(&s0, &a->c0, &a->c1);
  fq_add(&s1, &b->c0, &b->c1);
  fq_mul(&m, &s0, &s1);
  fq_sub(&r->c0, &t0, &t1);
  fq_sub(&m, &m, &t0);
  fq_sub(&r->c1, &m, &t1);
}
static inline void fq2_sqr(fq2_t* r, const fq2_t* a) {
  fq_t s, d, p;
  fq_add(&s, &a->c0, &a->c1);
  fq_sub(&d, &a->c0, &a->c1);
  fq_mul(&p, &a->c0, &a->c1);
  fq_mul(&r->c0, &s, &d);
  fq_dbl(&r->c1, &p);
}
static inline void fq2_inv(fq2_t* r, const fq2_t* a) {
  fq_t n, t, ni;
  fq_sqr(&n, &a->c0);
  fq_sqr(&t, &a->c1);
  fq_add(&n, &n, &t);
  fq_inv(&ni, &n);
  fq_mul(&r->c0, &a->c0, &ni);
  fq_mul(&t, &a->c1, &ni);
  fq_neg(&r->c1, &t);
}

/* ---- groups ---- */
#define F(x) fq_##x
#define G(x) g1_##x
#include "ec_tmpl.h"
#undef F
#undef G
#define F(x) fq2_##x
#define G(x) g2_##x
#include "ec_tmpl.h"
#undef F
#undef G

static void g1_aff_from_bytes(g1_aff* p, const uint8_t* b) { fq_from_bytes(&p->x, b); fq_from_bytes(&p->y, b + 32); }
static void g1_aff_to_bytes(uint8_t* b, const g1_aff* p) { fq_to_bytes(b, &p->x); fq_to_bytes(b + 32, &p->y); }
static void g2_aff_from_bytes(g2_aff* p, const uint8_t* b) {
  fq_from_bytes(&p->x.c0, b); fq_from_bytes(&p->x.c1, b + 32);
  fq_from_bytes(&p->y.c0, b + 64); fq_from_bytes(&p->y.c1, b + 96);
}
static void g2_aff_to_bytes(uint8_t* b, const g2_aff* p) {
  fq_to_bytes(b, &p->x.c0); fq_to_bytes(b + 32, &p->x.c1);
  fq_to_bytes(b + 64, &p->y.c0); fq_to_bytes(b + 96, &p->y.c1);
}

/* ---- tiny parallel-for ---- */
typedef void (*pf_fn)(void* arg, int tid, int nthreads);
typedef struct { pf_fn fn; void* arg; int tid, n; } pf_slot;
static void* pf_tramp(void* p) { pf_slot* s = (pf_slot*)p; s->fn(s->arg, s->tid, s->n); return NULL; }
static void parallel_run(pf_fn fn, void* arg, int nthreads) {
  if (nthreads <= 1) { fn(arg, 0, 1); return; }
  pthread_t th[256];
  pf_slot sl[256];
  if (nthreads > 256) nthreads = 256;
  for (int i = 0; i < nthreads; i++) {
    sl[i].fn = fn; sl[i].arg = arg; sl[i].tid = i; sl[i].n = nthreads;
    pthread_create(&th[i], NULL, pf_tramp, &sl[i]);
  }
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
}

/* ================= exported: field ops ================= */
void oc_field_op(int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    if (field == 0) {
      fr_t x, y, r; fr_from_bytes(&x, a + 32 * i); if (op != 3) fr_from_bytes(&y, b + 32 * i);
      if (op == 0) fr_add(&r, &x, &y); else if (op == 1) fr_sub(&r, &x, &y); else if (op == 2) fr_mul(&r, &x, &y);
      else { if (fr_is_zero(&x)) r = x; else fr_inv(&r, &x); }
      fr_to_bytes(out + 32 * i, &r);
    } else {
      fq_t x, y, r; fq_from_bytes(&x, a + 32 * i); if (op != 3) fq_from_bytes(&y, b + 32 * i);
      if (op == 0) fq_add(&r, &x, &y); else if (op == 1) fq_sub(&r, &x, &y); else if (op == 2) fq_mul(&r, &x, &y);
      else { if (fq_is_zero(&x)) r = x; else fq_inv(&r, &x); }
      fq_to_bytes(out + 32 * i, &r);
    }
  }
}

/* ================= MiMC7 ================= */
#define MIMC_ROUNDS 91
static fr_t g_mimc_c[MIMC_ROUNDS];
void oc_mimc7_set_constants(const uint8_t* c /* 91 x 32 canonical */) {
  for (int i = 0; i < MIMC_ROUNDS; i++) fr_from_bytes(&g_mimc_c[i], c + 32 * i);
}
static void mimc7_permute(fr_t* out, const fr_t* x, const fr_t* k) {
  fr_t r = *x, t, t2, t4, t6;
  for (int i = 0; i < MIMC_ROUNDS; i++) {
    fr_add(&t, &r, k); fr_add(&t, &t, &g_mimc_c[i]);
    fr_sqr(&t2, &t); fr_sqr(&t4, &t2); fr_mul(&t6, &t4, &t2); fr_mul(&r, &t6, &t);
  }
  fr_add(out, &r, k);
}
static void mimc7_hash2(fr_t* out, const fr_t* l, const fr_t* r) {
  fr_t k, e; fr_set_zero(&k);
  mimc7_permute(&e, l, &k); fr_add(&k, &k, l); fr_add(&k, &k, &e);
  mimc7_permute(&e, r, &k); fr_add(&k, &k, r); fr_add(&k, &k, &e);
  *out = k;
}
void oc_mimc7_hash2(const uint8_t* l, const uint8_t* r, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    fr_t a, b, h; fr_from_bytes(&a, l + 32 * i); fr_from_bytes(&b, r + 32 * i);
    mimc7_hash2(&h, &a, &b); fr_to_bytes(out + 32 * i, &h);
  }
}
typedef struct { const uint8_t* in; uint8_t* out; size_t n_out; } tree_job;
static void tree_level_worker(void* arg, int tid, int nt) {
  tree_job* j = (tree_job*)arg;
  size_t lo = j->n_out * tid / nt, hi = j->n_out * (tid + 1) / nt;
  for (size_t i = lo; i < hi; i++) {
    fr_t a, b, h; fr_from_bytes(&a, j->in + 64 * i); fr_from_bytes(&b, j->in + 64 * i + 32);
    mimc7_hash2(&h, &a, &b); fr_to_bytes(j->out + 32 * i, &h);
  }
}
void oc_mimc7_tree_build(const uint8_t* leaves, size_t n, uint8_t* nodes, int threads) {
  memcpy(nodes, leaves, n * 32);
  size_t off = 0;
  for (size_t w = n; w > 1; w >>= 1) {
    tree_job j = {nodes + off * 32, nodes + (off + w) * 32, w >> 1};
    parallel_run(tree_level_worker, &j, (w >> 1) >= 64 ? threads : 1);
    off += w;
  }
}

/* ================= NTT ================= */
static void fr_pow_u64(fr_t* r, const fr_t* a, uint64_t e) {
  fr_t acc, b = *a; fr_set_one(&acc);
  while (e) { if (e & 1) fr_mul(&acc, &acc, &b); fr_sqr(&b, &b); e >>= 1; }
  *r = acc;
}
static void fr_root_of_unity(fr_t* w, int log_n) { /* 7^((r-1)/2^log_n) */
  /* (r-1)/2^28 = odd cofactor T; compute g = 7^T via square-and-multiply over T's limbs */
  static const uint64_t T[4] = {0x9b9709143e1f593fULL, 0x181585d2833e8487ULL, 0x131a029b85045b68ULL, 0x000000030644e72eULL};
  fr_t seven, g; uint8_t sb[32] = {7}; fr_from_bytes(&seven, sb);
  fr_set_one(&g);
  for (int i = 255; i >= 0; i--) { fr_sqr(&g, &g); if ((T[i >> 6] >> (i & 63)) & 1) fr_mul(&g, &g, &seven); }
  for (int i = 28; i > log_n; i--) fr_sqr(&g, &g);
  *w = g;
}
static void ntt_inplace(fr_t* a, int log_n, int inverse) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) { /* bit reversal */
    size_t j = 0;
    for (int b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (j > i) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  fr_t w; fr_root_of_unity(&w, log_n);
  if (inverse) fr_inv(&w, &w);
  fr_t* tw = (fr_t*)malloc(sizeof(fr_t) * (n / 2 + 1));
  fr_set_one(&tw[0]);
  for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], &w);
  for (size_t m = 1; m < n; m <<= 1) {
    size_t step = n / (2 * m);
    for (size_t s = 0; s < n; s += 2 * m)
      for (size_t j = 0; j < m; j++) {
        fr_t u = a[s + j], v; fr_mul(&v, &a[s + j + m], &tw[j * step]);
        fr_add(&a[s + j], &u, &v); fr_sub(&a[s + j + m], &u, &v);
      }
  }
  free(tw);
  if (inverse) {
    fr_t ninv, nn; uint8_t nb[32] = {0}; uint64_t nv = n; memcpy(nb, &nv, 8);
    fr_from_bytes(&nn, nb); fr_inv(&ninv, &nn);
    for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
  }
}
static void coset_scale(fr_t* a, size_t n, int inverse) { /* a_i *= g^i (or g^-i), g = 7 */
  fr_t g, acc; uint8_t sb[32] = {7}; fr_from_bytes(&g, sb);
  if (inverse) fr_inv(&g, &g);
  fr_set_one(&acc);
  for (size_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &acc); fr_mul(&acc, &acc, &g); }
}
/* in-place on canonical bytes; coset != 0: forward = scale then NTT, inverse = iNTT then unscale */
void oc_ntt(uint8_t* data, int log_n, int inverse, int coset) {
  size_t n = (size_t)1 << log_n;
  fr_t* a = (fr_t*)malloc(sizeof(fr_t) * n);
  for (size_t i = 0; i < n; i++) fr_from_bytes(&a[i], data + 32 * i);
  if (coset && !inverse) coset_scale(a, n, 0);
  ntt_inplace(a, log_n, inverse);
  if (coset && inverse) coset_scale(a, n, 1);
  for (size_t i = 0; i < n; i++) fr_to_bytes(data + 32 * i, &a[i]);
  free(a);
}

/* ================= MSM ================= */
typedef struct {
  int is_g2; const void* bases; const uint8_t* scalars; size_t n; int c, nwin; void* win_out;
} msm_job;
static void msm_worker(void* arg, int tid, int nt) {
  msm_job* j = (msm_job*)arg;
  size_t nb = (size_t)1 << j->c;
  if (!j->is_g2) {
    g1_jac* buckets = (g1_jac*)malloc(sizeof(g1_jac) * nb);
    for (int w = tid; w < j->nwin; w += nt)
      g1_msm_window(&((g1_jac*)j->win_out)[w], (const g1_aff*)j->bases, j->scalars, j->n, j->c, w, buckets);
    free(buckets);
  } else {
    g2_jac* buckets = (g2_jac*)malloc(sizeof(g2_jac) * nb);
    for (int w = tid; w < j->nwin; w += nt)
      g2_msm_window(&((g2_jac*)j->win_out)[w], (const g2_aff*)j->bases, j->scalars, j->n, j->c, w, buckets);
    free(buckets);
  }
}
static int pick_c(size_t n) {
  int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = lg - 3; if (c < 2) c = 2; if (c > 16) c = 16; return c;
}
static void g1_msm(g1_jac* out, const g1_aff* bases, const uint8_t* scalars, size_t n, int threads) {
  g1_jac_set_inf(out);
  if (n == 0) return;
  int c = pick_c(n), nwin = (254 + c - 1) / c;
  g1_jac* wins = (g1_jac*)malloc(sizeof(g1_jac) * nwin);
  msm_job j = {0, bases, scalars, n, c, nwin, wins};
  parallel_run(msm_worker, &j, threads < nwin ? threads : nwin);
  for (int w = nwin - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) g1_jac_dbl(out, out);
    g1_jac_add(out, out, &wins[w]);
  }
  free(wins);
}
static void g2_msm(g2_jac* out, const g2_aff* bases, const uint8_t* scalars, size_t n, int threads) {
  g2_jac_set_inf(out);
  if (n == 0) return;
  int c = pick_c(n), nwin = (254 + c - 1) / c;
  g2_jac* wins = (g2_jac*)malloc(sizeof(g2_jac) * nwin);
  msm_job j = {1, bases, scalars, n, c, nwin, wins};
  parallel_run(msm_worker, &j, threads < nwin ? threads : nwin);
  for (int w = nwin - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) g2_jac_dbl(out, out);
    g2_jac_add(out, out, &wins[w]);
  }
  free(wins);
}
typedef struct { const uint8_t* in; void* out; size_t n; int is_g2; } conv_job;
static void conv_worker(void* arg, int tid, int nt) {
  conv_job* j = (conv_job*)arg;
  size_t lo = j->n * tid / nt, hi = j->n * (tid + 1) / nt;
  for (size_t i = lo; i < hi; i++) {
    if (j->is_g2) g2_aff_from_bytes(&((g2_aff*)j->out)[i], j->in + 128 * i);
    else g1_aff_from_bytes(&((g1_aff*)j->out)[i], j->in + 64 * i);
  }
}
/* opaque prepared bases (Montgomery form) so the timed MSM excludes byte conversion */
void* oc_bases_g1_new(const uint8_t* bases, size_t n, int threads) {
  g1_aff* p = (g1_aff*)malloc(sizeof(g1_aff) * (n ? n : 1));
  conv_job j = {bases, p, n, 0}; parallel_run(conv_worker, &j, threads); return p;
}
void* oc_bases_g2_new(const uint8_t* bases, size_t n, int threads) {
  g2_aff* p = (g2_aff*)malloc(sizeof(g2_aff) * (n ? n : 1));
  conv_job j = {bases, p, n, 1}; parallel_run(conv_worker, &j, threads); return p;
}
void oc_bases_free(void* p) { free(p); }
void oc_msm_g1_prepared(const void* bases, const uint8_t* scalars, size_t n, uint8_t out[64], int threads) {
  g1_jac r; g1_aff a; g1_msm(&r, (const g1_aff*)bases, scalars, n, threads);
  g1_jac_to_aff(&a, &r); g1_aff_to_bytes(out, &a);
}
void oc_msm_g2_prepared(const void* bases, const uint8_t* scalars, size_t n, uint8_t out[128], int threads) {
  g2_jac r; g2_aff a; g2_msm(&r, (const g2_aff*)bases, scalars, n, threads);
  g2_jac_to_aff(&a, &r); g2_aff_to_bytes(out, &a);
}
void oc_msm_g1(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[64], int threads) {
  void* p = oc_bases_g1_new(bases, n, threads); oc_msm_g1_prepared(p, scalars, n, out, threads); free(p);
}
void oc_msm_g2(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[128], int threads) {
  void* p = oc_bases_g2_new(bases, n, threads); oc_msm_g2_prepared(p, scalars, n, out, threads); free(p);
}

/* ---- fixed-base batch scalar multiplication: out[i] = k_i * base (used by the setup oracle) ---- */
typedef struct { int is_g2; const void* table; const uint8_t* scalars; size_t n; uint8_t* out; } fb_job;
#define FB_C 8
#define FB_NWIN 32
static void fb_worker(void* arg, int tid, int nt) {
  fb_job* j = (fb_job*)arg;
  size_t lo = j->n * tid / nt, hi = j->n * (tid + 1) / nt;
  for (size_t i = lo; i < hi; i++) {
    const uint8_t* s = j->scalars + 32 * i;
    if (!j->is_g2) {
      const g1_aff* tab = (const g1_aff*)j->table; g1_jac acc; g1_jac_set_inf(&acc);
      for (int w = 0; w < FB_NWIN; w++) if (s[w]) g1_jac_madd(&acc, &acc, &tab[w * 256 + s[w]]);
      g1_aff a; g1_jac_to_aff(&a, &acc); g1_aff_to_bytes(j->out + 64 * i, &a);
    } else {
      const g2_aff* tab = (const g2_aff*)j->table; g2_jac acc; g2_jac_set_inf(&acc);
      for (int w = 0; w < FB_NWIN; w++) if (s[w]) g2_jac_madd(&acc, &acc, &tab[w * 256 + s[w]]);
      g2_aff a; g2_jac_to_aff(&a, &acc); g2_aff_to_bytes(j->out + 128 * i, &a);
    }
  }
}
void oc_fixed_base_g1(const uint8_t base[64], const uint8_t* scalars, size_t n, uint8_t* out, int threads) {
  g1_aff* tab = (g1_aff*)malloc(sizeof(g1_aff) * FB_NWIN * 256);
  g1_aff b; g1_aff_from_bytes(&b, base); g1_jac cur; g1_jac_from_aff(&cur, &b);
  for (int w = 0; w < FB_NWIN; w++) {
    g1_jac acc; g1_jac_set_inf(&acc);
    fq_set_zero(&tab[w * 256].x); fq_set_zero(&tab[w * 256].y);
    for (int d = 1; d < 256; d++) { g1_jac_add(&acc, &acc, &cur); g1_jac_to_aff(&tab[w * 256 + d], &acc); }
    for (int k = 0; k < FB_C; k++) g1_jac_dbl(&cur, &cur);
  }
  fb_job j = {0, tab, scalars, n, out}; parallel_run(fb_worker, &j, threads); free(tab);
}
void oc_fixed_base_g2(const uint8_t base[128], const uint8_t* scalars, size_t n, uint8_t* out, int threads) {
  g2_aff* tab = (g2_aff*)malloc(sizeof(g2_aff) * FB_NWIN * 256);
  g2_aff b; g2_aff_from_bytes(&b, base); g2_jac cur; g2_jac_from_aff(&cur, &b);
  for (int w = 0; w < FB_NWIN; w++) {
    g2_jac acc; g2_jac_set_inf(&acc);
    fq2_set_zero(&tab[w * 256].x); fq2_set_zero(&tab[w * 256].y);
    for (int d = 1; d < 256; d++) { g2_jac_add(&acc, &acc, &cur); g2_jac_to_aff(&tab[w * 256 + d], &acc); }
    for (int k = 0; k < FB_C; k++) g2_jac_dbl(&cur, &cur);
  }
  fb_job j = {1, tab, scalars, n, out}; parallel_run(fb_worker, &j, threads); free(tab);
}

/* ================= Groth16 prove ================= */
typedef struct {
  uint64_t n_wires, n_pub, domain_log, n_rows;   /* n_rows = constraints + n_pub + 1 */
  const uint32_t *a_ptr, *a_col; const uint8_t* a_val; /* CSR over rows, values canonical */
  const uint32_t *b_ptr, *b_col; const uint8_t* b_val;
  const uint32_t *c_ptr, *c_col; const uint8_t* c_val;
  const uint8_t *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
  const uint8_t *a_query, *b_g1_query, *b_g2_query, *l_query, *h_query;
  uint64_t flags;   /* OWPK0001 header word 8; bit 0: no C matrix, C z := (A z) o (B z) row by row (a key imported from a .zkey) */
} oc_pk;

/* prepared key: bases converted once (setup cost, excluded from the timed prove) */
typedef struct {
  oc_pk pk;
  g1_aff *a_q, *b1_q, *l_q, *h_q; g2_aff* b2_q;
  fr_t *a_v, *b_v, *c_v;
} oc_pk_prepared;

void* oc_pk_prepare(const oc_pk* pk, int threads) {
  oc_pk_prepared* p = (oc_pk_prepared*)calloc(1, sizeof(*p));
  p->pk = *pk;
  size_t m = pk->n_wires, d = (size_t)1 << pk->domain_log;
  p->a_q = (g1_aff*)oc_bases_g1_new(pk->a_query, m, threads);
  p->b1_q = (g1_aff*)oc_bases_g1_new(pk->b_g1_query, m, threads);
  p->b2_q = (g2_aff*)oc_bases_g2_new(pk->b_g2_query, m, threads);
  p->l_q = (g1_aff*)oc_bases_g1_new(pk->l_query, m - pk->n_pub - 1, threads);
  p->h_q = (g1_aff*)oc_bases_g1_new(pk->h_query, d - 1, threads);
  size_t na = pk->a_ptr[pk->n_rows], nb = pk->b_ptr[pk->n_rows], nc = pk->c_ptr[pk->n_rows];
  p->a_v = (fr_t*)malloc(sizeof(fr_t) * (na ? na : 1));
  p->b_v = (fr_t*)malloc(sizeof(fr_t) * (nb ? nb : 1));
  p->c_v = (fr_t*)malloc(sizeof(fr_t) * (nc ? nc : 1));
  for (size_t i = 0; i < na; i++) fr_from_bytes(&p->a_v[i], pk->a_val + 32 * i);
  for (size_t i = 0; i < nb; i++) fr_from_bytes(&p->b_v[i], pk->b_val + 32 * i);
  for (size_t i = 0; i < nc; i++) fr_from_bytes(&p->c_v[i], pk->c_val + 32 * i);
  return p;
}
void oc_pk_free(void* pp) {
  oc_pk_prepared* p = (oc_pk_prepared*)pp;
  free(p->a_q); free(p->b1_q); free(p->b2_q); free(p->l_q); free(p->h_q); free(p->a_v); free(p->b_v); free(p->c_v); free(p);
}

static void spmv(fr_t* out, size_t n_rows, const uint32_t* ptr, const uint32_t* col, const fr_t* val, const fr_t* z) {
  for (size_t r = 0; r < n_rows; r++) {
    fr_t acc, t; fr_set_zero(&acc);
    for (uint32_t k = ptr[r]; k < ptr[r + 1]; k++) { fr_mul(&t, &val[k], &z[col[k]]); fr_add(&acc, &acc, &t); }
    out[r] = acc;
  }
}
typedef struct { oc_pk_prepared* p; const fr_t* z; fr_t* ev[3]; size_t d; int log_d; } abc_job;
static void abc_worker(void* arg, int tid, int nt) {
  abc_job* j = (abc_job*)arg;
  for (int k = tid; k < 3; k += nt) {
    const oc_pk* pk = &j->p->pk;
    const uint32_t* ptr = k == 0 ? pk->a_ptr : k == 1 ? pk->b_ptr : pk->c_ptr;
    const uint32_t* col = k == 0 ? pk->a_col : k == 1 ? pk->b_col : pk->c_col;
    const fr_t* val = k == 0 ? j->p->a_v : k == 1 ? j->p->b_v : j->p->c_v;
    memset(j->ev[k], 0, sizeof(fr_t) * j->d);
    if (k == 2 && (pk->flags & 1)) {   /* this worker's own A z and B z: the other two are being transformed in place */
      fr_t* az = (fr_t*)calloc(j->d, sizeof(fr_t));
      spmv(az, pk->n_rows, pk->a_ptr, pk->a_col, j->p->a_v, j->z);
      spmv(j->ev[2], pk->n_rows, pk->b_ptr, pk->b_col, j->p->b_v, j->z);
      for (size_t i = 0; i < pk->n_rows; i++) fr_mul(&j->ev[2][i], &j->ev[2][i], &az[i]);
      free(az);
    } else
    spmv(j->ev[k], pk->n_rows, ptr, col, val, j->z);
    ntt_inplace(j->ev[k], j->log_d, 1);
    coset_scale(j->ev[k], j->d, 0);
    ntt_inplace(j->ev[k], j->log_d, 0);
  }
}
/* The five MSMs of a proof as ONE pool of (MSM, window) tasks that `threads` workers pull from a shared counter -- G2 windows
 * (3x the cost) first -- so that any thread count is balanced: 4 threads per proof (bench.py runs one proof per core group)
 * are as busy as 85.  Same arithmetic per window and the same Horner combine as g1_msm / g2_msm: the bytes do not change. */
typedef struct { int is_g2; const void* bases; const uint8_t* scalars; size_t n; int c, nwin; void* wins; } msm_desc;
typedef struct { msm_desc d[5]; int ntask; int task_msm[5 * 128], task_win[5 * 128]; int next; } msm5_job;

static void msm5_worker(void* arg, int tid, int nt) {
  msm5_job* j = (msm5_job*)arg;
  (void)tid; (void)nt;
  g1_jac* b1 = NULL; g2_jac* b2 = NULL; size_t cap1 = 0, cap2 = 0;
  for (;;) {
    const int t = __sync_fetch_and_add(&j->next, 1);
    if (t >= j->ntask) break;
    const msm_desc* d = &j->d[j->task_msm[t]];
    const int w = j->task_win[t];
    const size_t nb = (size_t)1 << d->c;
    if (d->is_g2) {
      if (cap2 < nb) { free(b2); b2 = (g2_jac*)malloc(sizeof(g2_jac) * nb); cap2 = nb; }
      g2_msm_window(&((g2_jac*)d->wins)[w], (const g2_aff*)d->bases, d->scalars, d->n, d->c, w, b2);
    } else {
      if (cap1 < nb) { free(b1); b1 = (g1_jac*)malloc(sizeof(g1_jac) * nb); cap1 = nb; }
      g1_msm_window(&((g1_jac*)d->wins)[w], (const g1_aff*)d->bases, d->scalars, d->n, d->c, w, b1);
    }
  }
  free(b1); free(b2);
}
static void msm5_add(msm5_job* j, int k, int is_g2, const void* bases, const uint8_t* scalars, size_t n) {
  msm_desc* d = &j->d[k];
  d->is_g2 = is_g2; d->bases = bases; d->scalars = scalars; d->n = n;
  d->c = n ? pick_c(n) : 2; d->nwin = n ? (254 + d->c - 1) / d->c : 0;
  d->wins = malloc((is_g2 ? sizeof(g2_jac) : sizeof(g1_jac)) * (d->nwin ? d->nwin : 1));
  for (int w = 0; w < d->nwin; w++) { j->task_msm[j->ntask] = k; j->task_win[j->ntask] = w; j->ntask++; }
}
static void msm5_g1_result(g1_jac* out, msm5_job* j, int k) {
  msm_desc* d = &j->d[k];
  g1_jac_set_inf(out);
  for (int w = d->nwin - 1; w >= 0; w--) {
    for (int q = 0; q < d->c; q++) g1_jac_dbl(out, out);
    g1_jac_add(out, out, &((g1_jac*)d->wins)[w]);
  }
  free(d->wins);
}
static void msm5_g2_result(g2_jac* out, msm5_job* j, int k) {
  msm_desc* d = &j->d[k];
  g2_jac_set_inf(out);
  for (int w = d->nwin - 1; w >= 0; w--) {
    for (int q = 0; q < d->c; q++) g2_jac_dbl(out, out);
    g2_jac_add(out, out, &((g2_jac*)d->wins)[w]);
  }
  free(d->wins);
}
/* threads oc_groth16_prove keeps busy in its MSM phase for a given request (reported by the CPU baseline) */
int oc_prove_threads(uint64_t n_wires, int threads) {
  int c = pick_c((size_t)n_wires), nwin = (254 + c - 1) / c;
  return threads < 5 * nwin ? threads : 5 * nwin;  /* one (MSM, window) task per thread at most */
}

static void scalar_to_limbs(uint64_t k[4], const uint8_t* b) { memcpy(k, b, 32); }

/* returns 0 ok, -4 if the witness does not satisfy the circuit (h would have degree d-1) */
int oc_groth16_prove(void* prepared, const uint8_t* witness /* m x 32 */, const uint8_t r[32], const uint8_t s[32],
                     uint8_t proof[256], int threads) {
  oc_pk_prepared* p = (oc_pk_prepared*)prepared;
  const oc_pk* pk = &p->pk;
  size_t m = pk->n_wires; int log_d = (int)pk->domain_log; size_t d = (size_t)1 << log_d;
  fr_t* z = (fr_t*)malloc(sizeof(fr_t) * m);
  for (size_t i = 0; i < m; i++) fr_from_bytes(&z[i], witness + 32 * i);
  abc_job aj; aj.p = p; aj.z = z; aj.d = d; aj.log_d = log_d;
  for (int k = 0; k < 3; k++) aj.ev[k] = (fr_t*)malloc(sizeof(fr_t) * d);
  parallel_run(abc_worker, &aj, threads < 3 ? threads : 3);
  /* h = (a*b - c) / (g^d - 1) on the coset, then back to coefficients */
  fr_t g, gd, zinv, one; uint8_t sb[32] = {7}; fr_from_bytes(&g, sb);
  fr_pow_u64(&gd, &g, (uint64_t)d); fr_set_one(&one); fr_sub(&gd, &gd, &one); fr_inv(&zinv, &gd);
  fr_t* h = aj.ev[0];
  for (size_t i = 0; i < d; i++) {
    fr_t t; fr_mul(&t, &aj.ev[0][i], &aj.ev[1][i]); fr_sub(&t, &t, &aj.ev[2][i]); fr_mul(&h[i], &t, &zinv);
  }
  ntt_inplace(h, log_d, 1);
  coset_scale(h, d, 1);
  int bad = !fr_is_zero(&h[d - 1]);
  uint8_t* hb = (uint8_t*)malloc(32 * d);
  for (size_t i = 0; i < d; i++) fr_to_bytes(hb + 32 * i, &h[i]);
  free(aj.ev[0]); free(aj.ev[1]); free(aj.ev[2]); free(z);
  if (bad) { free(hb); return -4; }

  g1_jac A, B1, L, H, C, t1; g2_jac B2, t2;
  /* the five MSMs are independent: run them concurrently, each one window-parallel on its share of the threads */
  {
    const size_t l = pk->n_pub;
    msm5_job mj; mj.ntask = 0; mj.next = 0;
    msm5_add(&mj, 0, 1, p->b2_q, witness, m);   /* the G2 windows first: the longest tasks */
    msm5_add(&mj, 1, 0, p->a_q, witness, m);
    msm5_add(&mj, 2, 0, p->b1_q, witness, m);
    msm5_add(&mj, 3, 0, p->l_q, witness + 32 * (l + 1), m - l - 1);
    msm5_add(&mj, 4, 0, p->h_q, hb, d - 1);
    parallel_run(msm5_worker, &mj, threads < mj.ntask ? (threads < 1 ? 1 : threads) : (mj.ntask ? mj.ntask : 1));
    msm5_g2_result(&B2, &mj, 0); msm5_g1_result(&A, &mj, 1); msm5_g1_result(&B1, &mj, 2);
    msm5_g1_result(&L, &mj, 3); msm5_g1_result(&H, &mj, 4);
  }
  free(hb);

  g1_aff alpha, beta1, delta1; g2_aff beta2, delta2;
  g1_aff_from_bytes(&alpha, pk->alpha_g1); g1_aff_from_bytes(&beta1, pk->beta_g1); g1_aff_from_bytes(&delta1, pk->delta_g1);
  g2_aff_from_bytes(&beta2, pk->beta_g2); g2_aff_from_bytes(&delta2, pk->delta_g2);
  uint64_t rk[4], sk[4]; scalar_to_limbs(rk, r); scalar_to_limbs(sk, s);
  g1_jac d1j; g1_jac_from_aff(&d1j, &delta1); g2_jac d2j; g2_jac_from_aff(&d2j, &delta2);
  /* A = alpha + msmA + r*delta */
  g1_jac_madd(&A, &A, &alpha); g1_jac_mul(&t1, &d1j, rk); g1_jac_add(&A, &A, &t1);
  /* B = beta + msmB + s*delta (G2 and G1) */
  g2_jac_madd(&B2, &B2, &beta2); g2_jac_mul(&t2, &d2j, sk); g2_jac_add(&B2, &B2, &t2);
  g1_jac_madd(&B1, &B1, &beta1); g1_jac_mul(&t1, &d1j, sk); g1_jac_add(&B1, &B1, &t1);
  /* C = L + H + s*A + r*B1 - (r*s)*delta */
  g1_jac_add(&C, &L, &H);
  g1_jac_mul(&t1, &A, sk); g1_jac_add(&C, &C, &t1);
  g1_jac_mul(&t1, &B1, rk); g1_jac_add(&C, &C, &t1);
  fr_t rf, sf, rs; fr_from_bytes(&rf, r); fr_from_bytes(&sf, s); fr_mul(&rs, &rf, &sf);
  uint8_t rsb[32]; fr_to_bytes(rsb, &rs); uint64_t rsk[4]; scalar_to_limbs(rsk, rsb);
  g1_jac_mul(&t1, &d1j, rsk);
  g1_aff ta; g1_jac_to_aff(&ta, &t1); g1_aff_neg(&ta, &ta); g1_jac_madd(&C, &C, &ta);
  g1_aff Aa, Ca; g2_aff Ba;
  g1_jac_to_aff(&Aa, &A); g2_jac_to_aff(&Ba, &B2); g1_jac_to_aff(&Ca, &C);
  g1_aff_to_bytes(proof, &Aa); g2_aff_to_bytes(proof + 64, &Ba); g1_aff_to_bytes(proof + 192, &Ca);
  return 0;
}

/* h-polynomial only (for NTT-pipeline parity): evals a,b,c (d x 32 canonical) -> h coeffs */
void oc_h_poly(const uint8_t* a, const uint8_t* b, const uint8_t* c, int log_d, uint8_t* h_out) {
  size_t d = (size_t)1 << log_d;
  fr_t* ev[3]; const uint8_t* src[3] = {a, b, c};
  for (int k = 0; k < 3; k++) {
    ev[k] = (fr_t*)malloc(sizeof(fr_t) * d);
    for (size_t i = 0; i < d; i++) fr_from_bytes(&ev[k][i], src[k] + 32 * i);
    ntt_inplace(ev[k], log_d, 1); coset_scale(ev[k], d, 0); ntt_inplace(ev[k], log_d, 0);
  }
  fr_t g, gd, zinv, one; uint8_t sb[32] = {7}; fr_from_bytes(&g, sb);
  fr_pow_u64(&gd, &g, (uint64_t)d); fr_set_one(&one); fr_sub(&gd, &gd, &one); fr_inv(&zinv, &gd);
  for (size_t i = 0; i < d; i++) {
    fr_t t; fr_mul(&t, &ev[0][i], &ev[1][i]); fr_sub(&t, &t, &ev[2][i]); fr_mul(&ev[0][i], &t, &zinv);
  }
  ntt_inplace(ev[0], log_d, 1); coset_scale(ev[0], d, 1);
  for (size_t i = 0; i < d; i++) fr_to_bytes(h_out + 32 * i, &ev[0][i]);
  for (int k = 0; k < 3; k++) free(ev[k]);
}
