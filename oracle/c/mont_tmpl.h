/* TEST INFRASTRUCTURE ONLY -- 4x64-bit Montgomery field template for the C oracle.
 * Include with FNAME(x) and FIELD_* macros defined; generates one field's functions.
 * Fr restates the reference's `Fp`
 * (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11; the `ff`
 * derive produces the same [u64;4] Montgomery form with R = 2^256); Fq is EIP-196. */
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[4]; } FNAME(t);

static const uint64_t FNAME(N)[4] = FIELD_N;
static const uint64_t FNAME(R2)[4] = FIELD_R2;
static const uint64_t FNAME(ONE)[4] = FIELD_ONE;
#define FNAME_INV FIELD_INV

static inline int FNAME(is_zero)(const FNAME(t)* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int FNAME(eq)(const FNAME(t)* a, const FNAME(t)* b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline void FNAME(set_zero)(FNAME(t)* a) { memset(a, 0, sizeof(*a)); }
static inline void FNAME(set_one)(FNAME(t)* a) { memcpy(a->l, FNAME(ONE), 32); }

static inline uint64_t FNAME(sub_raw)(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  unsigned __int128 bw = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)a[i] - b[i] - (uint64_t)bw;
    r[i] = (uint64_t)d;
    bw = (d >> 64) & 1;
  }
  return (uint64_t)bw;
}
static inline uint64_t FNAME(add_raw)(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static inline void FNAME(reduce_once)(uint64_t r[4]) {
  uint64_t t[4];
  if (!FNAME(sub_raw)(t, r, FNAME(N))) memcpy(r, t, 32);
}
static inline void FNAME(add)(FNAME(t)* r, const FNAME(t)* a, const FNAME(t)* b) {
  FNAME(add_raw)(r->l, a->l, b->l);
  FNAME(reduce_once)(r->l);
}
static inline void FNAME(sub)(FNAME(t)* r, const FNAME(t)* a, const FNAME(t)* b) {
  uint64_t t[4];
  if (FNAME(sub_raw)(t, a->l, b->l)) FNAME(add_raw)(t, t, FNAME(N));
  memcpy(r->l, t, 32);
}
static inline void FNAME(neg)(FNAME(t)* r, const FNAME(t)* a) {
  if (FNAME(is_zero)(a)) { FNAME(set_zero)(r); return; }
  FNAME(sub_raw)(r->l, FNAME(N), a->l);
}
static inline void FNAME(dbl)(FNAME(t)* r, const FNAME(t)* a) { FNAME(add)(r, a, a); }

/* CIOS Montgomery multiplication, R = 2^256 */
static inline void FNAME(mul)(FNAME(t)* r, const FNAME(t)* a, const FNAME(t)* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (unsigned __int128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * FNAME_INV;
    c = (unsigned __int128)m * FNAME(N)[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (unsigned __int128)m * FNAME(N)[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  uint64_t o[4] = {t[0], t[1], t[2], t[3]};
  FNAME(reduce_once)(o);
  memcpy(r->l, o, 32);
}
static inline void FNAME(sqr)(FNAME(t)* r, const FNAME(t)* a) { FNAME(mul)(r, a, a); }

static inline void FNAME(from_bytes)(FNAME(t)* r, const uint8_t* b) { /* canonical LE -> Montgomery */
  FNAME(t) c, r2;
  memcpy(c.l, b, 32);
  memcpy(r2.l, FNAME(R2), 32);
  FNAME(mul)(r, &c, &r2);
}
static inline void FNAME(to_bytes)(uint8_t* b, const FNAME(t)* a) { /* Montgomery -> canonical LE */
  FNAME(t) one = {{1, 0, 0, 0}}, c;
  FNAME(mul)(&c, a, &one);
  memcpy(b, c.l, 32);
}
static inline void FNAME(pow_n_minus_2)(FNAME(t)* r, const FNAME(t)* a) {
  uint64_t e[4];
  memcpy(e, FNAME(N), 32);
  e[0] -= 2;
  FNAME(t) acc;
  FNAME(set_one)(&acc);
  for (int i = 255; i >= 0; i--) {
    FNAME(sqr)(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) FNAME(mul)(&acc, &acc, a);
  }
  *r = acc;
}
static inline void FNAME(inv)(FNAME(t)* r, const FNAME(t)* a) { FNAME(pow_n_minus_2)(r, a); }
#undef FNAME_INV
