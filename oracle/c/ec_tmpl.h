/* TEST INFRASTRUCTURE ONLY -- short-Weierstrass (a = 0) Jacobian group template for the C
 * oracle.  Define F(x) (field prefix), G(x) (group prefix) before including.  No reference
 * counterpart; formulas: EFD dbl-2009-l, add-2007-bl, madd-2007-bl. */

typedef struct { F(t) x, y; } G(aff);      /* infinity encoded as (0, 0) */
typedef struct { F(t) x, y, z; } G(jac);   /* infinity: z = 0 */

static inline int G(aff_is_inf)(const G(aff)* p) { return F(is_zero)(&p->x) && F(is_zero)(&p->y); }
static inline void G(jac_set_inf)(G(jac)* p) { F(set_one)(&p->x); F(set_one)(&p->y); F(set_zero)(&p->z); }
static inline int G(jac_is_inf)(const G(jac)* p) { return F(is_zero)(&p->z); }
static inline void G(jac_from_aff)(G(jac)* r, const G(aff)* p) {
  if (G(aff_is_inf)(p)) { G(jac_set_inf)(r); return; }
  r->x = p->x; r->y = p->y; F(set_one)(&r->z);
}

static inline void G(jac_dbl)(G(jac)* r, const G(jac)* p) {
  if (G(jac_is_inf)(p)) { *r = *p; return; }
  F(t) A, B, C, D, E, Fv, t, X3, Y3, Z3;
  F(sqr)(&A, &p->x);
  F(sqr)(&B, &p->y);
  F(sqr)(&C, &B);
  F(add)(&t, &p->x, &B); F(sqr)(&t, &t); F(sub)(&t, &t, &A); F(sub)(&t, &t, &C);
  F(dbl)(&D, &t);
  F(dbl)(&E, &A); F(add)(&E, &E, &A);
  F(sqr)(&Fv, &E);
  F(dbl)(&t, &D); F(sub)(&X3, &Fv, &t);
  F(sub)(&t, &D, &X3); F(mul)(&Y3, &E, &t);
  F(dbl)(&t, &C); F(dbl)(&t, &t); F(dbl)(&t, &t);
  F(sub)(&Y3, &Y3, &t);
  F(dbl)(&t, &p->y); F(mul)(&Z3, &t, &p->z);
  r->x = X3; r->y = Y3; r->z = Z3;
}

static inline void G(jac_add)(G(jac)* r, const G(jac)* p, const G(jac)* q) {
  if (G(jac_is_inf)(p)) { *r = *q; return; }
  if (G(jac_is_inf)(q)) { *r = *p; return; }
  F(t) Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
  F(sqr)(&Z1Z1, &p->z); F(sqr)(&Z2Z2, &q->z);
  F(mul)(&U1, &p->x, &Z2Z2); F(mul)(&U2, &q->x, &Z1Z1);
  F(mul)(&t, &q->z, &Z2Z2); F(mul)(&S1, &p->y, &t);
  F(mul)(&t, &p->z, &Z1Z1); F(mul)(&S2, &q->y, &t);
  if (F(eq)(&U1, &U2)) {
    if (F(eq)(&S1, &S2)) { G(jac_dbl)(r, p); return; }
    G(jac_set_inf)(r); return;
  }
  F(sub)(&H, &U2, &U1); F(sub)(&Rr, &S2, &S1);
  F(sqr)(&HH, &H); F(mul)(&HHH, &H, &HH); F(mul)(&V, &U1, &HH);
  F(sqr)(&X3, &Rr); F(sub)(&X3, &X3, &HHH); F(dbl)(&t, &V); F(sub)(&X3, &X3, &t);
  F(sub)(&t, &V, &X3); F(mul)(&Y3, &Rr, &t); F(mul)(&t, &S1, &HHH); F(sub)(&Y3, &Y3, &t);
  F(mul)(&Z3, &p->z, &q->z); F(mul)(&Z3, &Z3, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* mixed addition: q affine */
static inline void G(jac_madd)(G(jac)* r, const G(jac)* p, const G(aff)* q) {
  if (G(aff_is_inf)(q)) { *r = *p; return; }
  if (G(jac_is_inf)(p)) { G(jac_from_aff)(r, q); return; }
  F(t) Z1Z1, U2, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;
  F(sqr)(&Z1Z1, &p->z);
  F(mul)(&U2, &q->x, &Z1Z1);
  F(mul)(&t, &p->z, &Z1Z1); F(mul)(&S2, &q->y, &t);
  if (F(eq)(&p->x, &U2)) {
    if (F(eq)(&p->y, &S2)) { G(jac_dbl)(r, p); return; }
    G(jac_set_inf)(r); return;
  }
  F(sub)(&H, &U2, &p->x); F(sub)(&Rr, &S2, &p->y);
  F(sqr)(&HH, &H); F(mul)(&HHH, &H, &HH); F(mul)(&V, &p->x, &HH);
  F(sqr)(&X3, &Rr); F(sub)(&X3, &X3, &HHH); F(dbl)(&t, &V); F(sub)(&X3, &X3, &t);
  F(sub)(&t, &V, &X3); F(mul)(&Y3, &Rr, &t); F(mul)(&t, &p->y, &HHH); F(sub)(&Y3, &Y3, &t);
  F(mul)(&Z3, &p->z, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}

static inline void G(jac_to_aff)(G(aff)* r, const G(jac)* p) {
  if (G(jac_is_inf)(p)) { F(set_zero)(&r->x); F(set_zero)(&r->y); return; }
  F(t) zi, zi2, zi3;
  F(inv)(&zi, &p->z);
  F(sqr)(&zi2, &zi); F(mul)(&zi3, &zi2, &zi);
  F(mul)(&r->x, &p->x, &zi2); F(mul)(&r->y, &p->y, &zi3);
}

static inline void G(aff_neg)(G(aff)* r, const G(aff)* p) { r->x = p->x; F(neg)(&r->y, &p->y); }

/* k is a canonical 256-bit little-endian scalar (4 x u64) */
static void G(jac_mul)(G(jac)* r, const G(jac)* p, const uint64_t k[4]) {
  G(jac) acc; G(jac_set_inf)(&acc);
  for (int i = 255; i >= 0; i--) {
    G(jac_dbl)(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) G(jac_add)(&acc, &acc, p);
  }
  *r = acc;
}

/* one Pippenger window: sum over points of digit_w(s_i) * P_i, unsigned c-bit digits.
 * scalars: canonical LE 32 B each. */
static void G(msm_window)(G(jac)* out, const G(aff)* bases, const uint8_t* scalars, size_t n, int c, int w,
                          G(jac)* buckets) {
  size_t nb = ((size_t)1 << c);
  for (size_t b = 0; b < nb; b++) G(jac_set_inf)(&buckets[b]);
  int bit0 = w * c;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* s = scalars + i * 32;
    uint64_t lim[5] = {0, 0, 0, 0, 0};
    memcpy(lim, s, 32);
    int q = bit0 >> 6, sh = bit0 & 63;
    uint64_t d = lim[q] >> sh;
    if (sh && q < 4) d |= lim[q + 1] << (64 - sh);
    d &= nb - 1;
    if (d) G(jac_madd)(&buckets[d], &buckets[d], &bases[i]);
  }
  G(jac) run, acc;
  G(jac_set_inf)(&run); G(jac_set_inf)(&acc);
  for (size_t b = nb - 1; b >= 1; b--) {
    G(jac_add)(&run, &run, &buckets[b]);
    G(jac_add)(&acc, &acc, &run);
  }
  *out = acc;
}
