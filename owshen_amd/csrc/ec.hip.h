// BN254 G1 / G2 group law for gfx950.  No reference counterpart
// (the snapshot has no G1/G2); curves per EIP-196/197:
//   G1: y^2 = x^3 + 3 over Fq;   G2: y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1).
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed add 8M+2S, full add 12M+2S, double 6M+3S (EFD madd-2008-s / add-2008-s / dbl-2008-s-1),
// which minimises Montgomery multiplications -- the binding resource on CDNA4.
// Affine infinity is encoded as (0, 0) (not on either curve); XYZZ infinity as ZZ = 0.
#pragma once
#include "field.hip.h"

namespace og {

// ---- Fq2 --------------------------------------------------------------------
struct Fq2 {
  Fq c0, c1;
  OG_HD static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  OG_HD static Fq2 one() { return {Fq::one(), Fq::zero()}; }
  OG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  OG_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
};

// uniform field interface so the group law is written once
OG_HD Fq f_add(const Fq& a, const Fq& b) { return fe_add(a, b); }
OG_HD Fq f_sub(const Fq& a, const Fq& b) { return fe_sub(a, b); }
OG_HD Fq f_mul(const Fq& a, const Fq& b) { return fe_mul(a, b); }
OG_HD Fq f_sqr(const Fq& a) { return fe_sqr(a); }
OG_HD Fq f_dbl(const Fq& a) { return fe_dbl(a); }
OG_HD Fq f_neg(const Fq& a) { return fe_neg(a); }
OG_HD Fq f_inv(const Fq& a) { return fe_inv(a); }

OG_HD Fq2 f_add(const Fq2& a, const Fq2& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
OG_HD Fq2 f_sub(const Fq2& a, const Fq2& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
OG_HD Fq2 f_dbl(const Fq2& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
OG_HD Fq2 f_neg(const Fq2& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
// Schoolbook over the carry-free 64-bit columns with TWO reductions instead of Karatsuba's three reductions and five
// modular add/subs: c0 = a0 b0 + (8N - a1) b1, c1 = a0 b1 + a1 b0, each accumulated before one Montgomery
// reduction (field.hip.h).  A three-product Karatsuba in the columns (P3 - P1 + P2') saves 81 multiply-adds per
// product but needs three live column sets: measured on the G2 accumulation kernel it spills (scratch 172 -> 332 B
// per lane at 2 waves/SIMD) and runs 18 % slower, so the four-product form stays.
OG_HD Fq2 f_mul(const Fq2& a, const Fq2& b) {
  return {fe_mul_add(a.c0, b.c0, fe_neg_lazy(a.c1), b.c1), fe_mul_add(a.c0, b.c1, a.c1, b.c0)};
}
// re = a0^2 + (8N - a1) a1 (45 + 81 products, one reduction), im = (2 a0) a1; operands may be weak (< 8N)
OG_HD Fq2 f_sqr(const Fq2& a) {
  return {fe_sqr_add(a.c0, fe_neg_lazy(a.c1), a.c1), fe_mul(fe_dbl_lazy(a.c0), a.c1)};
}
// a b - c d with one reduction per component
OG_HD Fq f_mul_sub(const Fq& a, const Fq& b, const Fq& c, const Fq& d) {
  return fe_mul_add(a, b, fe_neg_lazy(c), d);
}
OG_HD Fq2 f_mul_sub(const Fq2& a, const Fq2& b, const Fq2& c, const Fq2& d) {
  const Fq na1 = fe_neg_lazy(a.c1), nc0 = fe_neg_lazy(c.c0), nc1 = fe_neg_lazy(c.c1);
  // re: a0 b0 - a1 b1 - c0 d0 + c1 d1     im: a0 b1 + a1 b0 - c0 d1 - c1 d0
  return {fe_mul_add4(a.c0, b.c0, na1, b.c1, nc0, d.c0, c.c1, d.c1), fe_mul_add4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
}
OG_HD Fq2 f_inv(const Fq2& a) {
  Fq n = fe_add(fe_sqr(a.c0), fe_sqr(a.c1));
  Fq ni = fe_inv(n);
  return {fe_mul(a.c0, ni), fe_neg(fe_mul(a.c1, ni))};
}

// weak forms (field.hip.h): normalized limbs, value bounded by a small multiple of N, never stored
OG_HD Fq f_sub_weak(const Fq& a, const Fq& b) { return fe_sub_weak(a, b); }
OG_HD Fq f_add2_weak(const Fq& a, const Fq& b) { return fe_add2_weak(a, b); }
OG_HD bool f_weak_diff_is_zero(const Fq& d) { return fe_weak_diff_is_zero(d); }
OG_HD Fq2 f_sub_weak(const Fq2& a, const Fq2& b) { return {fe_sub_weak(a.c0, b.c0), fe_sub_weak(a.c1, b.c1)}; }
OG_HD Fq2 f_add2_weak(const Fq2& a, const Fq2& b) { return {fe_add2_weak(a.c0, b.c0), fe_add2_weak(a.c1, b.c1)}; }
OG_HD bool f_weak_diff_is_zero(const Fq2& d) { return fe_weak_diff_is_zero(d.c0) && fe_weak_diff_is_zero(d.c1); }
// a b - x + 4N with ONE reduction and no carry pass (field.hip.h fe_mul_plus): for x < 2N the result lies in (2N, 6N) and is
// limb-for-limb what f_sub_weak(f_mul(a, b), x) gives.  In Fq2 the lazily negated operand is a's (a.c1).
OG_HD Fq f_mul_minus(const Fq& a, const Fq& b, const Fq& x) { return fe_mul_plus(a, b, fe_neg_lazy4(x)); }
OG_HD Fq2 f_mul_minus(const Fq2& a, const Fq2& b, const Fq2& x) {
  return {fe_mul_add_plus(a.c0, b.c0, fe_neg_lazy(a.c1), b.c1, fe_neg_lazy4(x.c0)), fe_mul_add_plus(a.c0, b.c1, a.c1, b.c0, fe_neg_lazy4(x.c1))};
}
// (neg ? -y : y) z - x + 4N, one reduction per component: the sign of a signed-digit entry is applied to the OPERANDS of the
// product (4N - y limb-wise, no carries), never to y itself.  In Fq2, c0 = y0 z0 - y1 z1 flips to (4N - y0) z0 + y1 z1.
OG_HD Fq f_mul_minus_y(const Fq& y, bool neg, const Fq& z, const Fq& x) {
  const Fq ny = fe_neg_lazy4(y);
  Fq a;
#pragma unroll
  for (int i = 0; i < 9; i++) a.l[i] = neg ? ny.l[i] : y.l[i];
  return fe_mul_plus(a, z, fe_neg_lazy4(x));
}
OG_HD Fq2 f_mul_minus_y(const Fq2& y, bool neg, const Fq2& z, const Fq2& x) {
  const Fq n0 = fe_neg_lazy4(y.c0), n1 = fe_neg_lazy4(y.c1), m1 = fe_neg_lazy(y.c1);
  Fq a0, a1, s1;  // a0 = +-y0, a1 = +-y1, s1 = -+y1 (the operand of the subtracted term of c0)
#pragma unroll
  for (int i = 0; i < 9; i++) {
    a0.l[i] = neg ? n0.l[i] : y.c0.l[i];
    a1.l[i] = neg ? n1.l[i] : y.c1.l[i];
    s1.l[i] = neg ? y.c1.l[i] : m1.l[i];
  }
  return {fe_mul_add_plus(a0, z.c0, s1, z.c1, fe_neg_lazy4(x.c0)), fe_mul_add_plus(a0, z.c1, a1, z.c0, fe_neg_lazy4(x.c1))};
}

// a a - c d with one reduction per component (for X3 = R^2 - PP (P + 2 X1))
// Bounds (multiples of N): a < 6, c < 2, d < 10.  Fq: 36 + 4 * 10 = 76 N^2.  Fq2: re 36 + 8*6 + 4*10 + 2*10 = 144,
// im 36 + 36 + 40 + 40 = 152, all < 169 (field.hip.h) -- which is why c is negated against 4N, not 8N.
OG_HD Fq f_sqr_sub(const Fq& a, const Fq& c, const Fq& d) { return fe_sqr_add(a, fe_neg_lazy4(c), d); }
OG_HD Fq2 f_sqr_sub(const Fq2& a, const Fq2& c, const Fq2& d) {
  // re: a0^2 - a1^2 - c0 d0 + c1 d1     im: 2 a0 a1 - c0 d1 - c1 d0
  const Fq na1 = fe_neg_lazy(a.c1), nc0 = fe_neg_lazy4(c.c0), nc1 = fe_neg_lazy4(c.c1);
  return {fe_sqr_add3(a.c0, na1, a.c1, nc0, d.c0, c.c1, d.c1), fe_mul_add3(fe_dbl_lazy(a.c0), a.c1, nc0, d.c1, nc1, d.c0)};
}

template <class T> struct FieldIO;
template <> struct FieldIO<Fq> {
  static constexpr int BYTES = 32;
  OG_HD static Fq load(const uint8_t* p) { return fe_load<FqParams>(p); }
  OG_HD static void store(uint8_t* p, const Fq& v) { fe_store(p, v); }
  OG_HD static Fq to_mont(const Fq& v) { return fe_to_mont(v); }
  OG_HD static Fq from_mont(const Fq& v) { return fe_from_mont(v); }
};
template <> struct FieldIO<Fq2> {
  static constexpr int BYTES = 64;
  OG_HD static Fq2 load(const uint8_t* p) { return {fe_load<FqParams>(p), fe_load<FqParams>(p + 32)}; }
  OG_HD static void store(uint8_t* p, const Fq2& v) { fe_store(p, v.c0); fe_store(p + 32, v.c1); }
  OG_HD static Fq2 to_mont(const Fq2& v) { return {fe_to_mont(v.c0), fe_to_mont(v.c1)}; }
  OG_HD static Fq2 from_mont(const Fq2& v) { return {fe_from_mont(v.c0), fe_from_mont(v.c1)}; }
};

// ---- points -------------------------------------------------------------------
template <class T>
struct Affine {
  T x, y;
  static constexpr int BYTES = 2 * FieldIO<T>::BYTES;
  OG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  OG_HD static Affine inf() { return {T::zero(), T::zero()}; }
  // raw load/store of the in-HBM (Montgomery) representation
  OG_HD static Affine load(const uint8_t* p) {
    return {FieldIO<T>::load(p), FieldIO<T>::load(p + FieldIO<T>::BYTES)};
  }
  OG_HD void store(uint8_t* p) const {
    FieldIO<T>::store(p, x);
    FieldIO<T>::store(p + FieldIO<T>::BYTES, y);
  }
};

template <class T>
struct XYZZ {
  T x, y, zz, zzz;
  static constexpr int BYTES = 4 * FieldIO<T>::BYTES;
  OG_HD bool is_inf() const { return zz.is_zero(); }
  OG_HD static XYZZ inf() { return {T::one(), T::one(), T::zero(), T::zero()}; }
  OG_HD static XYZZ from_affine(const Affine<T>& a) {
    if (a.is_inf()) return inf();
    return {a.x, a.y, T::one(), T::one()};
  }
  OG_HD static XYZZ load(const uint8_t* p) {
    constexpr int B = FieldIO<T>::BYTES;
    return {FieldIO<T>::load(p), FieldIO<T>::load(p + B), FieldIO<T>::load(p + 2 * B), FieldIO<T>::load(p + 3 * B)};
  }
  OG_HD void store(uint8_t* p) const {
    constexpr int B = FieldIO<T>::BYTES;
    FieldIO<T>::store(p, x);
    FieldIO<T>::store(p + B, y);
    FieldIO<T>::store(p + 2 * B, zz);
    FieldIO<T>::store(p + 3 * B, zzz);
  }
};

template <class T>
OG_HD Affine<T> affine_neg(const Affine<T>& p) {
  return {p.x, f_neg(p.y)};
}

// 2*P for affine P (mdbl-2008-s-1)
template <class T>
OG_HD XYZZ<T> xyzz_dbl_affine(const Affine<T>& p) {
  if (p.is_inf() || p.y.is_zero()) return XYZZ<T>::inf();
  T U = f_dbl(p.y);
  T V = f_sqr(U);
  T W = f_mul(U, V);
  T S = f_mul(p.x, V);
  T xx = f_sqr(p.x);
  T M = f_add(f_dbl(xx), xx);
  T X3 = f_sub(f_sqr(M), f_dbl(S));
  T Y3 = f_mul_sub(M, f_sub(S, X3), W, p.y);
  return {X3, Y3, V, W};
}

// 2*P (dbl-2008-s-1)
template <class T>
OG_HD XYZZ<T> xyzz_dbl(const XYZZ<T>& p) {
  if (p.is_inf() || p.y.is_zero()) return XYZZ<T>::inf();
  T U = f_dbl(p.y);
  T V = f_sqr(U);
  T W = f_mul(U, V);
  T S = f_mul(p.x, V);
  T xx = f_sqr(p.x);
  T M = f_add(f_dbl(xx), xx);
  T X3 = f_sub(f_sqr(M), f_dbl(S));
  T Y3 = f_mul_sub(M, f_sub(S, X3), W, p.y);
  return {X3, Y3, f_mul(V, p.zz), f_mul(W, p.zzz)};
}

// acc + (neg ? -q : q), q affine (madd-2008-s); complete: handles acc = inf, q = inf, q = +-acc.
// Inside, differences are "weak" (bounds in the comments, in multiples of N):
//   * P = U2 - X1, R = S2 - Y1 and D = Q - X3 come straight out of the Montgomery reduction of the product they follow
//     (f_mul_minus: the subtrahend rides in the high columns) -- no separate subtraction, no carry pass;
//   * X3 = R^2 - PPP - 2Q is the fused R R - PP (P + 2 X1), Y3 = R D - Y1 PPP one reduction per component;
//   * the sign of a signed-digit entry negates q.y lazily (4N - y limb-wise, a multiplication operand); only the two rare
//     branches that need y itself negate it properly.
// 8 product sets + 2 squarings with 9 reductions, 1 carry pass.  Outputs are < 2N again.
template <class T>
OG_HD XYZZ<T> xyzz_madd_signed(const XYZZ<T>& a, const Affine<T>& q, bool neg) {
  if (q.is_inf()) return a;
  if (a.is_inf()) return XYZZ<T>::from_affine(neg ? affine_neg(q) : q);
  T P = f_mul_minus(q.x, a.zz, a.x);       // U2 - X1 + 4N: (2, 6)
  T R = f_mul_minus_y(q.y, neg, a.zzz, a.y);  // S2 - Y1 + 4N: (2, 6)   (operands <= 8N x 2N: far below 169 N^2)
  if (f_weak_diff_is_zero(P)) {
    if (f_weak_diff_is_zero(R)) return xyzz_dbl_affine(neg ? affine_neg(q) : q);
    return XYZZ<T>::inf();
  }
  T PP = f_sqr(P);                         // Fq2 worst case 36 + 8 * 6 = 84 N^2 -> < 2
  T PPP = f_mul(PP, P);                    // the lazily negated operand inside an Fq2 product is PP's, not P's
  T W = f_add2_weak(P, a.x);               // P + 2 X1: < 10
  T X3 = f_sqr_sub(R, PP, W);              // R^2 - PP W: <= 152 N^2 -> < 2
  T D = f_mul_minus(a.x, PP, X3);          // Q - X3 + 4N: < 6
  T Y3 = f_mul_sub(R, D, a.y, PPP);        // R D - Y1 PPP: <= 36 + 48 + 16 + 4 = 104 N^2 -> < 2
  return {X3, Y3, f_mul(a.zz, PP), f_mul(a.zzz, PPP)};
}

template <class T>
OG_HD XYZZ<T> xyzz_madd(const XYZZ<T>& a, const Affine<T>& q) {
  return xyzz_madd_signed(a, q, false);
}

// a + b (add-2008-s); complete.  Same structure as xyzz_madd_signed: the three differences come out of reductions.
template <class T>
OG_HD XYZZ<T> xyzz_add(const XYZZ<T>& a, const XYZZ<T>& b) {
  if (b.is_inf()) return a;
  if (a.is_inf()) return b;
  T U1 = f_mul(a.x, b.zz);
  T S1 = f_mul(a.y, b.zzz);
  T P = f_mul_minus(b.x, a.zz, U1);        // U2 - U1 + 4N: (2, 6)
  T R = f_mul_minus(b.y, a.zzz, S1);       // S2 - S1 + 4N: (2, 6)
  if (f_weak_diff_is_zero(P)) {
    if (f_weak_diff_is_zero(R)) return xyzz_dbl(a);
    return XYZZ<T>::inf();
  }
  T PP = f_sqr(P);
  T PPP = f_mul(PP, P);
  T W = f_add2_weak(P, U1);
  T X3 = f_sqr_sub(R, PP, W);
  T D = f_mul_minus(U1, PP, X3);           // Q - X3 + 4N: < 6
  T Y3 = f_mul_sub(R, D, S1, PPP);
  return {X3, Y3, f_mul(f_mul(a.zz, b.zz), PP), f_mul(f_mul(a.zzz, b.zzz), PPP)};
}

template <class T>
OG_HD XYZZ<T> xyzz_neg(const XYZZ<T>& p) {
  return {p.x, f_neg(p.y), p.zz, p.zzz};
}

// x = X/ZZ, y = Y/ZZZ (one inversion: 1/ZZZ, then 1/ZZ = ZZZ^-1 ... via ZZ^3 = ZZZ^2)
template <class T>
OG_HD Affine<T> xyzz_to_affine(const XYZZ<T>& p) {
  if (p.is_inf()) return Affine<T>::inf();
  T izzz = f_inv(p.zzz);
  // 1/ZZ = ZZ^2 / ZZ^3 = ZZ^2 / ZZZ^2 = (ZZ * izzz)^2
  T t = f_mul(p.zz, izzz);
  T izz = f_sqr(t);
  return {f_mul(p.x, izz), f_mul(p.y, izzz)};
}

// NO OUT-OF-LINE DEVICE FUNCTIONS.  On ROCm 7.2 / gfx950 every kernel that CALLED a device function
// taking or returning XYZZ<Fq2> (by reference, by pointer or by value) hung, while kernels that inline
// the same group law run correctly (bisected on the GPU in round 1; reproducer in the git history of tools/dbg/).  The whole EC layer is
// therefore __forceinline__, and kernels keep code size in check by having ONE textual site per
// primitive (add / dbl / madd / to_affine) driven by a small rolled "op loop" -- see msm_impl.hip.h.

typedef Affine<Fq> G1Affine;
typedef Affine<Fq2> G2Affine;
typedef XYZZ<Fq> G1XYZZ;
typedef XYZZ<Fq2> G2XYZZ;

}  // namespace og
