// og_verify: Groth16 verification on the CPU (the `burn_tx` seam:
// /root/reference/src/blockchain/tx/burn_tx.rs:11-32 is where a sequencer would gate the debit on it).
// No reference counterpart (the snapshot verifies an ECDSA signature: contracts/src/Owshen.sol:66-78).
// The check is the EIP-197 predicate  e(-A, B) e(alpha, beta) e(vk_x, gamma) e(C, delta) == 1  with
// vk_x = IC_0 + sum x_i IC_i, evaluated with an optimal-ate Miller loop (affine line functions) over the
// tower Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(9+u)), Fq12 = Fq6[w]/(w^2-v) and a plain square-and-multiply
// final exponentiation.  It reuses the SAME 9 x 29-bit field layer and group law as the kernels, compiled for
// the host (no GPU is touched: the verifier must work on a sequencer without one).
#include "ctx.h"
#include "ec.hip.h"
#include <string.h>

namespace og {
namespace {

struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

inline Fq fq_from_u32(uint32_t v) { return fe_to_mont(fe_from_u32<FqParams>(v)); }
inline Fq2 fq2_conj(const Fq2& a) { return {a.c0, fe_neg(a.c1)}; }
inline Fq2 fq2_scale(const Fq2& a, const Fq& k) { return {fe_mul(a.c0, k), fe_mul(a.c1, k)}; }
// (a0 + a1 u)(9 + u) = 9 a0 - a1 + (a0 + 9 a1) u
inline Fq2 fq2_mul_xi(const Fq2& a) {
  const Fq nine = fq_from_u32(9);
  return {fe_sub(fe_mul(a.c0, nine), a.c1), fe_add(a.c0, fe_mul(a.c1, nine))};
}
inline bool fq2_eq(const Fq2& a, const Fq2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }

inline Fq6 f6_zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
inline Fq6 f6_one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
inline Fq6 f6_add(const Fq6& a, const Fq6& b) { return {f_add(a.c0, b.c0), f_add(a.c1, b.c1), f_add(a.c2, b.c2)}; }
inline Fq6 f6_sub(const Fq6& a, const Fq6& b) { return {f_sub(a.c0, b.c0), f_sub(a.c1, b.c1), f_sub(a.c2, b.c2)}; }
inline Fq6 f6_mul(const Fq6& a, const Fq6& b) {
  const Fq2 t0 = f_mul(a.c0, b.c0), t1 = f_mul(a.c1, b.c1), t2 = f_mul(a.c2, b.c2);
  const Fq2 c0 = f_add(t0, fq2_mul_xi(f_sub(f_mul(f_add(a.c1, a.c2), f_add(b.c1, b.c2)), f_add(t1, t2))));
  const Fq2 c1 = f_add(f_sub(f_mul(f_add(a.c0, a.c1), f_add(b.c0, b.c1)), f_add(t0, t1)), fq2_mul_xi(t2));
  const Fq2 c2 = f_add(f_sub(f_mul(f_add(a.c0, a.c2), f_add(b.c0, b.c2)), f_add(t0, t2)), t1);
  return {c0, c1, c2};
}
inline Fq6 f6_mul_v(const Fq6& a) { return {fq2_mul_xi(a.c2), a.c0, a.c1}; }

inline Fq12 f12_one() { return {f6_one(), f6_zero()}; }
inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
  const Fq6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
  return {f6_add(t0, f6_mul_v(t1)), f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), f6_add(t0, t1))};
}
inline bool f12_is_one(const Fq12& a) {
  const Fq2 z = Fq2::zero();
  return fq2_eq(a.c0.c0, Fq2::one()) && fq2_eq(a.c0.c1, z) && fq2_eq(a.c0.c2, z) && fq2_eq(a.c1.c0, z) && fq2_eq(a.c1.c1, z) &&
         fq2_eq(a.c1.c2, z);
}

// (p^12 - 1) / r, little-endian 64-bit words (2790 bits)
const uint64_t FINAL_EXP[44] = {
    0x86964b64ca86f120ull, 0x40a4efb7e54523a4ull, 0x837fa97896e84abbull, 0x361102b6b9b2b918ull,
    0xc0de81def35692daull, 0xbe04c7e8a6c3c760ull, 0xd766f9c9d570bb7full, 0xc230974d83561841ull,
    0x5bba1668c3be69a3ull, 0x7f3811c410526294ull, 0x29baee7ddadda71cull, 0xbf813b8d145da900ull,
    0x641bbadf423f9a2cull, 0xa80bb4ea44eacc5eull, 0xcd65664814fde37cull, 0x4a0364b9580291d2ull,
    0xee93dfb10826f0ddull, 0x6b42db8dc5514724ull, 0xbb10cf430b0f3785ull, 0x40494e406f804216ull,
    0x55cfe107acf3aafbull, 0x2088ec80e0ebae87ull, 0x846a3ed011a337a0ull, 0x48a45a4a1e3a5195ull,
    0xe5664568dfc50e16ull, 0xab6a41294c0cc4ebull, 0x82d0d602d268c7daull, 0x6668449aed3cc48aull,
    0x5062cd0fb2015dfcull, 0x7f2940a8b1ddb3d1ull, 0x77f5b63a2a226448ull, 0xfef0781361e443aeull,
    0xf977870e88d5c6c8ull, 0x790364a61f676baaull, 0x5887e72eceaddea3ull, 0x1377e563a09a1b70ull,
    0x0c54efee1bd8c3b2ull, 0x3ec3d15ad524d8f7ull, 0xdaf15466b2383a5dull, 0xe1e30a73bb94fec0ull,
    0x6a1c71015f3f7be2ull, 0x842d43bf6369b1ffull, 0x20fddadf107d20bcull, 0x0000002f4b6dc970ull};
// Frobenius constants xi^((p-1)/3), xi^((p-1)/2), xi^((p^2-1)/3), xi^((p^2-1)/2): canonical c0 || c1
static const uint8_t G12[64] = {0x3d, 0x55, 0x6f, 0x17, 0x57, 0x95, 0xe3, 0x99, 0x0c, 0x33, 0xc3, 0xc2, 0x10, 0xc3, 0x8c, 0xb7, 0x43, 0xb1, 0x59, 0xf5, 0x3c, 0xec, 0x0b, 0x4c, 0xf7, 0x11, 0x79, 0x4f, 0x98, 0x47, 0xb3, 0x2f,
    0xa2, 0xcb, 0x0f, 0x64, 0x1c, 0xd5, 0x65, 0x16, 0xce, 0x9d, 0x7c, 0x0b, 0x1d, 0x2a, 0xae, 0x32, 0x94, 0x07, 0x5a, 0xd7, 0x8b, 0xcc, 0xa4, 0x4b, 0x20, 0xae, 0xeb, 0x61, 0x50, 0xe5, 0xc9, 0x16};
static const uint8_t G13[64] = {0x5a, 0x13, 0xa0, 0x71, 0x46, 0x01, 0x54, 0xdc, 0x98, 0x59, 0xc9, 0xa9, 0xed, 0xe0, 0xaa, 0xdb, 0xb9, 0xf9, 0xe2, 0xb6, 0x98, 0xc6, 0x5e, 0xdc, 0xdc, 0xf5, 0x9a, 0x48, 0x05, 0xf3, 0x3c, 0x06,
    0xe3, 0xb0, 0x23, 0x26, 0x63, 0x7f, 0xd3, 0x82, 0xd2, 0x5b, 0xa2, 0x8f, 0xc9, 0x7d, 0x80, 0x21, 0x2b, 0x6f, 0x79, 0xec, 0xa7, 0xb5, 0x04, 0x07, 0x9a, 0x04, 0x41, 0xac, 0xbc, 0x3c, 0xc0, 0x07};
static const uint8_t G22[64] = {0x48, 0xfd, 0x7c, 0x60, 0xe5, 0x44, 0xbd, 0xe4, 0x3d, 0x6e, 0x96, 0xbb, 0x9f, 0x06, 0x8f, 0xc2, 0xb0, 0xcc, 0xac, 0xe0, 0xe7, 0xd9, 0x6d, 0x5e, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30,
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00};
static const uint8_t G23[64] = {0x46, 0xfd, 0x7c, 0xd8, 0x16, 0x8c, 0x20, 0x3c, 0x8d, 0xca, 0x71, 0x68, 0x91, 0x6a, 0x81, 0x97, 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30,
    0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00};
const uint64_t ATE_LOOP_LO = 0x9d797039be763ba8ull;  // 6x + 2 = 0x1_9d797039be763ba8 (65 bits)

// caller pointers (vk, proof, public inputs) have no alignment guarantee: go through memcpy, not fe_load's uint4 reads
template <class M>
Fe<M> ld_any(const uint8_t* p) {
  uint32_t w[8];
  memcpy(w, p, 32);
  return fe_from_words<M>(w);
}
Fq2 fq2_from_bytes(const uint8_t* p) { return {fe_to_mont(ld_any<FqParams>(p)), fe_to_mont(ld_any<FqParams>(p + 32))}; }

Fq12 final_exponentiation(const Fq12& f) {
  Fq12 r = f12_one();
  for (int i = 2789; i >= 0; i--) {
    r = f12_mul(r, r);
    if ((FINAL_EXP[i >> 6] >> (i & 63)) & 1) r = f12_mul(r, f);
  }
  return r;
}

struct G2A { Fq2 x, y; };  // affine twist point (never infinity inside the loop)
struct G1A { Fq x, y; };

// line through twist point t with slope lam, evaluated at the G1 point p
Fq12 line(const G2A& t, const Fq2& lam, const G1A& p) {
  Fq12 l;
  l.c0 = {Fq2{p.y, Fq::zero()}, Fq2::zero(), Fq2::zero()};
  l.c1 = {f_neg(fq2_scale(lam, p.x)), f_sub(f_mul(lam, t.x), t.y), Fq2::zero()};
  return l;
}

// t <- t + q (doubling when equal) and the line value; returns false for the degenerate vertical case
bool step(G2A& t, const G2A& q, const G1A& p, Fq12& l) {
  Fq2 lam;
  if (fq2_eq(t.x, q.x)) {
    if (!fq2_eq(t.y, q.y)) return false;
    const Fq2 xx = f_sqr(t.x);
    lam = f_mul(f_add(f_dbl(xx), xx), f_inv(f_dbl(t.y)));
  } else {
    lam = f_mul(f_sub(q.y, t.y), f_inv(f_sub(q.x, t.x)));
  }
  const Fq2 x3 = f_sub(f_sub(f_sqr(lam), t.x), q.x);
  const Fq2 y3 = f_sub(f_mul(lam, f_sub(t.x, x3)), t.y);
  l = line(t, lam, p);
  t = {x3, y3};
  return true;
}

bool miller_loop(const G1A& p, const G2A& q, Fq12& f_out) {
  Fq12 f = f12_one(), l;
  G2A t = q;
  for (int i = 63; i >= 0; i--) {  // bits below the leading one of 6x + 2
    if (!step(t, t, p, l)) return false;
    f = f12_mul(f12_mul(f, f), l);
    if ((ATE_LOOP_LO >> i) & 1) {
      if (!step(t, q, p, l)) return false;
      f = f12_mul(f, l);
    }
  }
  const G2A q1 = {f_mul(fq2_conj(q.x), fq2_from_bytes(G12)), f_mul(fq2_conj(q.y), fq2_from_bytes(G13))};
  const G2A q2 = {f_mul(q.x, fq2_from_bytes(G22)), f_neg(f_mul(q.y, fq2_from_bytes(G23)))};
  if (!step(t, q1, p, l)) return false;
  f = f12_mul(f, l);
  if (!step(t, q2, p, l)) return false;
  f_out = f12_mul(f, l);
  return true;
}

// canonical byte checks + curve membership
bool limbs_lt_modulus(const uint8_t* p32, const uint32_t N29[9]) {
  const Fq v = ld_any<FqParams>(p32);  // only the limb split is used
  for (int i = 8; i >= 0; i--) {
    if (v.l[i] != N29[i]) return v.l[i] < N29[i];
  }
  return false;
}
bool g1_decode(const uint8_t* b, G1A& out, bool& inf) {
  if (!limbs_lt_modulus(b, FqParams::N) || !limbs_lt_modulus(b + 32, FqParams::N)) return false;
  out = {fe_to_mont(ld_any<FqParams>(b)), fe_to_mont(ld_any<FqParams>(b + 32))};
  inf = out.x.is_zero() && out.y.is_zero();
  if (inf) return true;
  const Fq rhs = fe_add(fe_mul(fe_sqr(out.x), out.x), fq_from_u32(3));
  return fe_sqr(out.y) == rhs;
}
bool g2_decode(const uint8_t* b, G2A& out, bool& inf) {
  for (int k = 0; k < 4; k++)
    if (!limbs_lt_modulus(b + 32 * k, FqParams::N)) return false;
  out = {fq2_from_bytes(b), fq2_from_bytes(b + 64)};
  inf = out.x.is_zero() && out.y.is_zero();
  if (inf) return true;
  // y^2 = x^3 + 3/(9+u)
  const Fq2 xi = {fq_from_u32(9), Fq::one()};
  const Fq2 bt = fq2_scale(f_inv(xi), fq_from_u32(3));
  if (!fq2_eq(f_sqr(out.y), f_add(f_mul(f_sqr(out.x), out.x), bt))) return false;
  // subgroup check: r * Q = infinity (the twist has a large cofactor)
  XYZZ<Fq2> acc = XYZZ<Fq2>::inf();
  const Affine<Fq2> qa = {out.x, out.y};
  for (int i = 8; i >= 0; i--)
    for (int bit = 28; bit >= 0; bit--) {
      acc = xyzz_dbl(acc);
      if ((FrParams::N[i] >> bit) & 1) acc = xyzz_madd(acc, qa);
    }
  return acc.is_inf();
}

}  // namespace

// vk blob: "OWVK0001" | u64 n_pub | alpha_g1 (64) | beta_g2 (128) | gamma_g2 (128) | delta_g2 (128) | IC ((n_pub+1) x 64)
int verify_cpu(const uint8_t* vk, size_t vk_len, const uint8_t* pub, size_t n_pub, const uint8_t* proof, int* ok) {
  *ok = 0;
  OG_REQUIRE(vk_len >= 16 + 64 + 3 * 128 && memcmp(vk, "OWVK0001", 8) == 0, "og_verify: bad verifying key (want OWVK0001)");
  uint64_t n_vk;
  memcpy(&n_vk, vk + 8, 8);
  OG_REQUIRE(n_vk == n_pub, "og_verify: number of public inputs does not match the verifying key");
  OG_REQUIRE(n_pub <= ((size_t)1 << 24), "og_verify: too many public inputs");  // also keeps (n_pub + 1) * 64 from wrapping
  OG_REQUIRE(vk_len == 16 + 64 + 3 * 128 + (n_pub + 1) * 64, "og_verify: verifying key length does not match its header");
  const uint8_t *alpha_b = vk + 16, *beta_b = alpha_b + 64, *gamma_b = beta_b + 128, *delta_b = gamma_b + 128, *ic_b = delta_b + 128;
  G1A alpha, A, Cc, icp;
  G2A beta, gamma, delta, B;
  bool inf;
  OG_REQUIRE(g1_decode(alpha_b, alpha, inf) && !inf && g2_decode(beta_b, beta, inf) && !inf && g2_decode(gamma_b, gamma, inf) && !inf &&
                 g2_decode(delta_b, delta, inf) && !inf,
             "og_verify: verifying key holds an invalid point");
  // proof points: invalid encodings / off-curve / wrong subgroup / infinity => reject (ok = 0), not an error
  bool ia, ib, ic;
  if (!g1_decode(proof, A, ia) || !g2_decode(proof + 64, B, ib) || !g1_decode(proof + 192, Cc, ic)) return OG_OK;
  if (ia || ib || ic) return OG_OK;
  // vk_x = IC_0 + sum x_i IC_i   (public inputs must be canonical Fr elements)
  XYZZ<Fq> vkx = XYZZ<Fq>::inf();
  for (size_t i = 0; i <= n_pub; i++) {
    OG_REQUIRE(g1_decode(ic_b + 64 * i, icp, inf), "og_verify: verifying key holds an invalid IC point");
    // a public input >= r is a reject whatever its IC base is (checked BEFORE the infinity shortcut: an input whose base is
    // the point at infinity contributes nothing, but it still has to be a field element)
    const uint8_t* xb = i ? pub + 32 * (i - 1) : nullptr;
    if (xb && !limbs_lt_modulus(xb, FrParams::N)) return OG_OK;
    if (inf) continue;
    const Affine<Fq> p = {icp.x, icp.y};
    if (i == 0) {
      vkx = xyzz_madd(vkx, p);
      continue;
    }
    const Fr x = ld_any<FrParams>(xb);
    XYZZ<Fq> t = XYZZ<Fq>::inf();
    for (int w = 8; w >= 0; w--)
      for (int bit = 28; bit >= 0; bit--) {
        t = xyzz_dbl(t);
        if ((x.l[w] >> bit) & 1) t = xyzz_madd(t, p);
      }
    vkx = xyzz_add(vkx, t);
  }
  Fq12 f = f12_one(), m;
  const G1A negA = {A.x, fe_neg(A.y)};
  if (!miller_loop(negA, B, m)) return OG_OK;
  f = f12_mul(f, m);
  if (!miller_loop(alpha, beta, m)) return OG_OK;
  f = f12_mul(f, m);
  if (!vkx.is_inf()) {
    const Affine<Fq> v = xyzz_to_affine(vkx);
    if (!miller_loop({v.x, v.y}, gamma, m)) return OG_OK;
    f = f12_mul(f, m);
  }
  if (!miller_loop(Cc, delta, m)) return OG_OK;
  f = f12_mul(f, m);
  *ok = f12_is_one(final_exponentiation(f)) ? 1 : 0;
  return OG_OK;
}

}  // namespace og
