// Scalar multiplication + Groth16 proof assembly kernels, templated on the group and instantiated in
// msm_g1.hip / msm_g2.hip next to the MSM.  No reference counterpart.
//   A = alpha + sum z_i A_i + r delta
//   B = beta  + sum z_i B_i + s delta                      (G2; the G1 copy only feeds C)
//   C = sum z_i L_i + sum h_j H_j + s A + r B1 - r s delta
//     = L + H + s (alpha + Am) + r (beta1 + B1m) + (r s) delta1        (expanded, 4 independent muls)
#pragma once
#include "ec.hip.h"
#include "ctx.h"
#include "field_w9.hip.h"

namespace og {

struct Scalar256 {
  uint32_t l[8];
};

__device__ __forceinline__ Scalar256 scalar_load(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

__device__ __forceinline__ bool scalar_bit(const Scalar256& k, int i) { return (k.l[i >> 5] >> (i & 31)) & 1; }

// Fixed-base table of one point: tab[w * 16 + d] = d * 2^(4w) * base (affine Montgomery; d = 0 is the point at
// infinity), 64 windows of 4 bits.  Turns the 254 doublings + ~127 additions of k * base into <= 64 mixed additions:
// key generation multiplies the generators by one scalar per wire (keygen.hip), and the G2 blinding term s * delta2
// of every proof reads the key's own table (k_assemble_g2; there the chain is the single-proof latency of the step).
template <class T>
__global__ void __launch_bounds__(64) k_fixed_table(const uint8_t* __restrict__ base, uint8_t* __restrict__ tab) {
  OG_FILLER_PRIO();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 64 * 16) return;
  const int w = t >> 4, d = t & 15;
  XYZZ<T> pw = XYZZ<T>::from_affine(Affine<T>::load(base));
#pragma unroll 1
  for (int i = 0; i < 4 * w; i++) pw = xyzz_dbl(pw);
  XYZZ<T> r = XYZZ<T>::inf();
#pragma unroll 1
  for (int s = 0; s < 8; s++) {  // even s: r = 2 r (through the add site's equal-operand path); odd s: r += pw if the bit is set
    const int bit = 3 - (s >> 1);
    if ((s & 1) && !((d >> bit) & 1)) continue;
    r = xyzz_add(r, (s & 1) ? pw : r);
  }
  xyzz_to_affine(r).store(tab + (size_t)t * Affine<T>::BYTES);
}

// out[i] = k_i * base through the 64 x 16 table above; out canonical affine.  A lane carries SM_PER products and
// inverts once for all of them (Montgomery's trick on w = ZZ * ZZZ: x = X ZZZ / w, y = Y ZZ / w, the numerators parked
// in the output slots meanwhile), so the ~380-multiplication Fermat inversion is shared SM_PER ways.
constexpr int SM_PER = 8;
template <class T>
__global__ void __launch_bounds__(64) k_scalar_mul_fixed(const uint8_t* __restrict__ tab, const uint8_t* __restrict__ scalars,
                                                        size_t n, uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (size_t)gridDim.x * blockDim.x;
  if (lane >= n) return;
  T w[SM_PER], pre[SM_PER];
  T run = T::one();
  int cnt = 0;
#pragma unroll 1
  for (; cnt < SM_PER && lane + cnt * lanes < n; cnt++) {
    const size_t i = lane + cnt * lanes;
    const Scalar256 k = scalar_load(scalars + i * 32);
    XYZZ<T> acc = XYZZ<T>::inf();
#pragma unroll 1
    for (int j = 0; j < 64; j++) {
      const uint32_t d = (k.l[j >> 3] >> ((j & 7) * 4)) & 15u;
      if (d == 0) continue;
      acc = xyzz_madd(acc, Affine<T>::load(tab + (size_t)(j * 16 + d) * Affine<T>::BYTES));
    }
    const bool inf = acc.is_inf();
    Affine<T> num = {f_mul(acc.x, acc.zzz), f_mul(acc.y, acc.zz)};
    if (inf) num = Affine<T>::inf();
    num.store(out + i * Affine<T>::BYTES);
    w[cnt] = inf ? T::one() : f_mul(acc.zz, acc.zzz);
    pre[cnt] = run;
    run = f_mul(run, w[cnt]);
  }
  T inv = f_inv(run);
#pragma unroll 1
  for (int j = cnt - 1; j >= 0; j--) {
    const size_t i = lane + j * lanes;
    const T ij = f_mul(inv, pre[j]);
    inv = f_mul(inv, w[j]);
    Affine<T> a = Affine<T>::load(out + i * Affine<T>::BYTES);
    a.x = FieldIO<T>::from_mont(f_mul(a.x, ij));
    a.y = FieldIO<T>::from_mont(f_mul(a.y, ij));
    a.store(out + i * Affine<T>::BYTES);
  }
}

template <class T>
int fixed_table_t(og_ctx* ctx, const uint8_t* base_mont_d, uint8_t* tab_d) {
  hipLaunchKernelGGL(k_fixed_table<T>, dim3(16), dim3(64), 0, ctx->stream, base_mont_d, tab_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// tab_d: the base's 64 x 16 table (fixed_table_t)
template <class T>
int scalar_mul_fixed_t(og_ctx* ctx, const uint8_t* tab_d, const uint8_t* scalars_d, size_t n, uint8_t* out_d) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_scalar_mul_fixed<T>, dim3(grid_for((n + SM_PER - 1) / SM_PER, 64)), dim3(64), 0, ctx->stream, tab_d, scalars_d, n,
                     out_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// ---- proof assembly -----------------------------------------------------------------
// rs: n x 64 B canonical (r || s).  consts: G1 {alpha, beta, delta} / G2 {beta, delta}, affine Montgomery.

#ifdef OG_ECMUL_G1
// G1 step 1: tmp[g][j], j < 4 = { r delta, (r s) delta, s (alpha + Am), r (beta + B1m) }
__global__ void __launch_bounds__(64) k_assemble_g1_muls(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ rs,
                                                        const uint8_t* __restrict__ res_a, const uint8_t* __restrict__ res_b1,
                                                        size_t n, uint8_t* __restrict__ tmp) {
  OG_FILLER_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 4) return;
  const size_t g = t >> 2;
  const int j = (int)(t & 3);
  const Scalar256 r = scalar_load(rs + g * 64), s = scalar_load(rs + g * 64 + 32);
  const G1Affine delta = G1Affine::load(consts + 128);
  G1XYZZ p = G1XYZZ::from_affine(delta);
  Scalar256 k = r;
  if (j == 1) {
    const Fr rf = fe_from_words<FrParams>(r.l), sf = fe_from_words<FrParams>(s.l);
    const Fr prod = fe_from_mont(fe_mul(fe_to_mont(rf), fe_to_mont(sf)));  // canonical r s mod the group order
    fe_to_words(k.l, prod);
  }
  if (j >= 2) {
    p = G1XYZZ::load((j == 2 ? res_a : res_b1) + g * G1XYZZ::BYTES);
    p = xyzz_madd(p, G1Affine::load(consts + (j == 2 ? 0 : 64)));
    if (j == 2) k = s;
  }
  // 4-bit windows over a table of 1 p .. 15 p the lane keeps in the scratch behind tmp (n x 4 x 16 points): 1 doubling +
  // 13 additions for the table, then 63 x (4 doublings + an addition): ~3400 field multiplications instead of the ~5800 of
  // bit-by-bit double-and-add -- the four lanes of a proof share a wave, so an "if (bit)" addition was executed on nearly
  // every bit anyway.  This chain is a proof's last 2.7 ms (one request: profiles/r03_latency.json).
  uint8_t* tab = tmp + n * 4 * G1XYZZ::BYTES + t * 16 * G1XYZZ::BYTES;
  {
    G1XYZZ q = p;
    q.store(tab + G1XYZZ::BYTES);
#pragma unroll 1
    for (int d = 2; d < 16; d++) {
      q = d == 2 ? xyzz_dbl(p) : xyzz_add(q, p);
      q.store(tab + (size_t)d * G1XYZZ::BYTES);
    }
  }
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int w = 63; w >= 0; w--) {
    if (w != 63) {
#pragma unroll 1
      for (int e = 0; e < 4; e++) acc = xyzz_dbl(acc);
    }
    const uint32_t dgt = (k.l[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (dgt) acc = xyzz_add(acc, G1XYZZ::load(tab + (size_t)dgt * G1XYZZ::BYTES));
  }
  acc.store(tmp + t * G1XYZZ::BYTES);
}

// The same four products by EIGHT lanes per proof, each a half-length chain (round 4, latency-bound calls): the host hands over
// k = k1 + lambda k2 (glv.h: |k1|, |k2| < 2^127, signs in bit 127) for the four scalars r, r s, s, r, and lane 2 j + h walks
// |k_h| over P_j (h = 0) or phi(P_j) = (beta x, y) (h = 1), negated if k_h is negative: 32 windows of 4 bits instead of 64.
// tmp[g][2 j + h]; the finish kernel adds the halves.
// `beta`: the cube root of unity of Fq that goes with glv.h's lambda -- glv::BETA itself, passed by the launcher (ONE definition;
// groth16.hip checks once per process that (beta x_G, y_G) = [lambda] G and turns the GLV path off otherwise).
struct GlvBetaWords { uint32_t w[8]; };
__global__ void __launch_bounds__(64) k_assemble_g1_muls_glv(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ glv,
                                                            const uint8_t* __restrict__ res_a, const uint8_t* __restrict__ res_b1,
                                                            size_t n, uint8_t* __restrict__ tmp, GlvBetaWords beta) {
  OG_FILLER_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  const size_t g = t >> 3;
  const int j = (int)((t >> 1) & 3), h = (int)(t & 1);
  const uint4 kq = *reinterpret_cast<const uint4*>(glv + (g * 4 + j) * 32 + h * 16);
  uint32_t k[4] = {kq.x, kq.y, kq.z, kq.w};
  const bool neg = (k[3] >> 31) != 0;
  k[3] &= 0x7fffffffu;
  G1XYZZ p = G1XYZZ::from_affine(G1Affine::load(consts + 128));  // delta (j = 0: r delta, j = 1: (r s) delta)
  if (j >= 2) {
    p = G1XYZZ::load((j == 2 ? res_a : res_b1) + g * G1XYZZ::BYTES);
    p = xyzz_madd(p, G1Affine::load(consts + (j == 2 ? 0 : 64)));  // alpha + Am | beta + B1m
  }
  if (h) {  // phi: x -> beta x, i.e. X -> beta X in XYZZ coordinates
    p.x = fe_mul(p.x, fe_to_mont(fe_from_words<FqParams>(beta.w)));
  }
  if (neg) p = xyzz_neg(p);
  uint8_t* tab = tmp + n * 8 * G1XYZZ::BYTES + t * 16 * G1XYZZ::BYTES;
  {
    G1XYZZ q = p;
    q.store(tab + G1XYZZ::BYTES);
#pragma unroll 1
    for (int d = 2; d < 16; d++) {
      q = d == 2 ? xyzz_dbl(p) : xyzz_add(q, p);
      q.store(tab + (size_t)d * G1XYZZ::BYTES);
    }
  }
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int w = 31; w >= 0; w--) {
    if (w != 31) {
#pragma unroll 1
      for (int e = 0; e < 4; e++) acc = xyzz_dbl(acc);
    }
    const uint32_t dgt = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (dgt) acc = xyzz_add(acc, G1XYZZ::load(tab + (size_t)dgt * G1XYZZ::BYTES));
  }
  acc.store(tmp + t * G1XYZZ::BYTES);
}

// G1 step 2: proof[g][0:64] = A = alpha + Am + tmp0, proof[g][192:256] = C = L + H + tmp2 + tmp3 + tmp1
// (halves = 2: the eight-lane form above left every product as two halves, tmp[g][2 j], tmp[g][2 j + 1])
__global__ void __launch_bounds__(64) k_assemble_g1_finish(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ res_a,
                                                          const uint8_t* __restrict__ res_l, const uint8_t* __restrict__ res_h,
                                                          const uint8_t* __restrict__ tmp, size_t n, uint8_t* __restrict__ proofs, int halves) {
  OG_FILLER_PRIO();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* tg = tmp + g * 4 * halves * G1XYZZ::BYTES;
  G1XYZZ A = xyzz_madd(G1XYZZ::load(res_a + g * G1XYZZ::BYTES), G1Affine::load(consts));
  G1XYZZ Cc = G1XYZZ::load(res_l + g * G1XYZZ::BYTES);
#pragma unroll 1
  for (int s = 0; s < 1 + 4 * halves; s++) {  // one add site: C += H | then product j = (s - 1) / halves: A += tmp0.. | C += tmp1.., tmp2.., tmp3..
    const int j = (s - 1) / halves;
    const bool to_a = s >= 1 && j == 0;
    const uint8_t* ptr = s == 0 ? res_h + g * G1XYZZ::BYTES : tg + (size_t)(s - 1) * G1XYZZ::BYTES;
    const G1XYZZ res = xyzz_add(to_a ? A : Cc, G1XYZZ::load(ptr));
    if (to_a) A = res; else Cc = res;
  }
#pragma unroll 1
  for (int q = 0; q < 2; q++) {  // one to_affine site
    G1Affine a = xyzz_to_affine(q ? Cc : A);
    a.x = fe_from_mont(a.x);
    a.y = fe_from_mont(a.y);
    a.store(proofs + g * 256 + (q ? 192 : 0));
  }
}
// ---- the four products by a WAVE per half-length chain, the group law in the wave-wide form (calls of a handful of requests) ----------
// k_assemble_g1_muls_glv is 127 dependent doublings + ~46 additions per lane, ~1 670 lane-local Fq products of 205 instructions: 0.97
// ms on a lone wave whatever runs beside it, and the last chain a single request waits for.  Here one WAVE walks
// that chain with every coordinate spread over nine lanes (field_w9.hip.h): a product is ~90 instructions instead of 205, sums
// and differences are one instruction.  Strict products (29-bit digit: output < a b / (169 N) + N whatever the operands) keep the
// bounds closed without any extra reduction -- in multiples of N, coordinates between operations X < 9.3, Y < 5.7, ZZ, ZZZ < 1.2:
//   doubling (dbl-2008-s-1, 9 products):  U = 2Y < 11.4;  V = U^2 < 1.8;  W = U V < 1.2;  S = X V < 1.2;  M = 3 X^2 < 6.2;
//     X3 = M^2 - 2S + 8N < 9.3;  D = S - X3 + 16N < 17.2;  Y3 = M D - W Y + 4N < 5.7
//   addition (add-2008-s, 14 products):   U1, S1, U2, S2 < 1.2;  P = U2 - U1 + 4N, R = S2 - S1 + 4N < 5.2;  PP, PPP, Q, RR < 1.2;
//     X3 = RR - (PPP + 2Q) + 8N < 9.3;  D = Q - X3 + 16N < 17.2;  Y3 = R D - S1 PPP + 4N < 5.6
// Every difference adds K N in the borrow-free form (w9_kn_limb: K - 2 above the subtrahend's bound) and is carried before it is
// used again.  No equal / opposite operands can meet: the base has prime order r, the walked scalar is below 2^127 and the
// window digit below 16 (16 m = d mod r has no solution with 0 < m < 2^123, 0 < d < 16); infinity (the accumulator before the first
// non-zero digit, a base that is the point at infinity, a half that is zero) is a wave-uniform flag.  Results leave through a
// strict product by one (< 2 N) in the lane-local format k_assemble_g1_finish / _early / _late read.  tmp[g][2 j + h] as the GLV form.
struct W9Pt {
  uint32_t x, y, zz, zzz;
};
__device__ __forceinline__ uint32_t w9q_mul(const U9& a, uint32_t b, uint32_t nj) { return w9_mul<FqParams, true, false>(a, b, nj); }
__device__ __forceinline__ W9Pt w9_xyzz_dbl(const W9Pt& p, uint32_t nj, int lane) {
  const uint32_t c4 = w9_kn_limb<FqParams, 4>(lane), c8 = w9_kn_limb<FqParams, 8>(lane), c16 = w9_kn_limb<FqParams, 16>(lane);
  const uint32_t U = 2u * p.y;
  const U9 Ua = w9_gather(U);
  const uint32_t V = w9q_mul(Ua, U, nj), W = w9q_mul(Ua, V, nj);
  const U9 Xa = w9_gather(p.x);
  const uint32_t S = w9q_mul(Xa, V, nj), M = 3u * w9q_mul(Xa, p.x, nj);
  const U9 Ma = w9_gather(M);
  const uint32_t X3 = w9_carry(w9_sub(w9q_mul(Ma, M, nj), w9_carry(2u * S, lane), c8), lane);
  const uint32_t D = w9_carry(w9_sub(S, X3, c16), lane);
  const U9 Wa = w9_gather(W);
  const uint32_t Y3 = w9_carry(w9_sub(w9q_mul(Ma, D, nj), w9q_mul(Wa, p.y, nj), c4), lane);
  return {X3, Y3, w9q_mul(w9_gather(V), p.zz, nj), w9q_mul(Wa, p.zzz, nj)};
}
__device__ __forceinline__ W9Pt w9_xyzz_add(const W9Pt& a, const W9Pt& b, uint32_t nj, int lane) {
  const uint32_t c4 = w9_kn_limb<FqParams, 4>(lane), c8 = w9_kn_limb<FqParams, 8>(lane), c16 = w9_kn_limb<FqParams, 16>(lane);
  const U9 zz2 = w9_gather(b.zz), zzz2 = w9_gather(b.zzz), zz1 = w9_gather(a.zz), zzz1 = w9_gather(a.zzz);
  const uint32_t U1 = w9q_mul(zz2, a.x, nj), S1 = w9q_mul(zzz2, a.y, nj);
  const uint32_t P = w9_carry(w9_sub(w9q_mul(zz1, b.x, nj), U1, c4), lane), R = w9_carry(w9_sub(w9q_mul(zzz1, b.y, nj), S1, c4), lane);
  const U9 Pa = w9_gather(P), Ra = w9_gather(R);
  const uint32_t PP = w9q_mul(Pa, P, nj);
  const U9 PPa = w9_gather(PP);
  const uint32_t PPP = w9q_mul(PPa, P, nj), Q = w9q_mul(PPa, U1, nj);
  const uint32_t X3 = w9_carry(w9_sub(w9q_mul(Ra, R, nj), w9_carry(PPP + 2u * Q, lane), c8), lane);
  const uint32_t D = w9_carry(w9_sub(Q, X3, c16), lane);
  const U9 PPPa = w9_gather(PPP);
  const uint32_t Y3 = w9_carry(w9_sub(w9q_mul(Ra, D, nj), w9q_mul(PPPa, S1, nj), c4), lane);
  return {X3, Y3, w9q_mul(PPa, w9q_mul(zz1, b.zz, nj), nj), w9q_mul(PPPa, w9q_mul(zzz1, b.zzz, nj), nj)};
}
__global__ void __launch_bounds__(64) k_assemble_g1_muls_w9(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ glv,
                                                           const uint8_t* __restrict__ res_a, const uint8_t* __restrict__ res_b1,
                                                           size_t n, uint8_t* __restrict__ tmp, GlvBetaWords beta) {
  OG_FILLER_PRIO();
  __shared__ uint32_t tab[16 * 4 * 16];  // d P for d = 1 .. 15: four coordinates of nine limbs (lanes 9 .. 15: zeros)
  const size_t t = blockIdx.x;
  if (t >= n * 8) return;
  const int lane = threadIdx.x;
  const size_t g = t >> 3;
  const int j = (int)((t >> 1) & 3), h = (int)(t & 1);
  // the chain's base and scalar, lane-local, exactly as k_assemble_g1_muls_glv forms them (every lane the same)
  const uint4 kq = *reinterpret_cast<const uint4*>(glv + (g * 4 + j) * 32 + h * 16);
  uint32_t k[4] = {kq.x, kq.y, kq.z, kq.w};
  const bool neg = (k[3] >> 31) != 0;
  k[3] &= 0x7fffffffu;
  G1XYZZ p = G1XYZZ::from_affine(G1Affine::load(consts + 128));
  if (j >= 2) {
    p = G1XYZZ::load((j == 2 ? res_a : res_b1) + g * G1XYZZ::BYTES);
    p = xyzz_madd(p, G1Affine::load(consts + (j == 2 ? 0 : 64)));
  }
  if (h) p.x = fe_mul(p.x, fe_to_mont(fe_from_words<FqParams>(beta.w)));
  if (neg) p = xyzz_neg(p);
  uint8_t* out = tmp + t * G1XYZZ::BYTES;
  if (p.is_inf() || (k[0] | k[1] | k[2] | k[3]) == 0) {  // (wave-uniform)
    if (lane == 0) G1XYZZ::inf().store(out);
    return;
  }
  const uint32_t nj = w9_modulus_limb<FqParams>(lane);
  auto put = [&](int d, const W9Pt& q) {
    if (lane < 16) {
      tab[(d * 4 + 0) * 16 + lane] = q.x; tab[(d * 4 + 1) * 16 + lane] = q.y;
      tab[(d * 4 + 2) * 16 + lane] = q.zz; tab[(d * 4 + 3) * 16 + lane] = q.zzz;
    }
  };
  auto get = [&](int d) -> W9Pt {
    const int l = lane & 15;
    const bool on = lane < 16;
    return {on ? tab[(d * 4 + 0) * 16 + l] : 0u, on ? tab[(d * 4 + 1) * 16 + l] : 0u, on ? tab[(d * 4 + 2) * 16 + l] : 0u,
            on ? tab[(d * 4 + 3) * 16 + l] : 0u};
  };
  const W9Pt base = {w9_spread(p.x, lane), w9_spread(p.y, lane), w9_spread(p.zz, lane), w9_spread(p.zzz, lane)};
  put(1, base);
  W9Pt q = base;
#pragma unroll 1
  for (int d = 2; d < 16; d++) {  // 2 P by doubling, then + P
    q = d == 2 ? w9_xyzz_dbl(base, nj, lane) : w9_xyzz_add(q, base, nj, lane);
    put(d, q);
  }
  __syncthreads();
  W9Pt acc = base;
  bool acc_inf = true;
#pragma unroll 1
  for (int w = 31; w >= 0; w--) {
    if (!acc_inf) {
#pragma unroll 1
      for (int e = 0; e < 4; e++) acc = w9_xyzz_dbl(acc, nj, lane);
    }
    const uint32_t dgt = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (dgt == 0) continue;
    const W9Pt e = get((int)dgt);
    if (acc_inf) { acc = e; acc_inf = false; }
    else acc = w9_xyzz_add(acc, e, nj, lane);
  }
  // below 2 N and into the lane-local format: a strict product by one per coordinate
  const U9 one = w9_uniform(FqParams::ONE);
  const Fq ox = w9_collect<FqParams>(w9q_mul(one, acc.x, nj)), oy = w9_collect<FqParams>(w9q_mul(one, acc.y, nj));
  const Fq ozz = w9_collect<FqParams>(w9q_mul(one, acc.zz, nj)), ozzz = w9_collect<FqParams>(w9q_mul(one, acc.zzz, nj));
  if (lane == 0) {
    const G1XYZZ r = {fe_from_lazy_limbs<FqParams>(ox.l), fe_from_lazy_limbs<FqParams>(oy.l), fe_from_lazy_limbs<FqParams>(ozz.l),
                      fe_from_lazy_limbs<FqParams>(ozzz.l)};
    r.store(out);
  }
}

// The same sums in TWO parts, for a call whose queries fan out over the streams (groth16.hip `split`: one sub-batch).  The four
// products need A and B1 only: they run on a side stream while stream 0 is still in the quotient and the H query, and A's own
// sum and inversion follow them there (k_assemble_g1_early -> proof[g][0:64]).  Behind the H query there is C = L + H + the three
// products and one inversion left (k_assemble_g1_late, which waits for the products, not for A).
// The affine results are what k_assemble_g1_finish writes: a sum does not depend on its order.
__global__ void __launch_bounds__(64) k_assemble_g1_early(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ res_a,
                                                         const uint8_t* __restrict__ tmp, size_t n, uint8_t* __restrict__ proofs, int halves) {
  OG_FILLER_PRIO();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* tg = tmp + g * 4 * halves * G1XYZZ::BYTES;
  G1XYZZ A = xyzz_madd(G1XYZZ::load(res_a + g * G1XYZZ::BYTES), G1Affine::load(consts));
#pragma unroll 1
  for (int s = 0; s < halves; s++) A = xyzz_add(A, G1XYZZ::load(tg + (size_t)s * G1XYZZ::BYTES));  // + r delta
  G1Affine a = xyzz_to_affine(A);
  a.x = fe_from_mont(a.x);
  a.y = fe_from_mont(a.y);
  a.store(proofs + g * 256);
}
__global__ void __launch_bounds__(64) k_assemble_g1_late(const uint8_t* __restrict__ res_l, const uint8_t* __restrict__ res_h,
                                                        const uint8_t* __restrict__ tmp, size_t n, uint8_t* __restrict__ proofs, int halves) {
  OG_FILLER_PRIO();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* tg = tmp + g * 4 * halves * G1XYZZ::BYTES;
  G1XYZZ Cc = G1XYZZ::load(res_l + g * G1XYZZ::BYTES);
#pragma unroll 1
  for (int s = 0; s <= 3 * halves; s++)  // one add site: C += H, then the products (r s) delta, s (alpha + Am), r (beta + B1m)
    Cc = xyzz_add(Cc, G1XYZZ::load(s == 0 ? res_h + g * G1XYZZ::BYTES : tg + (size_t)(halves + s - 1) * G1XYZZ::BYTES));
  G1Affine a = xyzz_to_affine(Cc);
  a.x = fe_from_mont(a.x);
  a.y = fe_from_mont(a.y);
  a.store(proofs + g * 256 + 192);
}
#endif  // OG_ECMUL_G1

#ifdef OG_ECMUL_G2
// G2: proof[g][64:192] = B2m + (s delta2 + beta2), s delta2 from the fixed-base table
__global__ void __launch_bounds__(64) k_assemble_g2(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ fb_tab,
                                                   const uint8_t* __restrict__ rs, const uint8_t* __restrict__ res_b2, size_t n,
                                                   uint8_t* __restrict__ proofs) {
  OG_FILLER_PRIO();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const Scalar256 s = scalar_load(rs + g * 64 + 32);
  G2XYZZ acc = G2XYZZ::load(res_b2 + g * G2XYZZ::BYTES);
#pragma unroll 1
  for (int w = 0; w <= 64; w++) {  // w < 64: acc += tab[w][digit w of s]; w == 64: acc += beta2
    const uint32_t d = w < 64 ? (s.l[w >> 3] >> ((w & 7) * 4)) & 15u : 1u;
    if (d == 0) continue;
    acc = xyzz_madd(acc, G2Affine::load(w < 64 ? fb_tab + (size_t)(w * 16 + d) * G2Affine::BYTES : consts));
  }
  G2Affine b = xyzz_to_affine(acc);
  b.x = FieldIO<Fq2>::from_mont(b.x);
  b.y = FieldIO<Fq2>::from_mont(b.y);
  b.store(proofs + g * 256 + 64);
}
// The same sum by ONE WAVE per proof (calls of at most a thousand requests: the chain is what is waited for).  k_assemble_g2 walks 65
// dependent mixed additions in one lane; here lane w takes window w's table entry and six levels of a tree through the LDS add them
// up, B2m and beta2 join as two more levels of the same add site: 8 additions deep instead of 65 (one request: 0.95 -> ~0.35 ms of the
// B chain).  Same affine bytes: a sum does not depend on its order.
__global__ void __launch_bounds__(64) k_assemble_g2_tree(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ fb_tab,
                                                        const uint8_t* __restrict__ rs, const uint8_t* __restrict__ res_b2, size_t n,
                                                        uint8_t* __restrict__ proofs) {
  OG_FILLER_PRIO();
  __shared__ __align__(16) uint8_t ex[64 * G2XYZZ::BYTES];
  const size_t g = blockIdx.x;
  const int w = threadIdx.x;
  if (g >= n) return;
  const Scalar256 s = scalar_load(rs + g * 64 + 32);
  const uint32_t d = (s.l[w >> 3] >> ((w & 7) * 4)) & 15u;
  G2XYZZ acc = G2XYZZ::inf();
  if (d) acc = G2XYZZ::from_affine(G2Affine::load(fb_tab + (size_t)(w * 16 + d) * G2Affine::BYTES));
#pragma unroll 1
  for (int lvl = 0; lvl < 8; lvl++) {  // one add site: levels 0..5 the tree (lane w += lane w + 32 >> lvl), 6: + B2m, 7: + beta2
    acc.store(ex + (size_t)w * G2XYZZ::BYTES);
    __syncthreads();
    G2XYZZ other = G2XYZZ::load(ex + (size_t)((w + (32 >> (lvl < 6 ? lvl : 5))) & 63) * G2XYZZ::BYTES);
    if (lvl == 6) other = G2XYZZ::load(res_b2 + g * G2XYZZ::BYTES);
    if (lvl == 7) other = G2XYZZ::from_affine(G2Affine::load(consts));
    __syncthreads();
    acc = xyzz_add(acc, other);  // (only lane 0's sum is the proof's; the other lanes' sums are partial)
  }
  if (w != 0) return;
  G2Affine b = xyzz_to_affine(acc);
  b.x = FieldIO<Fq2>::from_mont(b.x);
  b.y = FieldIO<Fq2>::from_mont(b.y);
  b.store(proofs + g * 256 + 64);
}
#endif  // OG_ECMUL_G2

}  // namespace og
