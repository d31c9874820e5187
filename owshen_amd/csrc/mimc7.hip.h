// MiMC7 device primitives: circomlib convention, 91 rounds,
// x -> (x + k + c_i)^7, output + k; MultiMiMC7 = Miyaguchi-Preneel chaining.
// The reference has no MiMC7 (its only Fr "hash" is the placeholder product at
// /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:202-204).
// All values are Fr in Montgomery form; `consts` is the 91 x 8-limb Montgomery
// table in HBM, read with wave-uniform addresses (scalar loads).
#pragma once
#include "field.hip.h"
#include "field_w9.hip.h"

namespace og {

constexpr int MIMC7_ROUNDS = 91;

// The products of the lane-pair (latency-bound) forms.  -DOG_MIMC_LAT=1 builds them with fe_mul_lat / fe_sqr_lat (field.hip.h:
// 17 independent column accumulators, row-wise reduction -- the multiply-adds of one product no longer wait for each other).
// Measured in round 4 and NOT faster: one request's walk 12.6 -> 13.3 ms, the 2^20-leaf tree 8.83 -> 9.28 ms.  A lone wave
// issues a v_mad_u64_u32 every ~9.5 cycles whether or not it depends on the previous one (profiles/r02_probe_chains.json), so
// instruction-level parallelism inside one lane buys nothing; only more waves -- or splitting a product across lanes --
// would, and the carries of a split product cost about what it saves.  Default 0: the throughput products.
#ifndef OG_MIMC_LAT
#define OG_MIMC_LAT 0
#endif
#if OG_MIMC_LAT && !defined(OG_AB_HOOKS)
#error "-DOG_MIMC_LAT=1 is an A/B build: it needs -DOG_AB_HOOKS (the latency forms are not part of the shipped library)"
#endif
#if OG_MIMC_LAT
#define OG_MIMC_LAT_MUL(a, b) fe_mul_lat(a, b)
#define OG_MIMC_LAT_SQR(a) fe_sqr_lat(a)
#else
#define OG_MIMC_LAT_MUL(a, b) fe_mul(a, b)
#define OG_MIMC_LAT_SQR(a) fe_sqr(a)
#endif

__device__ __forceinline__ Fr mimc7_const(const uint32_t* __restrict__ consts, int i) {
  return fe_load<FrParams>(consts + i * 8);
}

// E_k(x)
__device__ __forceinline__ Fr mimc7_permute(const uint32_t* __restrict__ consts, Fr x, const Fr& k) {
  Fr r = x;
  for (int i = 0; i < MIMC7_ROUNDS; i++) {
    Fr t = fe_add3_weak(r, k, mimc7_const(consts, i));  // < 5N, only ever multiplied: no modular reduction needed
    Fr t2 = fe_sqr(t);
    Fr t4 = fe_sqr(t2);
    Fr t6 = fe_mul(t4, t2);
    r = fe_mul(t6, t);
  }
  return fe_add(r, k);
}

// ---- lane pairs: the latency-bound form -----------------------------------------------------------------------------------
// A permutation is a chain of 91 x 4 dependent multiplications, and a launch that cannot fill the chip (the upper levels of
// a tree, one request's Merkle path) runs at the latency of that chain: 0.33 ms per 2-to-1 hash whatever the launch size
// (profiles/r03_coresidency_real.json).  t^7 needs only THREE multiplications in sequence -- t^2, then t^3 and t^4 side by
// side, then t^4 t^3 -- so here lanes 2j and 2j + 1 walk the same permutation: both square t, the even lane forms t^4 and the
// odd lane t^3 with ONE instruction stream (t^2 times a per-lane selected operand), they swap results, and both form t^7.
// Twice the lanes, 0.78x the chain (477 multiply-adds per round instead of 612).  Used only where the lanes are idle anyway.
__device__ __forceinline__ Fr pair_swap(const Fr& v) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = OG_PAIR_SWAP32(v.l[i]);
  return r;
}
__device__ __forceinline__ Fr pair_select(bool take_a, const Fr& a, const Fr& b) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = take_a ? a.l[i] : b.l[i];
  return r;
}

// E_k(x) by a lane pair (same value in both lanes, in and out)
__device__ __forceinline__ Fr mimc7_permute_pair(const uint32_t* __restrict__ consts, Fr x, const Fr& k, bool odd) {
  Fr r = x;
  for (int i = 0; i < MIMC7_ROUNDS; i++) {
    const Fr t = fe_add3_weak(r, k, mimc7_const(consts, i));
    const Fr t2 = OG_MIMC_LAT_SQR(t);                           // (the latency forms of the products: field.hip.h)
    const Fr u = OG_MIMC_LAT_MUL(t2, pair_select(odd, t, t2));  // even lane: t^4, odd lane: t^3   (t < 5N, t2 < 2N: within fe_mul's bound)
    r = OG_MIMC_LAT_MUL(u, pair_swap(u));                       // t^7 in both
  }
  return fe_add(r, k);
}

// MultiMiMC7([l, r], key = 0)
template <bool PAIR = false>
__device__ __forceinline__ Fr mimc7_hash2(const uint32_t* __restrict__ consts, const Fr& l, const Fr& r, bool odd = false) {
  Fr k = Fr::zero();
  if constexpr (PAIR) {
    k = fe_add(fe_add(k, l), mimc7_permute_pair(consts, l, k, odd));
    k = fe_add(fe_add(k, r), mimc7_permute_pair(consts, r, k, odd));
  } else {
    k = fe_add(fe_add(k, l), mimc7_permute(consts, l, k));
    k = fe_add(fe_add(k, r), mimc7_permute(consts, r, k));
  }
  return k;
}

// ---- the wave-wide form: one hash per WAVE (field_w9.hip.h) ------------------------------------------------------------------------
// Where even lane pairs leave the chip idle -- the top of a tree, an append of a few leaves -- a hash can have a whole wave: the
// nine limbs in nine lanes, a round = two additions, four wave-wide products (616 cycles each on a lone wave against 904), ~2 600
// cycles against the pair's ~2 860.  `consts9`: the round constants as 91 x 16 u32 (nine Montgomery limbs, then zeros).
// ROWS (the form the launches use; field_w9.hip.h "two rows"): rows 0 and 1 of the wave both carry x and k, and a round is THREE
// products deep -- t^2 in both rows; then t^2 times (t^2 | t) = t^4 in row 0 beside t^3 in row 1; then t^4 times (t^2 | t^3) =
// t^6 in row 0 (a wire of the circuit, off the chain) beside t^7 in row 1, which one v_permlane16_swap_b32 hands to both rows.
// `lane` is the limb a lane holds (w9_row_limb(threadIdx.x) with ROWS, threadIdx.x without), `row1` = threadIdx.x & 16.

// the products of one round from t = x + k + c: t^2 (every row), t^4 and t^6 (row 0 with rows), x' = t^7 (every row)
// FORM 0: one row, four products deep (as first built); 1: two rows, three deep; 2: two rows and the 32-bit Montgomery digit
// (field_w9.hip.h LAZY) -- what the launches use; 0 and 1 exist in hooks builds (OG_W9_ROWS).  Bounds with FORM 2, in multiples of
// N: a product is below a b / 169 + 8.01; inputs x < 8.5 (an earlier round) or < 2 (a hash input), k < 10.5 (k1 = l + x with l < 2: every
// hash output passes the strict product of w9_renorm / w9_mimc7_hash2's `red`), c < 2: t < 21, t^2 < 10.5, t^3 < 9.3, t^4 < 8.7,
// t^6 < 8.6, t^7 < 8.5 -- all below 2^258, limbs < 2^31 + 2^9 going in (a_i b_j + m' N_j + carry < 2^62.6).
// With rows (FORM >= 1) t4 / t6 are what the two rows hold after the second / third product -- row 0: t^4, t^6; row 1: t^3, t^7 --
// and t6r0 is t^6 in BOTH rows (the other half of the swap that hands t^7 to both): the witness walk stores two wires per row from them.
template <int FORM>
__device__ __forceinline__ void w9_mimc7_round(uint32_t t, uint32_t nj, bool row1, uint32_t& t2, uint32_t& t4, uint32_t& t6, uint32_t& t6r0,
                                               uint32_t& x) {
  constexpr bool LAZY = FORM >= 2;
  const U9 ta = w9_gather(t);
  // (form 2: the per-row digit for this product too, although both rows carry the same t -- v_mul_lo_u32 + a DPP broadcast cost a
  // lone wave 17 cycles per step, v_readfirstlane_b32 + s_mul_i32 and the way back into a VALU operand 33: og_ubench kinds 210 ..)
  t2 = w9_mul<FrParams, FORM >= 2, LAZY>(ta, t, nj);
  const U9 t2a = w9_gather(t2);
  if constexpr (FORM >= 1) {
    t4 = w9_mul<FrParams, true, LAZY>(t2a, row1 ? t : t2, nj);                  // row 0: t^4, row 1: t^3
    t6 = w9_mul<FrParams, true, LAZY>(w9_gather(t4), row1 ? t4 : t2, nj);       // row 0: t^4 t^2, row 1: t^4 t^3
    OG_W9_ROWS01(t6, t6r0, x);
  } else {
    t4 = w9_mul<FrParams>(t2a, t2, nj);
    t6 = w9_mul<FrParams>(t2a, t4, nj);
    t6r0 = t6;
    x = w9_mul<FrParams>(ta, t6, nj);
  }
}
// E_k(x) without the final + k; x, k spread (limbs < 2^31; x < 2 N, k < 4 N).  The next round's constant is asked for a round ahead:
// on a lone wave a load that is waited for where it is issued costs its whole latency, every round.
template <int FORM>
__device__ __forceinline__ uint32_t w9_mimc7_rounds(const uint32_t* __restrict__ consts9, uint32_t x, uint32_t k, uint32_t nj, int lane, bool row1) {
  const int cl = lane < 15 ? lane : 15;
  uint32_t c = consts9[cl];
#pragma unroll 1
  for (int i = 0; i < MIMC7_ROUNDS; i++) {
    const uint32_t c_next = consts9[(i + 1 < MIMC7_ROUNDS ? i + 1 : i) * 16 + cl];
    uint32_t t2, t4, t6, t6r0;
    w9_mimc7_round<FORM>(x + k + c, nj, row1, t2, t4, t6, t6r0, x);
    c = c_next;
  }
  return x;
}
// MultiMiMC7([l, r], key 0) by one wave: l, r lane-local (Montgomery, < 2 N, the same in every lane); the result lane-local again
template <int FORM>
__device__ __forceinline__ Fr w9_mimc7_hash2(const uint32_t* __restrict__ consts9, const Fr& l, const Fr& r, int tid) {
  const int lane = FORM ? w9_row_limb(tid) : tid;
  const bool row1 = (tid & 16) != 0;
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  const uint32_t ls = w9_spread(l, lane), rs = w9_spread(r, lane);
  const uint32_t k1 = ls + w9_mimc7_rounds<FORM>(consts9, ls, 0u, nj, lane, row1);                 // l + E_0(l): < 4 N
  const uint32_t out = 2u * k1 + rs + w9_mimc7_rounds<FORM>(consts9, rs, k1, nj, lane, row1);      // 2 k1 + r + x_91: < 12 N, limbs < 2^32
  const uint32_t red = w9_mul<FrParams>(w9_uniform(FrParams::ONE), w9_carry(out, lane), nj);   // the same value below 2 N
  const Fr lazy = w9_collect<FrParams>(red);
  return fe_from_lazy_limbs<FrParams>(lazy.l);  // normalized limbs, < 2 N (v or v + N: callers take it out of Montgomery form, which is unique)
}

}  // namespace og
