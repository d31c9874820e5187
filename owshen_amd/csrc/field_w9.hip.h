// "w9": ONE field element spread over nine lanes of a wave -- the latency form of the Montgomery product (round 6).
//
// Why.  A MiMC7 permutation is 91 x 4 dependent products, and where a launch cannot fill the chip -- one request's Merkle
// walk (witness.hip), the upper levels of a tree, an append to the commitment tree (mimc7.hip) -- the time IS that chain:
// a lone wave issues one VALU instruction per ~4.8 cycles whatever it depends on, so a product costs its
// instruction count, 205 for fe_mul.  Instruction-level parallelism inside a lane buys nothing (fe_mul_lat, measured);
// what is left is fewer instructions per wave, i.e. the nine limbs in nine LANES:
//
//   lane j holds limb j of b (and N_j);  the limbs of a are wave-uniform (SGPRs, nine v_readlane_b32);  per limb a_i:
//       acc += a_i b_j                            v_mad_u64_u32 (SGPR operand)
//       m    = first lane of (acc * -N^-1) mod W  v_mul_lo_u32, v_and_b32, v_readfirstlane_b32
//       acc += m N_j                              v_mad_u64_u32 (SGPR operand)
//       acc  = (acc >> 29) + (acc mod W of lane j + 1)      v_lshrrev_b64, v_and_b32 with a DPP row_shl:1 source, 64-bit add
//   -- the division by W = 2^29 moves every column one lane down, the carry stays where it is -- and one carry pass at the end:
//   ~80 instructions instead of 205, the same Montgomery digits m_0 .. m_8 as fe_mul (CIOS order instead of column order: the
//   value is identical mod N and < 2N; the limbs are "almost normalized", < 2^29 + 32).
//
// Lanes 9 .. 63 of the wave carry zeros (b_j = N_j = 0 there: every accumulator stays 0), so the lane above the top limb
// feeds the shift a zero without a mask.  In this first form it is one product per WAVE: m and the a_i are wave-uniform
// (v_readfirstlane_b32 / v_readlane_b32 into SGPRs, the digit formed on the scalar unit).
//
// Operands: limbs < 2^31 on BOTH sides (2^31 x 2^31 + 2^29 x 2^29 + 2^34 < 2^64 per step: the accumulator is renormalized
// every step, so the 18-products-per-column bound of field.hip.h does not apply), values with a * b < 169 N^2.  So
// t = x + k + c needs no carry pass before it is squared.
//
// TWO ROWS (the last step of round 6).  Everything above is local to a DPP row of 16 lanes except the Montgomery digit, and that
// has a row-local form too on gfx90a and later: `row_newbcast:0` hands every lane of a row its row's lane 0.  With the digit taken
// that way (w9_mul<M, true>) rows 0 and 1 of the wave -- lanes 0 .. 8 and 16 .. 24 -- multiply the SAME uniform a by two DIFFERENT
// spread operands in the same ~80 instructions: the second lane group a MiMC7 round needs to be three products deep instead of
// four (t^2; then t^4 beside t^3; then t^6 beside t^7: mimc7.hip.h).  What crosses rows is a gather (v_readlane_b32 reads any
// lane) and one v_permlane16_swap_b32 per round (gfx950: a register swapped with its own copy leaves row 1 in rows 0 AND 1).
//
// On the CPU interpreter (tests/hipemu) the cross-lane reads are block-wide rendezvous, like OG_PAIR_SWAP32.
#pragma once
#include "field.hip.h"

namespace og {

#ifndef OG_W9_READLANE
#define OG_W9_READLANE(x, i) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (i)))
#define OG_W9_FIRST(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// lane j reads lane j + 1 (DPP row_shl:1) / lane j - 1 (row_shr:1); a lane without a source in its row of 16 reads 0
#define OG_W9_FROM_NEXT(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x101, 0xF, 0xF, true))
#define OG_W9_FROM_PREV(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x111, 0xF, 0xF, true))
// every lane reads lane 0 of its own row of 16 (DPP row_newbcast:0)
#define OG_W9_ROWFIRST(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x150, 0xF, 0xF, true))
// rows 0 and 1 both read row 1 (rows 2 and 3: row 3): v_permlane16_swap_b32 swaps the odd rows of its first operand with the
// even rows of its second; with the register and its copy the second result is (row 1, row 1, row 3, row 3)
#define OG_W9_FROM_ROW1(x) (__builtin_amdgcn_permlane16_swap((x), (x), false, false)[1])
// the same instruction's two results: r0 = (row 0, row 0, row 2, row 2), r1 = (row 1, row 1, row 3, row 3)
#define OG_W9_ROWS01(x, r0, r1) do { auto og_w9_sw_ = __builtin_amdgcn_permlane16_swap((x), (x), false, false); (r0) = og_w9_sw_[0]; (r1) = og_w9_sw_[1]; } while (0)
// the low 29 bits of lane j + 1's value (0 without a source in the row): the mask rides on the DPP move, one v_and_b32_dpp -- the
// compiler keeps a v_mov_b32_dpp and a v_and_b32 apart, and both sit on the chain of every step (s_nop 1: the two wait states
// between a VALU write and a DPP read, which the hazard recognizer does not insert in front of inline assembly)
#define OG_W9_NEXT_LOW29(x) og::w9_next_low29_(x)
#endif

#ifdef OG_W9_NEXT_LOW29
__device__ __forceinline__ uint32_t w9_next_low29_(uint32_t x) {
  uint32_t r;
  asm("s_nop 1\n\tv_and_b32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(0x1fffffffu));
  return r;
}
#else
#define OG_W9_NEXT_LOW29(x) (OG_W9_FROM_NEXT(x) & 0x1fffffffu)
#endif

// the wave-uniform copy of an element (SGPRs on the GPU)
struct U9 {
  uint32_t l[9];
};

// limb `lane` of a constant (9 limbs), 0 in lanes 9 ..
__device__ __forceinline__ uint32_t w9_const_limb(const uint32_t c[9], int lane) {
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) v = lane == i ? c[i] : v;
  return v;
}
template <class M>
__device__ __forceinline__ uint32_t w9_modulus_limb(int lane) { return w9_const_limb(M::N, lane); }

// the uniform form of a constant / of a spread element (nine v_readlane_b32)
__device__ __forceinline__ U9 w9_uniform(const uint32_t c[9]) {
  U9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c[i];
  return r;
}
__device__ __forceinline__ U9 w9_gather(uint32_t x) {
  U9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = OG_W9_READLANE(x, i);
  return r;
}

// one carry pass over a spread element: limbs < 2^32 in, < 2^29 + 8 out (the top limb keeps what it has)
__device__ __forceinline__ uint32_t w9_carry(uint32_t x, int lane) {
  const uint32_t lo = lane == 8 ? x : (x & MASK29), hi = lane == 8 ? 0u : (x >> 29);
  return lo + OG_W9_FROM_PREV(hi);
}

// the limb a lane holds when rows 0 AND 1 of the wave carry an element (lanes 0 .. 8 and 16 .. 24); 15 = none
__device__ __forceinline__ int w9_row_limb(int lane) { return lane < 32 ? (lane & 15) : 15; }

// a b 2^-261 mod N: a uniform, b and the result spread (header comment).  nj = w9_modulus_limb<M>(lane).
// ROWS = false: one Montgomery digit for the wave (lane 0's) -- every row that carries an element must carry the SAME b.
// ROWS = true: a digit per row of 16 lanes -- rows may carry different b (nj in each of them).
// LAZY = true: the digit keeps all 32 bits of acc * -N^-1 instead of the low 29 (lane 0's low limb vanishes either way: m' = m mod
// 2^29) -- one instruction less on the chain of every step, and the result is (a b + M' N) / 2^261 with M' < 2^264: the same
// residue, below a b / (169 N) + 8.01 N instead of a b / (169 N) + N.  mimc7.hip.h carries the bounds of a round built from it.
template <class M, bool ROWS = false, bool LAZY = false>
__device__ __forceinline__ uint32_t w9_mul(const U9& a, uint32_t b, uint32_t nj) {
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    acc += (uint64_t)a.l[i] * b;
    const uint32_t mi = LAZY ? (uint32_t)acc * M::INV : ((uint32_t)acc * M::INV) & MASK29;
    const uint32_t m = ROWS ? OG_W9_ROWFIRST(mi) : OG_W9_FIRST(mi);
    acc += (uint64_t)m * nj;  // lane 0: the low 29 bits are zero now
    acc = (acc >> 29) + (LAZY ? OG_W9_NEXT_LOW29((uint32_t)acc) : OG_W9_FROM_NEXT((uint32_t)acc & MASK29));
  }
  // acc < 2^34 (LAZY: 2^37): limb j = acc mod W + the carry of limb j - 1 (the top limb's carry is zero: the value is < 2N < 2^255;
  // LAZY: < 2^258)
  return ((uint32_t)acc & MASK29) + OG_W9_FROM_PREV((uint32_t)(acc >> 29));
}

// ---- sums and differences of spread elements (the group law in the wave-wide form: ecmul_impl.hip.h) ----------------------------
// Limb j of K N in a borrow-free form: every limb below the top inflated by 2 x 2^29 (two units borrowed from the limb above), so
// that C_j - b_j > 0 for every limb of a subtrahend b whose limbs are below 2^30 - 2 (a product's output, < 2^29 + 32, or a carried
// sum, < 2^29 + 8) and whose value is below (K - 2) N: then a + (C - b) = a - b + K N limb by limb, no borrow.  C_j < 2^29 + 2^30.
template <class M, int K>
__device__ __forceinline__ uint32_t w9_kn_limb(int lane) {
  uint32_t v = 0;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    c += (uint64_t)M::N[i] * (uint32_t)K;
    const uint32_t nk = i < 8 ? (uint32_t)c & MASK29 : (uint32_t)c;
    c >>= 29;
    const uint32_t limb = i == 0 ? nk + (2u << 29) : i < 8 ? nk + (2u << 29) - 2u : nk - 2u;
    v = lane == i ? limb : v;
  }
  return v;
}
// a - b + K N (ck = w9_kn_limb<M, K>(lane)); limbs < limbs(a) + 2^29 + 2^30: carry before the result is a subtrahend or a summand again
__device__ __forceinline__ uint32_t w9_sub(uint32_t a, uint32_t b, uint32_t ck) { return a + (ck - b); }

// spread element <-> the lane-local Fe<M> (tests, the seams of a kernel): lane j of the first nine takes / gives limb j
template <class M>
__device__ __forceinline__ uint32_t w9_spread(const Fe<M>& x, int lane) { return w9_const_limb(x.l, lane); }
template <class M>
__device__ __forceinline__ Fe<M> w9_collect(uint32_t x) {  // every lane gets the whole element (limbs as they are)
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = OG_W9_READLANE(x, i);
  return r;
}
// limbs < 2^32, value < 2^261 -> normalized limbs (for fe_mul / fe_from_mont, which want limbs < 2^30)
template <class M>
OG_HD Fe<M> fe_from_lazy_limbs(const uint32_t t[9]) {
  Fe<M> r;
  normalize29u(r.l, t);
  return r;
}

}  // namespace og
