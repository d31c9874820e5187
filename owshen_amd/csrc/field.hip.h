// BN254 Fq / Fr Montgomery arithmetic for gfx950.
//
// Fr is the reference's `Fp`
// (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11:
// modulus, generator 7, 32-byte little-endian canonical repr).  Fq is the
// EIP-196 base field; the reference has no Fq.
//
// Register layout: 9 limbs of 29 bits (radix 2^29, 261 bits), Montgomery form with R = 2^261.
// CDNA4 has no 64x64 multiply and -- unlike NVIDIA's IMAD.X -- no multiply-add that consumes a carry
// flag: with 32-bit limbs every partial product costs a v_mad_u64_u32 PLUS a 64-bit add PLUS register
// moves to form operand pairs (measured: ~600 VALU instructions per product).  With 29-bit limbs a 58-bit
// partial product leaves 6 spare bits, so the 9 products and 9 reduction products of a column accumulate
// into ONE 64-bit register with no carries at all: a product is 81 + 81 `v_mad_u64_u32 t, a, b, t`, one
// shift per column hands the carry to the next column as its starting addend, and the final conditional
// subtraction is dropped (R > 4N).  205 instructions per product, 162 of them the multiply-adds.
//
// Invariant of every Fe<M> value handed between functions: limbs < 2^29, value < 2N ("almost
// reduced"; 0, N and 2N... never 2N: [0, 2N)).  fe_mul tolerates operands up to 8N.
// In HBM (tables, buckets, scratch): the same value (< 2N < 2^256) as 32 bytes little-endian.
// At the C-ABI seam: canonical (non-Montgomery, < N) 32-byte little-endian.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// every routine is usable from host code too (og_verify runs the same field / curve layer on the CPU)
#define OG_HD __host__ __device__ __forceinline__

namespace og {

constexpr uint32_t MASK29 = (1u << 29) - 1;

struct FqParams {
  static constexpr uint32_t N[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                    0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t N2[9] = {0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u,
                                     0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
  static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod N
  static constexpr uint32_t R2[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                     0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};  // 2^522 mod N
  static constexpr uint32_t INV = 0x04866389u;  // -N^-1 mod 2^29
  // k N with every limb below the top inflated by 2^29 (borrowed from the next limb): C[i] - a[i] lies in (0, 2^30)
  // for any normalized a < k N, so k N - a needs no borrow propagation (fe_neg_lazy*, fe_sub_weak)
  static constexpr uint32_t NEG4[9] = {0x21f3f51cu, 0x241182dau, 0x31ca8d3bu, 0x2b548b42u, 0x361765dfu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
  static constexpr uint32_t NEG8[9] = {0x23e7ea38u, 0x282305b5u, 0x23951a77u, 0x36a91686u, 0x2c2ecbbfu, 0x36da0604u, 0x25370a07u, 0x32e1319fu, 0x01832272u};
  static constexpr uint32_t N3[9] = {0x0976f7d5u, 0x030d2224u, 0x1557e9edu, 0x087f6872u, 0x00918c68u, 0x0891c242u, 0x01f4a3c3u, 0x0b14729cu, 0x00912cebu};
  static constexpr uint32_t N4[9] = {0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
  static constexpr uint32_t N5[9] = {0x1a70f263u, 0x0515e391u, 0x0e3d308bu, 0x0e29ae14u, 0x0b9d3f58u, 0x0e4843c3u, 0x03426645u, 0x07ccbf04u, 0x00f1f588u};
};

struct FrParams {
  static constexpr uint32_t N[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                    0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t N2[9] = {0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u,
                                     0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
  static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                      0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
  static constexpr uint32_t R2[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                     0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
  static constexpr uint32_t INV = 0x0fffffffu;
  // k N with every limb below the top inflated by 2^29 (borrowed from the next limb): C[i] - a[i] lies in (0, 2^30)
  // for any normalized a < k N, so k N - a needs no borrow propagation (fe_neg_lazy*, fe_sub_weak)
  static constexpr uint32_t NEG4[9] = {0x20000004u, 0x3c3eb27du, 0x39709142u, 0x3f4243ccu, 0x36174a0bu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
  static constexpr uint32_t NEG8[9] = {0x20000008u, 0x387d64fbu, 0x32e12286u, 0x3e84879au, 0x2c2e9418u, 0x36da0604u, 0x25370a07u, 0x32e1319fu, 0x01832272u};
  static constexpr uint32_t N3[9] = {0x10000003u, 0x1d2f05deu, 0x0b146cf2u, 0x1771b2dau, 0x00917789u, 0x0891c242u, 0x01f4a3c3u, 0x0b14729cu, 0x00912cebu};
  static constexpr uint32_t N4[9] = {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
  static constexpr uint32_t N5[9] = {0x10000005u, 0x1b4e5f1du, 0x07ccb594u, 0x0712d4c1u, 0x0b9d1c90u, 0x0e4843c3u, 0x03426645u, 0x07ccbf04u, 0x00f1f588u};
};

template <class M>
struct Fe {
  uint32_t l[9];

  OG_HD static Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
  }
  OG_HD static Fe one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = M::ONE[i];
    return r;
  }
  // value == 0 mod N, i.e. the limbs are exactly 0 or exactly N (values are < 2N)
  OG_HD bool is_zero() const {
    // limb 0 first: the nine-limb comparison then sits in a branch a wave almost never takes (this test runs once per
    // mixed addition, on the accumulator and on the gathered base)
    if (l[0] != 0 && l[0] != M::N[0]) return false;
    uint32_t z = 0, n = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      z |= l[i];
      n |= l[i] ^ M::N[i];
    }
    return z == 0 || n == 0;
  }
  OG_HD bool operator==(const Fe& b) const;
  OG_HD bool operator!=(const Fe& b) const { return !(*this == b); }
};

// ---- carry handling -----------------------------------------------------------

// limbs are signed 32-bit quantities with |t_i| < 2^31 whose weighted sum is a value in [0, 2^261):
// propagate carries so that every limb lands in [0, 2^29)
OG_HD void normalize29(uint32_t r[9], const int32_t t[9]) {
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t v = t[i] + c;
    r[i] = (uint32_t)v & MASK29;
    c = v >> 29;  // arithmetic
  }
}

// t: signed limbs, value in [0, 4N): returns the value reduced into [0, 2N) with normalized limbs
template <class M>
OG_HD Fe<M> reduce_4n(const int32_t t[9]) {
  uint32_t n[9];
  normalize29(n, t);
  // u = n - 2N with borrow
  uint32_t u[9];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t v = (int32_t)n[i] - (int32_t)M::N2[i] + c;
    u[i] = (uint32_t)v & MASK29;
    c = v >> 29;
  }
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? n[i] : u[i];  // c = -1: n < 2N, keep n
  return r;
}

template <class M>
OG_HD Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) {
  int32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = (int32_t)(a.l[i] + b.l[i]);
  return reduce_4n<M>(t);
}

template <class M>
OG_HD Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) {
  int32_t t[9];  // a - b + 2N in (0, 4N)
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)M::N2[i];
  return reduce_4n<M>(t);
}

template <class M>
OG_HD Fe<M> fe_neg(const Fe<M>& a) {
  return fe_sub(Fe<M>::zero(), a);
}

template <class M>
OG_HD Fe<M> fe_dbl(const Fe<M>& a) {
  return fe_add(a, a);
}

template <class M>
OG_HD bool Fe<M>::operator==(const Fe<M>& b) const {
  return fe_sub(*this, b).is_zero();
}

// ---- column-serial Montgomery products ----------------------------------------------
// One 64-bit accumulator t walks the columns k = 0..16 of  sum(products) + sum_j m_j N 2^(29 j):  the carry out of column
// k - 1 (t >> 29) is the ADDEND the first multiply-add of column k starts from, so carry propagation costs no instruction
// of its own.  (Rounds 1-2 kept 17 independent column accumulators and paid a v_lshl_add_u64 per column: 152 of the 2321
// instructions of one G1 mixed addition, each issuing at the rate of a v_mad_u64_u32.  Measured on one box: 727 -> 758
// proofs/s.)  A product is one dependent chain; og_ubench_cycles shows a dependent v_mad_u64_u32 chain issuing at the same
// 9.5 cycles per wave as independent ones, whatever VGPR banks its operands sit in (profiles/r02_probe_chains.json).
// Column bound: 9 * 2^60 (products, one lazy operand) + 9 * 2^58 (m_j N) + carry (< 2^36) < 2^64.
//
// On the GPU each routine is ONE asm statement (mont_gfx950.inc, generated by tools/gen_mont_asm.py and interpreted on the
// CPU by tests/test_mont_asm.py): left to the optimizer, `t = carry; t += a b; ...` is reassociated so that the
// late-arriving carry is added LAST -- back to a separate 64-bit addition per column -- and asm statements smaller than
// the routine are each padded with an s_nop by the hazard recognizer.  The accumulator is the fixed pair v[30:31] because
// the text must name its halves, which an asm operand cannot.  Host code (og_verify, the CPU tests, the interpreter)
// runs the same columns as C: MontCols below; both give the same limbs.
#include "mont_gfx950.inc"
#define OG_FE_V(n, x)                                                                                                     \
  [n##0] "v"((x).l[0]), [n##1] "v"((x).l[1]), [n##2] "v"((x).l[2]), [n##3] "v"((x).l[3]), [n##4] "v"((x).l[4]),           \
      [n##5] "v"((x).l[5]), [n##6] "v"((x).l[6]), [n##7] "v"((x).l[7]), [n##8] "v"((x).l[8])
#define OG_FE_DBL(n, d)                                                                                                   \
  [n##d0] "v"(d[0]), [n##d1] "v"(d[1]), [n##d2] "v"(d[2]), [n##d3] "v"(d[3]), [n##d4] "v"(d[4]), [n##d5] "v"(d[5]),       \
      [n##d6] "v"(d[6]), [n##d7] "v"(d[7])
#define OG_MONT_OUT(r)                                                                                                    \
  [r0] "=&v"((r).l[0]), [r1] "=&v"((r).l[1]), [r2] "=&v"((r).l[2]), [r3] "=&v"((r).l[3]), [r4] "=&v"((r).l[4]),           \
      [r5] "=&v"((r).l[5]), [r6] "=&v"((r).l[6]), [r7] "=&v"((r).l[7]), [r8] "=&v"((r).l[8])
#define OG_MONT_MOD(M)                                                                                                    \
  [n0] "s"(M::N[0]), [n1] "s"(M::N[1]), [n2] "s"(M::N[2]), [n3] "s"(M::N[3]), [n4] "s"(M::N[4]), [n5] "s"(M::N[5]),       \
      [n6] "s"(M::N[6]), [n7] "s"(M::N[7]), [n8] "s"(M::N[8]), [inv] "s"(M::INV)
#define OG_MONT_CLOBBER "v30", "v31", "vcc"
#if defined(__HIP_DEVICE_COMPILE__)
#define OG_MONT_DEVICE 1
#else
#define OG_MONT_DEVICE 0
#endif
template <class M>
OG_HD void fe_doubled(uint32_t d[8], const Fe<M>& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = a.l[i] << 1;
}

template <class M>
struct MontCols {
  uint32_t m[9];
  uint64_t t;
  Fe<M> r;
  // add the a_i b_j with i + j = K
  template <int K>
  OG_HD void mul(const Fe<M>& a, const Fe<M>& b) {
    constexpr int lo = K > 8 ? K - 8 : 0, hi = K < 8 ? K : 8;
    for (int i = lo; i <= hi; i++) t += (uint64_t)a.l[i] * b.l[K - i];
  }
  // the 45-product squaring: cross terms through the doubled limb (a normalized, 2 a_i < 2^30)
  template <int K>
  OG_HD void sqr(const Fe<M>& a) {
    constexpr int lo = K > 8 ? K - 8 : 0;
    for (int i = lo; i < K - i; i++) t += (uint64_t)(a.l[i] << 1) * a.l[K - i];
    if constexpr ((K & 1) == 0) t += (uint64_t)a.l[K / 2] * a.l[K / 2];
  }
  // close column K: the reduction products m_j N_(K-j) of earlier rows, this row's multiplier (low half) or the result limb
  // (high half), then shift the carry down
  template <int K>
  OG_HD void close() {
    constexpr int lo = K > 8 ? K - 8 : 0, hi = K < 9 ? K : 9;  // rows lo <= j < hi
    for (int j = lo; j < hi; j++) t += (uint64_t)m[j] * M::N[K - j];
    if constexpr (K < 9) {
      m[K] = ((uint32_t)t * M::INV) & MASK29;
      t += (uint64_t)m[K] * M::N[0];  // the low 29 bits of t are zero now
    } else {
      r.l[K - 9] = (uint32_t)t & MASK29;
    }
    t >>= 29;
  }
};

// f(integral_constant<int, K>) for K = 0 .. 16
template <int K, class F>
OG_HD void for_columns(F&& f) {
  if constexpr (K < 17) {
    f(std::integral_constant<int, K>{});
    for_columns<K + 1>(f);
  }
}
#define OG_COL(kc) decltype(kc)::value

// a * b * 2^-261 mod N.  Operands: limbs < 2^30 (normalized is < 2^29), values a, b with a * b < 169 N^2;
// result < 2N, normalized.
// ONE operand may be lazier: limbs < 2^31 and a value up to 42 N, against a normalized (< 2^29-limb, < 2N) partner -- the
// radix-4 NTT's lazy butterflies multiply such a sum by a twiddle (ntt.hip, k_ntt_block4).  The column bound still holds:
// nine limb products of 2^31 x 2^29 = 9 x 2^60, nine reduction products of 2^29 x 2^29 = 9 x 2^58 and a carry < 2^35 stay
// below 2^64 (9 x 2^60 + 9 x 2^58 + 2^35 < 2^63.5), and 42 N x 2 N < 169 N^2.  That is under one bit of headroom: it relies on
// the column walk of this routine AND of its asm form (tools/gen_mont_asm.py: one 64-bit accumulator per column, carry = t >> 29),
// so any change to either must re-derive it; tests/test_emu_field29.py drives both forms with all-limbs-at-the-bound operands.
template <class M>
OG_HD Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template mul<OG_COL(kc)>(a, b);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

// ---- latency forms (hooks builds only: measured, not faster, never shipped) -------------------------------------------------
#ifdef OG_AB_HOOKS
// fe_mul above is ONE chain of 162 dependent multiply-adds: free when other waves fill the pipe (a dependent v_mad_u64_u32
// issues every 9.5 cycles, four waves hide that), but a request that walks a Merkle path is a single wave whose every
// product waits for the one before -- there the chain looked like the time (~1150 cycles per product).  [Measured, round 4:
// they are NOT faster on gfx950 -- a lone wave issues one v_mad_u64_u32 per ~9.5 cycles dependent or not -- and are kept only
// as the -DOG_MIMC_LAT=1 A/B build of mimc7.hip.h.]  These forms expose the
// parallelism instead: 17 independent column accumulators take the 81 (45) limb products in any order, then the reduction goes
// row by row -- a row's multiplier needs only its own column, its nine products m N_j land in nine different accumulators --
// so the multiply-adds issue back to back (4.4 cycles each) and only ~6 dependent instructions per row sit on the critical
// path.  Same Montgomery digits, same normalized limbs: bit-identical to fe_mul / fe_sqr (tests/test_emu_field29.py).
// 38 more registers, which a throughput kernel cannot afford; used by the lane-pair MiMC7 forms (mimc7.hip.h, witness.hip).
template <class M>
OG_HD Fe<M> fe_reduce_lat(uint64_t c[17]) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const uint32_t m = ((uint32_t)c[k] * M::INV) & MASK29;
#pragma unroll
    for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * M::N[j];
    c[k + 1] += c[k] >> 29;  // (the low 29 bits of column k are zero now)
  }
  Fe<M> r;
  uint64_t t = c[9];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.l[i] = (uint32_t)t & MASK29;
    t = (t >> 29) + (i + 10 < 17 ? c[i + 10] : 0);
  }
  r.l[8] = (uint32_t)t;
  return r;
}

template <class M>
OG_HD Fe<M> fe_mul_lat(const Fe<M>& a, const Fe<M>& b) {
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; k++) {
    c[k] = 0;
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) c[k] += (uint64_t)a.l[i] * b.l[k - i];
  }
  return fe_reduce_lat<M>(c);
}

template <class M>
OG_HD Fe<M> fe_sqr_lat(const Fe<M>& a) {  // a normalized (2 a_i < 2^30)
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; k++) {
    c[k] = 0;
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i < k - i; i++) c[k] += (uint64_t)(a.l[i] << 1) * a.l[k - i];
    if ((k & 1) == 0) c[k] += (uint64_t)a.l[k / 2] * a.l[k / 2];
  }
  return fe_reduce_lat<M>(c);
}

#endif  // OG_AB_HOOKS

// (a^2 + c d) 2^-261 mod N with one reduction and the 45-product squaring (a normalized; c may be lazy)
template <class M>
OG_HD Fe<M> fe_sqr_add(const Fe<M>& a, const Fe<M>& c, const Fe<M>& d) {
#if OG_MONT_DEVICE
  Fe<M> r;
  uint32_t ad[8];
  fe_doubled(ad, a);
  asm(OG_MONT_ASM_SQR_ADD : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(c, c), OG_FE_V(d, d), OG_FE_DBL(a, ad) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template sqr<OG_COL(kc)>(a);
    x.template mul<OG_COL(kc)>(c, d);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

// a + b limb-wise, no carries: limbs < 2^30, multiplication operand only
template <class M>
OG_HD Fe<M> fe_add_lazy(const Fe<M>& a, const Fe<M>& b) {
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}

// ---- lazy operands and fused products ----------------------------------------------
// 8N - a for a normalized a < 8N: limbs in (0, 2^30), value in (0, 8N].  NOT a normalized Fe: valid only
// as an operand of the multiplication routines (which accept limbs < 2^30).
template <class M>
OG_HD Fe<M> fe_neg_lazy(const Fe<M>& a) {
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = M::NEG8[i] - a.l[i];
  return r;
}
// 4N - a for a normalized a < 4N (tighter value bound than fe_neg_lazy, for sums of many products)
template <class M>
OG_HD Fe<M> fe_neg_lazy4(const Fe<M>& a) {
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = M::NEG4[i] - a.l[i];
  return r;
}

// ---- weak (carry-only) forms for the inside of the group-law formulas -------------------------
// They skip the conditional subtraction: results have normalized limbs but are only bounded by a small multiple
// of N (every product routine gives a result < 2N as long as its sum of products is < 169 N^2 = N 2^261).  Results must not be stored
// (the 32-byte format holds < 2^256 ~ 5.3 N) nor fed to fe_add / fe_sub / is_zero, which assume < 2N.
OG_HD void normalize29u(uint32_t r[9], const uint32_t t[9]) {  // limbs < 2^32, value < 2^261
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t v = t[i] + c;
    r[i] = v & MASK29;
    c = v >> 29;
  }
}
// a - b + 4N for normalized a, b with b < 4N: value in (0, bound(a) + 4N]
template <class M>
OG_HD Fe<M> fe_sub_weak(const Fe<M>& a, const Fe<M>& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = a.l[i] + (M::NEG4[i] - b.l[i]);
  Fe<M> r;
  normalize29u(r.l, t);
  return r;
}
// a + 2 b, normalized limbs, value = bound(a) + 2 bound(b)
template <class M>
OG_HD Fe<M> fe_add2_weak(const Fe<M>& a, const Fe<M>& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = a.l[i] + (b.l[i] << 1);
  Fe<M> r;
  normalize29u(r.l, t);
  return r;
}
// a + b + c, normalized limbs, value = sum of the bounds (no reduction): for sums that feed a product directly
template <class M>
OG_HD Fe<M> fe_add3_weak(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = a.l[i] + b.l[i] + c.l[i];
  Fe<M> r;
  normalize29u(r.l, t);
  return r;
}
// d = a - b + 4N with a, b in [0, 2N): d in (2N, 6N), and a == b (mod N) iff d is 3N, 4N or 5N
template <class M>
OG_HD bool fe_weak_diff_is_zero(const Fe<M>& d) {
  if (d.l[0] != M::N3[0] && d.l[0] != M::N4[0] && d.l[0] != M::N5[0]) return false;  // limb 0 first, as in is_zero()
  uint32_t x = 0, y = 0, z = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    x |= d.l[i] ^ M::N3[i];
    y |= d.l[i] ^ M::N4[i];
    z |= d.l[i] ^ M::N5[i];
  }
  return x == 0 || y == 0 || z == 0;
}

// 2a without reduction: limbs < 2^30, value < 4N; multiplication operand only
template <class M>
OG_HD Fe<M> fe_dbl_lazy(const Fe<M>& a) {
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
  return r;
}

// (a b + c d) 2^-261 mod N with ONE reduction.  At most one operand of each product may be lazy
// (limbs < 2^30); column bound 9 (2^59 + 2^59) + 9 2^58 < 2^64.  a b + c d < 169 N^2 => result < 2N
// (N^2 / 2^261 = N / 169.5, plus the < N of the reduction term).
template <class M>
OG_HD Fe<M> fe_mul_add(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c, const Fe<M>& d) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL_ADD : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(c, c), OG_FE_V(d, d) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template mul<OG_COL(kc)>(a, b);
    x.template mul<OG_COL(kc)>(c, d);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

// a b 2^-261 + c  with ONE reduction and no separate carry pass: c (limbs < 2^30, e.g. the lazy 4N - x) joins the HIGH
// columns, i.e. c 2^261 is added before the division by 2^261.  The reduction multipliers m_i depend only on the low
// columns, so the result is exactly fe_mul(a, b) + c limb-for-limb after normalisation: value < 2N + bound(c),
// normalized limbs.  This is how the group law takes U2 - X1, S2 - Y1 and Q - X3 (ec.hip.h) without a carry pass.
template <class M>
OG_HD Fe<M> fe_mul_plus(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL_PLUS : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(p, c) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    if constexpr (OG_COL(kc) >= 9) x.t += c.l[OG_COL(kc) - 9];
    x.template mul<OG_COL(kc)>(a, b);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t + c.l[8];
  return x.r;
#endif
}
// (a b + d e) 2^-261 + c, one reduction (the Fq2 components of the same)
template <class M>
OG_HD Fe<M> fe_mul_add_plus(const Fe<M>& a, const Fe<M>& b, const Fe<M>& d, const Fe<M>& e, const Fe<M>& c) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL_ADD_PLUS : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(d, d), OG_FE_V(e, e), OG_FE_V(p, c) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    if constexpr (OG_COL(kc) >= 9) x.t += c.l[OG_COL(kc) - 9];
    x.template mul<OG_COL(kc)>(a, b);
    x.template mul<OG_COL(kc)>(d, e);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t + c.l[8];
  return x.r;
#endif
}

// (a b + c d + e f + g h) 2^-261 mod N with one reduction; at most TWO of the four products may have a lazy
// operand: 9 (2 2^59 + 2 2^58) + 9 2^58 < 2^64.  Sum < 169 N^2 => result < 2N.
template <class M>
OG_HD Fe<M> fe_mul_add4(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c, const Fe<M>& d,
                                             const Fe<M>& e, const Fe<M>& f, const Fe<M>& g, const Fe<M>& h) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL_ADD4 : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(c, c), OG_FE_V(d, d), OG_FE_V(e, e), OG_FE_V(f, f), OG_FE_V(g, g), OG_FE_V(h, h) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template mul<OG_COL(kc)>(a, b);
    x.template mul<OG_COL(kc)>(c, d);
    x.template mul<OG_COL(kc)>(e, f);
    x.template mul<OG_COL(kc)>(g, h);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

// (a^2 + b c + d e + f g) 2^-261 mod N, one reduction: the real part of the Fq2 form of X3 = R^2 - PP W (ec.hip.h)
template <class M>
OG_HD Fe<M> fe_sqr_add3(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c, const Fe<M>& d, const Fe<M>& e, const Fe<M>& f, const Fe<M>& g) {
#if OG_MONT_DEVICE
  Fe<M> r;
  uint32_t ad[8];
  fe_doubled(ad, a);
  asm(OG_MONT_ASM_SQR_ADD3 : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(c, c), OG_FE_V(d, d), OG_FE_V(e, e), OG_FE_V(f, f), OG_FE_V(g, g), OG_FE_DBL(a, ad) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template sqr<OG_COL(kc)>(a);
    x.template mul<OG_COL(kc)>(b, c);
    x.template mul<OG_COL(kc)>(d, e);
    x.template mul<OG_COL(kc)>(f, g);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}
// (a b + c d + e f) 2^-261 mod N, one reduction (its imaginary part)
template <class M>
OG_HD Fe<M> fe_mul_add3(const Fe<M>& a, const Fe<M>& b, const Fe<M>& c, const Fe<M>& d, const Fe<M>& e, const Fe<M>& f) {
#if OG_MONT_DEVICE
  Fe<M> r;
  asm(OG_MONT_ASM_MUL_ADD3 : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_V(b, b), OG_FE_V(c, c), OG_FE_V(d, d), OG_FE_V(e, e), OG_FE_V(f, f) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template mul<OG_COL(kc)>(a, b);
    x.template mul<OG_COL(kc)>(c, d);
    x.template mul<OG_COL(kc)>(e, f);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

// a^2 * 2^-261 mod N with 45 instead of 81 partial products (cross terms use the doubled limb 2 a_i < 2^30)
template <class M>
OG_HD Fe<M> fe_sqr(const Fe<M>& a) {
#if OG_MONT_DEVICE
  Fe<M> r;
  uint32_t ad[8];
  fe_doubled(ad, a);
  asm(OG_MONT_ASM_SQR : OG_MONT_OUT(r) : OG_MONT_MOD(M), OG_FE_V(a, a), OG_FE_DBL(a, ad) : OG_MONT_CLOBBER);
  return r;
#else
  MontCols<M> x;
  x.t = 0;
  for_columns<0>([&](auto kc) {
    x.template sqr<OG_COL(kc)>(a);
    x.template close<OG_COL(kc)>();
  });
  x.r.l[8] = (uint32_t)x.t;
  return x.r;
#endif
}

template <class M>
OG_HD Fe<M> fe_to_mont(const Fe<M>& a) {  // a: any value < 2^256 (< 5.3 N)
  Fe<M> r2;
#pragma unroll
  for (int i = 0; i < 9; i++) r2.l[i] = M::R2[i];
  return fe_mul(a, r2);
}

// [0, 2N) -> [0, N): one conditional subtraction of N
template <class M>
OG_HD Fe<M> fe_canon(const Fe<M>& a) {
  Fe<M> x = a;
  uint32_t u[9];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t v = (int32_t)x.l[i] - (int32_t)M::N[i] + c;
    u[i] = (uint32_t)v & MASK29;
    c = v >> 29;
  }
#pragma unroll
  for (int i = 0; i < 9; i++) x.l[i] = c ? x.l[i] : u[i];
  return x;
}

// a < N for the normalized limbs of ANY 256-bit value (what fe_load returns): the boundary's canonical-encoding test, the
// device-side counterpart of `Fp::from_repr` rejecting bytes >= the modulus
// (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11)
template <class M>
OG_HD bool fe_lt_modulus(const Fe<M>& a) {
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) c = ((int32_t)a.l[i] - (int32_t)M::N[i] + c) >> 29;
  return c != 0;  // a - N borrows out of the top limb
}

// out of Montgomery form AND fully reduced: canonical, < N
template <class M>
OG_HD Fe<M> fe_from_mont(const Fe<M>& a) {
  Fe<M> o = Fe<M>::zero();
  o.l[0] = 1;
  return fe_canon(fe_mul(a, o));  // (a + m N) / R <= N
}

// a^(N-2) by square-and-multiply over the constant exponent (a != 0).  Inlined, with ROLLED loops
// (one squaring + one multiplication body): device-function calls are avoided throughout the EC code
// (see ec.hip.h), and rolled loops keep the code small.  The exponent is read from the 32-byte form of N.
template <class M>
OG_HD Fe<M> fe_inv(const Fe<M>& a) {
  Fe<M> r = Fe<M>::one();
#pragma unroll 1
  for (int w = 8; w >= 0; w--) {
    uint32_t limb = 0;  // limb w of N - 2, without dynamic indexing of a private array (N[0] >= 2)
#pragma unroll
    for (int k = 0; k < 9; k++) limb = (k == w) ? (k == 0 ? M::N[0] - 2u : M::N[k]) : limb;
#pragma unroll 1
    for (int b = 28; b >= 0; b--) {
      r = fe_sqr(r);
      if ((limb >> b) & 1) r = fe_mul(r, a);  // wave-uniform branch: the exponent is a constant
    }
  }
  return r;
}

// ---- 256-bit words <-> limbs -----------------------------------------------------

// w: 8 x u32 little-endian value < 2^256 -> 9 normalized limbs
template <class M>
OG_HD Fe<M> fe_from_words(const uint32_t w[8]) {
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, k = bit >> 5, s = bit & 31;
    uint32_t v = w[k] >> s;
    if (s > 3 && k + 1 < 8) v |= w[k + 1] << (32 - s);
    r.l[i] = v & MASK29;
  }
  return r;
}

// normalized limbs of a value < 2^256 -> 8 x u32
template <class M>
OG_HD void fe_to_words(uint32_t w[8], const Fe<M>& a) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, i = bit / 29, s = bit - 29 * i;  // word k starts at bit s of limb i
    uint32_t v = a.l[i] >> s;
    if (i + 1 < 9) v |= a.l[i + 1] << (29 - s);
    if (29 - s + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - s);
    w[k] = v;
  }
}

// small canonical constants
template <class M>
OG_HD Fe<M> fe_from_u32(uint32_t v) {
  Fe<M> r = Fe<M>::zero();
  r.l[0] = v & MASK29;
  r.l[1] = v >> 29;
  return r;
}

// ---- global / LDS memory (32 B per element, 16-byte vector accesses) ---------------

template <class M>
OG_HD Fe<M> fe_load(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return fe_from_words<M>(w);
}

template <class M>
OG_HD void fe_store(void* p, const Fe<M>& r) {
  uint32_t w[8];
  fe_to_words(w, r);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

typedef Fe<FqParams> Fq;
typedef Fe<FrParams> Fr;

}  // namespace og
