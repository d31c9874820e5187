// Radix-2 Fr NTT for gfx950 and the Groth16 H-polynomial pipeline.
// The field is the reference's `Fp` (/root/reference/src/blockchain/tx/owshen_airdrop/
// babyjubjub/mod.rs:7-11: generator 7, so w_n = 7^((r-1)/n)); the reference has no NTT.
//
// Stand-alone transform (og_ntt_fr_d, natural order in and out) = one permuting copy (bit reversal, fused with an
// optional per-index table multiply and Montgomery conversion) + ceil(log n / 10) in-place stage kernels, each running
// up to 10 DIT butterfly stages on a 1024-element tile staged in LDS, + an optional fused post-scale.
//
// The prover's quotient pipeline (h_poly_device) never permutes: an inverse transform runs as DIF (natural in,
// bit-reversed out), the forward coset transform that follows as DIT (bit-reversed in, natural out), and the two meet
// in ONE kernel on the contiguous low-stage tile (DIF stages 9..0, x g^i / n, DIT stages 0..9).  The pointwise
// (a b - c) / Z rides in the store of c's last pass, the final x g^-i / n and the Montgomery exit in the store of the
// last inverse pass, and h leaves in bit-reversed order (the H-query bases are stored in that order, groth16.hip).
// Per proof at d = 2^17: 11 tile passes (read + write 4 MB each) instead of 23, and inside a pass the tile lives in
// LDS as 9 x 29-bit limbs, so the 32-byte <-> limb conversions happen once per pass, not once per butterfly.
#include "ctx.h"
#include "field.hip.h"
#include "msm.hip.h"  // arena_get

namespace og {

constexpr int NTT_TILE_LOG = 10;
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;

// consts buffer layout (Fr, Montgomery): 0 w, 1 w^-1, 2 g, 3 g^-1, 4 n^-1, 5 1/(g^n - 1), 6 one
struct NttPlan {
  int log_n = 0;
  uint8_t* consts = nullptr;
  uint8_t* tw_fwd = nullptr;     // [n/2] w^k
  uint8_t* tw_inv = nullptr;     // [n/2] w^-k
  uint8_t* cs_fwd = nullptr;     // [n] g^i
  uint8_t* cs_fwd_ninv = nullptr;  // [n] g^i / n     (iNTT output -> coset NTT input, fused)
  uint8_t* cs_inv_ninv = nullptr;  // [n] g^-i / n    (coset iNTT post-scale)
  uint8_t* cs_fwd_ninv_br = nullptr;  // [n] g^rev(p) / n   (the same two tables indexed by bit-reversed position:
  uint8_t* cs_inv_ninv_br = nullptr;  // [n] g^-rev(p) / n   the quotient pipeline works on bit-reversed coefficients)
};

__global__ void k_ntt_consts(int log_n, uint8_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // T = (r-1) / 2^28, the odd cofactor
  const uint32_t T[8] = {0x3e1f593fu, 0x9b970914u, 0x833e8487u, 0x181585d2u, 0x85045b68u, 0x131a029bu, 0x0644e72eu, 0x00000003u};
  Fr seven = fe_to_mont(fe_from_u32<FrParams>(7));
  Fr w = Fr::one();
  for (int i = 255; i >= 0; i--) {
    w = fe_sqr(w);
    if ((T[i >> 5] >> (i & 31)) & 1) w = fe_mul(w, seven);
  }
  for (int i = 28; i > log_n; i--) w = fe_sqr(w);
  Fr nn = fe_to_mont(fe_from_u32<FrParams>(1u << log_n));  // log_n <= 28
  Fr gn = seven;
  for (int i = 0; i < log_n; i++) gn = fe_sqr(gn);
  Fr zc = fe_sub(gn, Fr::one());
  fe_store(out + 0 * 32, w);
  fe_store(out + 1 * 32, fe_inv(w));
  fe_store(out + 2 * 32, seven);
  fe_store(out + 3 * 32, fe_inv(seven));
  fe_store(out + 4 * 32, fe_inv(nn));
  fe_store(out + 5 * 32, fe_inv(zc));
  fe_store(out + 6 * 32, Fr::one());
}

// out[i] = consts[ic] * consts[ib]^e, e = i (rev_log = 0) or the rev_log-bit reversal of i
__global__ void __launch_bounds__(256) k_pow_table(const uint8_t* __restrict__ consts, int ib, int ic, uint8_t* __restrict__ out, size_t n,
                                                  int rev_log) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr b = fe_load<FrParams>(consts + ib * 32);
  Fr acc = fe_load<FrParams>(consts + ic * 32);
  for (size_t e = rev_log ? (size_t)(__brevll((unsigned long long)i) >> (64 - rev_log)) : i; e; e >>= 1) {
    if (e & 1) acc = fe_mul(acc, b);
    b = fe_sqr(b);
  }
  fe_store(out + i * 32, acc);
}

// out[g][bitrev(i)] = (to_mont?)(in[g][i]) * table[i]
__global__ void __launch_bounds__(256) k_ntt_prep(const uint8_t* __restrict__ in, size_t in_stride, uint8_t* __restrict__ out,
                                                 size_t out_stride, int log_n, const uint8_t* __restrict__ table, int to_mont) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)1 << log_n;
  if (i >= n) return;
  const int g = blockIdx.y;
  Fr v = fe_load<FrParams>(in + (size_t)g * in_stride + i * 32);
  if (to_mont) v = fe_to_mont(v);
  if (table) v = fe_mul(v, fe_load<FrParams>(table + i * 32));
  size_t j = log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
  fe_store(out + (size_t)g * out_stride + j * 32, v);
}

// data[g][i] = (from_mont?)(data[g][i] * table[i])      (table may be null)
__global__ void __launch_bounds__(256) k_ntt_post(uint8_t* __restrict__ data, size_t stride, size_t n,
                                                 const uint8_t* __restrict__ table, int from_mont) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = blockIdx.y;
  uint8_t* p = data + (size_t)g * stride + i * 32;
  Fr v = fe_load<FrParams>(p);
  if (table) v = fe_mul(v, fe_load<FrParams>(table + i * 32));
  if (from_mont) v = fe_from_mont(v);
  fe_store(p, v);
}

// Stages [s0, s0+ns) of the DIT network on bit-reversed data, in place.  Index split:
// i = (hi << (s0+ns)) | (m << s0) | lo; a workgroup owns all 2^ns values of m for LO_T
// consecutive values of lo (tile = 2^ns * LO_T <= 1024 elements).
__global__ void __launch_bounds__(256) k_ntt_stages(uint8_t* __restrict__ data, size_t stride, int log_n, int s0, int ns,
                                                   const uint8_t* __restrict__ tw) {
  __shared__ __align__(16) uint32_t lds[NTT_TILE * 8];
  const int g = blockIdx.y;
  uint8_t* base = data + (size_t)g * stride;
  const int tile_log = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
  const int tile = 1 << tile_log;
  const int lo_t_log = tile_log - ns;  // LO_T = 2^(tile_log - ns), <= 2^s0 by construction
  const int lo_t = 1 << lo_t_log;
  const size_t chunks_lo = ((size_t)1 << s0) >> lo_t_log;
  const size_t blk = blockIdx.x;
  const size_t hi = blk / chunks_lo, lo_base = (blk % chunks_lo) << lo_t_log;
  const size_t gbase = (hi << (s0 + ns)) | lo_base;
  // load tile
  for (int e = threadIdx.x; e < tile; e += 256) {
    int m = e >> lo_t_log, t = e & (lo_t - 1);
    const uint4* src = reinterpret_cast<const uint4*>(base + (gbase + ((size_t)m << s0) + t) * 32);
    uint4* dst = reinterpret_cast<uint4*>(&lds[e * 8]);
    dst[0] = src[0];
    dst[1] = src[1];
  }
  __syncthreads();
  for (int q = 0; q < ns; q++) {
    const int s = s0 + q;
    for (int b = threadIdx.x; b < tile / 2; b += 256) {
      int t = b & (lo_t - 1), mm = b >> lo_t_log;
      int m0 = ((mm >> q) << (q + 1)) | (mm & ((1 << q) - 1));
      int m1 = m0 | (1 << q);
      // twiddle exponent: j * n / 2^(s+1), j = low s bits of the global index
      size_t j = ((size_t)(m0 & ((1 << q) - 1)) << s0) | (lo_base + t);
      size_t te = j << (log_n - s - 1);
      uint32_t* p0 = &lds[(m0 * lo_t + t) * 8];
      uint32_t* p1 = &lds[(m1 * lo_t + t) * 8];
      Fr u = fe_load<FrParams>(p0);
      Fr v = fe_mul(fe_load<FrParams>(p1), fe_load<FrParams>(tw + te * 32));
      fe_store(p0, fe_add(u, v));
      fe_store(p1, fe_sub(u, v));
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < tile; e += 256) {
    int m = e >> lo_t_log, t = e & (lo_t - 1);
    uint4* dst = reinterpret_cast<uint4*>(base + (gbase + ((size_t)m << s0) + t) * 32);
    const uint4* src = reinterpret_cast<const uint4*>(&lds[e * 8]);
    dst[0] = src[0];
    dst[1] = src[1];
  }
}


// ---- fused stage blocks for the quotient pipeline --------------------------------------------------------------------
// One workgroup owns the tile of stage block [s0, s0 + ns) (same index split as k_ntt_stages) and runs, on data kept in
// LDS as 9 x 29-bit limbs:   [to_mont] -> [DIF stages s0+ns-1 .. s0 with tw_dif] -> [x mid[p]] -> [DIT stages s0 .. s0+ns-1
// with tw_dit] -> [pointwise: (pw_a[p] pw_b[p] - x) zinv] -> [from_mont] -> out.  p = global position of the element.
struct NttBlock {
  const uint8_t* in;
  uint8_t* out;
  size_t stride;
  int log_n, s0, ns;
  const uint8_t* tw_dif;
  const uint8_t* mid;
  const uint8_t* tw_dit;
  const uint8_t* pw_a;
  const uint8_t* pw_b;
  const uint8_t* consts;
  int to_mont, from_mont;
};

constexpr int NTT_LIMBS = 9;

__device__ __forceinline__ Fr lds_get(const uint32_t* p) {
  Fr r;
#pragma unroll
  for (int i = 0; i < NTT_LIMBS; i++) r.l[i] = p[i];
  return r;
}
__device__ __forceinline__ void lds_put(uint32_t* p, const Fr& v) {
#pragma unroll
  for (int i = 0; i < NTT_LIMBS; i++) p[i] = v.l[i];
}

__global__ void __launch_bounds__(256) k_ntt_block(NttBlock a) {
  OG_FILLER_PRIO();  // the quotient shares the chip with the next sub-batch's sorts (priority 3): at priority 0 it was the one starved
  __shared__ uint32_t lds[NTT_TILE * NTT_LIMBS];
  const int g = blockIdx.y;
  const int log_n = a.log_n, s0 = a.s0, ns = a.ns;
  const uint8_t* src = a.in + (size_t)g * a.stride;
  uint8_t* dst = a.out + (size_t)g * a.stride;
  const int tile_log = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
  const int tile = 1 << tile_log;
  const int lo_t_log = tile_log - ns;
  const int lo_t = 1 << lo_t_log;
  const size_t chunks_lo = ((size_t)1 << s0) >> lo_t_log;
  const size_t blk = blockIdx.x;
  const size_t hi = blk / chunks_lo, lo_base = (blk % chunks_lo) << lo_t_log;
  const size_t gbase = (hi << (s0 + ns)) | lo_base;
  for (int e = threadIdx.x; e < tile; e += 256) {
    const int m = e >> lo_t_log, t = e & (lo_t - 1);
    Fr v = fe_load<FrParams>(src + (gbase + ((size_t)m << s0) + t) * 32);
    if (a.to_mont) v = fe_to_mont(v);
    lds_put(&lds[e * NTT_LIMBS], v);
  }
  __syncthreads();
  // two sweeps over the block's stages: DIF (descending) then DIT (ascending); either may be absent
#pragma unroll 1
  for (int sweep = 0; sweep < 2; sweep++) {
    const uint8_t* tw = sweep == 0 ? a.tw_dif : a.tw_dit;
    if (sweep == 1 && a.mid && a.tw_dit) {
      for (int e = threadIdx.x; e < tile; e += 256) {
        const int m = e >> lo_t_log, t = e & (lo_t - 1);
        const size_t p = gbase + ((size_t)m << s0) + t;
        lds_put(&lds[e * NTT_LIMBS], fe_mul(lds_get(&lds[e * NTT_LIMBS]), fe_load<FrParams>(a.mid + p * 32)));
      }
      __syncthreads();
    }
    if (!tw) continue;
#pragma unroll 1
    for (int qq = 0; qq < ns; qq++) {
      const int q = sweep == 0 ? ns - 1 - qq : qq;
      const int s = s0 + q;
      for (int b = threadIdx.x; b < tile / 2; b += 256) {
        const int t = b & (lo_t - 1), mm = b >> lo_t_log;
        const int m0 = ((mm >> q) << (q + 1)) | (mm & ((1 << q) - 1));
        const int m1 = m0 | (1 << q);
        const size_t j = ((size_t)(m0 & ((1 << q) - 1)) << s0) | (lo_base + t);  // low s bits of the global index
        uint32_t* p0 = &lds[(m0 * lo_t + t) * NTT_LIMBS];
        uint32_t* p1 = &lds[(m1 * lo_t + t) * NTT_LIMBS];
        const Fr u = lds_get(p0), x = lds_get(p1);
        if (s == 0) {  // stage 0: every twiddle is w^0 = 1 (wave-uniform branch): no multiplication
          lds_put(p0, fe_add(u, x));
          lds_put(p1, fe_sub(u, x));
          continue;
        }
        const Fr w = fe_load<FrParams>(tw + (j << (log_n - s - 1)) * 32);
        if (sweep == 0) {  // DIF: (u + x, (u - x) w); the difference goes into the product unreduced (u - x + 4N < 6N)
          lds_put(p0, fe_add(u, x));
          lds_put(p1, fe_mul(fe_sub_weak(u, x), w));
        } else {           // DIT: (u + x w, u - x w)
          const Fr v = fe_mul(x, w);
          lds_put(p0, fe_add(u, v));
          lds_put(p1, fe_sub(u, v));
        }
      }
      __syncthreads();
    }
  }
  const uint8_t* pa = a.pw_a ? a.pw_a + (size_t)g * a.stride : nullptr;
  const uint8_t* pb = a.pw_b ? a.pw_b + (size_t)g * a.stride : nullptr;
  for (int e = threadIdx.x; e < tile; e += 256) {
    const int m = e >> lo_t_log, t = e & (lo_t - 1);
    const size_t p = gbase + ((size_t)m << s0) + t;
    Fr v = lds_get(&lds[e * NTT_LIMBS]);
    if (a.mid && !a.tw_dit) v = fe_mul(v, fe_load<FrParams>(a.mid + p * 32));  // no second sweep: scale on the way out
    if (pa) v = fe_mul(fe_sub(fe_mul(fe_load<FrParams>(pa + p * 32), fe_load<FrParams>(pb + p * 32)), v), fe_load<FrParams>(a.consts + 5 * 32));
    if (a.from_mont) v = fe_from_mont(v);
    fe_store(dst + p * 32, v);
  }
}

// ---- radix-4 stage blocks with lazy butterflies (round 4) -----------------------------------------------------------------
// k_ntt_block above spends more on everything around a butterfly's multiplication than on the multiplication: 205
// instructions for the product, ~150 for the fully reduced u + v / u - v (carry pass + conditional subtraction, twice), 36 LDS
// accesses, a twiddle load with its 32-byte -> limb unpacking, index arithmetic -- ~440 per butterfly.  Here a lane takes FOUR
// elements through TWO stages in registers (one LDS round trip, one carry pass, three twiddles per four butterflies: the
// first stage's two butterflies share theirs, the second stage's differ by the fourth root of unity, exponent + n / 4), and
// the additions are lazy:
//   DIT (forward, bit-reversed in): v = x w is a fresh product (< 2N), so u + v and u + (3N - v) only GROW a value by <= 3N per
//        stage: no reduction inside a pass (limbs may reach 2^31 as a product operand; 9 x 2^31 x 2^29 < 2^64 in a column), one
//        carry pass when the four elements go back to LDS, and at the END of a pass -- the 32-byte HBM format holds < 2^256 --
//        four conditional subtractions (32N, 16N, 8N, 4N) bring the <= 34N back under 4N.
//   DIF (inverse, natural in): the sum u + x of two unreduced values DOUBLES, so a double stage reduces its sums (< 8N: two
//        conditional subtractions; < 4N: one) and keeps the "everything stored is < 2N" contract of the old kernel.
// Every constant k N is subtracted through limbs inflated by 2^29 ("k N - b" needs no borrows for a normalized b < k N with a
// strictly smaller top limb: hence 3N for b < 2N, 5N for b < 4N, 40N for b < 34N).  Same values mod N as k_ntt_block at every
// point where the two are compared (pass outputs differ by multiples of N until the pipeline's last product / from_mont).
template <class M>
__host__ __device__ constexpr uint32_t kn_limb(uint32_t k, int i) {  // limb i of k N, normalized (the top limb takes the rest)
  uint64_t c = 0;
  uint32_t out = 0;
  for (int j = 0; j <= i; j++) {
    const uint64_t v = (uint64_t)M::N[j] * k + c;
    out = j == 8 ? (uint32_t)v : (uint32_t)(v & MASK29);
    c = v >> 29;
  }
  return out;
}
template <class M>
__host__ __device__ constexpr uint32_t kn_neg_limb(uint32_t k, int i) {  // k N with every limb below the top inflated by 2^29
  return kn_limb<M>(k, i) + (i < 8 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
}
static_assert(kn_neg_limb<FrParams>(4, 0) == FrParams::NEG4[0] && kn_neg_limb<FrParams>(4, 5) == FrParams::NEG4[5] &&
                  kn_neg_limb<FrParams>(4, 8) == FrParams::NEG4[8] && kn_limb<FrParams>(3, 2) == FrParams::N3[2] &&
                  kn_limb<FrParams>(5, 8) == FrParams::N5[8],
              "k N limbs");

template <uint32_t K>
struct KN {  // k N and its inflated form as compile-time tables
  static constexpr uint32_t L[9] = {kn_limb<FrParams>(K, 0), kn_limb<FrParams>(K, 1), kn_limb<FrParams>(K, 2), kn_limb<FrParams>(K, 3),
                                    kn_limb<FrParams>(K, 4), kn_limb<FrParams>(K, 5), kn_limb<FrParams>(K, 6), kn_limb<FrParams>(K, 7),
                                    kn_limb<FrParams>(K, 8)};
  static constexpr uint32_t NEG[9] = {kn_neg_limb<FrParams>(K, 0), kn_neg_limb<FrParams>(K, 1), kn_neg_limb<FrParams>(K, 2),
                                      kn_neg_limb<FrParams>(K, 3), kn_neg_limb<FrParams>(K, 4), kn_neg_limb<FrParams>(K, 5),
                                      kn_neg_limb<FrParams>(K, 6), kn_neg_limb<FrParams>(K, 7), kn_neg_limb<FrParams>(K, 8)};
};

__device__ __forceinline__ Fr nt_add(const Fr& a, const Fr& b) {  // limb-wise, no carries
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <uint32_t K>
__device__ __forceinline__ Fr nt_sub(const Fr& a, const Fr& b) {  // a + (K N - b), b normalized with a top limb below K N's
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (KN<K>::NEG[i] - b.l[i]);
  return r;
}
__device__ __forceinline__ Fr nt_norm(const Fr& a) {  // the carry pass: limbs < 2^29 below the top
  Fr r;
  normalize29u(r.l, a.l);
  return r;
}
template <uint32_t K>
__device__ __forceinline__ Fr nt_csub(const Fr& a) {  // a normalized: a - K N if that is >= 0, else a
  uint32_t u[9];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t v = (int32_t)a.l[i] - (int32_t)KN<K>::L[i] + c;
    u[i] = i < 8 ? ((uint32_t)v & MASK29) : (uint32_t)v;
    c = i < 8 ? (v >> 29) : (v < 0 ? -1 : 0);
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : u[i];
  return r;
}
__device__ __forceinline__ Fr nt_reduce_8n(const Fr& a) { return nt_csub<2>(nt_csub<4>(nt_norm(a))); }                          // < 8N -> < 2N
__device__ __forceinline__ Fr nt_reduce_64n(const Fr& a) { return nt_csub<4>(nt_csub<8>(nt_csub<16>(nt_csub<32>(a)))); }         // normalized < 64N -> < 4N

__global__ void __launch_bounds__(256) k_ntt_block4(NttBlock a) {
  OG_FILLER_PRIO();
  __shared__ uint32_t lds[NTT_TILE * NTT_LIMBS];
  const int g = blockIdx.y;
  const int log_n = a.log_n, s0 = a.s0, ns = a.ns;
  const uint8_t* src = a.in + (size_t)g * a.stride;
  uint8_t* dst = a.out + (size_t)g * a.stride;
  const int tile_log = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
  const int tile = 1 << tile_log;
  const int lo_t_log = tile_log - ns;
  const int lo_t = 1 << lo_t_log;
  const size_t chunks_lo = ((size_t)1 << s0) >> lo_t_log;
  const size_t blk = blockIdx.x;
  const size_t hi = blk / chunks_lo, lo_base = (blk % chunks_lo) << lo_t_log;
  const size_t gbase = (hi << (s0 + ns)) | lo_base;
  const size_t quarter = ((size_t)1 << log_n) >> 2;
  for (int e = threadIdx.x; e < tile; e += 256) {
    const int m = e >> lo_t_log, t = e & (lo_t - 1);
    Fr v = fe_load<FrParams>(src + (gbase + ((size_t)m << s0) + t) * 32);
    if (a.to_mont) v = fe_to_mont(v);
    lds_put(&lds[e * NTT_LIMBS], v);
  }
  __syncthreads();
  auto elem = [&](int m, int t) -> uint32_t* { return &lds[(m * lo_t + t) * NTT_LIMBS]; };
  // ---------------- DIF sweep (stages ns - 1 .. 0), everything stored < 2N ----------------
  if (a.tw_dif) {
    const uint8_t* tw = a.tw_dif;
    int q = ns - 2;
#pragma unroll 1
    for (; q >= 0; q -= 2) {  // stages q + 1, then q
      const int s = s0 + q;
      for (int gi = threadIdx.x; gi < tile / 4; gi += 256) {
        const int t = gi & (lo_t - 1), mm = gi >> lo_t_log;
        const int base = ((mm >> q) << (q + 2)) | (mm & ((1 << q) - 1));
        uint32_t *p00 = elem(base, t), *p01 = elem(base | (1 << q), t), *p10 = elem(base | (2 << q), t), *p11 = elem(base | (3 << q), t);
        const size_t j1 = ((size_t)(base & ((1 << q) - 1)) << s0) | (lo_base + t);
        const size_t te2 = j1 << (log_n - s - 2);
        Fr x0 = lds_get(p00), x1 = lds_get(p01), x2 = lds_get(p10), x3 = lds_get(p11);
        // stage q + 1: (x0, x2) and (x1, x3)
        const Fr w2 = fe_load<FrParams>(tw + te2 * 32), w3 = fe_load<FrParams>(tw + (te2 + quarter) * 32);
        const Fr a1 = nt_add(x0, x2), b1 = nt_add(x1, x3);                 // < 4N, limbs < 2^30
        const Fr c1 = fe_mul(nt_sub<4>(x0, x2), w2), d1 = fe_mul(nt_sub<4>(x1, x3), w3);  // operands < 6N; results < 2N
        // stage q: (a1, b1) and (c1, d1)
        const Fr b1n = nt_norm(b1);
        Fr y0 = nt_reduce_8n(nt_add(a1, b1)), y2 = nt_csub<2>(nt_norm(nt_add(c1, d1))), y1, y3;
        if (s == 0) {  // the global stage 0: every twiddle is 1 -- differences only, reduced
          y1 = nt_csub<2>(nt_csub<4>(nt_csub<8>(nt_norm(nt_sub<5>(a1, b1n)))));  // a1 - b1 + 5N < 9N
          y3 = fe_sub(c1, d1);
        } else {
          const Fr w1 = fe_load<FrParams>(tw + (j1 << (log_n - s - 1)) * 32);
          y1 = fe_mul(nt_sub<5>(a1, b1n), w1);   // operand < 9N, limbs < 2^31
          y3 = fe_mul(nt_sub<4>(c1, d1), w1);
        }
        lds_put(p00, y0); lds_put(p01, y1); lds_put(p10, y2); lds_put(p11, y3);
      }
      __syncthreads();
    }
    if (q == -1) {  // ns odd: stage 0 of the block alone
      const int s = s0;
      for (int b = threadIdx.x; b < tile / 2; b += 256) {
        const int t = b & (lo_t - 1), mm = b >> lo_t_log;
        uint32_t *p0 = elem(mm << 1, t), *p1 = elem((mm << 1) | 1, t);
        const Fr u = lds_get(p0), x = lds_get(p1);
        lds_put(p0, fe_add(u, x));
        if (s == 0) lds_put(p1, fe_sub(u, x));
        else lds_put(p1, fe_mul(nt_sub<4>(u, x), fe_load<FrParams>(tw + ((size_t)(lo_base + t) << (log_n - s - 1)) * 32)));
      }
      __syncthreads();
    }
  }
  if (a.mid && a.tw_dit) {
    for (int e = threadIdx.x; e < tile; e += 256) {
      const int m = e >> lo_t_log, t = e & (lo_t - 1);
      const size_t p = gbase + ((size_t)m << s0) + t;
      lds_put(&lds[e * NTT_LIMBS], fe_mul(lds_get(&lds[e * NTT_LIMBS]), fe_load<FrParams>(a.mid + p * 32)));
    }
    __syncthreads();
  }
  // ---------------- DIT sweep (stages 0 .. ns - 1), lazy: a value grows by <= 3N per stage ----------------
  if (a.tw_dit) {
    const uint8_t* tw = a.tw_dit;
    int q = 0;
#pragma unroll 1
    for (; q + 1 < ns; q += 2) {  // stages q, then q + 1
      const int s = s0 + q;
      for (int gi = threadIdx.x; gi < tile / 4; gi += 256) {
        const int t = gi & (lo_t - 1), mm = gi >> lo_t_log;
        const int base = ((mm >> q) << (q + 2)) | (mm & ((1 << q) - 1));
        uint32_t *p00 = elem(base, t), *p01 = elem(base | (1 << q), t), *p10 = elem(base | (2 << q), t), *p11 = elem(base | (3 << q), t);
        const size_t j1 = ((size_t)(base & ((1 << q) - 1)) << s0) | (lo_base + t);
        const size_t te2 = j1 << (log_n - s - 2);
        const Fr x0 = lds_get(p00), x1 = lds_get(p01), x2 = lds_get(p10), x3 = lds_get(p11);
        Fr v1 = x1, v3 = x3;
        if (s != 0) {  // (the global stage 0 has twiddle 1: x1, x3 enter as they are -- normalized, < 4N: nt_sub<5> below)
          const Fr w1 = fe_load<FrParams>(tw + (j1 << (log_n - s - 1)) * 32);
          v1 = fe_mul(x1, w1);
          v3 = fe_mul(x3, w1);
        }
        const Fr a1 = nt_add(x0, v1), c1 = nt_add(x2, v3);
        const Fr b1 = s != 0 ? nt_sub<3>(x0, v1) : nt_sub<5>(x0, v1), d1 = s != 0 ? nt_sub<3>(x2, v3) : nt_sub<5>(x2, v3);
        const Fr w2 = fe_load<FrParams>(tw + te2 * 32), w3 = fe_load<FrParams>(tw + (te2 + quarter) * 32);
        const Fr v2 = fe_mul(c1, w2), v4 = fe_mul(d1, w3);  // operand limbs < 2^31
        lds_put(p00, nt_norm(nt_add(a1, v2)));
        lds_put(p10, nt_norm(nt_sub<3>(a1, v2)));
        lds_put(p01, nt_norm(nt_add(b1, v4)));
        lds_put(p11, nt_norm(nt_sub<3>(b1, v4)));
      }
      __syncthreads();
    }
    if (q < ns) {  // ns odd: the block's last stage alone
      const int s = s0 + q;
      for (int b = threadIdx.x; b < tile / 2; b += 256) {
        const int t = b & (lo_t - 1), mm = b >> lo_t_log;
        const int m0 = ((mm >> q) << (q + 1)) | (mm & ((1 << q) - 1));
        uint32_t *p0 = elem(m0, t), *p1 = elem(m0 | (1 << q), t);
        const size_t j = ((size_t)(m0 & ((1 << q) - 1)) << s0) | (lo_base + t);
        const Fr u = lds_get(p0), x = lds_get(p1);
        if (s == 0) {
          lds_put(p0, nt_norm(nt_add(u, x)));
          lds_put(p1, nt_norm(nt_sub<5>(u, x)));
        } else {
          const Fr v = fe_mul(x, fe_load<FrParams>(tw + (j << (log_n - s - 1)) * 32));
          lds_put(p0, nt_norm(nt_add(u, v)));
          lds_put(p1, nt_norm(nt_sub<3>(u, v)));
        }
      }
      __syncthreads();
    }
  }
  const uint8_t* pa = a.pw_a ? a.pw_a + (size_t)g * a.stride : nullptr;
  const uint8_t* pb = a.pw_b ? a.pw_b + (size_t)g * a.stride : nullptr;
  const bool lazy = a.tw_dit != nullptr;  // the tile holds DIT outputs: normalized limbs, values < 4N + 3N ns (+ 2N) <= 36N
  for (int e = threadIdx.x; e < tile; e += 256) {
    const int m = e >> lo_t_log, t = e & (lo_t - 1);
    const size_t p = gbase + ((size_t)m << s0) + t;
    Fr v = lds_get(&lds[e * NTT_LIMBS]);
    if (a.mid && !a.tw_dit) v = fe_mul(v, fe_load<FrParams>(a.mid + p * 32));  // no second sweep: scale on the way out
    if (pa) {  // (pa pb - v) / Z: the difference through 40N - v (v < 36N), one product reduces it
      const Fr ab = fe_mul(fe_load<FrParams>(pa + p * 32), fe_load<FrParams>(pb + p * 32));
      v = fe_mul(lazy ? nt_sub<40>(ab, v) : nt_sub<4>(ab, v), fe_load<FrParams>(a.consts + 5 * 32));
    } else if (lazy) {
      v = nt_reduce_64n(v);  // the 32-byte format holds < 2^256 ~ 5.3N: back under 4N
    }
    if (a.from_mont) v = fe_from_mont(v);
    fe_store(dst + p * 32, v);
  }
}

// out[g][rev(i)] = in[g][i]   (only the C ABI's og_h_poly_d needs natural order)
__global__ void __launch_bounds__(256) k_bitrev_copy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t stride, int log_n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((size_t)1 << log_n)) return;
  const size_t j = log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
  const uint4* s4 = reinterpret_cast<const uint4*>(in + (size_t)blockIdx.y * stride + i * 32);
  uint4* d4 = reinterpret_cast<uint4*>(out + (size_t)blockIdx.y * stride + j * 32);
  d4[0] = s4[0];
  d4[1] = s4[1];
}

static int ntt_plan(og_ctx* ctx, int log_n, NttPlan* out) {
  OG_REQUIRE(log_n >= 0 && log_n <= 28, "ntt: log_n must be 0..28 (Fr 2-adicity)");
  const size_t n = (size_t)1 << log_n;
  std::string key = "ntt" + std::to_string(log_n);
  NttPlan p;
  p.log_n = log_n;
  bool fresh = !arena_has(ctx, (key + ".consts").c_str());
  OG_TRY(arena_get(ctx, (key + ".consts").c_str(), 8 * 32, (void**)&p.consts));
  OG_TRY(arena_get(ctx, (key + ".twf").c_str(), (n / 2 + 1) * 32, (void**)&p.tw_fwd));
  OG_TRY(arena_get(ctx, (key + ".twi").c_str(), (n / 2 + 1) * 32, (void**)&p.tw_inv));
  OG_TRY(arena_get(ctx, (key + ".csf").c_str(), n * 32, (void**)&p.cs_fwd));
  OG_TRY(arena_get(ctx, (key + ".csfn").c_str(), n * 32, (void**)&p.cs_fwd_ninv));
  OG_TRY(arena_get(ctx, (key + ".csin").c_str(), n * 32, (void**)&p.cs_inv_ninv));
  OG_TRY(arena_get(ctx, (key + ".csfnb").c_str(), n * 32, (void**)&p.cs_fwd_ninv_br));
  OG_TRY(arena_get(ctx, (key + ".csinb").c_str(), n * 32, (void**)&p.cs_inv_ninv_br));
  if (fresh) {
    hipLaunchKernelGGL(k_ntt_consts, dim3(1), dim3(64), 0, ctx->stream, log_n, p.consts);
    size_t h = n / 2 ? n / 2 : 1;
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(h, 256)), dim3(256), 0, ctx->stream, p.consts, 0, 6, p.tw_fwd, h, 0);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(h, 256)), dim3(256), 0, ctx->stream, p.consts, 1, 6, p.tw_inv, h, 0);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 2, 6, p.cs_fwd, n, 0);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 2, 4, p.cs_fwd_ninv, n, 0);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 2, 4, p.cs_fwd_ninv_br, n, log_n);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 3, 4, p.cs_inv_ninv_br, n, log_n);
    hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 3, 4, p.cs_inv_ninv, n, 0);
    OG_HIP(hipGetLastError());
  }
  *out = p;
  return OG_OK;
}

// consts buffer of the size-2^log_n domain (see NttPlan): used by the Groth16 setup helpers
int ntt_domain_consts(og_ctx* ctx, int log_n, uint8_t** consts_d) {
  NttPlan p;
  OG_TRY(ntt_plan(ctx, log_n, &p));
  *consts_d = p.consts;
  return OG_OK;
}

// in -> (prep: optional to_mont, optional table) -> out (bit reversed) -> stages in place on out
static int ntt_core(og_ctx* ctx, const NttPlan& p, const uint8_t* in, size_t in_stride, uint8_t* out, size_t out_stride,
                    int batch, bool inverse, const uint8_t* prep_table, int to_mont) {
  const int log_n = p.log_n;
  const size_t n = (size_t)1 << log_n;
  hipLaunchKernelGGL(k_ntt_prep, dim3(grid_for(n, 256), batch), dim3(256), 0, ctx->stream, in, in_stride, out, out_stride,
                     log_n, prep_table, to_mont);
  OG_HIP(hipGetLastError());
  const uint8_t* tw = inverse ? p.tw_inv : p.tw_fwd;
  const int tile_log = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
  for (int s0 = 0; s0 < log_n;) {
    int ns = log_n - s0 < tile_log ? log_n - s0 : tile_log;
    // the tile needs LO_T = 2^(tile_log - ns) <= 2^s0 low values; shrink ns is never needed because
    // s0 >= tile_log whenever ns < tile_log
    size_t nblocks = n >> tile_log;
    hipLaunchKernelGGL(k_ntt_stages, dim3((unsigned)nblocks, batch), dim3(256), 0, ctx->stream, out, out_stride, log_n, s0, ns, tw);
    OG_HIP(hipGetLastError());
    s0 += ns;
  }
  return OG_OK;
}

static int ntt_post(og_ctx* ctx, uint8_t* data, size_t stride, size_t n, int batch, const uint8_t* table, int from_mont) {
  hipLaunchKernelGGL(k_ntt_post, dim3(grid_for(n, 256), batch), dim3(256), 0, ctx->stream, data, stride, n, table, from_mont);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// Standalone transform on canonical data (C ABI og_ntt_fr_d): out-of-place into out_d.
int ntt_canonical(og_ctx* ctx, const uint8_t* in_d, uint8_t* out_d, int log_n, int batch, int inverse, int coset) {
  NttPlan p;
  OG_TRY(ntt_plan(ctx, log_n, &p));
  const size_t n = (size_t)1 << log_n, stride = n * 32;
  if (!inverse) {
    OG_TRY(ntt_core(ctx, p, in_d, stride, out_d, stride, batch, false, coset ? p.cs_fwd : nullptr, 1));
    OG_TRY(ntt_post(ctx, out_d, stride, n, batch, nullptr, 1));
  } else {
    OG_TRY(ntt_core(ctx, p, in_d, stride, out_d, stride, batch, true, nullptr, 1));
    // x n^-1 (and g^-i for the coset variant), back to canonical
    uint8_t* ninv_tab = nullptr;
    if (!coset) {
      OG_TRY(arena_get(ctx, ("ntt" + std::to_string(log_n) + ".ninv").c_str(), n * 32, (void**)&ninv_tab));
      hipLaunchKernelGGL(k_pow_table, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, p.consts, 6, 4, ninv_tab, n, 0);
      OG_HIP(hipGetLastError());
    }
    OG_TRY(ntt_post(ctx, out_d, stride, n, batch, coset ? p.cs_inv_ninv : ninv_tab, 1));
  }
  return OG_OK;
}

// H-polynomial: a, b, c = evaluations over the size-d domain (Montgomery form, batch x d x 32 B, destroyed);
// h_out = canonical coefficients of (A*B - C)/Z in BIT-REVERSED order (position p holds coefficient rev(p)),
// batch x d x 32 B.  3 iNTT + 3 coset NTT + pointwise + 1 coset iNTT (arkworks convention), as fused
// DIF / DIT stage blocks (see the head of this file).  tmp is unused (kept for the callers' scratch layout).
// gates (optional, 4 events): the launches are cut into four runs and run j waits for gates[j] first -- the pipelined prover
// passes "the math stream is about to launch the G1 accumulation A | B1 | L | H of the PREVIOUS sub-batch", so that the
// passes start beside a G1 accumulation (which leaves the LDS free) and not in front of the G2 one (whose accumulators fill it).
int h_poly_device(og_ctx* ctx, uint8_t* a, uint8_t* b, uint8_t* c, uint8_t* tmp, uint8_t* h_out, int log_d, int batch,
                  const hipEvent_t* gates) {
  (void)tmp;
  NttPlan p;
  OG_TRY(ntt_plan(ctx, log_d, &p));
  const size_t d = (size_t)1 << log_d, stride = d * 32;
  const int tile_log = log_d < NTT_TILE_LOG ? log_d : NTT_TILE_LOG;
  const unsigned nblocks = (unsigned)(d >> tile_log);
  // stage blocks, ascending: [0, tile_log), then chunks of up to tile_log stages
  int bs[8], bn[8], nb = 0;
  for (int s0 = 0; s0 < log_d || nb == 0;) {
    const int ns = log_d - s0 < tile_log ? log_d - s0 : tile_log;
    bs[nb] = s0; bn[nb] = ns; nb++;
    s0 += ns;
    if (ns == 0) break;
  }
  const int n_launch = 3 * (2 * nb - 1) + nb;
  int launched = 0, next_gate = 0;
  const bool radix4 = OG_HOOK_INT("OG_NTT_RADIX4", 1) != 0;  // hooks builds: 0 = the radix-2 kernel (read per call: tests run both)
  auto launch = [&](const NttBlock& blk) -> int {
    while (gates && next_gate < 4 && launched >= (next_gate * n_launch + 3) / 4) {  // 11 launches: runs of 3, 3, 3, 2
      if (gates[next_gate]) OG_HIP(hipStreamWaitEvent(ctx->stream, gates[next_gate], 0));
      next_gate++;
    }
    launched++;
    if (radix4)
      hipLaunchKernelGGL(k_ntt_block4, dim3(nblocks, batch), dim3(256), 0, ctx->stream, blk);
    else
      hipLaunchKernelGGL(k_ntt_block, dim3(nblocks, batch), dim3(256), 0, ctx->stream, blk);
    OG_HIP(hipGetLastError());
    return OG_OK;
  };
  auto block = [&](uint8_t* data, int k) {
    NttBlock x;
    x.in = data; x.out = data; x.stride = stride; x.log_n = log_d; x.s0 = bs[k]; x.ns = bn[k];
    x.tw_dif = nullptr; x.mid = nullptr; x.tw_dit = nullptr; x.pw_a = nullptr; x.pw_b = nullptr; x.consts = p.consts;
    x.to_mont = 0; x.from_mont = 0;
    return x;
  };
  uint8_t* arr[3] = {a, b, c};
  for (int k = 0; k < 3; k++) {
    // evaluations -> coefficients (DIF, high blocks first) -> x g^i / n -> coset evaluations (DIT, low block first)
    for (int j = nb - 1; j >= 1; j--) {
      NttBlock x = block(arr[k], j);
      x.tw_dif = p.tw_inv;
      OG_TRY(launch(x));
    }
    NttBlock m = block(arr[k], 0);
    m.tw_dif = p.tw_inv; m.mid = p.cs_fwd_ninv_br; m.tw_dit = p.tw_fwd;
    if (k == 2 && nb == 1) { m.pw_a = a; m.pw_b = b; }
    OG_TRY(launch(m));
    for (int j = 1; j < nb; j++) {
      NttBlock x = block(arr[k], j);
      x.tw_dit = p.tw_fwd;
      if (k == 2 && j == nb - 1) { x.pw_a = a; x.pw_b = b; }  // c <- (a b - c) / Z on the way out
      OG_TRY(launch(x));
    }
  }
  // coset evaluations of h (in c) -> coefficients, bit-reversed, x g^-i / n, canonical
  for (int j = nb - 1; j >= 1; j--) {
    NttBlock x = block(c, j);
    x.tw_dif = p.tw_inv;
    OG_TRY(launch(x));
  }
  NttBlock f = block(c, 0);
  f.tw_dif = p.tw_inv; f.mid = p.cs_inv_ninv_br; f.from_mont = 1; f.out = h_out;
  OG_TRY(launch(f));
  return OG_OK;
}


__global__ void __launch_bounds__(256) k_to_mont_copy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store(out + i * 32, fe_to_mont(fe_load<FrParams>(in + i * 32)));
}

// C ABI og_h_poly_d: canonical evaluations in (not modified), canonical coefficients out
int h_poly_canonical(og_ctx* ctx, const uint8_t* a, const uint8_t* b, const uint8_t* c, int log_d, int batch, uint8_t* h_out) {
  const size_t d = (size_t)1 << log_d, tot = d * (size_t)batch;
  uint8_t* buf[4];
  const char* names[4] = {"hpoly.a", "hpoly.b", "hpoly.c", "hpoly.t"};
  for (int k = 0; k < 4; k++) OG_TRY(arena_get(ctx, names[k], tot * 32, (void**)&buf[k]));
  const uint8_t* src[3] = {a, b, c};
  for (int k = 0; k < 3; k++) {
    hipLaunchKernelGGL(k_to_mont_copy, dim3(grid_for(tot, 256)), dim3(256), 0, ctx->stream, src[k], buf[k], tot);
    OG_HIP(hipGetLastError());
  }
  OG_TRY(h_poly_device(ctx, buf[0], buf[1], buf[2], buf[3], buf[3], log_d, batch, nullptr));  // bit-reversed coefficients
  hipLaunchKernelGGL(k_bitrev_copy, dim3(grid_for(d, 256), batch), dim3(256), 0, ctx->stream, buf[3], h_out, d * 32, log_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
