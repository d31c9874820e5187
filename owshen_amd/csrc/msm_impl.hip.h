// Templated MSM kernels + launchers, instantiated once per group in msm_g1.hip / msm_g2.hip
// (separate translation units so the two big instantiations compile in parallel).
#pragma once
#include "msm.hip.h"
#include "ec.hip.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace og {

constexpr int HEAVY = 2048;  // bucket sizes above this go to the workgroup-per-bucket path (OG_HEAVY overrides: tests)

// ---- bucket accumulation ----------------------------------------------------------
template <class T> struct AccCfg;
// HEAVY_MINW: the G1 heavy-bucket kernel is held to 96 registers (5 waves per SIMD; ~140 B of scratch per lane on a kernel
// that is 0.7 % of the step) so that it fits the 104 registers a persistent G1 accumulation leaves on every SIMD
// OG_TAIL_MINW (A/B build, round 4): waves per SIMD the G1 tail kernels (reduction levels, heavy combine, window combine) are
// held to -- 5 = at most 102 registers, i.e. what a persistent G1 accumulation leaves free, at the price of scratch spills
#ifndef OG_TAIL_MINW
#define OG_TAIL_MINW 1
#endif
// OG_RED_MINW (A/B build, round 5): waves per SIMD of the G1 running-sum kernel -- 1: 178 registers, two waves, no scratch;
// 3: 168 registers + 48 B of scratch, three waves
#ifndef OG_RED_MINW
#define OG_RED_MINW OG_TAIL_MINW
#endif
template <> struct AccCfg<Fq> { static constexpr int MINW = 1, ALT_MINW = 5, RED_MINW = OG_RED_MINW, RED_ALT = 2, HEAVY_MINW = 5, TAIL_MINW = OG_TAIL_MINW; };
template <> struct AccCfg<Fq2> { static constexpr int MINW = 2, ALT_MINW = 3, RED_MINW = 2, RED_ALT = 1, HEAVY_MINW = 2, TAIL_MINW = 1; };

// Gather-power probe (round 6, hooks build on the GPU only): OG_GATHER_MASK=m folds every table index onto
// `index & m`, i.e. onto a slice of the window tables small enough to stay in L2 / Infinity Cache.  The instruction stream and
// the VALU work are the same (one s_load + one v_and_b32 more per gather in BOTH arms: the unmasked arm runs with m = 2^32 - 1),
// the 1.4 TB/s of random 64-byte HBM gathers are gone -- and the results are of course wrong, so the probe times steps
// without verifying them (tools/gather_power_probe.sh).  What it measures is what the gathers cost the accumulation kernels
// through the package power cap: clock and time with and without them, on one box.
#if defined(OG_AB_HOOKS) && !defined(OG_HIPEMU)
static __device__ uint32_t og_gather_mask_d = 0xffffffffu;
#define OG_GATHER_INDEX(e) (((e) >> 1) & og_gather_mask_d)
static inline int gather_mask_apply() {  // (per translation unit: G1 and G2 each carry their own copy of the symbol)
  static long long applied = -1;
  const long long m = OG_HOOK_INT("OG_GATHER_MASK", 0xffffffffll);
  if (m == applied) return OG_OK;
  const uint32_t m32 = (uint32_t)m;
  OG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(og_gather_mask_d), &m32, sizeof m32));
  applied = m;
  return OG_OK;
}
#else
#define OG_GATHER_INDEX(e) ((e) >> 1)
static inline int gather_mask_apply() { return OG_OK; }
#endif

// entry e = (table index << 1) | sign: the base is gathered as stored, the sign goes to the group law (lazy negation)
template <class T>
__device__ __forceinline__ Affine<T> gather_base(const uint8_t* __restrict__ tab, uint32_t e) {
  return Affine<T>::load(tab + (size_t)OG_GATHER_INDEX(e) * Affine<T>::BYTES);
}

// ---- persistent form ------------------------------------------------------------------------------------------------------
// The grid form above launches one one-wave workgroup per 64 buckets (10^5 of them per launch) and lets the dispatcher
// refill every wave slot the moment it frees.  Two costs, both measured in round 2's rocprof trace: (1) whatever else is
// queued on the other streams -- the digit sorts of the next sub-batch, this MSM's own heavy-bucket / reduction tail --
// makes NO progress until the accumulation kernel has no workgroup left to dispatch (a 6.7 ms sort kernel took 167 ms
// under it and finished exactly when the accumulation drained), so the "pipeline" degenerates to time slicing at kernel
// boundaries; (2) residency is 3.1 of 4 waves per SIMD, the rest lost between a wave's exit and its successor's launch.
// Here a launch is `P` one-wave workgroups that stay resident and take (chunk, proof) work items from a counter in
// ctrl[1] -- chunk-major, i.e. every proof's largest buckets first -- until none is left.  P is chosen by the host
// (OG_ACC_WAVES_G1 / _G2 waves per CU): below the register limit it leaves wave slots, registers and LDS on every CU to
// the other streams for the whole length of the kernel.
// INTO: the buckets already hold sums (the first query of a merged pair, msm_run_phase) and this launch adds to them
template <class T, int MINW, bool CLAIM = true, bool INTO = false>
__global__ void __launch_bounds__(64, MINW) k_accumulate_p(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ entries, const uint32_t* __restrict__ order,
                                                         size_t nkeys, size_t ecap, uint8_t* __restrict__ buckets,
                                                         uint32_t* __restrict__ ctrl, uint32_t* __restrict__ heavy_list,
                                                         uint32_t heavy_cap, uint32_t heavy_min, uint32_t nchunk, uint32_t batch) {
  __shared__ uint32_t w_s;
  const uint32_t total = nchunk * batch;
  // G1: the body needs 127 registers, i.e. a 128-register allocation and FOUR waves per SIMD.  A launch of 12 waves per CU
  // is 3 per SIMD on an empty chip (og_ubench_coresidency: a second stream's kernels then run beside it at 0.7x of their
  // solo speed), but the prover launches it while the previous kernels still hold slots, the dispatcher packs 4 waves
  // onto the SIMDs that happen to be free, and a filler workgroup -- one wave on EACH SIMD of a CU -- no longer fits
  // anywhere (round 3 trace: a 0.05 ms scan took 56 ms).  Claiming register 135 makes the allocation 136: at most THREE
  // waves fit a SIMD wherever they land, and 104 registers + 5 wave slots per SIMD always remain for the fillers.
  // (CLAIM = false: a lone MSM has nothing queued beside it -- four waves per SIMD, 16 per CU)
  if constexpr (CLAIM && std::is_same<T, Fq>::value) OG_CLAIM_VGPR(135);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) w_s = atomicAdd(&ctrl[1], 1u);
    __syncthreads();
    const uint32_t w = w_s;
    if (w >= total) return;
    const uint32_t chunk = w / batch, g = w - chunk * batch;
    size_t key = (size_t)chunk * 64 + threadIdx.x;
    const bool live = key < nkeys;
    uint32_t lo = 0, hi = 0;
    const uint32_t* ent = entries + (size_t)g * ecap;
    if (live) {
      if (order) key = order[(size_t)g * nkeys + key];
      const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
      lo = off[key];
      hi = off[key + 1];
      if (hi - lo > heavy_min) {
        const uint32_t slot = atomicAdd(&ctrl[0], 1u);
        if (slot < heavy_cap) {
          heavy_list[2 * slot] = g;
          heavy_list[2 * slot + 1] = (uint32_t)key;
          hi = lo;
        }
      }
    }
    XYZZ<T> acc = XYZZ<T>::inf();
    if constexpr (INTO) {
      if (live) acc = XYZZ<T>::load(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
    }
#pragma unroll 1
    for (uint32_t p = lo; p < hi; p++) {  // (a software prefetch of the next base was measured in round 3: 144 registers, same time)
      const uint32_t e = ent[p];
      acc = xyzz_madd_signed(acc, gather_base<T>(tab, e), e & 1);
    }
    if (live) acc.store(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
  }
}


// ---- lone big MSM over plain bases: position-major pieces ---------------------------------------------------------------
// At 2^26 points the accumulation is not VALU-bound: every (point, window) pair gathers a 64-byte base once, 2^30 gathers from
// a 4 GB table -- far beyond L2 / Infinity Cache -- and both round 3's forms run at the chip's uncached random-gather rate
// (~0.8 TB/s of 64-byte reads: 11 - 13 G additions/s against 17 G/s when the table is cache resident).  But a base is needed
// by ALL 16 windows.  The sort leaves every bucket's entries ordered by point index, so piece q of P of a bucket -- entries
// [len q / P, len (q + 1) / P) -- lies near position q / P of the table whatever the bucket: work items are handed out
// piece-major (every bucket's piece 0, then every bucket's piece 1, ...), the resident waves of the whole chip sweep the table
// together, and a base fetched for one window is still in the Infinity Cache when the other 15 windows ask for it.  Pieces
// are cut by entry count, so the lanes of a wave walk chains of equal length; a piece's sum goes to pieces[q][key] and
// k_pieces_combine adds a bucket's P sums (P - 1 full additions per bucket: 0.3 % of the work).
template <class T, int MINW>
__global__ void __launch_bounds__(64, MINW) k_accumulate_pieces(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ entries, size_t nkeys, uint8_t* __restrict__ pieces,
                                                              uint32_t* __restrict__ ctrl, uint32_t* __restrict__ heavy_list, uint32_t heavy_cap,
                                                              uint32_t heavy_min, uint32_t nchunk, uint32_t npiece) {
  __shared__ uint32_t w_s;
  const uint32_t total = nchunk * npiece;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) w_s = atomicAdd(&ctrl[1], 1u);
    __syncthreads();
    const uint32_t w = w_s;
    if (w >= total) return;
    const uint32_t q = w / nchunk, chunk = w - q * nchunk;  // piece-major
    const size_t key = (size_t)chunk * 64 + threadIdx.x;
    const bool live = key < nkeys;
    uint32_t s0 = 0, s1 = 0;
    if (live) {
      const uint32_t lo = offsets[key], len = offsets[key + 1] - lo;
      if (len > heavy_min) {  // an outlier (the top window's few long buckets, the "digit 1" bucket): the heavy path, listed once
        if (q == 0) {
          const uint32_t slot = atomicAdd(&ctrl[0], 1u);
          if (slot < heavy_cap) {
            heavy_list[2 * slot] = 0u;
            heavy_list[2 * slot + 1] = (uint32_t)key;
          } else {
            s0 = lo; s1 = lo + len;  // list full: this lane walks the whole bucket as piece 0 (slow but correct)
          }
        }
      } else {
        s0 = lo + (uint32_t)((uint64_t)len * q / npiece);
        s1 = lo + (uint32_t)((uint64_t)len * (q + 1) / npiece);
      }
    }
    XYZZ<T> acc = XYZZ<T>::inf();
#pragma unroll 1
    for (uint32_t p = s0; p < s1; p++) {
      const uint32_t e = entries[p];
      acc = xyzz_madd_signed(acc, gather_base<T>(tab, e), e & 1);
    }
    if (live) acc.store(pieces + ((size_t)q * nkeys + key) * XYZZ<T>::BYTES);
  }
}

template <class T>
__global__ void __launch_bounds__(64) k_pieces_combine(const uint8_t* __restrict__ pieces, size_t nkeys, uint32_t npiece, uint8_t* __restrict__ buckets) {
  const size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= nkeys) return;
  XYZZ<T> acc = XYZZ<T>::load(pieces + key * XYZZ<T>::BYTES);
#pragma unroll 1
  for (uint32_t q = 1; q < npiece; q++) acc = xyzz_add(acc, XYZZ<T>::load(pieces + ((size_t)q * nkeys + key) * XYZZ<T>::BYTES));
  acc.store(buckets + key * XYZZ<T>::BYTES);
}

// Heavy buckets (the boolean-wire bucket: ~10 % of a proof's scalars land in it) are cut into HEAVY_SPLIT segments, one
// 128-lane workgroup each: a launch has one heavy bucket per proof, i.e. only a few hundred of them, and one workgroup per
// bucket left three quarters of the SIMDs without a wave (round 2 profile: 2.1 ms / 7.1 ms per launch in G1 / G2).
// k_accumulate_heavy writes one partial sum per segment, k_heavy_combine adds a bucket's partials.
constexpr int HEAVY_SPLIT = 8;
#ifndef OG_HEAVY_BLOCK
#define OG_HEAVY_BLOCK 128  // (A/B builds, 2^26-point MSM, round 3: 64 lanes 94-98 ms, 128 lanes 83-85 ms, 256 lanes 88 ms in this kernel)
#endif
constexpr int HEAVY_BLOCK = OG_HEAVY_BLOCK;

// Segment plan by SIZE (a lone MSM's heavy list mixes a few thousand buckets of 16 K entries with one of millions -- the
// "digit 1" bucket -- and eight segments of that one kept a single workgroup busy for 23 ms): bucket h gets
// min(HEAVY_SEG_MAX, ceil(len / HEAVY_SEG_ENTRIES)) segments, seg_off = their exclusive prefix sums (seg_off[nh] = total).
// If that would not fit the parts array, every bucket falls back to `split` segments.  One workgroup.
constexpr uint32_t HEAVY_SEG_ENTRIES = 4096, HEAVY_SEG_MAX = 128;
static __global__ void __launch_bounds__(1024) k_heavy_plan(const uint32_t* __restrict__ offsets, size_t nkeys, const uint32_t* __restrict__ heavy_count,
                                                    const uint32_t* __restrict__ heavy_list, uint32_t heavy_cap, uint32_t split,
                                                    uint32_t parts_cap, uint32_t* __restrict__ seg_off, uint32_t seg_entries) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t fallback;
  uint32_t nh = *heavy_count;
  if (nh > heavy_cap) nh = heavy_cap;
  const uint32_t t = threadIdx.x, per = (nh + 1023) / 1024;
  const uint32_t lo = t * per < nh ? t * per : nh, hi = lo + per < nh ? lo + per : nh;
  auto nseg = [&](uint32_t h, bool fb) -> uint32_t {
    if (fb) return split;
    const uint32_t* off = offsets + (size_t)heavy_list[2 * h] * (nkeys + 1);
    const uint32_t key = heavy_list[2 * h + 1], len = off[key + 1] - off[key];
    const uint32_t want = (len + seg_entries - 1) / seg_entries;
    return want < 1 ? 1 : (want > HEAVY_SEG_MAX ? HEAVY_SEG_MAX : want);
  };
  for (int pass = 0; pass < 2; pass++) {
    const bool fb = pass == 1;
    uint32_t s = 0;
    for (uint32_t h = lo; h < hi; h++) s += nseg(h, fb);
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t v = (int)t >= d ? part[t - d] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    if (t == 0) fallback = part[1023] > parts_cap ? 1u : 0u;
    __syncthreads();
    if (pass == 0 && fallback) continue;  // (uniform)
    uint32_t run = t ? part[t - 1] : 0;
    for (uint32_t h = lo; h < hi; h++) {
      seg_off[h] = run;
      run += nseg(h, fb);
    }
    if (t == 1023) seg_off[nh] = part[1023];
    break;
  }
}

// the heavy bucket that holds work item w: largest h with seg_off[h] <= w
__device__ __forceinline__ uint32_t heavy_of_item(const uint32_t* seg_off, uint32_t nh, uint32_t w) {
  uint32_t lo = 0, hi = nh;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_off[mid] <= w) lo = mid; else hi = mid;
  }
  return lo;
}

template <class T, int MINW>
__global__ void __launch_bounds__(HEAVY_BLOCK, MINW) k_accumulate_heavy(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ entries, size_t nkeys, size_t ecap,
                                                         uint8_t* __restrict__ parts, const uint32_t* __restrict__ heavy_count,
                                                         const uint32_t* __restrict__ heavy_list, uint32_t heavy_cap, uint32_t split_,
                                                         const uint32_t* __restrict__ seg_off) {
  OG_FILLER_PRIO();
  OG_DYN_LDS(smem);
  uint32_t nh = *heavy_count;
  if (nh > heavy_cap) nh = heavy_cap;
  const uint32_t total = seg_off ? (nh ? seg_off[nh] : 0u) : nh * split_;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    uint32_t h, seg, split;
    if (seg_off) {
      h = heavy_of_item(seg_off, nh, w);
      seg = w - seg_off[h];
      split = seg_off[h + 1] - seg_off[h];
    } else {
      h = w / split_; seg = w % split_; split = split_;
    }
    const uint32_t g = heavy_list[2 * h], key = heavy_list[2 * h + 1];
    const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
    const uint32_t* ent = entries + (size_t)g * ecap;
    const uint32_t lo = off[key], len = off[key + 1] - lo;
    const uint32_t s0 = lo + (uint32_t)((uint64_t)len * seg / split), s1 = lo + (uint32_t)((uint64_t)len * (seg + 1) / split);
    XYZZ<T> acc = XYZZ<T>::inf();
    for (uint32_t p = s0 + threadIdx.x; p < s1; p += blockDim.x) {
      const uint32_t e = ent[p];
      acc = xyzz_madd_signed(acc, gather_base<T>(tab, e), e & 1);
    }
#pragma unroll 1
    for (int d = blockDim.x / 2; d >= 1; d >>= 1) {
      __syncthreads();
      if ((int)threadIdx.x >= d && (int)threadIdx.x < 2 * d) acc.store(smem + (size_t)(threadIdx.x - d) * XYZZ<T>::BYTES);
      __syncthreads();
      if ((int)threadIdx.x < d) acc = xyzz_add(acc, XYZZ<T>::load(smem + (size_t)threadIdx.x * XYZZ<T>::BYTES));
    }
    if (threadIdx.x == 0) acc.store(parts + (size_t)w * XYZZ<T>::BYTES);
    __syncthreads();
  }
}

template <class T>
__global__ void __launch_bounds__(64, AccCfg<T>::TAIL_MINW) k_heavy_combine(const uint8_t* __restrict__ parts, const uint32_t* __restrict__ heavy_count,
                                                     const uint32_t* __restrict__ heavy_list, uint32_t heavy_cap, size_t nkeys,
                                                     uint8_t* __restrict__ buckets, uint32_t split_, const uint32_t* __restrict__ seg_off,
                                                     uint32_t into) {
  OG_FILLER_PRIO();
  uint32_t nh = *heavy_count;
  if (nh > heavy_cap) nh = heavy_cap;
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nh) return;
  const size_t first = seg_off ? seg_off[h] : (size_t)h * split_;
  const uint32_t split = seg_off ? seg_off[h + 1] - seg_off[h] : split_;
  const uint32_t g = heavy_list[2 * h], key = heavy_list[2 * h + 1];
  uint8_t* bucket = buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES;
  XYZZ<T> acc = XYZZ<T>::load(parts + first * XYZZ<T>::BYTES);
#pragma unroll 1
  for (uint32_t s = 1; s < split + into; s++)  // (into: the bucket holds the first query's sum -- the accumulation left it untouched)
    acc = xyzz_add(acc, XYZZ<T>::load(s < split ? parts + (first + s) * XYZZ<T>::BYTES : bucket));
  acc.store(bucket);
}

// ---- bucket reduction ---------------------------------------------------------------
// V(x) = sum_i i x_i and T(x) = sum_i x_i over `nsets` independent arrays (the MSM wants
// sum_b (b+1) B_b = V(B) + T(B)).  Split x into segments of SEG elements: with t_j the segment totals and
// v_j = sum_{i in seg j} (i - j SEG) x_i the local zero-based weighted sums,
//     V(x) = T(v) + SEG * V(t),      T(x) = T(t),
// so each level needs two group additions per element (descending running sum: run += x_i; acc += run),
// half of what a radix-4 tree of (run, acc) pairs costs, and the recursion continues on the SEG-times
// shorter array t.  The "T(v)" terms of all levels are folded into one carry array
//     u_1 = v_1,   u_k[j] = sum_{i in seg j} u_{k-1}[i] + SEG^(k-1) v_k[j]      =>   V(x) = u_L[0], T(x) = t_L[0].
// Every kernel has ONE inlined group-law site driven by a rolled op loop (see ec.hip.h).
#ifndef OG_SEG_LOG
#define OG_SEG_LOG 3  // A/B builds: make EXTRA=-DOG_SEG_LOG=4
#endif
constexpr int SEG_LOG = OG_SEG_LOG;
constexpr int SEG = 1 << SEG_LOG;

// One XYZZ<Fq2> per lane parked in LDS, limb-major / lane-minor (conflict-free).  The G2 running-sum kernel keeps `run` and
// `acc` there: two live extended points are 144 registers, and with the 250 the addition itself wants the 256-register
// budget spilled 776 B per lane to scratch (round 2 listing); a slot index is LDS address arithmetic, so the one inlined
// addition site can still serve both.
struct LdsXyzz2 {
  uint32_t* p;  // &lds[threadIdx.x]; word w of slot s at p[(s * 72 + w) * 64]
  __device__ __forceinline__ Fq get1(int s, int c) const {
    Fq r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = p[((s * 8 + c) * 9 + k) * 64];
    return r;
  }
  __device__ __forceinline__ void put1(int s, int c, const Fq& v) const {
#pragma unroll
    for (int k = 0; k < 9; k++) p[((s * 8 + c) * 9 + k) * 64] = v.l[k];
  }
  __device__ __forceinline__ XYZZ<Fq2> get(int s) const {
    return {{get1(s, 0), get1(s, 1)}, {get1(s, 2), get1(s, 3)}, {get1(s, 4), get1(s, 5)}, {get1(s, 6), get1(s, 7)}};
  }
  __device__ __forceinline__ void put(int s, const XYZZ<Fq2>& v) const {
    put1(s, 0, v.x.c0); put1(s, 1, v.x.c1); put1(s, 2, v.y.c0); put1(s, 3, v.y.c1);
    put1(s, 4, v.zz.c0); put1(s, 5, v.zz.c1); put1(s, 6, v.zzz.c0); put1(s, 7, v.zzz.c1);
  }
};


// ---- G2 bucket accumulation with the accumulator in LDS ----------------------------------------------------------------
// The Fq2 mixed addition wants ~370 registers; capped at 256 (2 waves / SIMD) the register version spills 108 B per lane
// and writes ~50 GB of scratch per launch.  Here the running XYZZ<Fq2> accumulator (72 limbs) lives in LDS (LdsXyzz2,
// slot 0; limb-major / lane-minor: conflict-free), each coordinate fetched where the formula consumes it and written back
// as soon as it is final: 28 B of scratch left, 18 KiB of LDS per one-wave workgroup (8 per CU = the same 2 waves / SIMD).
// Measured: 404 -> 390 ms per 1024 proofs once the kernel ran in one-wave workgroups (with 4-wave groups, where residency
// was the limit, it made no difference).  OG_G2_LDS=0 selects the register version.
// PERSIST: the launch is a fixed number of resident one-wave workgroups that take (chunk, proof) work items from the
// counter in heavy_count[1], chunk-major (see k_accumulate_p); otherwise one workgroup per work item (blockIdx).
template <int MINW, bool PERSIST>
__global__ void __launch_bounds__(64, MINW) k_accumulate_g2_lds(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ entries, const uint32_t* __restrict__ order,
                                                              size_t nkeys, size_t ecap, uint8_t* __restrict__ buckets,
                                                              uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list,
                                                              uint32_t heavy_cap, uint32_t heavy_min, uint32_t nchunk, uint32_t batch) {
  __shared__ uint32_t lds[72 * 64];
  __shared__ uint32_t w_s;
  typedef Fq2 T;
  for (;;) {
  size_t key;
  uint32_t g;
  if constexpr (PERSIST) {
    __syncthreads();
    if (threadIdx.x == 0) w_s = atomicAdd(&heavy_count[1], 1u);
    __syncthreads();
    const uint32_t w = w_s;
    if (w >= nchunk * batch) return;
    const uint32_t chunk = w / batch;
    g = w - chunk * batch;
    key = (size_t)chunk * 64 + threadIdx.x;
  } else {
    key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    g = blockIdx.y;
  }
  const bool live = key < nkeys;
  uint32_t lo = 0, hi = 0;
  const uint32_t* ent = entries + (size_t)g * ecap;
  if (live) {
    if (order) key = order[(size_t)g * nkeys + key];
    const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
    lo = off[key];
    hi = off[key + 1];
    if (hi - lo > heavy_min) {
      const uint32_t slot = atomicAdd(heavy_count, 1u);
      if (slot < heavy_cap) {
        heavy_list[2 * slot] = (uint32_t)g;
        heavy_list[2 * slot + 1] = (uint32_t)key;
        hi = lo;
      }
    }
  }
  const LdsXyzz2 acc{&lds[threadIdx.x]};
  auto coord = [&](int j) -> T { return {acc.get1(0, 2 * j), acc.get1(0, 2 * j + 1)}; };
  auto set = [&](int j, const T& v) {
    acc.put1(0, 2 * j, v.c0);
    acc.put1(0, 2 * j + 1, v.c1);
  };
  bool inf = true;
#pragma unroll 1
  for (uint32_t p = lo; p < hi; p++) {
    const uint32_t e = ent[p];
    const bool neg = e & 1;
    const Affine<T> q = gather_base<T>(tab, e);
    if (q.is_inf()) continue;
    if (inf) {
      acc.put(0, XYZZ<T>::from_affine(neg ? affine_neg(q) : q));
      inf = false;
      continue;
    }
    const T P = f_mul_minus(q.x, coord(2), coord(0));          // U2 - X1 + 4N
    const T R = f_mul_minus_y(q.y, neg, coord(3), coord(1));   // S2 - Y1 + 4N
    if (f_weak_diff_is_zero(P)) {                              // q = +-acc: rare
      if (f_weak_diff_is_zero(R)) acc.put(0, xyzz_dbl_affine(neg ? affine_neg(q) : q)); else inf = true;
      continue;
    }
    const T PP = f_sqr(P);
    const T PPP = f_mul(PP, P);
    set(2, f_mul(coord(2), PP));                               // ZZ3
    set(3, f_mul(coord(3), PPP));                              // ZZZ3
    const T X1 = coord(0);
    const T X3 = f_sqr_sub(R, PP, f_add2_weak(P, X1));         // R^2 - PP (P + 2 X1)
    const T D = f_mul_minus(X1, PP, X3);                       // Q - X3 + 4N
    set(0, X3);
    set(1, f_mul_sub(R, D, coord(1), PPP));                    // R D - Y1 PPP
  }
  XYZZ<T> out = XYZZ<T>::inf();
  if (!inf) out = acc.get(0);
  if (live) out.store(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
  if constexpr (!PERSIST) return;
  }
}

// t_out[set][j] = sum of segment j, v_out[set][j] = sum_{i} i_local * x_i      (G2: run / acc live in LDS, see LdsXyzz2)
template <int MINW>
__global__ void __launch_bounds__(64, MINW) k_seg_runacc_g2(const uint8_t* __restrict__ items, size_t n_in, size_t n_out, size_t nsets,
                                                          uint8_t* __restrict__ t_out, uint8_t* __restrict__ v_out) {
  OG_FILLER_PRIO();
  typedef Fq2 T;
  __shared__ uint32_t lds[2 * 72 * 64];
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * nsets) return;
  const size_t set = t / n_out, u = t % n_out;
  const size_t base = set * n_in + u * SEG;
  const LdsXyzz2 slot{&lds[threadIdx.x]};
  slot.put(0, XYZZ<T>::inf());
  slot.put(1, XYZZ<T>::inf());
#pragma unroll 1
  for (int s = 0; s < 2 * SEG - 1; s++) {  // even s: run (slot 0) += x[SEG-1 - s/2]; odd s: acc (slot 1) += run
    const bool to_run = !(s & 1);
    const int i = SEG - 1 - (s >> 1);
    if (to_run && u * SEG + i >= n_in) continue;
    XYZZ<T> rhs;
    if (to_run) rhs = XYZZ<T>::load(items + (base + i) * XYZZ<T>::BYTES); else rhs = slot.get(0);
    slot.put(to_run ? 0 : 1, xyzz_add(slot.get(to_run ? 0 : 1), rhs));
  }
  slot.get(0).store(t_out + (set * n_out + u) * XYZZ<T>::BYTES);
  slot.get(1).store(v_out + (set * n_out + u) * XYZZ<T>::BYTES);
}

// One inlined addition site serves both sums WITHOUT selecting its left operand at run time: the state is the pair (U, V) =
// (run, acc) before an even step and (acc, run) before an odd one; every step computes W = U + rhs (rhs = the next element on
// even steps, V = run on odd ones) and rotates (U, V) := (V, W).  Round 4's form -- xyzz_add(to_run ? run : acc, ...), then
// "if (to_run) run = res; else acc = res" -- made the compiler keep run and acc in scratch memory (179 registers + 292 B per
// lane: two 36-word points), and every step of this latency-bound kernel (two waves per SIMD) then waited for a scratch
// round trip.  An element past the end is the point at infinity (the addition returns its other operand).
template <class T, int MINW>
__global__ void __launch_bounds__(64, MINW) k_seg_runacc(const uint8_t* __restrict__ items, size_t n_in, size_t n_out, size_t nsets,
                                                  uint8_t* __restrict__ t_out, uint8_t* __restrict__ v_out) {
  OG_FILLER_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * nsets) return;
  const size_t set = t / n_out, u = t % n_out;
  const size_t base = set * n_in + u * SEG;
  XYZZ<T> U = XYZZ<T>::inf(), V = XYZZ<T>::inf();
#pragma unroll 1
  for (int s = 0; s < 2 * SEG - 1; s++) {  // even s: run += x[SEG-1 - s/2]; odd s: acc += run
    const int i = SEG - 1 - (s >> 1);
    XYZZ<T> rhs = V;
    if (!(s & 1)) {
      rhs = XYZZ<T>::inf();
      if (u * SEG + i < n_in) rhs = XYZZ<T>::load(items + (base + i) * XYZZ<T>::BYTES);
    }
    const XYZZ<T> W = xyzz_add(U, rhs);
    U = V;
    V = W;
  }
  V.store(t_out + (set * n_out + u) * XYZZ<T>::BYTES);  // the loop ends on an even step: (U, V) = (acc, run)
  U.store(v_out + (set * n_out + u) * XYZZ<T>::BYTES);
}

// u_out[set][j] = (carry ? sum_{i in seg j} carry[set][i] : 0) + 2^shift * v[set][j]
template <class T, int MINW>
__global__ void __launch_bounds__(64, MINW) k_seg_carry(const uint8_t* __restrict__ carry, size_t n_in, const uint8_t* __restrict__ v,
                                                 size_t n_out, size_t nsets, int shift, uint8_t* __restrict__ u_out) {
  OG_FILLER_PRIO();
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * nsets) return;
  const size_t set = t / n_out, u = t % n_out;
  XYZZ<T> acc = XYZZ<T>::load(v + (set * n_out + u) * XYZZ<T>::BYTES);
#pragma unroll 1
  for (int s = 0; s < shift + SEG; s++) {  // s < shift: acc += acc; then acc += carry[u SEG + (s - shift)]
    const int i = s - shift;
    if (i >= 0 && (!carry || u * SEG + i >= n_in)) continue;
    XYZZ<T> rhs = acc;
    if (i >= 0) rhs = XYZZ<T>::load(carry + (set * n_in + u * SEG + i) * XYZZ<T>::BYTES);
    acc = xyzz_add(acc, rhs);
  }
  acc.store(u_out + (set * n_out + u) * XYZZ<T>::BYTES);
}

// ---- scan-shaped bucket reduction for launches of a few bucket sets (round 4) ----------------------------------------------
// The segmented running sums above are work-efficient (2.15 additions per bucket) but DEEP: 137 dependent additions for 2^15
// buckets, whatever the number of sets -- 2.5 ms of one request's G2 query, the longest chain of its proof after the witness.
// With one set (a request) or sixteen (a lone MSM over plain bases) the chip is idle anyway, so depth is what counts:
//     sum_b (b + 1) B_b = sum_j S_j,   S_j = sum_{b >= j} B_b   (suffix sums)
// k_scan_reduce: a workgroup takes a block of bs <= 256 buckets into LDS, suffix-scans it (log2 bs steps, every lane adds) and
// tree-sums the suffixes (log2 bs steps): total_blk = S'_0 and P_blk = sum_i S'_i.  Across blocks
//     sum_j S_j = sum_blk P_blk + bs * sum_{m >= 1} m total_m,   and   sum_m m total_m = (sum_m (m + 1) total_m) - sum_m total_m
// is the same problem on the totals: the SAME kernel runs once more, on [totals | parts] as 2 x nsets arrays of nb <= 256
// elements (totals -> T2 = sum total, P2 = sum (m + 1) total; parts -> T3 = sum P), and k_scan_reduce_final gives
// T3 + bs (P2 - T2).  Depth 2 log2(bs) + 2 log2(nb) + log2(bs) + 2 = 40 additions for 2^15 buckets, ~12 additions of work per
// bucket: used for <= SCAN_SETS sets.  One inlined addition site per kernel (ec.hip.h).
template <class T>
__global__ void __launch_bounds__(256) k_scan_reduce(const uint8_t* __restrict__ items, uint32_t n_in, uint32_t bs, uint8_t* __restrict__ totals,
                                                    uint8_t* __restrict__ parts) {
  OG_FILLER_PRIO();
  OG_DYN_LDS(smem);
  const uint32_t nb = n_in / bs, blk = blockIdx.x, i = threadIdx.x;
  const size_t set = blockIdx.y;
  XYZZ<T> x = XYZZ<T>::inf();
  if (i < bs) x = XYZZ<T>::load(items + (set * n_in + (size_t)blk * bs + i) * XYZZ<T>::BYTES);
  int lg = 0;
  while ((1u << lg) < bs) lg++;
#pragma unroll 1
  for (int s2 = 0; s2 < 2 * lg; s2++) {
    const bool scan = s2 < lg;
    const uint32_t d = scan ? (1u << s2) : (bs >> (s2 - lg + 1));
    if (s2 == lg && i == 0) x.store(totals + (set * nb + blk) * XYZZ<T>::BYTES);  // the block's own suffix sum S'_0 = its total
    // publish: the scan publishes every lane's value in place, the tree the upper half of what is left
    if (scan ? i < bs : (i >= d && i < 2 * d)) x.store(smem + (size_t)(scan ? i : i - d) * XYZZ<T>::BYTES);
    __syncthreads();
    const bool active = scan ? i + d < bs : i < d;
    XYZZ<T> y = XYZZ<T>::inf();
    if (active) y = XYZZ<T>::load(smem + (size_t)(scan ? i + d : i) * XYZZ<T>::BYTES);
    __syncthreads();
    x = xyzz_add(x, y);
  }
  if (i == 0) {
    if (lg == 0) x.store(totals + (set * nb + blk) * XYZZ<T>::BYTES);
    x.store(parts + (set * nb + blk) * XYZZ<T>::BYTES);
  }
}

// out[set] = T3 + bs (P2 - T2): tp2 / pp2 = the second level's totals / parts, [2][nsets] (totals' row first); ssum[set] = infinity
template <class T>
__global__ void __launch_bounds__(64) k_scan_reduce_final(const uint8_t* __restrict__ t2, const uint8_t* __restrict__ p2, uint32_t nsets,
                                                         int log_bs, uint8_t* __restrict__ gsum, uint8_t* __restrict__ ssum) {
  OG_FILLER_PRIO();
  const uint32_t set = blockIdx.x * blockDim.x + threadIdx.x;
  if (set >= nsets) return;
  XYZZ<T> acc = XYZZ<T>::load(p2 + (size_t)set * XYZZ<T>::BYTES);                 // P2 = sum (m + 1) total_m
#pragma unroll 1
  for (int s2 = 0; s2 < log_bs + 2; s2++) {  // s2 = 0: acc -= T2; 1 .. log_bs: acc += acc; last: acc += T3
    XYZZ<T> rhs = acc;
    if (s2 == 0) rhs = xyzz_neg(XYZZ<T>::load(t2 + (size_t)set * XYZZ<T>::BYTES));
    if (s2 == log_bs + 1) rhs = XYZZ<T>::load(t2 + (size_t)(nsets + set) * XYZZ<T>::BYTES);
    acc = xyzz_add(acc, rhs);
  }
  acc.store(gsum + (size_t)set * XYZZ<T>::BYTES);
  XYZZ<T>::inf().store(ssum + (size_t)set * XYZZ<T>::BYTES);
}

// result[g] = sum_k 2^(c k) * (G_k + S_k) over the nsets_per_g window sets (Horner), one lane per g
template <class T>
__global__ void __launch_bounds__(64, AccCfg<T>::TAIL_MINW) k_window_combine(const uint8_t* __restrict__ gsum, const uint8_t* __restrict__ ssum,
                                                      int nsets_per_g, int c, int batch, uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  // Horner over the windows with one inlined add site: per window c doublings (acc += acc, skipped for the
  // top window), then acc += G_k, acc += S_k.
  XYZZ<T> acc = XYZZ<T>::inf();
  const int per = c + 2;
#pragma unroll 1
  for (int s = 0; s < nsets_per_g * per; s++) {
    const int k = nsets_per_g - 1 - s / per, q = s % per;
    if (q < c && k == nsets_per_g - 1) continue;
    const size_t idx = (size_t)g * nsets_per_g + k;
    if (q < c) {  // (a doubling proper: 2M + 5S instead of the general addition's detour through "the operands are equal")
      acc = xyzz_dbl(acc);
      continue;
    }
    const XYZZ<T> rhs = XYZZ<T>::load((q == c ? gsum : ssum) + idx * XYZZ<T>::BYTES);
    acc = xyzz_add(acc, rhs);
  }
  acc.store(out + (size_t)g * XYZZ<T>::BYTES);
}

// window-sharded MSM, this rank's share: out[g][k] = G + S of window k's bucket set if this rank owns window k, else the
// point at infinity (slots = nwin); with precomputed tables there is one bucket set and one slot
template <class T>
__global__ void __launch_bounds__(64) k_window_points(const uint8_t* __restrict__ gsum, const uint8_t* __restrict__ ssum, int nsets_per_g,
                                                     int slots, uint32_t own, int precomp, int batch, uint8_t* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * slots) return;
  const int g = t / slots, k = t % slots;
  XYZZ<T> acc = XYZZ<T>::inf();
  if (precomp || ((own >> k) & 1u)) {
    const size_t idx = (size_t)g * nsets_per_g + (precomp ? 0 : (size_t)__popc(own & ((1u << k) - 1u)));
    acc = xyzz_add(XYZZ<T>::load(gsum + idx * XYZZ<T>::BYTES), XYZZ<T>::load(ssum + idx * XYZZ<T>::BYTES));
  }
  acc.store(out + (size_t)t * XYZZ<T>::BYTES);
}

// out[g] = sum_k 2^(c k) sum_r gathered[r][g][k]: Horner over the window slots, every rank's partial added per slot
// (one inlined add site: per slot c doublings, skipped for the top slot, then `world` additions)
template <class T>
__global__ void __launch_bounds__(64) k_partial_combine(const uint8_t* __restrict__ gathered, int world, int slots, int c, int batch,
                                                       uint8_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  XYZZ<T> acc = XYZZ<T>::inf();
  const int per = c + world;
#pragma unroll 1
  for (int s = 0; s < slots * per; s++) {
    const int k = slots - 1 - s / per, q = s % per;
    if (q < c && k == slots - 1) continue;
    XYZZ<T> rhs = acc;
    if (q >= c) rhs = XYZZ<T>::load(gathered + (((size_t)(q - c) * batch + g) * slots + k) * XYZZ<T>::BYTES);
    acc = xyzz_add(acc, rhs);
  }
  acc.store(out + (size_t)g * XYZZ<T>::BYTES);
}

// out[g] = sum_r partial[r][g]: the ranks' partial sums of one query after an all-gather (window-sharded PROVING, groth16.hip:
// over per-window tables every rank's share of an MSM is ONE point per proof, so there is no Horner step -- the partials just
// add).  `rank_stride` = bytes between two ranks' arrays (a rank's block holds all five queries' arrays).  One addition site.
template <class T>
__global__ void __launch_bounds__(64) k_sum_ranks(const uint8_t* __restrict__ gathered, size_t rank_stride, int world, int batch,
                                                 uint8_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  XYZZ<T> acc = XYZZ<T>::load(gathered + (size_t)g * XYZZ<T>::BYTES);
#pragma unroll 1
  for (int r = 1; r < world; r++) acc = xyzz_add(acc, XYZZ<T>::load(gathered + (size_t)r * rank_stride + (size_t)g * XYZZ<T>::BYTES));
  acc.store(out + (size_t)g * XYZZ<T>::BYTES);
}

template <class T>
int msm_sum_ranks_t(og_ctx* ctx, const uint8_t* gathered_d, size_t rank_stride, int world, int batch, uint8_t* out_d) {
  hipLaunchKernelGGL(k_sum_ranks<T>, dim3(grid_for(batch, 64)), dim3(64), 0, ctx->stream, gathered_d, rank_stride, world, batch, out_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

template <class T>
static void launch_runacc(bool alt, dim3 grid, hipStream_t st, const uint8_t* items, size_t n_in, size_t n_out, size_t nsets, uint8_t* to,
                          uint8_t* vo) {
#ifdef OG_AB_HOOKS
  if (alt) {  // OG_RED_ALT: G2 the register version (776 B of scratch per lane at 2 waves / SIMD), G1 two waves per SIMD
    if constexpr (std::is_same<T, Fq2>::value)
      hipLaunchKernelGGL((k_seg_runacc<T, AccCfg<T>::RED_MINW>), grid, dim3(64), 0, st, items, n_in, n_out, nsets, to, vo);
    else
      hipLaunchKernelGGL((k_seg_runacc<T, AccCfg<T>::RED_ALT>), grid, dim3(64), 0, st, items, n_in, n_out, nsets, to, vo);
    return;
  }
#endif
  (void)alt;
  if constexpr (std::is_same<T, Fq2>::value)
    hipLaunchKernelGGL((k_seg_runacc_g2<1>), grid, dim3(64), 0, st, items, n_in, n_out, nsets, to, vo);  // 308 registers, no scratch
  else
    hipLaunchKernelGGL((k_seg_runacc<T, AccCfg<T>::RED_MINW>), grid, dim3(64), 0, st, items, n_in, n_out, nsets, to, vo);
}

template <class T>
int msm_combine_t(og_ctx* ctx, const og_bases* bases, const uint8_t* gathered_d, int world, int batch, uint8_t* out_d) {
  const int slots = bases->precomp ? 1 : bases->nwin;
  hipLaunchKernelGGL(k_partial_combine<T>, dim3(grid_for(batch, 64)), dim3(64), 0, ctx->stream, gathered_d, world, slots, bases->c,
                     batch, out_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
#include "msm_ab.hip.h"
namespace og {

// phase (msm.hip.h): MSM_FULL one query; MSM_FIRST / MSM_SECOND the two halves of a merged pair that share ONE bucket set -- the
// first accumulates (and folds its heavy buckets in, on the issuing stream) and stops; the second adds to the same buckets and
// runs the one reduction.
template <class T>
int msm_run_t(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_d, bool partial, int phase) {
  const size_t PB = XYZZ<T>::BYTES;
  const size_t B = (size_t)1 << (ds.c - 1);
  const int nsets_per_g = ds.precomp ? 1 : std::max(1, ds.n_own);
  const size_t nsets = (size_t)ds.batch * nsets_per_g;
  const char* sfx = bases->is_g2 ? "2" : "1";
  uint8_t* buckets = nullptr;
  uint32_t* heavy = nullptr;
  // (hooks builds: OG_HEAVY (threshold) and OG_HEAVY_CAP (list capacity) make the overflow path reachable at toy sizes)
  const uint32_t heavy_cap = (uint32_t)std::max<long long>(1, OG_HOOK_INT("OG_HEAVY_CAP", 1 << 16));
  const bool heavy_forced = OG_HOOK_SET("OG_HEAVY");
  uint32_t heavy_min = (uint32_t)std::max<long long>(1, OG_HOOK_INT("OG_HEAVY", HEAVY));
  // A launch of a few bucket sets (one request, a handful of requests) does not fill the chip, and its length is the
  // longest chain a single lane walks: a bucket of 190 entries -- the "digit 1" bucket of the selector bits -- is 2.3 ms
  // of dependent G2 additions.  There, everything above twice the average bucket goes to the workgroup-per-bucket path
  // (128 lanes x 8 segments, a tree and a combine: ~0.1 ms whatever the size).  Throughput launches keep the 2048 bound:
  // their waves hold buckets of similar size (`order`), so a medium bucket costs no more than its additions.
  if (!heavy_forced && (double)nsets * (double)B <= 4.0 * ctx->n_cu * 256) {
    const double avg = (double)ds.n * (ds.precomp ? ds.n_own : 1) / (double)B;
    heavy_min = (uint32_t)std::min<double>((double)HEAVY, std::max(32.0, 2.0 * avg));
  }
  // A lone big MSM over plain bases (2^26 points: 2^19 buckets of ~2048 entries): the ordinary buckets are walked one lane
  // each -- chains of equal length -- and only the outliers (the top window's few, long buckets; the "digit 1" bucket) go to
  // the heavy path: everything above four times the average bucket.
  const double lone_avg = (double)ds.n / (double)B;
  const bool lone_plain = !ds.precomp && ds.batch == 1 && lone_avg >= OG_HOOK_DBL("OG_LONE_AVG", 256.0);
  if (!heavy_forced && lone_plain) heavy_min = (uint32_t)std::max<double>((double)HEAVY, 4.0 * lone_avg);
  // With a side stream for the tail (ctx->tail_stream, set by the batched prover) the heavy buckets, the bucket reduction
  // and the window combine of THIS MSM run under the bucket accumulation of the NEXT one, so the buffers they read get a
  // per-query name (ctx->msm_tag) instead of being shared by consecutive MSMs.
  const std::string tag = ctx->tail_stream ? std::string(sfx) + "." + std::to_string(ctx->msm_tag) : std::string(sfx);
  // (a merged pair keeps its bucket set under a name of its own: on one stream other G1 queries may run between its halves)
  OG_TRY(arena_get(ctx, ("msm.buckets" + tag + (phase == MSM_FULL ? "" : ".pair")).c_str(), nsets * B * PB, (void**)&buckets));
  const std::string ph = phase == MSM_SECOND ? "b" : "";  // (the first half's heavy kernels may still be reading their list)
  OG_TRY(arena_get(ctx, ("msm.heavy" + (ctx->tail_stream ? tag : std::string()) + ph).c_str(), (size_t)(2 * heavy_cap + 4) * 4, (void**)&heavy));
  static const uint32_t heavy_split = (uint32_t)std::max(1, std::min(HEAVY_SPLIT, (int)OG_HOOK_INT("OG_HEAVY_SPLIT", HEAVY_SPLIT)));
  uint8_t* heavy_parts = nullptr;
  OG_TRY(arena_get(ctx, ("msm.heavyparts" + tag + ph).c_str(), std::min<size_t>(heavy_cap, nsets * B) * HEAVY_SPLIT * PB,
                   (void**)&heavy_parts));
  OG_HIP(hipMemsetAsync(heavy, 0, 8, ctx->stream));  // [0] heavy-bucket count, [1] work-item counter of a persistent launch
  OG_TRY(gather_mask_apply());
  uint32_t* heavy_count = heavy;
  uint32_t* heavy_list = heavy + 4;
  {
    ProfScope ps(ctx, bases->is_g2 ? PROF_ACC_G2 : PROF_ACC_G1, (double)ds.n * ds.batch);
    // ONE wave per workgroup, PERSISTENT: `pw` resident one-wave workgroups per CU take (64 buckets, proof) work items from the
    // counter in heavy[1] (see k_accumulate_p).  One wave, because the lanes never cooperate and a 4-wave workgroup needs four
    // wave slots on its CU at once (round 2: G2 500 -> 405 ms, G1 756 -> 717 ms per 1024 proofs); persistent, because whatever
    // is queued on the other streams only runs beside a launch that leaves slots free for its whole length (round 3).
    // G1: 127 registers + the claim = 3 waves per SIMD, 12 per CU; G2 (accumulator in LDS): 8 per CU is the limit.
    // (hooks builds: OG_ACC_WAVES_G1 / _G2 = 0 selects round 2's grid launch, msm_ab.hip.h)
    const int pw = (int)(std::is_same<T, Fq2>::value ? OG_HOOK_INT("OG_ACC_WAVES_G2", 8) : OG_HOOK_INT("OG_ACC_WAVES_G1", 12));
    const int pw_lone = (int)OG_HOOK_INT("OG_ACC_WAVES_LONE", 16);  // nothing runs beside a lone MSM: 16 waves per CU, no register claim
    const uint32_t nchunk = grid_for(ds.nkeys, 64);
    OG_REQUIRE((size_t)nchunk * ds.batch < ((size_t)1 << 32), "msm: too many (bucket chunk, proof) work items for one launch");
    // (A lone huge MSM over per-window TABLES -- 2^26 points: 2^15 buckets of 2^15 entries each -- has ONLY heavy buckets and is
    // accumulated by the heavy-bucket kernel below.  Cutting every bucket into "virtual buckets" of 512 entries for THIS kernel
    // was measured in round 3: 388 ms against 85 ms.  One lane per (virtual) bucket makes the 64 lanes of a wave gather from 64
    // unrelated places of the 68 GB of window tables -- a TLB miss per lane -- whereas the heavy kernel's lanes walk
    // CONSECUTIVE entries of one bucket, whose bases are neighbours in the table.)
    // pieces per bucket of a lone plain-bases MSM: ~256 entries each (hooks builds: OG_LONE_PIECES; 1 = whole buckets)
    const uint32_t npiece = !lone_plain ? 1u
                            : OG_HOOK_SET("OG_LONE_PIECES") ? (uint32_t)std::max<long long>(1, std::min<long long>(64, OG_HOOK_INT("OG_LONE_PIECES", 1)))
                                                            : (uint32_t)std::max(1.0, std::min(64.0, lone_avg / 256.0));
    bool launched = false;
#ifdef OG_AB_HOOKS
    if (phase != MSM_SECOND) OG_TRY(ab_accumulate<T>(ctx, bases, ds, tag, pw, lone_plain, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, &launched));
#endif
    if (launched) {
    } else if (lone_plain && npiece > 1 && std::is_same<T, Fq>::value && (size_t)nchunk * npiece < ((size_t)1 << 32)) {
      uint8_t* pieces = nullptr;
      OG_TRY(arena_get(ctx, ("msm.pieces" + tag).c_str(), (size_t)npiece * ds.nkeys * PB, (void**)&pieces));
      const unsigned lgrid = (unsigned)std::min<size_t>((size_t)nchunk * npiece, (size_t)pw_lone * ctx->n_cu);
      hipLaunchKernelGGL((k_accumulate_pieces<T, AccCfg<T>::MINW>), dim3(lgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries,
                         ds.nkeys, pieces, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, npiece);
      OG_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_pieces_combine<T>, dim3(grid_for(ds.nkeys, 64)), dim3(64), 0, ctx->stream, pieces, ds.nkeys, npiece, buckets);
    } else {
      const unsigned pgrid = (unsigned)std::min<size_t>((size_t)nchunk * ds.batch, (size_t)std::max(1, lone_plain ? pw_lone : pw) * ctx->n_cu);
      if constexpr (std::is_same<T, Fq2>::value)
        hipLaunchKernelGGL((k_accumulate_g2_lds<AccCfg<T>::MINW, true>), dim3(pgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets,
                           ds.entries, ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
      else if (phase == MSM_SECOND)
        hipLaunchKernelGGL((k_accumulate_p<T, AccCfg<T>::MINW, true, true>), dim3(pgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets,
                           ds.entries, ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
      else if (lone_plain)
        hipLaunchKernelGGL((k_accumulate_p<T, AccCfg<T>::MINW, false>), dim3(pgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets,
                           ds.entries, ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
      else
        hipLaunchKernelGGL((k_accumulate_p<T, AccCfg<T>::MINW, true>), dim3(pgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets,
                           ds.entries, ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
    }
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "accumulate");
  }
  // the tail: latency-bound kernels with big register footprints.  On the side stream they wait for this accumulation and
  // then fill the slots the NEXT accumulation kernel leaves idle (its ramp-down), instead of standing between the two.
  hipStream_t main_stream = ctx->stream;
  struct StreamGuard {
    og_ctx* c;
    hipStream_t s;
    ~StreamGuard() { c->stream = s; }
  } stream_guard{ctx, main_stream};
  if (ctx->tail_stream && phase != MSM_FIRST) {  // (the first half of a merged pair folds its heavy buckets in before the second half starts: same stream)
    hipEvent_t e = ctx->tail_ev[ctx->tail_ev_next++ & 7];
    OG_HIP(hipEventRecord(e, main_stream));
    OG_HIP(hipStreamWaitEvent(ctx->tail_stream, e, 0));
    ctx->stream = ctx->tail_stream;
  }
  // the accumulation region is exactly ONE k_accumulate launch (bench.py's roofline); the heavy buckets have a region of
  // their own (a lone 2^26-point MSM has ONLY heavy buckets: 2^15 buckets of 2^15 entries each)
  {
    ProfScope ps_heavy(ctx, bases->is_g2 ? PROF_HEAVY_G2 : PROF_HEAVY_G1, (double)ds.n * ds.batch);
    // a lone huge MSM has only heavy buckets (the AVERAGE bucket is above the threshold)
    const bool all_heavy = (double)ds.n * ds.nwin / (double)ds.nkeys > (double)heavy_min;
    // (one segment per bucket for that case -- each lane walking a 8x longer stretch, one LDS tree per bucket instead of
    // eight -- was measured in round 3: the 2^26-point MSM went from 85 to 197 ms in this kernel, so the split stays)
    const uint32_t split = heavy_split;
    // a lone huge MSM does ALL its additions here: the full-register build; the prover's few heavy buckets: the 96-register
    // build that fits beside the next query's persistent accumulation (AccCfg::HEAVY_MINW)
    // a lone plain-bases MSM plans its segments by size (k_heavy_plan); everything else keeps `split` segments per bucket
    uint32_t* seg_off = nullptr;
    if (lone_plain && OG_HOOK_INT("OG_HEAVY_PLAN", 1)) {
      OG_TRY(arena_get(ctx, ("msm.heavyplan" + tag).c_str(), ((size_t)heavy_cap + 2) * 4, (void**)&seg_off));
      const uint32_t parts_cap = (uint32_t)(std::min<size_t>(heavy_cap, nsets * B) * HEAVY_SPLIT);
      hipLaunchKernelGGL(k_heavy_plan, dim3(1), dim3(1024), 0, ctx->stream, ds.offsets, ds.nkeys, heavy_count, heavy_list, heavy_cap, split, parts_cap,
                         seg_off, (uint32_t)std::max<long long>(64, OG_HOOK_INT("OG_HEAVY_SEG", HEAVY_SEG_ENTRIES)));
      OG_HIP(hipGetLastError());
    }
    if (all_heavy || lone_plain)
      hipLaunchKernelGGL((k_accumulate_heavy<T, AccCfg<T>::MINW>), dim3(16 * ctx->n_cu), dim3(HEAVY_BLOCK), (HEAVY_BLOCK / 2) * PB, ctx->stream,
                         bases->tab_d, ds.offsets, ds.entries, ds.nkeys, ds.ecap, heavy_parts, heavy_count, heavy_list, heavy_cap, split, seg_off);
    else
      hipLaunchKernelGGL((k_accumulate_heavy<T, AccCfg<T>::HEAVY_MINW>), dim3(16 * ctx->n_cu), dim3(HEAVY_BLOCK), (HEAVY_BLOCK / 2) * PB, ctx->stream,
                         bases->tab_d, ds.offsets, ds.entries, ds.nkeys, ds.ecap, heavy_parts, heavy_count, heavy_list, heavy_cap, split, seg_off);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_heavy_combine<T>, dim3(grid_for(std::min<size_t>(heavy_cap, nsets * B), 64)), dim3(64), 0, ctx->stream,
                       heavy_parts, heavy_count, heavy_list, heavy_cap, ds.nkeys, buckets, split, seg_off, phase == MSM_SECOND ? 1u : 0u);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "accumulate_heavy");
  }
  // (nothing below reads the digit sort any more: the stage pipeline lets the next-but-one sub-batch's preparation reuse it)
  if (ctx->after_heavy_ev && phase != MSM_FIRST) OG_HIP(hipEventRecord(ctx->after_heavy_ev, ctx->stream));
  if (phase == MSM_FIRST) return OG_OK;
  ProfScope ps_red(ctx, bases->is_g2 ? PROF_REDUCE_G2 : PROF_REDUCE_G1, (double)nsets * B);
  // weighted reduction: sum_b (b+1) B_b = G(B) + S(B)
  const size_t lvl_cap = nsets * ((B + SEG - 1) / SEG);
  uint8_t *tb[2], *ub[2], *vb[2];
  OG_TRY(arena_get(ctx, (std::string("msm.t0") + sfx).c_str(), lvl_cap * PB, (void**)&tb[0]));
  OG_TRY(arena_get(ctx, (std::string("msm.t1") + sfx).c_str(), lvl_cap * PB, (void**)&tb[1]));
  OG_TRY(arena_get(ctx, (std::string("msm.u0") + sfx).c_str(), lvl_cap * PB, (void**)&ub[0]));
  OG_TRY(arena_get(ctx, (std::string("msm.u1") + sfx).c_str(), lvl_cap * PB, (void**)&ub[1]));
  OG_TRY(arena_get(ctx, (std::string("msm.v0") + sfx).c_str(), lvl_cap * PB, (void**)&vb[0]));
  OG_TRY(arena_get(ctx, (std::string("msm.v1") + sfx).c_str(), lvl_cap * PB, (void**)&vb[1]));
  const uint8_t* items = buckets;
  const uint8_t* carry = nullptr;
  size_t n_in = B;
  int lvl = 0;
  // a few bucket sets: the scan-shaped reduction (depth ~40 additions instead of ~137); OG_SCAN_REDUCE=0 | 1 forces either
  const size_t scan_sets = std::is_same<T, Fq2>::value ? 8 : 16;
  const bool scan_reduce = OG_HOOK_SET("OG_SCAN_REDUCE") ? OG_HOOK_INT("OG_SCAN_REDUCE", 0) != 0 : nsets <= scan_sets;
  if (scan_reduce && B >= 2 && B <= 65536) {
    const uint32_t bs = (uint32_t)std::min<size_t>(256, B), nb = (uint32_t)(B / bs);
    int log_bs = 0;
    while ((1u << log_bs) < bs) log_bs++;
    uint8_t *tp = nullptr, *tp2 = nullptr, *pp2 = nullptr;
    OG_TRY(arena_get(ctx, (std::string("msm.scan.tp") + sfx).c_str(), 2 * nsets * nb * PB, (void**)&tp));     // [totals | parts][nsets][nb]
    OG_TRY(arena_get(ctx, (std::string("msm.scan.t2") + sfx).c_str(), 2 * nsets * PB, (void**)&tp2));
    OG_TRY(arena_get(ctx, (std::string("msm.scan.p2") + sfx).c_str(), 2 * nsets * PB, (void**)&pp2));
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb, (unsigned)nsets), dim3(256), (size_t)bs * PB, ctx->stream, buckets, (uint32_t)B, bs, tp,
                       tp + nsets * nb * PB);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3(1, (unsigned)(2 * nsets)), dim3(256), (size_t)nb * PB, ctx->stream, tp, nb, nb, tp2, pp2);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scan_reduce_final<T>, dim3(grid_for(nsets, 64)), dim3(64), 0, ctx->stream, tp2, pp2, (uint32_t)nsets, log_bs, ub[0], tb[0]);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "scan_reduce");
    carry = ub[0];
    items = tb[0];
    n_in = 1;
  }
  while (n_in > 1) {
    const size_t n_out = (n_in + SEG - 1) / SEG;
    uint8_t* to = tb[lvl & 1];
    uint8_t* vo = vb[lvl & 1];
    const unsigned gsz = grid_for(n_out * nsets, 64);
    static const bool red_alt = OG_HOOK_INT("OG_RED_ALT", 0) != 0;
    launch_runacc<T>(red_alt, dim3(gsz), ctx->stream, items, n_in, n_out, nsets, to, vo);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "seg_runacc");
    if (lvl == 0) {
      carry = vo;  // u_1 = v_1
    } else {
      uint8_t* uo = ub[lvl & 1];
#ifdef OG_AB_HOOKS
      if (red_alt)
        hipLaunchKernelGGL((k_seg_carry<T, AccCfg<T>::RED_ALT>), dim3(gsz), dim3(64), 0, ctx->stream, carry, n_in, vo, n_out, nsets,
                           lvl * SEG_LOG, uo);
      else
#endif
        hipLaunchKernelGGL((k_seg_carry<T, AccCfg<T>::RED_MINW>), dim3(gsz), dim3(64), 0, ctx->stream, carry, n_in, vo, n_out, nsets,
                           lvl * SEG_LOG, uo);
      OG_HIP(hipGetLastError());
      OG_STEP(ctx, "seg_carry");
      carry = uo;
    }
    items = to;
    n_in = n_out;
    lvl++;
  }
  const uint8_t* pin = carry;     // V per set
  const uint8_t* sitems = items;  // T per set
  // c >= 8 so B >= 128 and at least one level ran: pin = G per set, sitems = S per set
  if (partial) {
    const int slots = ds.precomp ? 1 : ds.nwin;
    hipLaunchKernelGGL(k_window_points<T>, dim3(grid_for((size_t)ds.batch * slots, 64)), dim3(64), 0, ctx->stream, pin, sitems,
                       nsets_per_g, slots, ds.own_mask, ds.precomp, ds.batch, out_d);
  } else {
    hipLaunchKernelGGL(k_window_combine<T>, dim3(grid_for(ds.batch, 64)), dim3(64), 0, ctx->stream, pin, sitems, nsets_per_g,
                       ds.c, ds.batch, out_d);
  }
  OG_HIP(hipGetLastError());
  OG_STEP(ctx, "window_combine");
  return OG_OK;
}

}  // namespace og
#include "msm_bases.hip.h"  // table import, the per-window shifts, XYZZ -> affine bytes
