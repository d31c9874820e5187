// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950.
//
// Pipeline (all on the issuing lane's stream, no host round trips):
//   1. digit sort       signed c-bit digits of every scalar, counting-sorted by bucket.  With precomputed window
//                       tables (one bucket set of 2^(c-1) keys per proof) the histogram lives in LDS:
//                       k_digit_hist_lds / k_scan_chunks / k_digit_scatter_lds, no global atomics; otherwise
//                       k_digit_hist / k_scan_offsets / k_digit_scatter with global atomics.  An optional wire map
//                       restricts the sort to a query's non-infinity bases (density compaction).
//   2. k_bucket_order   bucket ids by descending size (the lane -> bucket map of step 3)
//   3. k_accumulate     one lane per bucket: XYZZ += affine base (8M + 2S, gathered from the Montgomery-form table);
//                       buckets larger than HEAVY are deferred to
//   4. k_accumulate_heavy  one 256-lane workgroup per heavy bucket, LDS tree combine
//   5. k_seg_runacc / k_seg_carry   sum_b (b+1) B_b by segmented running sums (2 additions per bucket), log8(#buckets) levels
//   6. k_window_combine Horner over windows when the bases have no precomputed window tables
//
// With `precomp` bases (tab[k][i] = 2^(ck) P_i, affordable in 288 GB of HBM) every window
// shares ONE bucket set, so step 5 runs once instead of nwin times.  Everything is batched:
// `batch` independent scalar vectors (proofs) over the same bases go through each launch
// together, which keeps the latency-bound reduction levels throughput-bound.
#include "msm.hip.h"
#include "field.hip.h"
#include <algorithm>

namespace og {

// window choice: cost ~ nwin(c) * n mixed additions + ~10 addition-equivalents per bucket (2^(c-1) buckets);
// 16-bit windows win above ~16k points (nwin 16 vs 22 outweighs the 8x bucket reduction), 12-bit below, 8-bit for toy sizes
size_t msm_pick_c(size_t n) { return n < (1u << 9) ? 8 : (n < 16384 ? 12 : 16); }
// Window for a proving-key query (precomputed tables, many proofs per launch).
//
// 17 bits = 15 windows instead of 16: one sixteenth fewer mixed additions per point against twice the buckets to reduce
// (2.15 full additions each, in kernels that run at ~0.7 of the accumulation's efficiency): it pays from ~150 k points on.
//   n = 2^18: 15 n + 3 x 2^16 = 4.13 M addition-equivalents against 16 n + 3 x 2^15 = 4.29 M (-3.8 %).
// Measured in round 3 (1024 dense 2^18-wire proofs, same box, interleaved; profiles/r03_ab_query_window.txt): while every
// sub-batch still ran three witness digit sorts beside the accumulation, the step did not move (1859 -> 1866 ms, and
// 1911 -> 1889 ms on a second box, profiles/r03_ab_reduce_variants.txt) -- the 300-register reduction kernels do not fit
// beside the persistent accumulation waves, and with the sorts' filler waves also in the way the doubled reductions gave the
// saving back.  With ONE sort per sub-batch (groth16.hip, sort_src) it does: 566.3 / 564.6 -> 577.2 / 575.9 proofs/s
// (+1.9 %).  So: 17 bits from 160 k points on (the dense padding's A, B and L queries; the H query's 131 071 stay at 16).
// Cost: the bucket sets double (sub-batch scratch ~54 -> ~65 GB per slot, three slots).  OG_QUERY_C = 16 | 17 forces either.
//
// A query of 8 k .. 16 k points (the B query of the natural depth-32 statement: 13 205) gets 16-bit windows too.  By the
// addition count 12 bits is level with 16 there, but 12 bits means 2048 buckets of ~140 entries, and one request's proof
// then waits for single lanes walking 140 dependent G2 additions: measured on the natural statement (same box, A/B/A/B)
// one request 18.8 -> 16.8 ms (best; G2 accumulation 2.3 -> 0.3 ms), 8 requests 26.5 -> 24.5 ms, 64: 35 -> 34 ms;
// 4096 requests: 4140 -> 4083 proofs/s.  OG_QUERY_C16_MIN moves that bound (A/B).
//
// 15 bits for queries of 8 k .. 72 k points (round 5: the natural depth-32 withdraw statement's queries -- B 13 205 points,
// A / L 26 k, H 32 k).  There the bucket reduction is a third to a half of a query's work: measured on 4096 such proofs, one
// bucket costs what 4.4 G1 (5.1 G2) mixed additions cost -- 2.15 full additions in kernels at ~0.7 of the accumulation's
// efficiency -- and n x ceil(255 / c) + 4.5 x 2^(c-1) puts the 15 | 16 crossover at 73 728 points (16 | 17 at 147 k; 160 000 above).
// Same box, interleaved, natural statement (tools/ab_query_c15.sh, profiles/r05_ab_query_c15.txt):
//     batch 4096:   16 bits 4 474 / 4 487 proofs/s      15 bits 4 689 / 4 704 (+4.8 %)      B at 13 bits, the rest 15: 4 753 / 4 751
//     1 / 8 / 64 requests (median ms):   16 bits 11.6 / 18.4 / 31.1      15 bits 10.9 / 17.6 / 30.2      B at 13: 12.3 / 19.3 / 38.3
// 13 bits for the B query is what the cost model asks for (20 windows, 2^12 buckets) and it does cut the G2 reduction to a
// third, but its 64-entry buckets make one request wait for 64 dependent G2 additions per lane (accumulate_g2 0.4 -> 1.3 ms) and
// slow the batched G2 accumulation by 45 % instead of 25 %: +1.3 % throughput for +12 .. +27 % latency.  Not kept; 15 bits is
// better than 16 on every count.  OG_QUERY_C = 15 | 16 | 17 forces a size, OG_QUERY_C15_MAX moves the bound (hooks builds).
size_t msm_pick_query_c(size_t n) {
  const bool fits17 = n >= (1u << 16) && (double)n * 15 < (double)(1u << 23);
  if (const char* e = OG_HOOK_STR("OG_QUERY_C")) {
    if (atoi(e) == 17 && fits17) return 17;
    if ((atoi(e) == 16 || atoi(e) == 15) && n >= 512) return (size_t)atoi(e);
  } else if (fits17 && n >= 160000) {
    return 17;
  }
  const size_t c16_min = (size_t)OG_HOOK_INT("OG_QUERY_C16_MIN", 8192);
  if (!OG_HOOK_SET("OG_QUERY_C") && n >= c16_min && n >= 512 && n < (size_t)OG_HOOK_INT("OG_QUERY_C15_MAX", 73728)) return 15;
  if (n >= c16_min && n >= 512) return 16;
  return msm_pick_c(n);
}
int msm_nwin(int c) { return (255 + c - 1) / c; }

static std::string arena_key(og_ctx* ctx, const char* name) { return std::string(1, (char)('0' + ctx->lane)) + ":" + name; }

bool arena_has(og_ctx* ctx, const char* name) { return ctx->arena.find(arena_key(ctx, name)) != ctx->arena.end(); }

int arena_get(og_ctx* ctx, const char* name_, size_t bytes, void** out) {
  const std::string name = arena_key(ctx, name_);  // scratch is private to the lane that issues the work
  auto it = ctx->arena.find(name);
  if (it != ctx->arena.end() && it->second.second >= bytes) {
    *out = it->second.first;
    return OG_OK;
  }
  if (it != ctx->arena.end()) {
    // growing a buffer frees the old one: nothing of this ctx may still be using it -- on ANY of its streams (a previous
    // call may be in flight, og_withdraw_prove_batch_submit_d)
    OG_HIP(hipStreamSynchronize(ctx->stream));
    OG_HIP(drain_streams(ctx));
    OG_HIP(hipFree(it->second.first));
    ctx->arena.erase(it);
  }
  void* p = nullptr;
  size_t cap = bytes + bytes / 8 + 256;
  OG_HIP(hipMalloc(&p, cap));
  ctx->arena[name] = {p, cap};
  *out = p;
  return OG_OK;
}

// ---- digits -------------------------------------------------------------------

// signed digits of a canonical 254-bit scalar; calls f(k, bucket_index, negate) for non-zero digits
template <int C, class F>
__device__ __forceinline__ void for_each_digit(const uint32_t l[8], F&& f) {
  constexpr int NWIN = (255 + C - 1) / C;
  constexpr uint32_t HALF = 1u << (C - 1);
  uint32_t carry = 0;
#pragma unroll
  for (int k = 0; k < NWIN; k++) {
    const int bit = k * C;
    const int w = bit >> 5, s = bit & 31;
    uint32_t raw = 0;
    if (w < 8) {
      uint64_t v = l[w];
      if (w + 1 < 8) v |= (uint64_t)l[w + 1] << 32;
      raw = (uint32_t)(v >> s) & ((1u << C) - 1);
    }
    raw += carry;
    bool neg = raw > HALF;
    uint32_t mag = neg ? (1u << C) - raw : raw;
    carry = neg ? 1u : 0u;
    if (mag != 0) f(k, mag - 1, neg);
  }
}

// window ownership (window-sharded MSM across GPUs): bit k of `own` set = this rank accumulates window k;
// win_slot = index of window k among the owned windows (its bucket-set slot when there is one set per window)
__device__ __forceinline__ bool win_owned(uint32_t own, int k) { return (own >> k) & 1u; }
__device__ __forceinline__ uint32_t win_slot(uint32_t own, int k) { return (uint32_t)__popc(own & ((1u << k) - 1u)); }

__device__ __forceinline__ void load_scalar(const uint8_t* p, uint32_t l[8]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  l[0] = a.x; l[1] = a.y; l[2] = a.z; l[3] = a.w;
  l[4] = b.x; l[5] = b.y; l[6] = b.z; l[7] = b.w;
}

template <int C>
__global__ void __launch_bounds__(256) k_digit_hist(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                   const uint32_t* __restrict__ map, int precomp, uint32_t own,
                                                   uint32_t* __restrict__ counts, size_t nkeys) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= n) return;
  uint32_t l[8];
  load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
  uint32_t* cnt = counts + (size_t)g * (nkeys + 1);
  constexpr uint32_t B = 1u << (C - 1);
  for_each_digit<C>(l, [&](int k, uint32_t b, bool) {
    if (win_owned(own, k)) atomicAdd(&cnt[(precomp ? 0u : win_slot(own, k) * B) + b], 1u);
  });
}

// exclusive scan of counts[g][0..nkeys) in place -> offsets (offsets[nkeys] = total); cursor = copy
__global__ void __launch_bounds__(1024) k_scan_offsets(uint32_t* __restrict__ counts, uint32_t* __restrict__ cursor, size_t nkeys) {
  __shared__ uint32_t part[1024];
  const int g = blockIdx.x;
  uint32_t* c = counts + (size_t)g * (nkeys + 1);
  uint32_t* cur = cursor + (size_t)g * nkeys;
  const int t = threadIdx.x;
  size_t per = (nkeys + 1023) / 1024;
  size_t lo = (size_t)t * per, hi = lo + per < nkeys ? lo + per : nkeys;
  uint32_t s = 0;
  for (size_t i = lo; i < hi; i++) s += c[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  for (size_t i = lo; i < hi; i++) {
    uint32_t v = c[i];
    c[i] = run;
    cur[i] = run;
    run += v;
  }
  if (t == 1023) c[nkeys] = part[1023];
}

template <int C>
__global__ void __launch_bounds__(256) k_digit_scatter(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                      const uint32_t* __restrict__ map, int precomp, uint32_t own,
                                                      uint32_t* __restrict__ cursor, size_t nkeys,
                                                      uint32_t* __restrict__ entries, size_t ecap) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= n) return;
  uint32_t l[8];
  load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
  uint32_t* cur = cursor + (size_t)g * nkeys;
  uint32_t* ent = entries + (size_t)g * ecap;
  constexpr uint32_t B = 1u << (C - 1);
  for_each_digit<C>(l, [&](int k, uint32_t b, bool neg) {
    if (!win_owned(own, k)) return;
    uint32_t key = (precomp ? 0u : win_slot(own, k) * B) + b;
    uint32_t pos = atomicAdd(&cur[key], 1u);
    uint32_t idx = (precomp ? (uint32_t)k * (uint32_t)n : 0u) + (uint32_t)i;
    ent[pos] = (idx << 1) | (neg ? 1u : 0u);
  });
}

// ---- LDS-staged counting sort (precomputed-table mode: one bucket set of 2^(C-1) keys per proof) ----------
// The per-proof histogram fits in LDS (2^15 counters = 128 KiB of the CU's 160 KiB), so both passes run their
// atomics in LDS: a workgroup owns a chunk of SORT_CHUNK scalars; pass 1 writes its private histogram to
// hist[key][chunk]; an exclusive scan over (key-major, chunk-minor) turns that into every chunk's private
// write cursor per key; pass 2 reloads the cursors into LDS and scatters.  No global atomics at all.
constexpr int SORT_CHUNK = 16384;
constexpr int SORT_BLOCK = 1024;

template <int C>
__global__ void __launch_bounds__(SORT_BLOCK) k_digit_hist_lds(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                              const uint32_t* __restrict__ map, uint32_t own,
                                                              uint32_t* __restrict__ hist, uint32_t nchunks) {
  constexpr uint32_t B = 1u << (C - 1);
  __shared__ uint32_t cnt[B];
  const uint32_t chunk = blockIdx.x;
  const int g = blockIdx.y;
  for (uint32_t k = threadIdx.x; k < B; k += SORT_BLOCK) cnt[k] = 0;
  __syncthreads();
  const size_t lo = (size_t)chunk * SORT_CHUNK, hi = lo + SORT_CHUNK < n ? lo + SORT_CHUNK : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += SORT_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool) {
      if (win_owned(own, k)) atomicAdd(&cnt[b], 1u);
    });
  }
  __syncthreads();
  uint32_t* hg = hist + (size_t)g * ((size_t)B * nchunks + 1);
  for (uint32_t k = threadIdx.x; k < B; k += SORT_BLOCK) hg[(size_t)k * nchunks + chunk] = cnt[k];
}

// in-place exclusive scan of hist[g][0 .. len) (hist[g][len] = total); offsets[g][key] = hist[g][key * nchunks]
// 256 lanes per workgroup, not 1024: a 16-wave workgroup needs four free wave slots on every SIMD of one CU at once, and
// beside a persistent accumulation (3 per SIMD) plus two resident sort workgroups (2 per SIMD) there are only three --
// round 3 trace: this 0.05 ms kernel waited 56 ms, and with it the H query's sort and the math stream.
constexpr int SCAN_BLOCK = 256;
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_chunks(uint32_t* __restrict__ hist, size_t len, uint32_t nchunks,
                                                           uint32_t* __restrict__ offsets, size_t nkeys) {
  OG_FILLER_PRIO();
  __shared__ uint32_t part[SCAN_BLOCK];
  const int g = blockIdx.x;
  uint32_t* h = hist + (size_t)g * (len + 1);
  uint32_t* off = offsets + (size_t)g * (nkeys + 1);
  const int t = threadIdx.x;
  const size_t per = (len + SCAN_BLOCK - 1) / SCAN_BLOCK;
  const size_t lo = (size_t)t * per < len ? (size_t)t * per : len, hi = lo + per < len ? lo + per : len;
  uint32_t s = 0;
  for (size_t i = lo; i < hi; i++) s += h[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  for (size_t i = lo; i < hi; i++) {
    const uint32_t v = h[i];
    h[i] = run;
    if (i % nchunks == 0) off[i / nchunks] = run;
    run += v;
  }
  if (t == SCAN_BLOCK - 1) {
    h[len] = part[SCAN_BLOCK - 1];
    off[nkeys] = part[SCAN_BLOCK - 1];
  }
}

// Multi-block variant of the scan for small batches (a single huge MSM): the array is cut into `nblk` slices per
// proof; pass A sums every slice, pass B scans the slice totals (one block per proof), pass C rescans each slice
// from its base.  Same outputs as k_scan_chunks.
__global__ void __launch_bounds__(1024) k_scan_slice_sums(const uint32_t* __restrict__ hist, size_t len, uint32_t nblk,
                                                         uint32_t* __restrict__ sums) {
  __shared__ uint32_t part[1024];
  const int g = blockIdx.y, t = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const uint32_t* h = hist + (size_t)g * (len + 1);
  const size_t slice = (len + nblk - 1) / nblk, lo = (size_t)b * slice, hi = lo + slice < len ? lo + slice : len;
  uint32_t s = 0;
  for (size_t i = lo + t; i < hi; i += 1024) s += h[i];
  part[t] = s;
  __syncthreads();
  for (int d = 512; d >= 1; d >>= 1) {
    if (t < d) part[t] += part[t + d];
    __syncthreads();
  }
  if (t == 0) sums[(size_t)g * nblk + b] = part[0];
}

__global__ void __launch_bounds__(1024) k_scan_slice_bases(uint32_t* __restrict__ sums, uint32_t nblk) {
  __shared__ uint32_t part[1024];
  const int g = blockIdx.x, t = threadIdx.x;
  uint32_t* s = sums + (size_t)g * nblk;
  const uint32_t v = (uint32_t)t < nblk ? s[t] : 0;
  part[t] = v;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t x = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  if ((uint32_t)t < nblk) s[t] = part[t] - v;  // exclusive
}

__global__ void __launch_bounds__(1024) k_scan_slices(uint32_t* __restrict__ hist, size_t len, uint32_t nblk, uint32_t nchunks,
                                                     const uint32_t* __restrict__ bases, uint32_t* __restrict__ offsets, size_t nkeys) {
  __shared__ uint32_t part[1024];
  const int g = blockIdx.y, t = threadIdx.x;
  const uint32_t b = blockIdx.x;
  uint32_t* h = hist + (size_t)g * (len + 1);
  uint32_t* off = offsets + (size_t)g * (nkeys + 1);
  const size_t slice = (len + nblk - 1) / nblk, s_lo = (size_t)b * slice, s_hi = s_lo + slice < len ? s_lo + slice : len;
  const size_t span = s_hi > s_lo ? s_hi - s_lo : 0, per = (span + 1023) / 1024;
  const size_t lo = s_lo + ((size_t)t * per < span ? (size_t)t * per : span), hi = lo + per < s_hi ? lo + per : s_hi;
  uint32_t s = 0;
  for (size_t i = lo; i < hi; i++) s += h[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = bases[(size_t)g * nblk + b] + (t ? part[t - 1] : 0);
  for (size_t i = lo; i < hi; i++) {
    const uint32_t v = h[i];
    h[i] = run;
    if (i % nchunks == 0) off[i / nchunks] = run;
    run += v;
  }
  if (b == nblk - 1 && t == 1023) {
    const uint32_t total = bases[(size_t)g * nblk + b] + part[1023];
    h[len] = total;
    off[nkeys] = total;
  }
}

template <int C>
__global__ void __launch_bounds__(SORT_BLOCK) k_digit_scatter_lds(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                                 const uint32_t* __restrict__ map, uint32_t own,
                                                                 const uint32_t* __restrict__ hist, uint32_t nchunks,
                                                                 uint32_t* __restrict__ entries, size_t ecap) {
  constexpr uint32_t B = 1u << (C - 1);
  __shared__ uint32_t cur[B];
  const uint32_t chunk = blockIdx.x;
  const int g = blockIdx.y;
  const uint32_t* hg = hist + (size_t)g * ((size_t)B * nchunks + 1);
  for (uint32_t k = threadIdx.x; k < B; k += SORT_BLOCK) cur[k] = hg[(size_t)k * nchunks + chunk];
  __syncthreads();
  uint32_t* ent = entries + (size_t)g * ecap;
  const size_t lo = (size_t)chunk * SORT_CHUNK, hi = lo + SORT_CHUNK < n ? lo + SORT_CHUNK : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += SORT_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool neg) {
      if (!win_owned(own, k)) return;
      const uint32_t pos = atomicAdd(&cur[b], 1u);
      ent[pos] = (((uint32_t)k * (uint32_t)n + (uint32_t)i) << 1) | (neg ? 1u : 0u);
    });
  }
}

template <int C>
static int digit_sort_lds(og_ctx* ctx, const std::string& tag, const uint8_t* scalars_d, size_t stride, size_t n,
                          const uint32_t* map_d, int batch, DigitSort& ds) {
  const uint32_t nchunks = (uint32_t)((n + SORT_CHUNK - 1) / SORT_CHUNK);
  const size_t B = (size_t)1 << (C - 1), len = B * (nchunks ? nchunks : 1);
  uint32_t* hist = nullptr;
  OG_TRY(arena_get(ctx, (tag + ".hist").c_str(), (size_t)batch * (len + 1) * 4, (void**)&hist));
  if (n == 0) {
    OG_HIP(hipMemsetAsync(ds.offsets, 0, (size_t)batch * (ds.nkeys + 1) * 4, ctx->stream));
    return OG_OK;
  }
  hipLaunchKernelGGL(k_digit_hist_lds<C>, dim3(nchunks, batch), dim3(SORT_BLOCK), 0, ctx->stream, scalars_d, stride, n, map_d,
                     ds.own_mask, hist, nchunks);
  OG_HIP(hipGetLastError());
  // one block per proof is enough when many proofs are sorted together; a lone big MSM gets a multi-block scan
  uint32_t nblk = batch >= 32 ? 1u : (uint32_t)std::min<size_t>(1024, std::max<size_t>(1, len >> 18));
  if (const char* e = OG_HOOK_STR("OG_SCAN_NBLK")) nblk = (uint32_t)std::min(1024, std::max(1, atoi(e)));  // test hook
  if (nblk == 1) {
    hipLaunchKernelGGL(k_scan_chunks, dim3(batch), dim3(SCAN_BLOCK), 0, ctx->stream, hist, len, nchunks, ds.offsets, ds.nkeys);
  } else {
    uint32_t* sums = nullptr;
    OG_TRY(arena_get(ctx, (tag + ".ssum").c_str(), (size_t)batch * nblk * 4, (void**)&sums));
    hipLaunchKernelGGL(k_scan_slice_sums, dim3(nblk, batch), dim3(1024), 0, ctx->stream, hist, len, nblk, sums);
    hipLaunchKernelGGL(k_scan_slice_bases, dim3(batch), dim3(1024), 0, ctx->stream, sums, nblk);
    hipLaunchKernelGGL(k_scan_slices, dim3(nblk, batch), dim3(1024), 0, ctx->stream, hist, len, nblk, nchunks, sums, ds.offsets,
                       ds.nkeys);
  }
  OG_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_digit_scatter_lds<C>, dim3(nchunks, batch), dim3(SORT_BLOCK), 0, ctx->stream, scalars_d, stride, n, map_d,
                     ds.own_mask, hist, nchunks, ds.entries, ds.ecap);
  OG_HIP(hipGetLastError());
  return OG_OK;
}


// ---- two-level radix sort (the prover's shape: 16-bit windows, one bucket set of 2^15 keys per proof) ------------------
// The single-level LDS sort above scatters 4-byte entries over 2^15 destinations: each bucket receives ~8 entries per
// chunk, i.e. 32-byte runs -- partial cache lines, measured 13x write amplification at the memory side (round 1 PMC).
// Two levels keep every run long:
//   k_digit_hist_hi / scan / k_digit_scatter_hi   partition the digits by the HIGH 8 bits of the bucket (256 bins): a chunk of
//        RS_CHUNK scalars sends ~256 entries (1 KB) to each bin; the low 7 bucket bits ride in the entry's top bits
//   k_sort_lo    one workgroup per (proof, bin): the bin's ~16 K entries are counted and scattered by the LOW 7 bits inside a
//        region of ~64 KB that this workgroup alone writes completely (L2 write-combines it); it also emits the bucket offsets
// Needs (table index << 1 | sign) < 2^25, i.e. n * nwin < 2^24: true for every Groth16 query here (n <= 2^18).
constexpr int RS_CHUNK = 4096;
constexpr int RS_BLOCK = 256;
// 256 bins whatever the window: the low LO = C - 9 bucket bits are sorted inside a bin and ride in the top bits of the
// partitioned entry, above the IDX = 32 - LO bits of (table index << 1 | sign).  16-bit windows: 7 + 25; 17-bit: 8 + 24; 15-bit: 6 + 26.
template <int C> struct RsBits { static constexpr int LO = C - 9, IDX = 32 - (C - 9); };

template <int C>
__global__ void __launch_bounds__(RS_BLOCK) k_digit_hist_hi(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                           const uint32_t* __restrict__ map, uint32_t own,
                                                           uint32_t* __restrict__ hist, uint32_t nchunks) {
  OG_FILLER_PRIO();
  constexpr uint32_t NBIN = 1u << (C - 1 - RsBits<C>::LO);
  __shared__ uint32_t cnt[NBIN];
  const uint32_t chunk = blockIdx.x;
  const int g = blockIdx.y;
  for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) cnt[k] = 0;
  __syncthreads();
  const size_t lo = (size_t)chunk * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += RS_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool) {
      if (win_owned(own, k)) atomicAdd(&cnt[b >> RsBits<C>::LO], 1u);
    });
  }
  __syncthreads();
  uint32_t* hg = hist + (size_t)g * ((size_t)NBIN * nchunks + 1);
  for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) hg[(size_t)k * nchunks + chunk] = cnt[k];
}

// first position p in off[0 .. n] (exclusive prefix sums, off[n] = total) with off[p + 1] > t: the bin that holds sorted slot t
__device__ __forceinline__ uint32_t bin_of_slot(const uint32_t* off, uint32_t n, uint32_t t) {
  uint32_t lo = 0, hi = n;  // invariant: off[lo] <= t < off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

// exclusive scan of cnt[0 .. N) into off[0 .. N] (off[N] = total) by the first N lanes; N a power of two <= blockDim
template <uint32_t N>
__device__ __forceinline__ void lds_excl_scan(const uint32_t* cnt, uint32_t* off, uint32_t* tmp) {
  const uint32_t t = threadIdx.x;
  if (t < N) tmp[t] = cnt[t];
  __syncthreads();
  for (uint32_t d = 1; d < N; d <<= 1) {
    const uint32_t v = (t < N && t >= d) ? tmp[t - d] : 0;
    __syncthreads();
    if (t < N) tmp[t] += v;
    __syncthreads();
  }
  if (t < N) off[t + 1] = tmp[t];
  if (t == 0) off[0] = 0;
  __syncthreads();
}

// The partition by high bucket bits, writing RUNS: a tile of RS_TILE scalars (<= 16 K entries) is counting-sorted by bin
// inside LDS, then every (tile, bin) run -- ~64 entries, 256 B -- leaves through consecutive lanes as whole 64-byte
// sectors.  (Storing each entry where its cursor pointed, the first version of this kernel, kept ~0.5 M partially written
// lines in flight across the launch, more than L2 holds: WRITE_SIZE was 3.8x the entry bytes.)
constexpr int RS_TILE = 1024;

template <int C>
__global__ void __launch_bounds__(RS_BLOCK) k_digit_scatter_hi(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                              const uint32_t* __restrict__ map, uint32_t own,
                                                              const uint32_t* __restrict__ hist, uint32_t nchunks,
                                                              uint32_t* __restrict__ tmp, size_t ecap) {
  OG_FILLER_PRIO();
  constexpr uint32_t NBIN = 1u << (C - 1 - RsBits<C>::LO);
  constexpr int NWIN = (255 + C - 1) / C;
  static_assert(NBIN <= RS_BLOCK, "one lane per bin");
  __shared__ uint32_t buf[RS_TILE * NWIN];
  __shared__ uint32_t cur[NBIN], cnt[NBIN], fill[NBIN], off[NBIN + 1], scan_tmp[NBIN];
  const uint32_t chunk = blockIdx.x;
  const int g = blockIdx.y;
  const uint32_t* hg = hist + (size_t)g * ((size_t)NBIN * nchunks + 1);
  for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) cur[k] = hg[(size_t)k * nchunks + chunk];
  uint32_t* out = tmp + (size_t)g * ecap;
  const size_t c_lo = (size_t)chunk * RS_CHUNK, c_hi = c_lo + RS_CHUNK < n ? c_lo + RS_CHUNK : n;
  for (size_t t_lo = c_lo; t_lo < c_hi; t_lo += RS_TILE) {
    const size_t t_hi = t_lo + RS_TILE < c_hi ? t_lo + RS_TILE : c_hi;
    for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) { cnt[k] = 0; fill[k] = 0; }
    __syncthreads();
    for (size_t i = t_lo + threadIdx.x; i < t_hi; i += RS_BLOCK) {
      uint32_t l[8];
      load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
      for_each_digit<C>(l, [&](int k, uint32_t b, bool) {
        if (win_owned(own, k)) atomicAdd(&cnt[b >> RsBits<C>::LO], 1u);
      });
    }
    __syncthreads();
    lds_excl_scan<NBIN>(cnt, off, scan_tmp);
    for (size_t i = t_lo + threadIdx.x; i < t_hi; i += RS_BLOCK) {
      uint32_t l[8];
      load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
      for_each_digit<C>(l, [&](int k, uint32_t b, bool neg) {
        if (!win_owned(own, k)) return;
        const uint32_t bin = b >> RsBits<C>::LO;
        buf[off[bin] + atomicAdd(&fill[bin], 1u)] =
            ((b & ((1u << RsBits<C>::LO) - 1u)) << RsBits<C>::IDX) | (((uint32_t)k * (uint32_t)n + (uint32_t)i) << 1) | (neg ? 1u : 0u);
      });
    }
    __syncthreads();
    const uint32_t total = off[NBIN];
    for (uint32_t t = threadIdx.x; t < total; t += RS_BLOCK) {
      const uint32_t bin = bin_of_slot(off, NBIN, t);
      out[cur[bin] + (t - off[bin])] = buf[t];
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) cur[k] += cnt[k];
    __syncthreads();
  }
}

// one workgroup per (bin, proof): counting sort of the bin's entries by the low bucket bits + the bucket offsets of the bin.
// Pass 1 counts the whole bin (bucket offsets); pass 2 goes through the bin in tiles of SL_TILE entries, sorts a tile inside
// LDS and writes its per-bucket runs through consecutive lanes.
constexpr int SL_TILE = 8192;

template <int LO>
__global__ void __launch_bounds__(RS_BLOCK) k_sort_lo(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binoff, uint32_t nbin,
                                                     uint32_t* __restrict__ entries, size_t ecap, uint32_t* __restrict__ offsets,
                                                     size_t nkeys) {
  OG_FILLER_PRIO();
  constexpr uint32_t NLO = 1u << LO;
  constexpr int IDX = 32 - LO;
  __shared__ uint32_t buf[SL_TILE];
  __shared__ uint32_t cnt[NLO], cur[NLO], fill[NLO], off[NLO + 1], scan_tmp[NLO];
  const uint32_t bin = blockIdx.x, t = threadIdx.x;
  const int g = blockIdx.y;
  const uint32_t* bo = binoff + (size_t)g * (nbin + 1);
  const uint32_t lo = bo[bin], hi = bo[bin + 1];
  const uint32_t* in = tmp + (size_t)g * ecap;
  uint32_t* out = entries + (size_t)g * ecap;
  if (t < NLO) cnt[t] = 0;
  __syncthreads();
  for (uint32_t p = lo + t; p < hi; p += RS_BLOCK) atomicAdd(&cnt[in[p] >> IDX], 1u);
  __syncthreads();
  lds_excl_scan<NLO>(cnt, off, scan_tmp);
  if (t < NLO) {
    cur[t] = lo + off[t];
    offsets[(size_t)g * (nkeys + 1) + (size_t)bin * NLO + t] = lo + off[t];
  }
  if (bin == nbin - 1 && t == 0) offsets[(size_t)g * (nkeys + 1) + nkeys] = hi;
  __syncthreads();
  for (uint32_t t_lo = lo; t_lo < hi; t_lo += SL_TILE) {
    const uint32_t t_hi = t_lo + SL_TILE < hi ? t_lo + SL_TILE : hi;
    if (t < NLO) { cnt[t] = 0; fill[t] = 0; }
    __syncthreads();
    for (uint32_t p = t_lo + t; p < t_hi; p += RS_BLOCK) atomicAdd(&cnt[in[p] >> IDX], 1u);
    __syncthreads();
    lds_excl_scan<NLO>(cnt, off, scan_tmp);
    for (uint32_t p = t_lo + t; p < t_hi; p += RS_BLOCK) {
      const uint32_t e = in[p], b = e >> IDX;
      buf[off[b] + atomicAdd(&fill[b], 1u)] = e & ((1u << IDX) - 1u);
    }
    __syncthreads();
    const uint32_t total = t_hi - t_lo;
    for (uint32_t s = t; s < total; s += RS_BLOCK) {
      const uint32_t b = bin_of_slot(off, NLO, s);
      out[cur[b] + (s - off[b])] = buf[s];
    }
    __syncthreads();
    if (t < NLO) cur[t] += cnt[t];
    __syncthreads();
  }
}

// Second level for a lone MSM's bins (~65 K entries each, but SKEWED: the bin of the "digit 1" bucket holds millions, the top
// window's bins a million each).  One workgroup per bin walked such a bin tile after tile -- ~1000 tiles of 4 K entries one
// after the other, 20 ms for the one giant bin while the chip idled -- so a bin of more than SB_SLICE entries is cut into up to
// SB_SMAX slices (grid.y), each a workgroup that takes every nsl-th tile:
//   k_sub_count    per (bin, slice): counts of the 2^LO sub-buckets, added into gcnt[bin][.]           (batched loads: a lane
//   k_sub_offsets  per bin: bucket offsets = bin start + exclusive scan of gcnt; gcur = the same        pulls SB_PER entries of
//   k_sub_scatter  per (bin, slice): tiles sorted in LDS, runs written through consecutive lanes        a tile into registers)
// A bin with ONE slice keeps its running cursors in LDS and its tiles in order: a bucket's entries stay ordered by source
// position, which is what makes the position-major accumulation's table sweep coherent -- a PERFORMANCE property only: any
// order of a bucket's entries gives the same sum.  Slices of a cut bin claim their runs from gcur with one global atomic per
// sub-bucket and tile, so their buckets lose that order -- those are the skewed bins, whose buckets feed the heavy-bucket path.
constexpr int SB_PER = 16;
constexpr int SB_TILE = RS_BLOCK * SB_PER;  // 4096 entries
constexpr uint32_t SB_SLICE = 1u << 17;
constexpr int SB_SMAX = 16;

__device__ __forceinline__ uint32_t sub_slices(uint32_t len) {
  const uint32_t n = (len + SB_SLICE - 1) / SB_SLICE;
  return n < 1 ? 1u : (n > (uint32_t)SB_SMAX ? (uint32_t)SB_SMAX : n);
}

template <int LO>
__global__ void __launch_bounds__(RS_BLOCK) k_sub_count(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binoff, uint32_t nbin,
                                                       uint32_t* __restrict__ gcnt) {
  constexpr uint32_t NLO = 1u << LO;
  constexpr int IDX = 32 - LO;
  __shared__ uint32_t cnt[NLO];
  const uint32_t bin = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = binoff[bin], hi = binoff[bin + 1];
  const uint32_t nsl = sub_slices(hi - lo);
  if (blockIdx.y >= nsl) return;
  if (t < NLO) cnt[t] = 0;
  __syncthreads();
  for (uint32_t t_lo = lo + blockIdx.y * SB_TILE; t_lo < hi; t_lo += nsl * SB_TILE) {
    uint32_t v[SB_PER];
#pragma unroll
    for (int j = 0; j < SB_PER; j++) {
      const uint32_t p = t_lo + (uint32_t)j * RS_BLOCK + t;
      v[j] = p < hi ? tmp[p] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < SB_PER; j++)
      if (t_lo + (uint32_t)j * RS_BLOCK + t < hi) (void)OG_LDS_ATOMIC_INC_AGG(cnt, v[j] >> IDX);
  }
  __syncthreads();
  if (t < NLO && cnt[t]) atomicAdd(&gcnt[(size_t)bin * NLO + t], cnt[t]);
}

// offsets[bin * NLO + s] = gcur[bin * NLO + s] = binoff[bin] + (exclusive scan of gcnt[bin][.])[s]; one lane per bin
template <int LO>
__global__ void __launch_bounds__(256) k_sub_offsets(const uint32_t* __restrict__ binoff, uint32_t nbin, const uint32_t* __restrict__ gcnt,
                                                    uint32_t* __restrict__ gcur, uint32_t* __restrict__ offsets, size_t nkeys) {
  constexpr uint32_t NLO = 1u << LO;
  const uint32_t bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= nbin) return;
  uint32_t run = binoff[bin];
  for (uint32_t s2 = 0; s2 < NLO; s2++) {
    const size_t k = (size_t)bin * NLO + s2;
    const uint32_t c = gcnt[k];
    gcur[k] = run;
    offsets[k] = run;
    run += c;
  }
  if (bin == nbin - 1) offsets[nkeys] = binoff[nbin];
}

template <int LO>
__global__ void __launch_bounds__(RS_BLOCK) k_sub_scatter(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binoff, uint32_t nbin,
                                                         uint32_t* __restrict__ gcur, uint32_t* __restrict__ out) {
  constexpr uint32_t NLO = 1u << LO;
  constexpr int IDX = 32 - LO;
  __shared__ uint32_t buf[SB_TILE];
  __shared__ uint32_t cnt[NLO], cur[NLO], fill[NLO], off[NLO + 1], scan_tmp[NLO];
  const uint32_t bin = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = binoff[bin], hi = binoff[bin + 1];
  const uint32_t nsl = sub_slices(hi - lo);
  if (blockIdx.y >= nsl) return;
  if (t < NLO) cur[t] = gcur[(size_t)bin * NLO + t];  // (a single slice: running cursors, tiles stay in order)
  __syncthreads();
  for (uint32_t t_lo = lo + blockIdx.y * SB_TILE; t_lo < hi; t_lo += nsl * SB_TILE) {
    const uint32_t t_hi = t_lo + SB_TILE < hi ? t_lo + SB_TILE : hi;
    uint32_t v[SB_PER];
#pragma unroll
    for (int j = 0; j < SB_PER; j++) {
      const uint32_t p = t_lo + (uint32_t)j * RS_BLOCK + t;
      v[j] = p < t_hi ? tmp[p] : 0xffffffffu;
    }
    if (t < NLO) { cnt[t] = 0; fill[t] = 0; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SB_PER; j++)
      if (t_lo + (uint32_t)j * RS_BLOCK + t < t_hi) (void)OG_LDS_ATOMIC_INC_AGG(cnt, v[j] >> IDX);
    __syncthreads();
    lds_excl_scan<NLO>(cnt, off, scan_tmp);
    if (nsl > 1 && t < NLO && cnt[t]) cur[t] = atomicAdd(&gcur[(size_t)bin * NLO + t], cnt[t]);  // a cut bin: claim this tile's runs
#pragma unroll
    for (int j = 0; j < SB_PER; j++)
      if (t_lo + (uint32_t)j * RS_BLOCK + t < t_hi) {
        const uint32_t b = v[j] >> IDX;
        buf[off[b] + OG_LDS_ATOMIC_INC_AGG(fill, b)] = v[j] & ((1u << IDX) - 1u);
      }
    __syncthreads();
    const uint32_t total = t_hi - t_lo;
    for (uint32_t s2 = t; s2 < total; s2 += RS_BLOCK) {
      const uint32_t b = bin_of_slot(off, NLO, s2);
      out[cur[b] + (s2 - off[b])] = buf[s2];
    }
    __syncthreads();
    if (nsl == 1 && t < NLO) cur[t] += cnt[t];
    __syncthreads();
  }
}

// Direct variants: every entry is stored where its cursor points, no LDS staging.  Shorter dependency chains per
// workgroup (no tile loop, no scans), so they win when a launch does not fill the chip (a single request, small
// sub-batches: one proof 29.6 -> 27.6 ms, batch 8 65 -> 46 ms); at scale their partially written lines outlive L2 (WRITE_SIZE
// 3.8x the entry bytes) and the run-staging kernels above take over (digit_sort_radix picks by grid size).
template <int C>
__global__ void __launch_bounds__(RS_BLOCK) k_digit_scatter_hi_direct(const uint8_t* __restrict__ scalars, size_t stride, size_t n,
                                                              const uint32_t* __restrict__ map, uint32_t own,
                                                              const uint32_t* __restrict__ hist, uint32_t nchunks,
                                                              uint32_t* __restrict__ tmp, size_t ecap) {
  OG_FILLER_PRIO();
  constexpr uint32_t NBIN = 1u << (C - 1 - RsBits<C>::LO);
  __shared__ uint32_t cur[NBIN];
  const uint32_t chunk = blockIdx.x;
  const int g = blockIdx.y;
  const uint32_t* hg = hist + (size_t)g * ((size_t)NBIN * nchunks + 1);
  for (uint32_t k = threadIdx.x; k < NBIN; k += RS_BLOCK) cur[k] = hg[(size_t)k * nchunks + chunk];
  __syncthreads();
  uint32_t* out = tmp + (size_t)g * ecap;
  const size_t lo = (size_t)chunk * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += RS_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + (size_t)g * stride + (size_t)(map ? map[i] : (uint32_t)i) * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool neg) {
      if (!win_owned(own, k)) return;
      const uint32_t pos = atomicAdd(&cur[b >> RsBits<C>::LO], 1u);
      out[pos] = ((b & ((1u << RsBits<C>::LO) - 1u)) << RsBits<C>::IDX) | (((uint32_t)k * (uint32_t)n + (uint32_t)i) << 1) | (neg ? 1u : 0u);
    });
  }
}

// (direct variant) one workgroup per (bin, proof): counting sort by the low bucket bits + the bucket offsets of the bin
template <int LO>
__global__ void __launch_bounds__(RS_BLOCK) k_sort_lo_direct(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binoff, uint32_t nbin,
                                                     uint32_t* __restrict__ entries, size_t ecap, uint32_t* __restrict__ offsets,
                                                     size_t nkeys) {
  OG_FILLER_PRIO();
  constexpr uint32_t NLO = 1u << LO;
  constexpr int IDX = 32 - LO;
  __shared__ uint32_t cnt[NLO];
  __shared__ uint32_t cur[NLO];
  const uint32_t bin = blockIdx.x, t = threadIdx.x;
  const int g = blockIdx.y;
  const uint32_t* bo = binoff + (size_t)g * (nbin + 1);
  const uint32_t lo = bo[bin], hi = bo[bin + 1];
  const uint32_t* in = tmp + (size_t)g * ecap;
  uint32_t* out = entries + (size_t)g * ecap;
  if (t < NLO) cnt[t] = 0;
  __syncthreads();
  for (uint32_t p = lo + t; p < hi; p += RS_BLOCK) atomicAdd(&cnt[in[p] >> IDX], 1u);
  __syncthreads();
  // exclusive scan of the NLO counters (Hillis-Steele on the first NLO lanes)
  uint32_t own_cnt = t < NLO ? cnt[t] : 0;
  for (uint32_t d = 1; d < NLO; d <<= 1) {
    uint32_t v = (t < NLO && t >= d) ? cnt[t - d] : 0;
    __syncthreads();
    if (t < NLO) cnt[t] += v;
    __syncthreads();
  }
  if (t < NLO) {
    const uint32_t start = lo + cnt[t] - own_cnt;
    cur[t] = start;
    offsets[(size_t)g * (nkeys + 1) + (size_t)bin * NLO + t] = start;
  }
  if (bin == nbin - 1 && t == 0) offsets[(size_t)g * (nkeys + 1) + nkeys] = hi;
  __syncthreads();
  for (uint32_t p = lo + t; p < hi; p += RS_BLOCK) {
    const uint32_t e = in[p];
    out[atomicAdd(&cur[e >> IDX], 1u)] = e & ((1u << IDX) - 1u);
  }
}

template <int C>
static int digit_sort_radix(og_ctx* ctx, const std::string& tag, const uint8_t* scalars_d, size_t stride, size_t n,
                            const uint32_t* map_d, int batch, DigitSort& ds) {
  constexpr uint32_t NBIN = 1u << (C - 1 - RsBits<C>::LO);
  const uint32_t nchunks = (uint32_t)((n + RS_CHUNK - 1) / RS_CHUNK);
  const size_t len = (size_t)NBIN * (nchunks ? nchunks : 1);
  uint32_t *hist = nullptr, *binoff = nullptr, *tmp = nullptr;
  OG_TRY(arena_get(ctx, (tag + ".hist").c_str(), (size_t)batch * (len + 1) * 4, (void**)&hist));
  OG_TRY(arena_get(ctx, (tag + ".binoff").c_str(), (size_t)batch * (NBIN + 1) * 4, (void**)&binoff));
  OG_TRY(arena_get(ctx, (tag + ".tmp").c_str(), (size_t)batch * (ds.ecap ? ds.ecap : 1) * 4, (void**)&tmp));
  if (n == 0) {
    OG_HIP(hipMemsetAsync(ds.offsets, 0, (size_t)batch * (ds.nkeys + 1) * 4, ctx->stream));
    return OG_OK;
  }
  hipLaunchKernelGGL(k_digit_hist_hi<C>, dim3(nchunks, batch), dim3(RS_BLOCK), 0, ctx->stream, scalars_d, stride, n, map_d, ds.own_mask,
                     hist, nchunks);
  OG_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_scan_chunks, dim3(batch), dim3(SCAN_BLOCK), 0, ctx->stream, hist, len, nchunks, binoff, (size_t)NBIN);
  OG_HIP(hipGetLastError());
  // Launches that fill the chip many times over stage their runs in LDS (coalesced writes: WRITE_SIZE 1.0x the entry bytes
  // against 3.8x); small ones go direct (latency).  Inside the pipelined prover the sorts run BESIDE the persistent bucket
  // accumulation, and what counts there is what fits next to it: the staged kernels want 64 registers and 71 KB / 35 KB of
  // LDS per workgroup (one workgroup per CU beside a G1 accumulation, none beside a G2 one, whose accumulators fill the
  // LDS), the direct ones 22 / 8 registers and 1 KB (four per CU beside G1, two beside G2).  Round 3, same box: digit sorts
  // 1730 -> 330 ms of wall time per 1024 proofs beside the accumulation, 532 -> 545 proofs/s; HBM has the headroom for the
  // partial lines (the accumulation moves 1.3 of 8 TB/s).
  static const int force = (int)OG_HOOK_INT("OG_SORT_DIRECT", -1);
  const bool direct = force >= 0 ? force != 0 : (ctx->sort_beside_acc || (size_t)nchunks * batch < 8192);
  if (direct) {
    hipLaunchKernelGGL(k_digit_scatter_hi_direct<C>, dim3(nchunks, batch), dim3(RS_BLOCK), 0, ctx->stream, scalars_d, stride, n, map_d,
                       ds.own_mask, hist, nchunks, tmp, ds.ecap);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sort_lo_direct<RsBits<C>::LO>, dim3(NBIN, batch), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, NBIN, ds.entries, ds.ecap,
                       ds.offsets, ds.nkeys);
  } else {
    hipLaunchKernelGGL(k_digit_scatter_hi<C>, dim3(nchunks, batch), dim3(RS_BLOCK), 0, ctx->stream, scalars_d, stride, n, map_d,
                       ds.own_mask, hist, nchunks, tmp, ds.ecap);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sort_lo<RsBits<C>::LO>, dim3(NBIN, batch), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, NBIN, ds.entries, ds.ecap, ds.offsets,
                       ds.nkeys);
  }
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// ---- lone big MSM over PLAIN bases (no per-window tables): two-level radix sort over (window, bucket) keys ---------------
// BASELINE.json configs[2]: one MSM over 2^26 points.  Per-window tables buy nothing at that size -- with or without them the
// accumulation is n x 16 mixed additions, and the 16 bucket-set reductions they save are ~10^6 additions against 2^30 -- but
// cost 68.7 GB and a 4.6 s build.  Plain bases need one bucket set per window, i.e. 16 x 2^15 = 2^19 keys: too many for an
// LDS histogram of keys, so round 1's path fell back to global atomics.  Here, two levels again:
//   k_lone_hist / scan / k_lone_scatter   ONE pass over the scalars partitions the digits of ALL windows by
//        bin = (window slot, high 10 bits of the bucket): 16 x 1024 = 16 384 bins, LDS counters / cursors (64 KB per workgroup);
//        the low LN_LO = 5 bucket bits ride in the top bits of the entry, above 27 bits of (point index << 1 | sign)
//   k_sort_lo_direct<5> / k_sort_lo<5>    one workgroup per bin sorts its ~65 K entries by the low 5 bits inside a region it
//        alone writes, and emits the 32 bucket offsets of the bin (key = bin x 32 + low bits = slot x 2^15 + bucket)
// Needs n <= 2^26 (27-bit entries).  A bucket's entries end up ordered by source chunk, so the 64 lanes of an accumulation
// wave -- 64 buckets of one window walked in lockstep -- gather their bases from one neighbourhood of the 4 GB table at a time.
constexpr int LN_LO = 5;
constexpr int LN_CHUNK = 32768;
constexpr int LN_BLOCK = 1024;

template <int C>
__global__ void __launch_bounds__(LN_BLOCK) k_lone_hist(const uint8_t* __restrict__ scalars, size_t n, uint32_t own, uint32_t nbins,
                                                       uint32_t* __restrict__ hist, uint32_t nchunks, uint32_t chunk_sz) {
  constexpr uint32_t NB = 1u << (C - 1 - LN_LO);  // bins per window
  OG_DYN_LDS(smem);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
  const uint32_t chunk = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < nbins; k += LN_BLOCK) cnt[k] = 0;
  __syncthreads();
  const size_t lo = (size_t)chunk * chunk_sz, hi = lo + chunk_sz < n ? lo + chunk_sz : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += LN_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + i * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool) {
      if (win_owned(own, k)) atomicAdd(&cnt[win_slot(own, k) * NB + (b >> LN_LO)], 1u);
    });
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nbins; k += LN_BLOCK) hist[(size_t)k * nchunks + chunk] = cnt[k];
}

template <int C>
__global__ void __launch_bounds__(LN_BLOCK) k_lone_scatter(const uint8_t* __restrict__ scalars, size_t n, uint32_t own, uint32_t nbins,
                                                          const uint32_t* __restrict__ hist, uint32_t nchunks, uint32_t chunk_sz,
                                                          uint32_t spg, uint32_t* __restrict__ tmp) {
  // blockIdx.y = window group: the workgroup scatters only the digits of window slots [s0, s1) of its chunk, so that it
  // has (s1 - s0) x 1024 runs open instead of 16 384 (spg = slots per group; one group = every slot)
  constexpr uint32_t NB = 1u << (C - 1 - LN_LO);
  OG_DYN_LDS(smem);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem);
  const uint32_t chunk = blockIdx.x, s0 = blockIdx.y * spg, nslots = nbins / NB, s1 = s0 + spg < nslots ? s0 + spg : nslots;
  const uint32_t mine = (s1 - s0) * NB;
  for (uint32_t k = threadIdx.x; k < mine; k += LN_BLOCK) cur[k] = hist[(size_t)(s0 * NB + k) * nchunks + chunk];
  __syncthreads();
  const size_t lo = (size_t)chunk * chunk_sz, hi = lo + chunk_sz < n ? lo + chunk_sz : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += LN_BLOCK) {
    uint32_t l[8];
    load_scalar(scalars + i * 32, l);
    for_each_digit<C>(l, [&](int k, uint32_t b, bool neg) {
      if (!win_owned(own, k)) return;
      const uint32_t slot = win_slot(own, k);
      if (slot < s0 || slot >= s1) return;
      const uint32_t pos = atomicAdd(&cur[(slot - s0) * NB + (b >> LN_LO)], 1u);
      tmp[pos] = ((b & ((1u << LN_LO) - 1u)) << (32 - LN_LO)) | ((uint32_t)i << 1) | (neg ? 1u : 0u);
    });
  }
}

// ---- round 5: the first level through per-window DIGIT ARRAYS, runs staged in LDS ------------------------------------------
// Round 4's first level (k_lone_hist / k_lone_scatter above) stored every 4-byte entry where its LDS cursor pointed: 4 GB of
// 4-byte stores into 16 384 open runs per workgroup (4 096 with the four window groups), at ~1 TB/s -- 14.7 of the sort's
// 15.9 ms, the one HBM-bound stage of the lone MSM and it ran at a sixth of what the chip can write.  Staging the runs needs
// all the entries of a tile to belong to FEW bins, which a pass over scalars cannot give (a scalar feeds every window).  So:
//   k_lone_digits   ONE pass over the scalars: the signed digit of every owned window as a 16-bit value into dig[slot][i]
//                   (coalesced 2-byte stores: consecutive lanes, consecutive points) + the (slot, bin) histogram of the chunk.
//                   v = raw digit mod 2^16: 0 = no entry, 1 .. 2^15 = +v, above = -(2^16 - v).
//   scan            as before (chunk-minor exclusive scan -> every (bin, chunk) run's start)
//   k_lone_scatter_runs   one workgroup per (chunk, slot): ONE window's digits, so 1024 bins.  A tile of LN_TILE digits is
//                   counting-sorted by bin in LDS and every (tile, bin) run leaves through consecutive lanes -- ~16 entries,
//                   a whole 64-byte sector -- instead of one 4-byte store per lane-chosen address.  The point index is the
//                   position in dig[]: the digit arrays cost 2 B per (point, window) where the re-read scalars cost 32 B per
//                   point and pass.
// Traffic at 2^26 points: 2 + 2 GB (digits), 2 + 4 GB (runs) against 2 + 8 + 4 GB, and the 4 GB of entries go out in sectors.
constexpr int LN_TILE = 16384;

template <int C>
__global__ void __launch_bounds__(LN_BLOCK) k_lone_digits(const uint8_t* __restrict__ scalars, size_t n, uint32_t own, uint32_t nbins,
                                                         uint16_t* __restrict__ dig, uint32_t* __restrict__ hist, uint32_t nchunks,
                                                         uint32_t chunk_sz) {
  constexpr uint32_t NB = 1u << (C - 1 - LN_LO);  // bins per window
  OG_DYN_LDS(smem);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);
  const uint32_t chunk = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < nbins; k += LN_BLOCK) cnt[k] = 0;
  __syncthreads();
  const size_t lo = (size_t)chunk * chunk_sz, hi = lo + chunk_sz < n ? lo + chunk_sz : n;
  for (size_t i0 = lo; i0 < hi; i0 += LN_BLOCK) {  // (every lane of the workgroup takes part in the aggregated increments)
    const size_t i = i0 + threadIdx.x;
    uint32_t l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < hi) load_scalar(scalars + i * 32, l);
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < (255 + C - 1) / C; k++) {  // the digits of for_each_digit, zeros included
      const int bit = k * C, w = bit >> 5, sh = bit & 31;
      uint32_t raw = 0;
      if (w < 8) {
        uint64_t v = l[w];
        if (w + 1 < 8) v |= (uint64_t)l[w + 1] << 32;
        raw = (uint32_t)(v >> sh) & ((1u << C) - 1);
      }
      raw += carry;
      const bool neg = raw > (1u << (C - 1));
      carry = neg ? 1u : 0u;
      if (!win_owned(own, k)) continue;
      const uint32_t slot = win_slot(own, k), mag = neg ? (1u << C) - raw : raw;
      if (i < hi) dig[(size_t)slot * n + i] = (uint16_t)(raw & ((1u << C) - 1u));
      const bool entry = i < hi && mag != 0;
      if (entry) (void)OG_LDS_ATOMIC_INC_AGG(cnt, slot * NB + ((mag - 1) >> LN_LO));
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nbins; k += LN_BLOCK) hist[(size_t)k * nchunks + chunk] = cnt[k];
}

// A lane pulls its LN_TILE / LN_BLOCK = 16 consecutive digits of a tile into registers with two 16-byte loads (the first
// version re-read the digit array 2-byte by 2-byte in both LDS passes: 32 dependent global loads per lane and tile, one
// workgroup per CU -- 7.7 ms for 6 GB, rocprofv3 round 5), and the counters share storage (counts -> their scan -> the fill
// cursors: 12 KB instead of 20), so that TWO workgroups fit a CU's 160 KB and one's LDS passes overlap the other's loads / stores.
template <int C>
__global__ void __launch_bounds__(LN_BLOCK) k_lone_scatter_runs(const uint16_t* __restrict__ dig, size_t n, const uint32_t* __restrict__ hist,
                                                               uint32_t nchunks, uint32_t chunk_sz, uint32_t tile, uint32_t* __restrict__ tmp) {
  constexpr uint32_t NB = 1u << (C - 1 - LN_LO);
  constexpr int PER = LN_TILE / LN_BLOCK;  // digits per lane and tile
  static_assert(NB == LN_BLOCK, "one lane per bin");
  static_assert(PER == 16, "two uint4 loads per lane");
  __shared__ uint32_t buf[LN_TILE];                                   // 64 KB + 12 KB of counters: two workgroups per CU
  __shared__ uint32_t cur[NB], pos[NB], off[NB + 1];                  // pos: bin counts, then (in place) their scan, then the fill cursors
  const uint32_t chunk = blockIdx.x, slot = blockIdx.y, t = threadIdx.x;
  cur[t] = hist[(size_t)(slot * NB + t) * nchunks + chunk];
  const uint16_t* d = dig + (size_t)slot * n;
  const size_t c_lo = (size_t)chunk * chunk_sz, c_hi = c_lo + chunk_sz < n ? c_lo + chunk_sz : n;
  const uint32_t per = tile / LN_BLOCK > 0 ? tile / LN_BLOCK : 1;     // (hooks builds shrink the tile; per <= PER)
  for (size_t t_lo = c_lo; t_lo < c_hi; t_lo += (size_t)per * LN_BLOCK) {
    const size_t t_hi = t_lo + (size_t)per * LN_BLOCK < c_hi ? t_lo + (size_t)per * LN_BLOCK : c_hi;
    const size_t i0 = t_lo + (size_t)t * per;                         // this lane's digits: [i0, i0 + per)
    uint32_t v[PER];
    if (per == PER && i0 + PER <= t_hi && (((size_t)(d + i0)) & 15) == 0) {
      const uint4 q0 = *reinterpret_cast<const uint4*>(d + i0), q1 = *reinterpret_cast<const uint4*>(d + i0 + 8);
      const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int j = 0; j < PER; j++) v[j] = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
    } else {
#pragma unroll
      for (int j = 0; j < PER; j++) v[j] = ((uint32_t)j < per && i0 + j < t_hi) ? d[i0 + j] : 0u;
    }
    pos[t] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const uint32_t mag = v[j] > (1u << (C - 1)) ? (1u << C) - v[j] : v[j];
      if (mag) (void)OG_LDS_ATOMIC_INC_AGG(pos, (mag - 1) >> LN_LO);
    }
    __syncthreads();
    lds_excl_scan<NB>(pos, off, pos);                                  // (in place: the scan copies its input first)
    const uint32_t mine = off[t + 1] - off[t];                         // this bin's entries in the tile
    pos[t] = off[t];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const bool neg = v[j] > (1u << (C - 1));
      const uint32_t mag = neg ? (1u << C) - v[j] : v[j];
      if (mag) {
        const uint32_t b = mag - 1, bin = b >> LN_LO;
        buf[OG_LDS_ATOMIC_INC_AGG(pos, bin)] = ((b & ((1u << LN_LO) - 1u)) << (32 - LN_LO)) | ((uint32_t)(i0 + j) << 1) | (neg ? 1u : 0u);
      }
    }
    __syncthreads();
    const uint32_t total = off[NB];
    for (uint32_t s2 = t; s2 < total; s2 += LN_BLOCK) {
      const uint32_t bin = bin_of_slot(off, NB, s2);
      tmp[cur[bin] + (s2 - off[bin])] = buf[s2];
    }
    __syncthreads();
    cur[t] += mine;
    __syncthreads();
  }
}

// order[g][.] = bucket ids sorted by descending size (counting sort on min(size, ORDER_BINS - 1), in LDS).  256-lane
// workgroups for the same reason as k_scan_chunks: a 16-wave workgroup does not fit beside the accumulation.
constexpr int ORDER_BINS = 2048;
constexpr int ORDER_BLOCK = 256;
__global__ void __launch_bounds__(ORDER_BLOCK) k_bucket_order(const uint32_t* __restrict__ offsets, size_t nkeys,
                                                             uint32_t* __restrict__ order) {
  OG_FILLER_PRIO();
  constexpr int PER = ORDER_BINS / ORDER_BLOCK;
  __shared__ uint32_t bins[ORDER_BINS];
  __shared__ uint32_t part[ORDER_BLOCK];
  const int g = blockIdx.x, t = threadIdx.x;
  const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
  uint32_t* ord = order + (size_t)g * nkeys;
  for (int k = t; k < ORDER_BINS; k += ORDER_BLOCK) bins[k] = 0;
  __syncthreads();
  for (size_t k = t; k < nkeys; k += ORDER_BLOCK) {
    uint32_t sz = off[k + 1] - off[k];
    if (sz > ORDER_BINS - 1) sz = ORDER_BINS - 1;
    atomicAdd(&bins[ORDER_BINS - 1 - sz], 1u);
  }
  __syncthreads();
  // exclusive scan of the 2048 bins: PER consecutive bins per lane
  uint32_t loc[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    loc[j] = bins[PER * t + j];
    sum += loc[j];
  }
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < ORDER_BLOCK; d <<= 1) {
    uint32_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    bins[PER * t + j] = run;
    run += loc[j];
  }
  __syncthreads();
  for (size_t k = t; k < nkeys; k += ORDER_BLOCK) {
    uint32_t sz = off[k + 1] - off[k];
    if (sz > ORDER_BINS - 1) sz = ORDER_BINS - 1;
    ord[atomicAdd(&bins[ORDER_BINS - 1 - sz], 1u)] = (uint32_t)k;
  }
}

// lone big MSM over plain bases (see k_lone_hist): n <= 2^26, batch == 1, C = 16
static int digit_sort_lone(og_ctx* ctx, const std::string& tag, const uint8_t* scalars_d, size_t n, DigitSort& ds) {
  constexpr int C = 16;
  constexpr uint32_t NB = 1u << (C - 1 - LN_LO);
  const uint32_t nbins = (uint32_t)std::max(1, ds.n_own) * NB;
  // scalars per workgroup: a (chunk, bin) run is chunk / 1024 entries, and runs shorter than a few cache lines are written as
  // partial lines (the 16 384 runs a workgroup has open outlive L2): OG_LONE_CHUNK moves it (A/B)
  // Measured at 2^26 points (same box): 32 K scalars per workgroup 20.6 ms of sort, 128 K 19.6, 256 K 17.9 -- one workgroup per CU.
  const uint32_t chunk_sz = OG_HOOK_SET("OG_LONE_CHUNK") ? (uint32_t)std::max(64, (int)OG_HOOK_INT("OG_LONE_CHUNK", 0))
                                                      : (uint32_t)std::min<size_t>(8 * LN_CHUNK, std::max<size_t>(LN_CHUNK, n / 256));
  const uint32_t nchunks = (uint32_t)((n + chunk_sz - 1) / chunk_sz);
  const size_t len = (size_t)nbins * nchunks;
  uint32_t *hist = nullptr, *binoff = nullptr, *tmp = nullptr;
  OG_TRY(arena_get(ctx, (tag + ".lhist").c_str(), (len + 1) * 4, (void**)&hist));
  OG_TRY(arena_get(ctx, (tag + ".lbinoff").c_str(), ((size_t)nbins + 1) * 4, (void**)&binoff));
  OG_TRY(arena_get(ctx, (tag + ".ltmp").c_str(), (ds.ecap ? ds.ecap : 1) * 4, (void**)&tmp));
  const size_t lds = (size_t)nbins * 4;
  // round 5: per-window digit arrays + LDS-staged runs (k_lone_digits / k_lone_scatter_runs); hooks builds: OG_LONE_SORT_V1=1
  // keeps round 4's direct scatter for A/Bs
  const bool staged_runs = !OG_HOOK_INT("OG_LONE_SORT_V1", 0);
  uint16_t* dig = nullptr;
  if (staged_runs) {
    OG_TRY(arena_get(ctx, (tag + ".ldig").c_str(), (size_t)std::max(1, ds.n_own) * n * 2, (void**)&dig));
    hipLaunchKernelGGL(k_lone_digits<C>, dim3(nchunks), dim3(LN_BLOCK), lds, ctx->stream, scalars_d, n, ds.own_mask, nbins, dig, hist, nchunks, chunk_sz);
  } else {
    hipLaunchKernelGGL(k_lone_hist<C>, dim3(nchunks), dim3(LN_BLOCK), lds, ctx->stream, scalars_d, n, ds.own_mask, nbins, hist, nchunks, chunk_sz);
  }
  OG_HIP(hipGetLastError());
  uint32_t nblk = (uint32_t)std::min<size_t>(1024, std::max<size_t>(1, len >> 16));
  if (const char* e = OG_HOOK_STR("OG_SCAN_NBLK")) nblk = (uint32_t)std::min(1024, std::max(1, atoi(e)));  // test hook
  uint32_t* sums = nullptr;
  OG_TRY(arena_get(ctx, (tag + ".lssum").c_str(), (size_t)nblk * 4, (void**)&sums));
  hipLaunchKernelGGL(k_scan_slice_sums, dim3(nblk, 1), dim3(1024), 0, ctx->stream, hist, len, nblk, sums);
  hipLaunchKernelGGL(k_scan_slice_bases, dim3(1), dim3(1024), 0, ctx->stream, sums, nblk);
  hipLaunchKernelGGL(k_scan_slices, dim3(nblk, 1), dim3(1024), 0, ctx->stream, hist, len, nblk, nchunks, sums, binoff, (size_t)nbins);
  OG_HIP(hipGetLastError());
  // window groups of the scatter: four passes over the scalars, each with a quarter of the runs open (4 096 instead of
  // 16 384 per workgroup, 16 KB of cursors instead of 64).  Measured at 2^26 points, same box: groups 1 / 2 / 4 / 8 / 16 ->
  // 18.2 / 16.7 / 16.4 / 17.1 / 18.7 ms of sort (the extra reads of the scalars catch up).  OG_LONE_WGROUPS overrides.
  const uint32_t nslots = nbins / NB;
  const uint32_t wgroups = (uint32_t)std::max(1, std::min<int>((int)nslots, (int)OG_HOOK_INT("OG_LONE_WGROUPS", 4)));
  const uint32_t spg = (nslots + wgroups - 1) / wgroups;
  if (staged_runs) {
    const uint32_t tile = (uint32_t)std::min<long long>(LN_TILE, std::max<long long>(64, OG_HOOK_INT("OG_LONE_TILE", LN_TILE)));  // (hook: several tiles at toy size)
    hipLaunchKernelGGL(k_lone_scatter_runs<C>, dim3(nchunks, nslots), dim3(LN_BLOCK), 0, ctx->stream, dig, n, hist, nchunks, chunk_sz, tile, tmp);
  } else {
    hipLaunchKernelGGL(k_lone_scatter<C>, dim3(nchunks, (nslots + spg - 1) / spg), dim3(LN_BLOCK), (size_t)spg * NB * 4, ctx->stream, scalars_d, n,
                       ds.own_mask, nbins, hist, nchunks, chunk_sz, spg, tmp);
  }
  OG_HIP(hipGetLastError());
  // bins of >= 16 K entries: the run-staging kernel (whole-line writes); smaller ones go direct
  static const int force = (int)OG_HOOK_INT("OG_SORT_DIRECT", -1);
  const bool direct = force >= 0 ? force != 0 : (double)n * ds.n_own / nbins < 16384.0;
  if (direct)
    hipLaunchKernelGGL(k_sort_lo_direct<LN_LO>, dim3(nbins, 1), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, nbins, ds.entries, ds.ecap, ds.offsets,
                       ds.nkeys);
  else if (OG_HOOK_INT("OG_LONE_SORT_OLD", 0))  // A/B hook: the unbatched second level
    hipLaunchKernelGGL(k_sort_lo<LN_LO>, dim3(nbins, 1), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, nbins, ds.entries, ds.ecap, ds.offsets,
                       ds.nkeys);
  else {
    uint32_t *gcnt = nullptr, *gcur = nullptr;
    OG_TRY(arena_get(ctx, (tag + ".lgcnt").c_str(), ds.nkeys * 4, (void**)&gcnt));
    OG_TRY(arena_get(ctx, (tag + ".lgcur").c_str(), ds.nkeys * 4, (void**)&gcur));
    OG_HIP(hipMemsetAsync(gcnt, 0, ds.nkeys * 4, ctx->stream));
    hipLaunchKernelGGL(k_sub_count<LN_LO>, dim3(nbins, SB_SMAX), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, nbins, gcnt);
    hipLaunchKernelGGL(k_sub_offsets<LN_LO>, dim3(grid_for(nbins, 256)), dim3(256), 0, ctx->stream, binoff, nbins, gcnt, gcur, ds.offsets, ds.nkeys);
    hipLaunchKernelGGL(k_sub_scatter<LN_LO>, dim3(nbins, SB_SMAX), dim3(RS_BLOCK), 0, ctx->stream, tmp, binoff, nbins, gcur, ds.entries);
  }
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int msm_digit_sort(og_ctx* ctx, int slot, const uint8_t* scalars_d, size_t stride, size_t n, const uint32_t* map_d,
                   int batch, int c, int precomp, DigitSort* out) {
  return msm_digit_sort_windows(ctx, slot, scalars_d, stride, n, map_d, batch, c, precomp, 0, 1, out);
}

int msm_digit_sort_windows(og_ctx* ctx, int slot, const uint8_t* scalars_d, size_t stride, size_t n, const uint32_t* map_d,
                           int batch, int c, int precomp, int win_rank, int win_world, DigitSort* out) {
  OG_REQUIRE(c == 8 || c == 12 || c == 15 || c == 16 || c == 17, "msm: window must be 8, 12, 15, 16 or 17 bits");
  OG_REQUIRE(batch >= 1 && batch <= 65535, "msm: batch out of range");
  const int nwin = msm_nwin(c);
  OG_REQUIRE((double)n * nwin < 2147483648.0, "msm: n * nwin must be < 2^31");
  ProfScope ps(ctx, PROF_SORT, (double)n * batch);
  DigitSort ds;
  ds.n = n; ds.batch = batch; ds.c = c; ds.nwin = nwin; ds.precomp = precomp;
  OG_REQUIRE(win_world >= 1 && win_rank >= 0 && win_rank < win_world, "msm: bad window shard (rank, world)");
  ds.own_mask = 0;
  for (int k = 0; k < nwin; k++)
    if (k % win_world == win_rank) ds.own_mask |= 1u << k;  // round-robin: every rank gets high and low windows
  ds.n_own = __builtin_popcount(ds.own_mask);
  ds.nkeys = (size_t)(precomp ? 1 : std::max(1, ds.n_own)) << (c - 1);
  ds.ecap = n * (size_t)nwin;
  std::string tag = "ds" + std::to_string(slot);
  OG_TRY(arena_get(ctx, (tag + ".off").c_str(), (size_t)batch * (ds.nkeys + 1) * 4, (void**)&ds.offsets));
  OG_TRY(arena_get(ctx, (tag + ".cur").c_str(), (size_t)batch * ds.nkeys * 4, (void**)&ds.cursor));
  OG_TRY(arena_get(ctx, (tag + ".ent").c_str(), (size_t)batch * (ds.ecap ? ds.ecap : 1) * 4, (void**)&ds.entries));
  OG_TRY(arena_get(ctx, (tag + ".ord").c_str(), (size_t)batch * ds.nkeys * 4, (void**)&ds.order));
  static const bool use_order = !OG_HOOK_INT("OG_NO_ORDER", 0);
  auto finish = [&]() -> int {
    if (!use_order) {
      ds.order = nullptr;
      return OG_OK;
    }
    hipLaunchKernelGGL(k_bucket_order, dim3(batch), dim3(ORDER_BLOCK), 0, ctx->stream, ds.offsets, ds.nkeys, ds.order);
    OG_HIP(hipGetLastError());
    return OG_OK;
  };
  static const bool use_lds = !OG_HOOK_INT("OG_SORT_GLOBAL", 0);
  static const bool use_radix = !OG_HOOK_INT("OG_SORT_LEGACY", 0);
  if (precomp && use_lds && use_radix && (c == 15 || c == 16 || c == 17) && (double)n * nwin < (double)(1u << (32 - (c - 9) - 1))) {
    // the prover's shape (many proofs, n <= 2^18): two-level radix sort  (entry: c - 9 low bucket bits above (index << 1 | sign))
    OG_TRY(c == 15   ? digit_sort_radix<15>(ctx, tag, scalars_d, stride, n, map_d, batch, ds)
           : c == 16 ? digit_sort_radix<16>(ctx, tag, scalars_d, stride, n, map_d, batch, ds)
                     : digit_sort_radix<17>(ctx, tag, scalars_d, stride, n, map_d, batch, ds));
    OG_TRY(finish());
    *out = ds;
    return OG_OK;
  }
  OG_REQUIRE(c != 17, "msm: 17-bit windows need precomputed window tables and n x 15 < 2^23 (the two-level radix sort)");
  OG_REQUIRE(c != 15, "msm: 15-bit windows need precomputed window tables (the two-level radix sort)");
  // a lone big MSM over plain bases: one bucket set per window, sorted in two levels without global atomics
  const size_t lone_min = (size_t)OG_HOOK_INT("OG_LONE_MIN", (long long)1 << 18);  // (test hook, read per call)
  if (!precomp && use_lds && use_radix && c == 16 && batch == 1 && map_d == nullptr && n >= lone_min && n <= ((size_t)1 << 26) && ds.n_own >= 1) {
    OG_TRY(digit_sort_lone(ctx, tag, scalars_d, n, ds));
    // no size ordering: the 2^19 buckets hold n / 2^15 entries each give or take a few per cent (the outliers -- the top
    // window's, the "digit 1" bucket -- go to the heavy path), and a one-workgroup sort of 2^19 keys would cost milliseconds
    ds.order = nullptr;
    *out = ds;
    return OG_OK;
  }
  if (precomp && use_lds) {  // one bucket set per proof: the LDS-staged sort
    int r = c == 8 ? digit_sort_lds<8>(ctx, tag, scalars_d, stride, n, map_d, batch, ds)
                   : c == 12 ? digit_sort_lds<12>(ctx, tag, scalars_d, stride, n, map_d, batch, ds)
                             : digit_sort_lds<16>(ctx, tag, scalars_d, stride, n, map_d, batch, ds);
    OG_TRY(r);
    OG_TRY(finish());
    *out = ds;
    return OG_OK;
  }
  OG_HIP(hipMemsetAsync(ds.offsets, 0, (size_t)batch * (ds.nkeys + 1) * 4, ctx->stream));
  if (n > 0) {
    dim3 grid(grid_for(n, 256), batch), blk(256);
#define LAUNCH_C(CC)                                                                                         \
  hipLaunchKernelGGL(k_digit_hist<CC>, grid, blk, 0, ctx->stream, scalars_d, stride, n, map_d, precomp, ds.own_mask, \
                     ds.offsets, ds.nkeys)
    if (c == 8) LAUNCH_C(8); else if (c == 12) LAUNCH_C(12); else LAUNCH_C(16);
#undef LAUNCH_C
    OG_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_scan_offsets, dim3(batch), dim3(1024), 0, ctx->stream, ds.offsets, ds.cursor, ds.nkeys);
  OG_HIP(hipGetLastError());
  if (n > 0) {
    dim3 grid(grid_for(n, 256), batch), blk(256);
#define LAUNCH_C(CC)                                                                                           \
  hipLaunchKernelGGL(k_digit_scatter<CC>, grid, blk, 0, ctx->stream, scalars_d, stride, n, map_d, precomp, ds.own_mask, \
                     ds.cursor, ds.nkeys, ds.entries, ds.ecap)
    if (c == 8) LAUNCH_C(8); else if (c == 12) LAUNCH_C(12); else LAUNCH_C(16);
#undef LAUNCH_C
    OG_HIP(hipGetLastError());
  }
  OG_TRY(finish());
  *out = ds;
  return OG_OK;
}

// ---- dispatch to the per-group translation units ---------------------------------------
int msm_run_g1(og_ctx*, const og_bases*, const DigitSort&, uint8_t*, bool, int);
int msm_run_g2(og_ctx*, const og_bases*, const DigitSort&, uint8_t*, bool, int);
int msm_combine_g1(og_ctx*, const og_bases*, const uint8_t*, int, int, uint8_t*);
int msm_combine_g2(og_ctx*, const og_bases*, const uint8_t*, int, int, uint8_t*);
int msm_sum_ranks_g1(og_ctx*, const uint8_t*, size_t, int, int, uint8_t*);
int msm_sum_ranks_g2(og_ctx*, const uint8_t*, size_t, int, int, uint8_t*);
int bases_fill_g1(og_ctx*, og_bases*, const uint8_t*);
int bases_fill_g2(og_ctx*, og_bases*, const uint8_t*);
int xyzz_to_affine_bytes_g1(og_ctx*, const uint8_t*, uint8_t*, size_t);
int xyzz_to_affine_bytes_g2(og_ctx*, const uint8_t*, uint8_t*, size_t);

static int msm_run_any(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_xyzz_d, bool partial, int phase = MSM_FULL) {
  OG_REQUIRE(bases->c == ds.c && bases->precomp == ds.precomp, "msm: bases/digit-sort window mismatch");
  OG_REQUIRE(bases->n >= ds.n, "msm: more scalars than bases");
  OG_REQUIRE(!ds.precomp || bases->n == ds.n, "msm: precomputed tables need n == bases.n");
  OG_REQUIRE(phase == MSM_FULL || (!bases->is_g2 && ds.precomp && !partial), "msm: a merged pair is two G1 queries over precomputed window tables");
  return bases->is_g2 ? msm_run_g2(ctx, bases, ds, out_xyzz_d, partial, phase) : msm_run_g1(ctx, bases, ds, out_xyzz_d, partial, phase);
}

// (Over per-window tables -- one bucket set whatever the number of windows -- a digit sort restricted to SOME windows is a
// legitimate input of msm_run / msm_run_phase: the result is this rank's partial sum of the query, and the partials of the
// ranks simply add (window-sharded proving, groth16.hip).  Over plain bases the window points need the Horner combine, so a
// restricted sort must go through msm_run_partial.)
int msm_run_phase(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_xyzz_d, int phase) {
  OG_REQUIRE(ds.n_own == ds.nwin || ds.precomp, "msm_run_phase: the digit sort covers only some windows");
  OG_REQUIRE(phase == MSM_FULL || phase == MSM_FIRST || phase == MSM_SECOND, "msm_run_phase: bad phase");
  return msm_run_any(ctx, bases, ds, out_xyzz_d, false, phase);
}

int msm_run(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_xyzz_d) {
  OG_REQUIRE(ds.n_own == ds.nwin || ds.precomp, "msm_run: the digit sort covers only some windows (use msm_run_partial)");
  return msm_run_any(ctx, bases, ds, out_xyzz_d, false);
}

int msm_partial_slots(const og_bases* bases) { return bases->precomp ? 1 : bases->nwin; }

int msm_run_partial(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* partial_xyzz_d) {
  return msm_run_any(ctx, bases, ds, partial_xyzz_d, true);
}

int msm_combine(og_ctx* ctx, const og_bases* bases, const uint8_t* gathered_xyzz_d, int world, int batch, uint8_t* out_xyzz_d) {
  OG_REQUIRE(world >= 1 && batch >= 1, "msm_combine: bad world / batch");
  return bases->is_g2 ? msm_combine_g2(ctx, bases, gathered_xyzz_d, world, batch, out_xyzz_d)
                      : msm_combine_g1(ctx, bases, gathered_xyzz_d, world, batch, out_xyzz_d);
}

int msm_sum_ranks(og_ctx* ctx, int is_g2, const uint8_t* gathered_d, size_t rank_stride, int world, int batch, uint8_t* out_d) {
  OG_REQUIRE(world >= 1 && batch >= 1, "msm_sum_ranks: bad world / batch");
  return is_g2 ? msm_sum_ranks_g2(ctx, gathered_d, rank_stride, world, batch, out_d) : msm_sum_ranks_g1(ctx, gathered_d, rank_stride, world, batch, out_d);
}

int xyzz_to_affine_bytes(og_ctx* ctx, int is_g2, const uint8_t* xyzz_d, uint8_t* out_d, size_t count) {
  return is_g2 ? xyzz_to_affine_bytes_g2(ctx, xyzz_d, out_d, count) : xyzz_to_affine_bytes_g1(ctx, xyzz_d, out_d, count);
}

int bases_create(og_ctx* ctx, int is_g2, const uint8_t* points_d, size_t n, int c, int precomp, og_bases** out) {
  OG_REQUIRE(c == 8 || c == 12 || c == 15 || c == 16 || c == 17, "bases: window must be 8, 12, 15, 16 or 17 bits");
  OG_REQUIRE((c != 17 && c != 15) || precomp, "bases: 15- and 17-bit windows need precomputed window tables");
  og_bases* b = new og_bases();
  b->is_g2 = is_g2; b->n = n; b->c = c; b->nwin = msm_nwin(c); b->precomp = precomp ? 1 : 0; b->device = ctx->device;
  const size_t pb = is_g2 ? 128 : 64;
  const size_t ntab = b->precomp ? b->nwin : 1;
  hipError_t e = hipMalloc((void**)&b->tab_d, (n ? n : 1) * ntab * pb);
  if (e != hipSuccess) {
    delete b;
    set_error(std::string("bases: hipMalloc failed: ") + hipGetErrorString(e));
    return OG_ERR_HIP;
  }
  int r = is_g2 ? bases_fill_g2(ctx, b, points_d) : bases_fill_g1(ctx, b, points_d);
  if (r != OG_OK) {
    (void)hipFree(b->tab_d);
    delete b;
    return r;
  }
  *out = b;
  return OG_OK;
}

void bases_destroy(og_bases* b) {
  if (!b) return;
  if (b->tab_d) (void)hipFree(b->tab_d);
  delete b;
}

}  // namespace og
