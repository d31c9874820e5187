// Templated MSM kernels + launchers, instantiated once per group in msm_g1.hip / msm_g2.hip
// (separate translation units so the two big instantiations compile in parallel).
#pragma once
#include "msm.cuh"
#include "ec.cuh"

namespace og {

constexpr int HEAVY = 2048;  // bucket sizes above this go to the workgroup-per-bucket path
constexpr int RS = 4;        // reduction radix

// ---- bucket accumulation ----------------------------------------------------------
template <class T> struct AccCfg;
template <> struct AccCfg<Fq> { static constexpr int MINW = 1, ALT_MINW = 5; };
template <> struct AccCfg<Fq2> { static constexpr int MINW = 2, ALT_MINW = 3; };

template <class T>
__device__ __forceinline__ Affine<T> gather_base(const uint8_t* __restrict__ tab, uint32_t e) {
  Affine<T> p = Affine<T>::load(tab + (size_t)(e >> 1) * Affine<T>::BYTES);
  if (e & 1) p.y = f_neg(p.y);
  return p;
}

// MINW = minimum waves per SIMD the register allocator must leave room for (launch_bounds' second argument):
// the G2 body wants ~370 registers (1 wave/SIMD); MINW = 2 caps it at 256 and trades spills for occupancy.
template <class T, int MINW>
__global__ void __launch_bounds__(256, MINW) k_accumulate(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ entries, const uint32_t* __restrict__ order,
                                                   size_t nkeys, size_t ecap, uint8_t* __restrict__ buckets,
                                                   uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list,
                                                   uint32_t heavy_cap) {
  size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (key >= nkeys) return;
  if (order) key = order[(size_t)g * nkeys + key];  // lanes of a wave get buckets of (nearly) equal size
  const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
  const uint32_t* ent = entries + (size_t)g * ecap;
  uint32_t lo = off[key], hi = off[key + 1];
  XYZZ<T> acc = XYZZ<T>::inf();
  if (hi - lo > (uint32_t)HEAVY) {
    uint32_t slot = atomicAdd(heavy_count, 1u);
    if (slot < heavy_cap) {
      heavy_list[2 * slot] = (uint32_t)g;
      heavy_list[2 * slot + 1] = (uint32_t)key;
    }
  } else {
    for (uint32_t p = lo; p < hi; p++) acc = xyzz_madd(acc, gather_base<T>(tab, ent[p]));
  }
  acc.store(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
}

template <class T>
__global__ void __launch_bounds__(256) k_accumulate_heavy(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ entries, size_t nkeys, size_t ecap,
                                                         uint8_t* __restrict__ buckets, const uint32_t* __restrict__ heavy_count,
                                                         const uint32_t* __restrict__ heavy_list, uint32_t heavy_cap) {
  OG_DYN_LDS(smem);
  uint32_t nh = *heavy_count;
  if (nh > heavy_cap) nh = heavy_cap;
  for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
    const uint32_t g = heavy_list[2 * h], key = heavy_list[2 * h + 1];
    const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
    const uint32_t* ent = entries + (size_t)g * ecap;
    uint32_t lo = off[key], hi = off[key + 1];
    XYZZ<T> acc = XYZZ<T>::inf();
    for (uint32_t p = lo + threadIdx.x; p < hi; p += blockDim.x) acc = xyzz_madd(acc, gather_base<T>(tab, ent[p]));
#pragma unroll 1
    for (int d = blockDim.x / 2; d >= 1; d >>= 1) {
      __syncthreads();
      if ((int)threadIdx.x >= d && (int)threadIdx.x < 2 * d) acc.store(smem + (size_t)(threadIdx.x - d) * XYZZ<T>::BYTES);
      __syncthreads();
      if ((int)threadIdx.x < d) acc = xyzz_add(acc, XYZZ<T>::load(smem + (size_t)threadIdx.x * XYZZ<T>::BYTES));
    }
    if (threadIdx.x == 0) acc.store(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
    __syncthreads();
  }
}

// ---- bucket reduction ---------------------------------------------------------------
// One level of  G(items) = sum_i i * item_i  over `nsets` independent item arrays of length
// n_in (n_out = ceil(n_in / RS) segments each).  Per segment u:
//   R'_u = RS * sum_j item_{u RS + j}            (pre-scaled so deeper levels carry RS^level)
//   P'_u = sum_j j * item_{u RS + j} + sum_j P_{u RS + j}
// After the last level (n_out == 1): G = P'_0.   The MSM wants sum_b (b+1) * bucket_b
// = G(buckets) + sum(buckets) = P_final + R_final / RS^levels; to avoid the division the first
// level is called with `plus_one`, which uses weights j+1 on level 0 ... see msm_run.
template <class T>
__global__ void __launch_bounds__(64) k_reduce_level(const uint8_t* __restrict__ items, const uint8_t* __restrict__ p_in,
                                                    size_t n_in, size_t n_out, size_t nsets, uint8_t* __restrict__ r_out,
                                                    uint8_t* __restrict__ p_out, int has_p) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * nsets) return;
  size_t set = t / n_out, u = t % n_out;
  const size_t base = set * n_in + u * RS;
  // One inlined group-law site, driven by a rolled op loop (see ec.cuh on why nothing is out of line):
  //   s = 0,2,4: run += item[3 - s/2]     s = 1,3,5: acc += run     s = 6: run += item[0]
  //   s = 7..10: acc += P[s - 7]          s = 11,12: run += run  (R' = 4 * run)
  static_assert(RS == 4, "the op table below is written for radix 4");
  XYZZ<T> run = XYZZ<T>::inf(), acc = XYZZ<T>::inf();
#pragma unroll 1
  for (int s = 0; s < 13; s++) {
    bool to_run, rhs_run, en = true;
    const uint8_t* ptr = items;
    if (s < 7) {
      if (s & 1) {
        to_run = false; rhs_run = true;
      } else {
        const int j = 3 - (s >> 1);
        to_run = true; rhs_run = false;
        ptr = items + (base + j) * XYZZ<T>::BYTES;
        en = u * RS + j < n_in;
      }
    } else if (s < 11) {
      const int j = s - 7;
      to_run = false; rhs_run = false;
      en = has_p && (u * RS + j < n_in);
      ptr = p_in + (base + j) * XYZZ<T>::BYTES;
    } else {
      to_run = true; rhs_run = true;
    }
    if (!en) continue;
    XYZZ<T> rhs = run;
    if (!rhs_run) rhs = XYZZ<T>::load(ptr);
    const XYZZ<T> res = xyzz_add(to_run ? run : acc, rhs);
    if (to_run) run = res; else acc = res;
  }
  run.store(r_out + (set * n_out + u) * XYZZ<T>::BYTES);
  acc.store(p_out + (set * n_out + u) * XYZZ<T>::BYTES);
}

// plain sum of every set's items (tree by RS): used for the "+ sum(buckets)" term
template <class T>
__global__ void __launch_bounds__(64) k_sum_level(const uint8_t* __restrict__ items, size_t n_in, size_t n_out, size_t nsets,
                                                 uint8_t* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * nsets) return;
  size_t set = t / n_out, u = t % n_out;
  XYZZ<T> acc = XYZZ<T>::inf();
#pragma unroll 1
  for (int j = 0; j < RS; j++)
    if (u * RS + j < n_in) acc = xyzz_add(acc, XYZZ<T>::load(items + (set * n_in + u * RS + j) * XYZZ<T>::BYTES));
  acc.store(out + (set * n_out + u) * XYZZ<T>::BYTES);
}

// result[g] = sum_k 2^(c k) * (G_k + S_k) over the nsets_per_g window sets (Horner), one lane per g
template <class T>
__global__ void __launch_bounds__(64) k_window_combine(const uint8_t* __restrict__ gsum, const uint8_t* __restrict__ ssum,
                                                      int nsets_per_g, int c, int batch, uint8_t* __restrict__ out) {
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
    XYZZ<T> rhs = acc;
    if (q == c) rhs = XYZZ<T>::load(gsum + idx * XYZZ<T>::BYTES);
    if (q == c + 1) rhs = XYZZ<T>::load(ssum + idx * XYZZ<T>::BYTES);
    acc = xyzz_add(acc, rhs);
  }
  acc.store(out + (size_t)g * XYZZ<T>::BYTES);
}

template <class T>
int msm_run_t(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_d) {
  const size_t PB = XYZZ<T>::BYTES;
  const size_t B = (size_t)1 << (ds.c - 1);
  const int nsets_per_g = ds.precomp ? 1 : ds.nwin;
  const size_t nsets = (size_t)ds.batch * nsets_per_g;
  const char* sfx = bases->is_g2 ? "2" : "1";
  uint8_t* buckets = nullptr;
  uint32_t* heavy = nullptr;
  const uint32_t heavy_cap = 1u << 16;
  OG_TRY(arena_get(ctx, (std::string("msm.buckets") + sfx).c_str(), nsets * B * PB, (void**)&buckets));
  OG_TRY(arena_get(ctx, "msm.heavy", (size_t)(2 * heavy_cap + 4) * 4, (void**)&heavy));
  OG_HIP(hipMemsetAsync(heavy, 0, 4, ctx->stream));
  uint32_t* heavy_count = heavy;
  uint32_t* heavy_list = heavy + 4;
  {
    ProfScope ps(ctx, bases->is_g2 ? PROF_ACC_G2 : PROF_ACC_G1, (double)ds.n * ds.batch);
    dim3 grid(grid_for(ds.nkeys, 256), ds.batch), blk(256);
    static const int variant = getenv("OG_ACC_MINW") ? atoi(getenv("OG_ACC_MINW")) : 0;
    if (variant == 2)
      hipLaunchKernelGGL((k_accumulate<T, AccCfg<T>::ALT_MINW>), grid, blk, 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries,
                         ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap);
    else
      hipLaunchKernelGGL((k_accumulate<T, AccCfg<T>::MINW>), grid, blk, 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries,
                         ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "accumulate");
    hipLaunchKernelGGL(k_accumulate_heavy<T>, dim3(512), dim3(256), 128 * PB, ctx->stream, bases->tab_d, ds.offsets,
                       ds.entries, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "accumulate_heavy");
  }
  // weighted reduction: sum_b (b+1) B_b = G(B) + S(B)
  ProfScope ps_red(ctx, bases->is_g2 ? PROF_REDUCE_G2 : PROF_REDUCE_G1, (double)nsets * B);
  size_t lvl_cap = nsets * ((B + RS - 1) / RS);
  uint8_t *r0, *r1, *p0, *p1, *s0, *s1;
  OG_TRY(arena_get(ctx, (std::string("msm.r0") + sfx).c_str(), lvl_cap * PB, (void**)&r0));
  OG_TRY(arena_get(ctx, (std::string("msm.r1") + sfx).c_str(), lvl_cap * PB, (void**)&r1));
  OG_TRY(arena_get(ctx, (std::string("msm.p0") + sfx).c_str(), lvl_cap * PB, (void**)&p0));
  OG_TRY(arena_get(ctx, (std::string("msm.p1") + sfx).c_str(), lvl_cap * PB, (void**)&p1));
  OG_TRY(arena_get(ctx, (std::string("msm.s0") + sfx).c_str(), lvl_cap * PB, (void**)&s0));
  OG_TRY(arena_get(ctx, (std::string("msm.s1") + sfx).c_str(), lvl_cap * PB, (void**)&s1));
  const uint8_t* items = buckets;
  const uint8_t* sitems = buckets;
  const uint8_t* pin = nullptr;
  size_t n_in = B;
  int lvl = 0;
  while (n_in > 1) {
    size_t n_out = (n_in + RS - 1) / RS;
    uint8_t* ro = (lvl & 1) ? r1 : r0;
    uint8_t* po = (lvl & 1) ? p1 : p0;
    uint8_t* so = (lvl & 1) ? s1 : s0;
    unsigned gsz = grid_for(n_out * nsets, 64);
    hipLaunchKernelGGL(k_reduce_level<T>, dim3(gsz), dim3(64), 0, ctx->stream, items, pin, n_in, n_out, nsets, ro, po,
                       pin ? 1 : 0);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "reduce_level");
    hipLaunchKernelGGL(k_sum_level<T>, dim3(gsz), dim3(64), 0, ctx->stream, sitems, n_in, n_out, nsets, so);
    OG_HIP(hipGetLastError());
    OG_STEP(ctx, "sum_level");
    items = ro; pin = po; sitems = so;
    n_in = n_out;
    lvl++;
  }
  // c >= 8 so B >= 128 and at least one level ran: pin = G per set, sitems = S per set
  hipLaunchKernelGGL(k_window_combine<T>, dim3(grid_for(ds.batch, 64)), dim3(64), 0, ctx->stream, pin, sitems, nsets_per_g,
                     ds.c, ds.batch, out_d);
  OG_HIP(hipGetLastError());
  OG_STEP(ctx, "window_combine");
  return OG_OK;
}

// ---- bases -------------------------------------------------------------------------

template <class T>
__global__ void __launch_bounds__(256) k_bases_import(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<T> p = Affine<T>::load(in + i * Affine<T>::BYTES);
  p.x = FieldIO<T>::to_mont(p.x);
  p.y = FieldIO<T>::to_mont(p.y);
  p.store(out + i * Affine<T>::BYTES);
}

// out[i] = 2^c * in[i]  (affine in, affine out)
template <class T>
__global__ void __launch_bounds__(256) k_bases_shift(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n, int c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<T> p = Affine<T>::load(in + i * Affine<T>::BYTES);
  XYZZ<T> a = XYZZ<T>::from_affine(p);
#pragma unroll 1
  for (int k = 0; k < c; k++) a = xyzz_dbl(a);
  xyzz_to_affine(a).store(out + i * Affine<T>::BYTES);
}

template <class T>
__global__ void __launch_bounds__(256) k_xyzz_to_affine_bytes(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<T> a = xyzz_to_affine(XYZZ<T>::load(in + i * XYZZ<T>::BYTES));
  a.x = FieldIO<T>::from_mont(a.x);
  a.y = FieldIO<T>::from_mont(a.y);
  a.store(out + i * Affine<T>::BYTES);
}


template <class T>
int xyzz_to_affine_bytes_t(og_ctx* ctx, const uint8_t* xyzz_d, uint8_t* out_d, size_t count) {
  if (count == 0) return OG_OK;
  hipLaunchKernelGGL(k_xyzz_to_affine_bytes<T>, dim3(grid_for(count, 256)), dim3(256), 0, ctx->stream, xyzz_d, out_d, count);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

template <class T>
int bases_fill_t(og_ctx* ctx, og_bases* b, const uint8_t* points_d) {
  const size_t n = b->n, pb = Affine<T>::BYTES;
  const size_t ntab = b->precomp ? b->nwin : 1;
  if (n == 0) return OG_OK;
  dim3 grid(grid_for(n, 256)), blk(256);
  hipLaunchKernelGGL(k_bases_import<T>, grid, blk, 0, ctx->stream, points_d, b->tab_d, n);
  for (size_t k = 1; k < ntab; k++)
    hipLaunchKernelGGL(k_bases_shift<T>, grid, blk, 0, ctx->stream, b->tab_d + (k - 1) * n * pb, b->tab_d + k * n * pb, n, b->c);
  OG_HIP(hipGetLastError());
  OG_HIP(hipStreamSynchronize(ctx->stream));
  return OG_OK;
}

}  // namespace og
