// A/B variants of the bucket accumulation -- compiled ONLY into hooks builds (-DOG_AB_HOOKS: libowshen_gpu_hooks.so and the CPU
// interpreter build), never into the shipped library.  Every form here was measured against the default and lost; it stays
// because the tests run the forms against each other and the oracles (tests/test_gpu_msm.py, tests/test_emu_kernels.py):
//   k_accumulate            round 1-2's grid launch, one workgroup per 64 buckets          (OG_ACC_WAVES_G1 / _G2 = 0)
//   k_accumulate_g2_lds<., false>  the same for the G2 kernel with its accumulator in LDS
//   k_accumulate_p<Fq2>     the G2 register version: 256 registers, 108 B of scratch        (OG_G2_LDS = 0)
//   k_accumulate_affine     batched affine additions, built for G2 in round 4: 2.2x slower  (OG_G2_AFFINE = 1)
// Included by msm_impl.hip.h after the default kernels.
#pragma once
#ifdef OG_AB_HOOKS

namespace og {

// MINW = minimum waves per SIMD the register allocator must leave room for (launch_bounds' second argument):
// the G2 body wants ~370 registers (1 wave/SIMD); MINW = 2 caps it at 256 and trades spills for occupancy.
template <class T, int MINW>
__global__ void __launch_bounds__(256, MINW) k_accumulate(const uint8_t* __restrict__ tab, const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ entries, const uint32_t* __restrict__ order,
                                                   size_t nkeys, size_t ecap, uint8_t* __restrict__ buckets,
                                                   uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list,
                                                   uint32_t heavy_cap, uint32_t heavy_min) {
  size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (key >= nkeys) return;
  if (order) key = order[(size_t)g * nkeys + key];  // lanes of a wave get buckets of (nearly) equal size
  const uint32_t* off = offsets + (size_t)g * (nkeys + 1);
  const uint32_t* ent = entries + (size_t)g * ecap;
  uint32_t lo = off[key], hi = off[key + 1];
  XYZZ<T> acc = XYZZ<T>::inf();
  if (hi - lo > heavy_min) {
    // deferred to k_accumulate_heavy -- unless the list is full: then the bucket is accumulated right here
    // (slow but correct; heavy_count keeps counting, the consumer clamps it to heavy_cap)
    const uint32_t slot = atomicAdd(heavy_count, 1u);
    if (slot < heavy_cap) {
      heavy_list[2 * slot] = (uint32_t)g;
      heavy_list[2 * slot + 1] = (uint32_t)key;
      hi = lo;
    }
  }
  for (uint32_t p = lo; p < hi; p++) {
    const uint32_t e = ent[p];
    acc = xyzz_madd_signed(acc, gather_base<T>(tab, e), e & 1);
  }
  acc.store(buckets + ((size_t)g * nkeys + key) * XYZZ<T>::BYTES);
}

// ---- bucket accumulation by batched affine additions ---------------------------------------------------------------------
// An affine addition costs 2M + 1S and one inversion; Montgomery's trick turns n inversions into one plus 3M each, so an
// addition is 5M + 1S (+ its share of the one inversion) against the 8M + 2S of the XYZZ mixed addition.  The inversion
// (Fermat: 261 squarings + 130 products of Fq, ~67 k instructions) does not spread over lanes, so each LANE amortises its own:
// a lane owns AFF_K buckets and per round adds ONE entry to each of them -- AFF_K independent additions, one inversion.
// That is where G1 and G2 part: the inversion of an Fq2 element is ONE Fq inversion plus six products, but an Fq2 product
// is three times an Fq product, so for G2 the inversion weighs a third as much: with 128 buckets per lane an addition is
// ~4 000 instructions instead of 5 380; for G1 the same layout would not pay (1 340 + 67 000 / 128 against 1 956).
//   Layout: rank = position of a bucket in `order` (descending size).  Wave w of a bucket set takes ranks
// [w * 64 K, (w + 1) * 64 K), lane l slot k = rank w * 64 K + k * 64 + l: the 64 lanes of a wave work on 64 buckets of
// (nearly) equal size at every slot, and the slots of a lane die out together.  Per round and slot: forward pass -- gather the
// entry's base P, load the bucket's running sum A (affine, in the first half of the bucket's XYZZ slot), d = x_P - x_A (or
// 2 y_A when P = A; 1 when there is nothing to invert: A or P the point at infinity, P = -A), park the running product in
// the second half of the slot, multiply d in; then ONE inversion; backward pass -- the same loads again, 1 / d from the parked
// prefix, the addition.  Traffic ~770 B per addition (two gathers, three slot reads, two writes), all lane-private lines.
#ifndef OG_AFF_K
#define OG_AFF_K 128
#endif
constexpr int AFF_K = OG_AFF_K;

// meta[g][rank] = (bucket key, first entry, entries, -): one 16-byte record per slot, coalesced across the lanes of a wave.
// Buckets above heavy_min go to the heavy list exactly as in k_accumulate_p and get 0 entries here.
static __global__ void __launch_bounds__(256) k_affine_meta(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ order, size_t nkeys,
                                                           size_t nk_pad, uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list,
                                                           uint32_t heavy_cap, uint32_t heavy_min, uint4* __restrict__ meta) {
  const size_t rank = (size_t)blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
  if (rank >= nk_pad) return;
  uint4 m = {0xffffffffu, 0u, 0u, 0u};
  if (rank < nkeys) {
    const uint32_t key = order ? order[g * nkeys + rank] : (uint32_t)rank;
    const uint32_t* off = offsets + g * (nkeys + 1);
    const uint32_t lo = off[key];
    uint32_t len = off[key + 1] - lo;
    if (len > heavy_min) {
      const uint32_t slot = atomicAdd(heavy_count, 1u);
      if (slot < heavy_cap) {
        heavy_list[2 * slot] = (uint32_t)g;
        heavy_list[2 * slot + 1] = key;
        len = 0;
      }
    }
    m = {key, lo, len, 0u};
  }
  meta[g * nk_pad + rank] = m;
}

template <class T, int K, int MINW>
__global__ void __launch_bounds__(64, MINW) k_accumulate_affine(const uint8_t* __restrict__ tab, const uint4* __restrict__ meta,
                                                              const uint32_t* __restrict__ entries, size_t nkeys, size_t nk_pad, size_t ecap,
                                                              uint8_t* __restrict__ buckets) {
  constexpr size_t PB = XYZZ<T>::BYTES, AB = Affine<T>::BYTES;
  const size_t g = blockIdx.y;
  const uint4* mt = meta + g * nk_pad + (size_t)blockIdx.x * 64 * K + threadIdx.x;  // slot k of this lane: mt[k * 64]
  const uint32_t* ent = entries + g * ecap;
  uint8_t* bk = buckets + g * nkeys * PB;
  // round 0: the running sum starts as the bucket's first entry
  uint32_t rounds = 0;
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    const uint4 m = mt[(size_t)k * 64];
    if (m.z == 0) continue;
    rounds = m.z > rounds ? m.z : rounds;
    const uint32_t e = ent[m.y];
    const Affine<T> p = gather_base<T>(tab, e);
    ((e & 1) && !p.is_inf() ? affine_neg(p) : p).store(bk + (size_t)m.x * PB);
  }
#pragma unroll 1
  for (uint32_t r = 1; r < rounds; r++) {
    T run = T::one();
#pragma unroll 1
    for (int k = 0; k < K; k++) {  // forward: the denominators and their running product
      const uint4 m = mt[(size_t)k * 64];
      if (r >= m.z) continue;
      uint8_t* slot = bk + (size_t)m.x * PB;
      const uint32_t e = ent[m.y + r];
      const Affine<T> p = gather_base<T>(tab, e);
      const Affine<T> a = Affine<T>::load(slot);
      T d = T::one();
      if (!p.is_inf() && !a.is_inf()) {
        const T dx = f_sub(p.x, a.x);
        if (!dx.is_zero()) d = dx;
        else if (((e & 1) ? f_neg(p.y) : p.y) == a.y) d = f_dbl(a.y);  // P = A: the tangent (y = 0 is not on the curve)
      }
      FieldIO<T>::store(slot + AB, run);
      run = f_mul(run, d);
    }
    T inv = f_inv(run);
#pragma unroll 1
    for (int k = K - 1; k >= 0; k--) {  // backward: 1 / d from the parked prefix, then the addition
      const uint4 m = mt[(size_t)k * 64];
      if (r >= m.z) continue;
      uint8_t* slot = bk + (size_t)m.x * PB;
      const uint32_t e = ent[m.y + r];
      Affine<T> p = gather_base<T>(tab, e);
      if (p.is_inf()) continue;                     // A + O = A
      if (e & 1) p.y = f_neg(p.y);
      const Affine<T> a = Affine<T>::load(slot);
      if (a.is_inf()) { p.store(slot); continue; }  // O + P = P
      const T dx = f_sub(p.x, a.x);
      T d = dx, num;
      if (dx.is_zero()) {
        if (!(p.y == a.y)) { Affine<T>::inf().store(slot); continue; }  // P = -A
        d = f_dbl(a.y);
        const T xx = f_sqr(a.x);
        num = f_add(f_dbl(xx), xx);                 // 3 x^2 (the curves have a = 0)
      } else {
        num = f_sub(p.y, a.y);
      }
      const T inv_d = f_mul(inv, FieldIO<T>::load(slot + AB));
      inv = f_mul(inv, d);
      const T lam = f_mul(num, inv_d);
      const T x3 = f_sub(f_sub(f_sqr(lam), a.x), p.x);
      const T y3 = f_sub(f_mul(lam, f_sub(a.x, x3)), a.y);
      Affine<T>{x3, y3}.store(slot);
    }
  }
#pragma unroll 1
  for (int k = 0; k < K; k++) {  // the reduction kernels read XYZZ
    const uint4 m = mt[(size_t)k * 64];
    if (m.x == 0xffffffffu) continue;
    uint8_t* slot = bk + (size_t)m.x * PB;
    XYZZ<T> out = XYZZ<T>::inf();
    if (m.z) out = XYZZ<T>::from_affine(Affine<T>::load(slot));
    out.store(slot);
  }
}

// launches the A/B form the environment asks for, if any (*launched says whether it did)
template <class T>
static int ab_accumulate(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, const std::string& tag, int pw, bool lone_plain, uint8_t* buckets,
                         uint32_t* heavy_count, uint32_t* heavy_list, uint32_t heavy_cap, uint32_t heavy_min, bool* launched) {
  *launched = false;
  if (lone_plain) return OG_OK;
  const uint32_t nchunk = grid_for(ds.nkeys, 64);
  const bool g2_lds = OG_HOOK_INT("OG_G2_LDS", 1) != 0;
  if constexpr (std::is_same<T, Fq2>::value) {
    if (OG_HOOK_INT("OG_G2_AFFINE", 0)) {  // batched affine additions: AFF_K buckets per lane, one inversion per lane and round
      const size_t per_wave = (size_t)64 * AFF_K, nk_pad = grid_for(ds.nkeys, per_wave) * per_wave;
      uint4* meta = nullptr;
      OG_TRY(arena_get(ctx, ("msm.affmeta" + tag).c_str(), (size_t)ds.batch * nk_pad * sizeof(uint4), (void**)&meta));
      hipLaunchKernelGGL(k_affine_meta, dim3(grid_for(nk_pad, 256), ds.batch), dim3(256), 0, ctx->stream, ds.offsets, ds.order, ds.nkeys, nk_pad,
                         heavy_count, heavy_list, heavy_cap, heavy_min, meta);
      OG_HIP(hipGetLastError());
      hipLaunchKernelGGL((k_accumulate_affine<T, AFF_K, AccCfg<T>::MINW>), dim3((unsigned)(nk_pad / per_wave), ds.batch), dim3(64), 0, ctx->stream,
                         bases->tab_d, meta, ds.entries, ds.nkeys, nk_pad, ds.ecap, buckets);
      *launched = true;
      return OG_OK;
    }
    if (pw > 0 && !g2_lds) {  // persistent, accumulator in registers
      const unsigned pgrid = (unsigned)std::min<size_t>((size_t)nchunk * ds.batch, (size_t)pw * ctx->n_cu);
      hipLaunchKernelGGL((k_accumulate_p<T, AccCfg<T>::MINW, false>), dim3(pgrid), dim3(64), 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries,
                         ds.order, ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
      *launched = true;
      return OG_OK;
    }
  }
  if (pw > 0) return OG_OK;  // the default persistent launch
  const dim3 grid(nchunk, ds.batch), blk(64);
  if constexpr (std::is_same<T, Fq2>::value) {
    if (g2_lds) {
      hipLaunchKernelGGL((k_accumulate_g2_lds<AccCfg<T>::MINW, false>), grid, blk, 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries, ds.order,
                         ds.nkeys, ds.ecap, buckets, heavy_count, heavy_list, heavy_cap, heavy_min, nchunk, (uint32_t)ds.batch);
      *launched = true;
      return OG_OK;
    }
  }
  hipLaunchKernelGGL((k_accumulate<T, AccCfg<T>::MINW>), grid, blk, 0, ctx->stream, bases->tab_d, ds.offsets, ds.entries, ds.order, ds.nkeys, ds.ecap,
                     buckets, heavy_count, heavy_list, heavy_cap, heavy_min);
  *launched = true;
  return OG_OK;
}

}  // namespace og
#endif  // OG_AB_HOOKS
