// The table side of the MSM layer, included by msm_impl.hip.h (one instantiation per group): import of canonical affine points
// into Montgomery form, the per-window shifted tables tab[k][i] = 2^(c k) P_i (one inversion per lane for SHIFT_PER points),
// and the XYZZ -> canonical affine bytes conversion of results.  Key-load / result kernels, not proving kernels.
#pragma once

namespace og {

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

// out[i] = 2^c * in[i]  (affine in, affine out).  A lane carries SHIFT_PER points (strided by the lane count, so loads
// stay coalesced) and inverts once for all of them: with w = ZZ * ZZZ, x = X ZZZ / w and y = Y ZZ / w; the numerators
// wait in the output slots while the lane forms the running product of the w, one Fermat inversion (~380
// multiplications, more than the c doublings themselves) undoes it, and a backward walk hands each point its 1 / w.
// The two small per-lane arrays live in scratch memory -- this is a key-load kernel, not a proving kernel.
constexpr int SHIFT_PER = 8;
template <class T>
__global__ void __launch_bounds__(256) k_bases_shift(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n, int c) {
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (size_t)gridDim.x * blockDim.x;
  if (lane >= n) return;
  T w[SHIFT_PER], pre[SHIFT_PER];
  T run = T::one();
  int cnt = 0;
#pragma unroll 1
  for (; cnt < SHIFT_PER && lane + cnt * lanes < n; cnt++) {
    const size_t i = lane + cnt * lanes;
    XYZZ<T> a = XYZZ<T>::from_affine(Affine<T>::load(in + i * Affine<T>::BYTES));
#pragma unroll 1
    for (int k = 0; k < c; k++) a = xyzz_dbl(a);
    const bool inf = a.is_inf();
    Affine<T> num = {f_mul(a.x, a.zzz), f_mul(a.y, a.zz)};
    if (inf) num = Affine<T>::inf();
    num.store(out + i * Affine<T>::BYTES);
    w[cnt] = inf ? T::one() : f_mul(a.zz, a.zzz);
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
    a.x = f_mul(a.x, ij);
    a.y = f_mul(a.y, ij);
    a.store(out + i * Affine<T>::BYTES);
  }
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
  const dim3 sgrid(grid_for((n + SHIFT_PER - 1) / SHIFT_PER, 256));
  for (size_t k = 1; k < ntab; k++)
    hipLaunchKernelGGL(k_bases_shift<T>, sgrid, blk, 0, ctx->stream, b->tab_d + (k - 1) * n * pb, b->tab_d + k * n * pb, n, b->c);
  OG_HIP(hipGetLastError());
  OG_HIP(hipStreamSynchronize(ctx->stream));
  return OG_OK;
}

}  // namespace og
