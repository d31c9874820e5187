// Elementwise field kernels behind og_field_op_d / og_field_mulchain_d:
// the parity surface for Montgomery add/sub/mul/inv and the mulmod/s micro-benchmark.
#include "ctx.h"
#include "field.hip.h"
#include "field_w9.hip.h"

namespace og {

template <class M>
__global__ void __launch_bounds__(256) k_field_op(int op, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                 uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<M> x = fe_to_mont(fe_load<M>(a + i * 32));
  Fe<M> y = op == 3 ? Fe<M>::zero() : fe_to_mont(fe_load<M>(b + i * 32));
  Fe<M> r;
  switch (op) {
    case 0: r = fe_add(x, y); break;
    case 1: r = fe_sub(x, y); break;
    case 2: r = fe_mul(x, y); break;
    default: r = x.is_zero() ? x : fe_inv(x); break;
  }
  fe_store(out + i * 32, fe_from_mont(r));
}

template <class M>
__global__ void __launch_bounds__(256) k_mulchain(uint8_t* __restrict__ x, const uint8_t* __restrict__ y, size_t n, int iters) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<M> a = fe_load<M>(x + i * 32);
  Fe<M> b = fe_load<M>(y + i * 32);
  for (int k = 0; k < iters; k++) a = fe_mul(a, b);
  fe_store(x + i * 32, a);
}

// The same chain with ONE element per wave in the w9 form (field_w9.hip.h): the limbs of x over nine lanes, y wave-uniform.
// Same Montgomery digits, so the same bytes as k_mulchain; *cycles = the longest wave's loop in shader cycles.
template <class M>
__global__ void __launch_bounds__(64) k_mulchain_w9(uint8_t* __restrict__ x, const uint8_t* __restrict__ y, size_t n, int iters,
                                                   unsigned long long* __restrict__ cycles) {
  const size_t i = blockIdx.x;
  const int lane = threadIdx.x;
  if (i >= n) return;
  const Fe<M> b = fe_load<M>(y + i * 32);
  U9 yu;
#pragma unroll
  for (int k = 0; k < 9; k++) yu.l[k] = OG_W9_FIRST(b.l[k]);
  const uint32_t nj = w9_modulus_limb<M>(lane);
  uint32_t xs = w9_spread(fe_load<M>(x + i * 32), lane);
  const unsigned long long t0 = OG_SHADER_CYCLES();
  for (int k = 0; k < iters; k++) xs = w9_mul<M>(yu, xs, nj);
  const unsigned long long t1 = OG_SHADER_CYCLES();
  const Fe<M> lazy = w9_collect<M>(xs);
  if (lane == 0) {
    fe_store(x + i * 32, fe_from_lazy_limbs<M>(lazy.l));
    if (cycles) atomicMax(cycles, t1 - t0);
  }
}

// the lane-local chain of one wave, timed the same way (the yardstick for the w9 form: n lanes of ONE wave)
template <class M>
__global__ void __launch_bounds__(64) k_mulchain_cycles(uint8_t* __restrict__ x, const uint8_t* __restrict__ y, size_t n, int iters,
                                                       unsigned long long* __restrict__ cycles) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<M> a = fe_load<M>(x + i * 32);
  const Fe<M> b = fe_load<M>(y + i * 32);
  const unsigned long long t0 = OG_SHADER_CYCLES();
  for (int k = 0; k < iters; k++) a = fe_mul(a, b);
  const unsigned long long t1 = OG_SHADER_CYCLES();
  fe_store(x + i * 32, a);
  if (cycles && (threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

int field_op(og_ctx* ctx, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  if (n == 0) return OG_OK;
  dim3 g(grid_for(n, 256)), blk(256);
  if (field == 0)
    hipLaunchKernelGGL(k_field_op<FrParams>, g, blk, 0, ctx->stream, op, a, b, out, n);
  else
    hipLaunchKernelGGL(k_field_op<FqParams>, g, blk, 0, ctx->stream, op, a, b, out, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int field_mulchain(og_ctx* ctx, int field, uint8_t* x, const uint8_t* y, size_t n, int iters, float* ms) {
  dim3 g(grid_for(n, 256)), blk(256);
  OG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  if (field == 0)
    hipLaunchKernelGGL(k_mulchain<FrParams>, g, blk, 0, ctx->stream, x, y, n, iters);
  else
    hipLaunchKernelGGL(k_mulchain<FqParams>, g, blk, 0, ctx->stream, x, y, n, iters);
  OG_HIP(hipGetLastError());
  OG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  OG_HIP(hipEventSynchronize(ctx->ev1));
  OG_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return OG_OK;
}

// form 0: lane-local fe_mul, one lane per element (64-lane workgroups); form 1: w9, one wave per element
int field_mulchain_lat(og_ctx* ctx, int field, int form, uint8_t* x, const uint8_t* y, size_t n, int iters, float* ms, uint64_t* cycles_out) {
  unsigned long long* cyc = nullptr;
  OG_HIP(hipMalloc((void**)&cyc, 8));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{cyc};
  OG_HIP(hipMemsetAsync(cyc, 0, 8, ctx->stream));
  OG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  if (form == 1) {
    if (field == 0)
      hipLaunchKernelGGL(k_mulchain_w9<FrParams>, dim3((unsigned)n), dim3(64), 0, ctx->stream, x, y, n, iters, cyc);
    else
      hipLaunchKernelGGL(k_mulchain_w9<FqParams>, dim3((unsigned)n), dim3(64), 0, ctx->stream, x, y, n, iters, cyc);
  } else {
    if (field == 0)
      hipLaunchKernelGGL(k_mulchain_cycles<FrParams>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, x, y, n, iters, cyc);
    else
      hipLaunchKernelGGL(k_mulchain_cycles<FqParams>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, x, y, n, iters, cyc);
  }
  OG_HIP(hipGetLastError());
  OG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  OG_HIP(hipEventSynchronize(ctx->ev1));
  OG_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  unsigned long long c = 0;
  OG_HIP(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  if (cycles_out) *cycles_out = c;
  return OG_OK;
}

}  // namespace og
