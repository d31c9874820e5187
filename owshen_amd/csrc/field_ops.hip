// Elementwise field kernels behind og_field_op_d / og_field_mulchain_d:
// the parity surface for Montgomery add/sub/mul/inv and the mulmod/s micro-benchmark.
#include "ctx.h"
#include "field.hip.h"

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

}  // namespace og
