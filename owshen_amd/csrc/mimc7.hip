// Batched MiMC7 kernels: 2-to-1 hash, Merkle paths, full-tree build.
// One lane per hash; purely VALU-bound (728 mulmod per 2-to-1 hash vs 96 B of traffic).
#include "ctx.h"
#include "mimc7.cuh"
#include <string.h>

namespace og {

__global__ void __launch_bounds__(256) k_to_mont_fr(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store(out + i * 32, fe_to_mont(fe_load<FrParams>(in + i * 32)));
}

__global__ void __launch_bounds__(256) k_mimc7_hash2(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ left,
                                                    const uint8_t* __restrict__ right, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr l = fe_to_mont(fe_load<FrParams>(left + i * 32));
  Fr r = fe_to_mont(fe_load<FrParams>(right + i * 32));
  fe_store(out + i * 32, fe_from_mont(mimc7_hash2(consts, l, r)));
}

// one tree level: out[i] = H(in[2i], in[2i+1]); canonical in/out
__global__ void __launch_bounds__(256) k_mimc7_tree_level(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ in,
                                                         uint8_t* __restrict__ out, size_t n_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  Fr l = fe_to_mont(fe_load<FrParams>(in + (2 * i) * 32));
  Fr r = fe_to_mont(fe_load<FrParams>(in + (2 * i + 1) * 32));
  fe_store(out + i * 32, fe_from_mont(mimc7_hash2(consts, l, r)));
}

__global__ void __launch_bounds__(64) k_mimc7_merkle_paths(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ leaves,
                                                          const uint64_t* __restrict__ indices, const uint8_t* __restrict__ siblings,
                                                          int depth, uint8_t* __restrict__ nodes, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t idx = indices[i];
  uint8_t* o = nodes + i * (size_t)(depth + 1) * 32;
  Fr cur = fe_load<FrParams>(leaves + i * 32);
  fe_store(o, cur);
  cur = fe_to_mont(cur);
  for (int l = 0; l < depth; l++) {
    Fr sib = fe_to_mont(fe_load<FrParams>(siblings + (i * (size_t)depth + l) * 32));
    bool right = (idx >> l) & 1;
    Fr a = right ? sib : cur;
    Fr b = right ? cur : sib;
    cur = mimc7_hash2(consts, a, b);
    fe_store(o + (size_t)(l + 1) * 32, fe_from_mont(cur));
  }
}

int mimc7_init(og_ctx* ctx) {
  // c_0 = 0; c_i = keccak256^(i+1)("mimc") (big-endian integer) mod r
  static const uint64_t RMOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                   0x30644e72e131a029ULL};
  uint8_t h[32];
  keccak256((const uint8_t*)"mimc", 4, h);
  memset(ctx->mimc_consts_canon, 0, sizeof(ctx->mimc_consts_canon));
  for (int i = 1; i < MIMC7_ROUNDS; i++) {
    uint8_t h2[32];
    keccak256(h, 32, h2);
    memcpy(h, h2, 32);
    uint64_t v[4];
    for (int w = 0; w < 4; w++) {  // big-endian bytes -> LE limbs
      uint64_t x = 0;
      for (int k = 0; k < 8; k++) x = (x << 8) | h[(3 - w) * 8 + k];
      v[w] = x;
    }
    for (;;) {  // v mod r by repeated subtraction (2^256 / r < 6)
      bool ge = true;
      for (int w = 3; w >= 0; w--) {
        if (v[w] != RMOD[w]) { ge = v[w] > RMOD[w]; break; }
      }
      if (!ge) break;
      unsigned __int128 bw = 0;
      for (int w = 0; w < 4; w++) {
        unsigned __int128 d = (unsigned __int128)v[w] - RMOD[w] - (uint64_t)bw;
        v[w] = (uint64_t)d;
        bw = (d >> 64) & 1;
      }
    }
    for (int w = 0; w < 4; w++)
      for (int k = 0; k < 8; k++) ctx->mimc_consts_canon[i * 32 + w * 8 + k] = (uint8_t)(v[w] >> (8 * k));
  }
  OG_HIP(hipMalloc((void**)&ctx->mimc_consts_d, MIMC7_ROUNDS * 32));
  uint8_t* tmp = nullptr;
  OG_HIP(hipMalloc((void**)&tmp, MIMC7_ROUNDS * 32));
  OG_HIP(hipMemcpyAsync(tmp, ctx->mimc_consts_canon, MIMC7_ROUNDS * 32, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_to_mont_fr, dim3(1), dim3(256), 0, ctx->stream, tmp, ctx->mimc_consts_d, (size_t)MIMC7_ROUNDS);
  OG_HIP(hipGetLastError());
  OG_HIP(hipStreamSynchronize(ctx->stream));
  OG_HIP(hipFree(tmp));
  return OG_OK;
}

int mimc7_hash2(og_ctx* ctx, const uint8_t* l, const uint8_t* r, uint8_t* out, size_t n) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_mimc7_hash2, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)ctx->mimc_consts_d, l, r, out, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int mimc7_merkle_paths(og_ctx* ctx, const uint8_t* leaves, const uint64_t* idx, const uint8_t* sib, int depth,
                       uint8_t* nodes, size_t n) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_mimc7_merkle_paths, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream,
                     (const uint32_t*)ctx->mimc_consts_d, leaves, idx, sib, depth, nodes, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int mimc7_tree_build(og_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t* nodes) {
  OG_HIP(hipMemcpyAsync(nodes, leaves, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
  size_t off = 0;
  for (size_t w = n; w > 1; w >>= 1) {
    size_t n_out = w >> 1;
    // small blocks near the root keep every CU busy a little longer
    unsigned block = n_out >= 65536 ? 256 : 64;
    hipLaunchKernelGGL(k_mimc7_tree_level, dim3(grid_for(n_out, block)), dim3(block), 0, ctx->stream,
                       (const uint32_t*)ctx->mimc_consts_d, nodes + off * 32, nodes + (off + w) * 32, n_out);
    OG_HIP(hipGetLastError());
    off += w;
  }
  return OG_OK;
}

}  // namespace og
