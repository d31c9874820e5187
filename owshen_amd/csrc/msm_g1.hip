// G1 instantiation of the MSM / scalar-mul / proof-assembly templates (msm_impl.hip.h, ecmul_impl.hip.h).
#define OG_ECMUL_G1 1
#include "msm_impl.hip.h"
#include "ecmul_impl.hip.h"
#include "glv.h"

namespace og {

int msm_run_g1(og_ctx* ctx, const og_bases* b, const DigitSort& ds, uint8_t* out, bool partial, int phase) { return msm_run_t<Fq>(ctx, b, ds, out, partial, phase); }
int msm_combine_g1(og_ctx* ctx, const og_bases* b, const uint8_t* gathered, int world, int batch, uint8_t* out) {
  return msm_combine_t<Fq>(ctx, b, gathered, world, batch, out);
}
int msm_sum_ranks_g1(og_ctx* ctx, const uint8_t* gathered, size_t rank_stride, int world, int batch, uint8_t* out) {
  return msm_sum_ranks_t<Fq>(ctx, gathered, rank_stride, world, batch, out);
}
int bases_fill_g1(og_ctx* ctx, og_bases* b, const uint8_t* pts) { return bases_fill_t<Fq>(ctx, b, pts); }
int xyzz_to_affine_bytes_g1(og_ctx* ctx, const uint8_t* in, uint8_t* out, size_t n) { return xyzz_to_affine_bytes_t<Fq>(ctx, in, out, n); }
int fixed_table_g1(og_ctx* ctx, const uint8_t* base_mont_d, uint8_t* tab_d) { return fixed_table_t<Fq>(ctx, base_mont_d, tab_d); }
int scalar_mul_fixed_g1(og_ctx* ctx, const uint8_t* tab_d, const uint8_t* k_d, size_t n, uint8_t* out_d) {
  return scalar_mul_fixed_t<Fq>(ctx, tab_d, k_d, n, out_d);
}
int import_points_g1(og_ctx* ctx, const uint8_t* in_d, uint8_t* out_d, size_t n) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_bases_import<Fq>, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in_d, out_d, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// proofs_d[g][0:64] = A, [192:256] = C   (tmp_d: n x 4 x 17 x 128 B scratch: four products + their four window tables per proof)
// glv_d (optional): n x 4 x 32 B, the GLV halves of r, r s, s, r per proof (glv.h) -- then eight lanes per proof walk
// half-length chains (tmp_d: n x 8 x 17 x 128 B)
int assemble_g1(og_ctx* ctx, const uint8_t* consts_d, const uint8_t* rs_d, const uint8_t* res_a, const uint8_t* res_b1,
                const uint8_t* res_l, const uint8_t* res_h, size_t n, uint8_t* tmp_d, uint8_t* proofs_d, const uint8_t* glv_d) {
  if (n == 0) return OG_OK;
  if (glv_d) {
    GlvBetaWords beta;
    for (int i = 0; i < 4; i++) { beta.w[2 * i] = (uint32_t)glv::BETA[i]; beta.w[2 * i + 1] = (uint32_t)(glv::BETA[i] >> 32); }
    // a handful of requests: a WAVE per half-length chain, the group law in the wave-wide form (one request 5.6 -> 5.4 ms, 8: 7.6 ->
    // 7.5, 16: level -- profiles/r06m_ab_asm_w9.txt; ~19 x the wave-instructions of a lane per chain, which more requests would
    // pay for out of their accumulations); OG_ASM_W9_MAX moves the bound in hooks builds
    if (n <= (size_t)OG_HOOK_INT("OG_ASM_W9_MAX", 8))
      hipLaunchKernelGGL(k_assemble_g1_muls_w9, dim3((unsigned)(n * 8)), dim3(64), 0, ctx->stream, consts_d, glv_d, res_a, res_b1, n, tmp_d, beta);
    else
      hipLaunchKernelGGL(k_assemble_g1_muls_glv, dim3(grid_for(n * 8, 64)), dim3(64), 0, ctx->stream, consts_d, glv_d, res_a, res_b1, n, tmp_d, beta);
  }
  else
    hipLaunchKernelGGL(k_assemble_g1_muls, dim3(grid_for(n * 4, 64)), dim3(64), 0, ctx->stream, consts_d, rs_d, res_a, res_b1, n, tmp_d);
  OG_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_assemble_g1_finish, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, consts_d, res_a, res_l, res_h, tmp_d, n,
                     proofs_d, glv_d ? 2 : 1);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// assemble_g1 in two parts (ecmul_impl.hip.h), each on the ctx's current stream: the products (need the A and B1 results), then A's
// sum (`early`); C's sum (`late`: needs the L and H results and the PRODUCTS -- `products_done`, if given, is recorded between the two)
int assemble_g1_early(og_ctx* ctx, const uint8_t* consts_d, const uint8_t* rs_d, const uint8_t* res_a, const uint8_t* res_b1, size_t n,
                      uint8_t* tmp_d, uint8_t* proofs_d, const uint8_t* glv_d, hipEvent_t products_done) {
  if (n == 0) return OG_OK;
  if (glv_d) {
    GlvBetaWords beta;
    for (int i = 0; i < 4; i++) { beta.w[2 * i] = (uint32_t)glv::BETA[i]; beta.w[2 * i + 1] = (uint32_t)(glv::BETA[i] >> 32); }
    // a handful of requests: a WAVE per half-length chain, the group law in the wave-wide form (one request 5.6 -> 5.4 ms, 8: 7.6 ->
    // 7.5, 16: level -- profiles/r06m_ab_asm_w9.txt; ~19 x the wave-instructions of a lane per chain, which more requests would
    // pay for out of their accumulations); OG_ASM_W9_MAX moves the bound in hooks builds
    if (n <= (size_t)OG_HOOK_INT("OG_ASM_W9_MAX", 8))
      hipLaunchKernelGGL(k_assemble_g1_muls_w9, dim3((unsigned)(n * 8)), dim3(64), 0, ctx->stream, consts_d, glv_d, res_a, res_b1, n, tmp_d, beta);
    else
      hipLaunchKernelGGL(k_assemble_g1_muls_glv, dim3(grid_for(n * 8, 64)), dim3(64), 0, ctx->stream, consts_d, glv_d, res_a, res_b1, n, tmp_d, beta);
  }
  else
    hipLaunchKernelGGL(k_assemble_g1_muls, dim3(grid_for(n * 4, 64)), dim3(64), 0, ctx->stream, consts_d, rs_d, res_a, res_b1, n, tmp_d);
  OG_HIP(hipGetLastError());
  if (products_done) OG_HIP(hipEventRecord(products_done, ctx->stream));
  hipLaunchKernelGGL(k_assemble_g1_early, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, consts_d, res_a, (const uint8_t*)tmp_d, n, proofs_d,
                     glv_d ? 2 : 1);
  OG_HIP(hipGetLastError());
  return OG_OK;
}
int assemble_g1_late(og_ctx* ctx, const uint8_t* res_l, const uint8_t* res_h, size_t n, const uint8_t* tmp_d, uint8_t* proofs_d, bool glv) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_assemble_g1_late, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, res_l, res_h, tmp_d, n, proofs_d, glv ? 2 : 1);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
