// G2 instantiation of the MSM / scalar-mul / proof-assembly templates (msm_impl.hip.h, ecmul_impl.hip.h).
#define OG_ECMUL_G2 1
#include "msm_impl.hip.h"
#include "ecmul_impl.hip.h"

namespace og {

int msm_run_g2(og_ctx* ctx, const og_bases* b, const DigitSort& ds, uint8_t* out, bool partial, int phase) { return msm_run_t<Fq2>(ctx, b, ds, out, partial, phase); }
int msm_combine_g2(og_ctx* ctx, const og_bases* b, const uint8_t* gathered, int world, int batch, uint8_t* out) {
  return msm_combine_t<Fq2>(ctx, b, gathered, world, batch, out);
}
int msm_sum_ranks_g2(og_ctx* ctx, const uint8_t* gathered, size_t rank_stride, int world, int batch, uint8_t* out) {
  return msm_sum_ranks_t<Fq2>(ctx, gathered, rank_stride, world, batch, out);
}
int bases_fill_g2(og_ctx* ctx, og_bases* b, const uint8_t* pts) { return bases_fill_t<Fq2>(ctx, b, pts); }
int xyzz_to_affine_bytes_g2(og_ctx* ctx, const uint8_t* in, uint8_t* out, size_t n) { return xyzz_to_affine_bytes_t<Fq2>(ctx, in, out, n); }
int scalar_mul_fixed_g2(og_ctx* ctx, const uint8_t* tab_d, const uint8_t* k_d, size_t n, uint8_t* out_d) {
  return scalar_mul_fixed_t<Fq2>(ctx, tab_d, k_d, n, out_d);
}
int import_points_g2(og_ctx* ctx, const uint8_t* in_d, uint8_t* out_d, size_t n) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_bases_import<Fq2>, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in_d, out_d, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// tab_d: 64 x 16 x 128 B fixed-base table of the G2 point at base_mont_d (see k_fixed_table)
int fixed_table_g2(og_ctx* ctx, const uint8_t* base_mont_d, uint8_t* tab_d) {
  return fixed_table_t<Fq2>(ctx, base_mont_d, tab_d);
}

// proofs_d[g][64:192] = B
int assemble_g2(og_ctx* ctx, const uint8_t* consts_d, const uint8_t* fb_tab_d, const uint8_t* rs_d, const uint8_t* res_b2, size_t n,
                uint8_t* proofs_d, bool wave_per_proof) {
  if (n == 0) return OG_OK;
  // a wave per proof (8 additions deep) for a call that is waited for (groth16.hip: the fanned-out calls of one sub-batch); a lane
  // per proof (65 deep) for a sub-batch of the pipeline, whose assembly hides under the next sub-batch's work
  if (wave_per_proof)
    hipLaunchKernelGGL(k_assemble_g2_tree, dim3((unsigned)n), dim3(64), 0, ctx->stream, consts_d, fb_tab_d, rs_d, res_b2, n, proofs_d);
  else
    hipLaunchKernelGGL(k_assemble_g2, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, consts_d, fb_tab_d, rs_d, res_b2, n, proofs_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
