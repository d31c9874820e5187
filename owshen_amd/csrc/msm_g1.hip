// G1 instantiation of the MSM templates (see msm_impl.cuh).
#include "msm_impl.cuh"

namespace og {

int msm_run_g1(og_ctx* ctx, const og_bases* b, const DigitSort& ds, uint8_t* out) { return msm_run_t<Fq>(ctx, b, ds, out); }
int bases_fill_g1(og_ctx* ctx, og_bases* b, const uint8_t* pts) { return bases_fill_t<Fq>(ctx, b, pts); }
int xyzz_to_affine_bytes_g1(og_ctx* ctx, const uint8_t* in, uint8_t* out, size_t n) { return xyzz_to_affine_bytes_t<Fq>(ctx, in, out, n); }

}  // namespace og
