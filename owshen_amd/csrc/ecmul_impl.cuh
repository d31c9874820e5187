// Scalar multiplication + Groth16 proof assembly kernels, templated on the group and instantiated in
// msm_g1.hip / msm_g2.hip next to the MSM.  No reference counterpart.
//   A = alpha + sum z_i A_i + r delta
//   B = beta  + sum z_i B_i + s delta                      (G2; the G1 copy only feeds C)
//   C = sum z_i L_i + sum h_j H_j + s A + r B1 - r s delta
//     = L + H + s (alpha + Am) + r (beta1 + B1m) + (r s) delta1        (expanded, 4 independent muls)
#pragma once
#include "ec.cuh"
#include "ctx.h"

namespace og {

struct Scalar256 {
  uint32_t l[8];
};

__device__ __forceinline__ Scalar256 scalar_load(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}

template <class T>
__device__ __noinline__ XYZZ<T> xyzz_madd_nv(XYZZ<T> a, Affine<T> q) {
  return xyzz_madd(a, q);
}

// k * p, k < 2^254 canonical (left-to-right double-and-add; by-value on purpose, see ec.cuh)
template <class T>
__device__ __noinline__ XYZZ<T> xyzz_scalar_mul(XYZZ<T> p, Scalar256 k) {
  XYZZ<T> acc = XYZZ<T>::inf();
#pragma unroll 1
  for (int i = 253; i >= 0; i--) {
    acc = xyzz_dbl_nv(acc);
    if ((k.l[i >> 5] >> (i & 31)) & 1) acc = xyzz_add_nv(acc, p);
  }
  return acc;
}

// out[i] = k_i * base; base: affine Montgomery (device, one point); out canonical affine
template <class T>
__global__ void __launch_bounds__(64) k_scalar_mul_fixed(const uint8_t* __restrict__ base, const uint8_t* __restrict__ scalars,
                                                        size_t n, uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<T> b = Affine<T>::load(base);
  Scalar256 k = scalar_load(scalars + i * 32);
  XYZZ<T> acc = XYZZ<T>::inf();
#pragma unroll 1
  for (int j = 253; j >= 0; j--) {
    acc = xyzz_dbl_nv(acc);
    if ((k.l[j >> 5] >> (j & 31)) & 1) acc = xyzz_madd_nv(acc, b);
  }
  Affine<T> a = xyzz_to_affine(acc);
  a.x = FieldIO<T>::from_mont(a.x);
  a.y = FieldIO<T>::from_mont(a.y);
  a.store(out + i * Affine<T>::BYTES);
}

template <class T>
int scalar_mul_fixed_t(og_ctx* ctx, const uint8_t* base_mont_d, const uint8_t* scalars_d, size_t n, uint8_t* out_d) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_scalar_mul_fixed<T>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, base_mont_d, scalars_d, n, out_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

// ---- proof assembly -----------------------------------------------------------------
// rs: n x 64 B canonical (r || s).  consts: G1 {alpha, beta, delta} / G2 {beta, delta}, affine Montgomery.

#ifdef OG_ECMUL_G1
// G1 step 1: tmp[g][j], j < 4 = { r delta, (r s) delta, s (alpha + Am), r (beta + B1m) }
__global__ void __launch_bounds__(64) k_assemble_g1_muls(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ rs,
                                                        const uint8_t* __restrict__ res_a, const uint8_t* __restrict__ res_b1,
                                                        size_t n, uint8_t* __restrict__ tmp) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 4) return;
  const size_t g = t >> 2;
  const int j = (int)(t & 3);
  Scalar256 r = scalar_load(rs + g * 64), s = scalar_load(rs + g * 64 + 32);
  G1Affine alpha = G1Affine::load(consts), beta = G1Affine::load(consts + 64), delta = G1Affine::load(consts + 128);
  G1XYZZ p;
  Scalar256 k;
  if (j == 0) {
    p = G1XYZZ::from_affine(delta);
    k = r;
  } else if (j == 1) {
    p = G1XYZZ::from_affine(delta);
    Fr rf, sf;
#pragma unroll
    for (int i = 0; i < 8; i++) { rf.l[i] = r.l[i]; sf.l[i] = s.l[i]; }
    Fr prod = fe_mul(fe_to_mont(rf), sf);  // (r R)(s) / R = r s, canonical
#pragma unroll
    for (int i = 0; i < 8; i++) k.l[i] = prod.l[i];
  } else if (j == 2) {
    p = xyzz_madd_nv(G1XYZZ::load(res_a + g * G1XYZZ::BYTES), alpha);
    k = s;
  } else {
    p = xyzz_madd_nv(G1XYZZ::load(res_b1 + g * G1XYZZ::BYTES), beta);
    k = r;
  }
  xyzz_scalar_mul(p, k).store(tmp + t * G1XYZZ::BYTES);
}

// G1 step 2: proof[g][0:64] = A, proof[g][192:256] = C
__global__ void __launch_bounds__(64) k_assemble_g1_finish(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ res_a,
                                                          const uint8_t* __restrict__ res_l, const uint8_t* __restrict__ res_h,
                                                          const uint8_t* __restrict__ tmp, size_t n, uint8_t* __restrict__ proofs) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  G1Affine alpha = G1Affine::load(consts);
  const uint8_t* tg = tmp + g * 4 * G1XYZZ::BYTES;
  G1XYZZ A = xyzz_madd_nv(G1XYZZ::load(res_a + g * G1XYZZ::BYTES), alpha);
  A = xyzz_add_nv(A, G1XYZZ::load(tg));
  G1XYZZ Cc = xyzz_add_nv(G1XYZZ::load(res_l + g * G1XYZZ::BYTES), G1XYZZ::load(res_h + g * G1XYZZ::BYTES));
  Cc = xyzz_add_nv(Cc, G1XYZZ::load(tg + 2 * G1XYZZ::BYTES));
  Cc = xyzz_add_nv(Cc, G1XYZZ::load(tg + 3 * G1XYZZ::BYTES));
  Cc = xyzz_add_nv(Cc, G1XYZZ::load(tg + 1 * G1XYZZ::BYTES));
  G1Affine a = xyzz_to_affine(A), c = xyzz_to_affine(Cc);
  a.x = fe_from_mont(a.x); a.y = fe_from_mont(a.y);
  c.x = fe_from_mont(c.x); c.y = fe_from_mont(c.y);
  a.store(proofs + g * 256);
  c.store(proofs + g * 256 + 192);
}

#endif  // OG_ECMUL_G1

#ifdef OG_ECMUL_G2
// G2: proof[g][64:192] = beta2 + B2m + s delta2
__global__ void __launch_bounds__(64) k_assemble_g2(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ rs,
                                                   const uint8_t* __restrict__ res_b2, size_t n, uint8_t* __restrict__ proofs) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  G2Affine beta = G2Affine::load(consts), delta = G2Affine::load(consts + 128);
  Scalar256 s = scalar_load(rs + g * 64 + 32);
  G2XYZZ B = xyzz_madd_nv(G2XYZZ::load(res_b2 + g * G2XYZZ::BYTES), beta);
  B = xyzz_add_nv(B, xyzz_scalar_mul(G2XYZZ::from_affine(delta), s));
  G2Affine b = xyzz_to_affine(B);
  b.x = FieldIO<Fq2>::from_mont(b.x);
  b.y = FieldIO<Fq2>::from_mont(b.y);
  b.store(proofs + g * 256 + 64);
}

#endif  // OG_ECMUL_G2

}  // namespace og
