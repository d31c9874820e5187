// Batched EdDSA verification on BabyJubJub with the MiMC7 sponge.
//
// Reference: /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs -- curve constants :174-189 (twisted
// Edwards a = 168700, d = 168696 over Fr, base point = the prime-order "Base8"), `verify` :99-115
// (s * BASE == R + h * pk with h = hash(R.x, R.y, pk.x, pk.y, message)), and the PLACEHOLDER hash :202-204 (a product of
// its inputs).  This kernel is that `verify` with the placeholder replaced by MultiMiMC7 (circomlib convention, the hash
// of the withdraw circuit), one signature per lane:
//   h   = MultiMiMC7([R.x, R.y, pk.x, pk.y, msg], key 0)                               5 x 91 rounds
//   acc = s * BASE - h * pk   by one joint double-and-add over 254 bits (Shamir)        254 doublings + <= 254 additions
//   ok  = on_curve(pk) && on_curve(R) && acc == R                                        (projective comparison, no inversion)
// Projective twisted-Edwards coordinates with the unified addition law (add-2008-bbjlp: the reference's projective add
// :118-143 without its affine equality test; the law also doubles).  Everything is Fr arithmetic on the 9 x 29-bit layer.
// Records: pk.x | pk.y | R.x | R.y | s | msg = 6 x 32 B canonical little-endian (`Fp::to_repr()`, mod.rs:10).
#include "ctx.h"
#include "mimc7.hip.h"

namespace og {

struct EdPoint {
  Fr x, y, z;
};

__device__ __forceinline__ Fr ed_const(uint32_t v) { return fe_to_mont(fe_from_u32<FrParams>(v)); }

// unified projective addition on a x^2 + y^2 = 1 + d x^2 y^2
__device__ __forceinline__ EdPoint ed_add(const EdPoint& p, const EdPoint& q, const Fr& ca, const Fr& cd) {
  const Fr A = fe_mul(p.z, q.z);
  const Fr B = fe_sqr(A);
  const Fr C = fe_mul(p.x, q.x);
  const Fr D = fe_mul(p.y, q.y);
  const Fr E = fe_mul(cd, fe_mul(C, D));
  const Fr F = fe_sub(B, E);
  const Fr G = fe_add(B, E);
  const Fr t = fe_sub(fe_sub(fe_mul(fe_add(p.x, p.y), fe_add(q.x, q.y)), C), D);
  EdPoint r;
  r.x = fe_mul(fe_mul(A, F), t);
  r.y = fe_mul(fe_mul(A, G), fe_sub(D, fe_mul(ca, C)));
  r.z = fe_mul(F, G);
  return r;
}

__device__ __forceinline__ bool ed_on_curve(const Fr& x, const Fr& y, const Fr& ca, const Fr& cd) {
  const Fr xx = fe_sqr(x), yy = fe_sqr(y);
  return fe_add(fe_mul(ca, xx), yy) == fe_add(Fr::one(), fe_mul(cd, fe_mul(xx, yy)));
}

__device__ __forceinline__ bool canonical_fr(const uint8_t* p) {  // value < r
  const Fr v = fe_load<FrParams>(p);
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) c = ((int32_t)v.l[i] - (int32_t)FrParams::N[i] + c) >> 29;
  return c != 0;  // borrow out: v < N
}

__global__ void __launch_bounds__(64) k_eddsa_verify(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ recs, size_t n,
                                                    uint32_t* __restrict__ ok) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* rec = recs + g * 192;
  bool good = true;
#pragma unroll 1
  for (int k = 0; k < 6; k++) good = good && canonical_fr(rec + 32 * k);
  const Fr ca = ed_const(168700), cd = ed_const(168696);
  const Fr pkx = fe_to_mont(fe_load<FrParams>(rec)), pky = fe_to_mont(fe_load<FrParams>(rec + 32));
  const Fr rx = fe_to_mont(fe_load<FrParams>(rec + 64)), ry = fe_to_mont(fe_load<FrParams>(rec + 96));
  const Fr msg = fe_to_mont(fe_load<FrParams>(rec + 160));
  good = good && ed_on_curve(pkx, pky, ca, cd) && ed_on_curve(rx, ry, ca, cd);
  // h = MultiMiMC7([R.x, R.y, pk.x, pk.y, msg]): r <- r + x + E_r(x), one inlined permutation body
  Fr h = Fr::zero();
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    const Fr x = k == 0 ? rx : k == 1 ? ry : k == 2 ? pkx : k == 3 ? pky : msg;
    Fr t = x;
#pragma unroll 1
    for (int i = 0; i < MIMC7_ROUNDS; i++) {
      const Fr u = fe_add3_weak(t, h, mimc7_const(consts, i));
      const Fr u2 = fe_sqr(u);
      const Fr u4 = fe_sqr(u2);
      t = fe_mul(fe_mul(u4, u2), u);
    }
    h = fe_add(fe_add(h, x), fe_add(t, h));  // r + x + (t + r)
  }
  const Fr hc = fe_from_mont(h);      // the scalar h as an integer < r
  const Fr sc = fe_load<FrParams>(rec + 128);
  // BASE (mod.rs:177-188), -pk, BASE - pk
  const uint32_t BX[8] = {0xbb957051u, 0x2893f3f6u, 0x0534e0b6u, 0x2ab8d801u, 0x9d6277c1u, 0x4eacb2e0u, 0xd63e739bu, 0x0bb77a6au};
  const uint32_t BY[8] = {0x872d7d8bu, 0x4b3c257au, 0xb9e13377u, 0xfce0051fu, 0xd16bf9edu, 0x25572e1cu, 0xf7a0b249u, 0x25797203u};
  EdPoint base, npk, both;
  base.x = fe_to_mont(fe_from_words<FrParams>(BX));
  base.y = fe_to_mont(fe_from_words<FrParams>(BY));
  base.z = Fr::one();
  npk.x = fe_neg(pkx);
  npk.y = pky;
  npk.z = Fr::one();
  both = ed_add(base, npk, ca, cd);
  EdPoint acc;
  acc.x = Fr::zero();
  acc.y = Fr::one();
  acc.z = Fr::one();
  // joint double-and-add, MSB first: one inlined addition site (op 0: acc += acc, op 1: acc += {BASE, -pk, BASE - pk})
#pragma unroll 1
  for (int step = 2 * 261 - 1; step >= 0; step--) {
    const int bit = step >> 1, op = !(step & 1);   // per bit: first the doubling (odd step), then the addition (even step)
    const int w = bit / 29, sh = bit % 29;
    uint32_t sl = 0, hl = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      sl = k == w ? sc.l[k] : sl;
      hl = k == w ? hc.l[k] : hl;
    }
    const uint32_t sel = ((sl >> sh) & 1u) | (((hl >> sh) & 1u) << 1);
    if (op && sel == 0) continue;
    EdPoint rhs = acc;
    if (op) rhs = sel == 1 ? base : sel == 2 ? npk : both;
    acc = ed_add(acc, rhs, ca, cd);
  }
  // acc == R  <=>  acc.x == R.x acc.z and acc.y == R.y acc.z  (acc.z != 0 on the curve's complete addition domain)
  good = good && !acc.z.is_zero() && acc.x == fe_mul(rx, acc.z) && acc.y == fe_mul(ry, acc.z);
  ok[g] = good ? 1u : 0u;
}

int eddsa_verify(og_ctx* ctx, const uint8_t* recs_d, size_t n, uint32_t* ok_d) {
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_eddsa_verify, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, recs_d, n, ok_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
