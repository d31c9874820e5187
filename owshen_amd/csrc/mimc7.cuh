// MiMC7 device primitives: circomlib convention, 91 rounds,
// x -> (x + k + c_i)^7, output + k; MultiMiMC7 = Miyaguchi-Preneel chaining.
// The reference has no MiMC7 (its only Fr "hash" is the placeholder product at
// /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:202-204).
// All values are Fr in Montgomery form; `consts` is the 91 x 8-limb Montgomery
// table in HBM, read with wave-uniform addresses (scalar loads).
#pragma once
#include "field.cuh"

namespace og {

constexpr int MIMC7_ROUNDS = 91;

__device__ __forceinline__ Fr mimc7_const(const uint32_t* __restrict__ consts, int i) {
  return fe_load<FrParams>(consts + i * 8);
}

// E_k(x)
__device__ __forceinline__ Fr mimc7_permute(const uint32_t* __restrict__ consts, Fr x, const Fr& k) {
  Fr r = x;
  for (int i = 0; i < MIMC7_ROUNDS; i++) {
    Fr t = fe_add3_weak(r, k, mimc7_const(consts, i));  // < 5N, only ever multiplied: no modular reduction needed
    Fr t2 = fe_sqr(t);
    Fr t4 = fe_sqr(t2);
    Fr t6 = fe_mul(t4, t2);
    r = fe_mul(t6, t);
  }
  return fe_add(r, k);
}

// MultiMiMC7([l, r], key = 0)
__device__ __forceinline__ Fr mimc7_hash2(const uint32_t* __restrict__ consts, const Fr& l, const Fr& r) {
  Fr k = Fr::zero();
  k = fe_add(fe_add(k, l), mimc7_permute(consts, l, k));
  k = fe_add(fe_add(k, r), mimc7_permute(consts, r, k));
  return k;
}

}  // namespace og
