// BN254 Fq / Fr Montgomery arithmetic for gfx950.
//
// Fr is the reference's `Fp`
// (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11:
// modulus, generator 7, 32-byte little-endian canonical repr).  Fq is the
// EIP-196 base field; the reference has no Fq.
//
// Layout: 8 x u32 limbs, little-endian, Montgomery form (R = 2^256) in
// registers and in HBM scratch; canonical (non-Montgomery) 32-byte LE at the
// C-ABI seam.  CDNA4 has no 64x64 VALU multiply: everything is built on
// v_mad_u64_u32 (32x32+64 -> 64) which hipcc emits for `(u64)a*b + c`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace og {

struct FqParams {
  static constexpr uint32_t N[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr uint32_t INV = 0xe4866389u;  // -N^-1 mod 2^32
};

struct FrParams {
  static constexpr uint32_t N[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t INV = 0xefffffffu;
};

template <class M>
struct Fe {
  uint32_t l[8];

  __device__ __forceinline__ static Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  __device__ __forceinline__ static Fe one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = M::ONE[i];
    return r;
  }
  __device__ __forceinline__ bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i];
    return o == 0;
  }
  __device__ __forceinline__ bool operator==(const Fe& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
    return o == 0;
  }
  __device__ __forceinline__ bool operator!=(const Fe& b) const { return !(*this == b); }
};

// ---- 256-bit helpers --------------------------------------------------------

// r = a + b, returns carry
__device__ __forceinline__ uint32_t add256(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a[i] + b[i];
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}

// r = a - b, returns borrow (1 if a < b)
__device__ __forceinline__ uint32_t sub256(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (int64_t)a[i] - (int64_t)b[i];
    r[i] = (uint32_t)c;
    c >>= 32;  // arithmetic shift: 0 or -1
  }
  return (uint32_t)(c & 1);
}

template <class M>
__device__ __forceinline__ void cond_sub_mod(uint32_t r[8]) {
  uint32_t t[8];
  uint32_t bw = sub256(t, r, M::N);
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = bw ? r[i] : t[i];
}

template <class M>
__device__ __forceinline__ Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) {
  Fe<M> r;
  add256(r.l, a.l, b.l);  // < 2^255, no carry out (N < 2^254)
  cond_sub_mod<M>(r.l);
  return r;
}

template <class M>
__device__ __forceinline__ Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) {
  Fe<M> r;
  uint32_t bw = sub256(r.l, a.l, b.l);
  uint32_t t[8];
  add256(t, r.l, M::N);
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = bw ? t[i] : r.l[i];
  return r;
}

template <class M>
__device__ __forceinline__ Fe<M> fe_neg(const Fe<M>& a) {
  Fe<M> r;
  sub256(r.l, M::N, a.l);
  bool z = a.is_zero();
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = z ? 0u : r.l[i];
  return r;
}

template <class M>
__device__ __forceinline__ Fe<M> fe_dbl(const Fe<M>& a) {
  Fe<M> r;
#pragma unroll
  for (int i = 7; i > 0; i--) r.l[i] = (a.l[i] << 1) | (a.l[i - 1] >> 31);
  r.l[0] = a.l[0] << 1;
  cond_sub_mod<M>(r.l);
  return r;
}

// Montgomery product a*b*2^-256 mod N, CIOS over 32-bit limbs.
// N < 2^254 so the running sum never needs a 10th word.
template <class M>
__device__ __forceinline__ Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t t8 = t[8] + (uint32_t)c;  // no overflow: total < 2N * 2^32
    const uint32_t m = t[0] * M::INV;
    c = (uint64_t)m * M::N[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c = (uint64_t)m * M::N[j] + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t8;
    t[7] = (uint32_t)c;
    t[8] = (uint32_t)(c >> 32);
  }
  Fe<M> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  cond_sub_mod<M>(r.l);
  return r;
}

template <class M>
__device__ __forceinline__ Fe<M> fe_sqr(const Fe<M>& a) {
  return fe_mul(a, a);
}

template <class M>
__device__ __forceinline__ Fe<M> fe_to_mont(const Fe<M>& a) {
  Fe<M> r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.l[i] = M::R2[i];
  return fe_mul(a, r2);
}

template <class M>
__device__ __forceinline__ Fe<M> fe_from_mont(const Fe<M>& a) {
  Fe<M> o = Fe<M>::zero();
  o.l[0] = 1;
  return fe_mul(a, o);
}

// a^(N-2) by square-and-multiply over the constant exponent (a != 0).  Inlined, with a ROLLED loop
// (one squaring + one multiplication body): device-function calls are avoided throughout the EC code
// (see ec.cuh), and a rolled loop keeps the code small.
template <class M>
__device__ __forceinline__ Fe<M> fe_inv(const Fe<M>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = M::N[i];
  e[0] -= 2;  // N[0] >= 2 for both moduli
  Fe<M> r = Fe<M>::one();
#pragma unroll 1
  for (int w = 7; w >= 0; w--) {
    uint32_t limb = 0;  // e[w] without dynamic indexing of a private array
#pragma unroll
    for (int k = 0; k < 8; k++) limb = (k == w) ? e[k] : limb;
#pragma unroll 1
    for (int b = 31; b >= 0; b--) {
      r = fe_sqr(r);
      if ((limb >> b) & 1) r = fe_mul(r, a);  // wave-uniform branch: the exponent is a constant
    }
  }
  return r;
}

// ---- global memory (32 B per element, 16-byte vector accesses) ---------------

template <class M>
__device__ __forceinline__ Fe<M> fe_load(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fe<M> r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}

template <class M>
__device__ __forceinline__ void fe_store(void* p, const Fe<M>& r) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
  q[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
}

typedef Fe<FqParams> Fq;
typedef Fe<FrParams> Fr;

}  // namespace og
