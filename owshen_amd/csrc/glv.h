// GLV decomposition of a BN254 scalar for the proof assembly of latency-bound calls (host code).
// BN254 G1 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y): a scalar k < r splits as k = k1 + lambda k2 (mod r)
// with |k1|, |k2| < 2^127, so k P = k1 P + k2 phi(P) is TWO half-length double-and-add chains side by side instead of one of
// 254 bits -- the assembly's four scalar multiplications are a single request's last ~1.6 ms.  Constants: tools/derive_glv.py
// (derived from the two moduli and checked on the generator; tests/test_glv.py re-derives them).  Two checks stand between a
// wrong constant and a wrong proof: every decomposition is VERIFIED here (k1 + lambda k2 == k mod r in the field layer's own
// arithmetic, both halves below 2^127) -- that covers the lattice constants and lambda as an integer -- and the PAIRING of
// lambda with beta, phi(P) = (beta x, y) = [lambda] P, which no decomposition can see, is verified once per process on the
// generator (groth16.hip: glv_pair_ok, the group law of ec.hip.h on the host); BETA below is the only definition of beta, the
// assembly kernel receives it as an argument.  A failure of either check makes the caller take the plain 254-bit path.
#pragma once
#include <stdint.h>
#include <string.h>
#include "field.hip.h"

namespace og {
namespace glv {

typedef unsigned __int128 u128;
// lattice basis (a1, b1), (a2, b2) with a + b lambda = 0 (mod r); b1 is negative: B1N = -b1
static const uint64_t A1 = 0x89d3256894d213e3ull;
static const uint64_t B1N[2] = {0x8211bbeb7d4f1128ull, 0x6f4d8248eeb859fcull};
static const uint64_t A2[2] = {0x0be4e1541221250bull, 0x6f4d8248eeb859fdull};
static const uint64_t B2 = 0x89d3256894d213e3ull;
// g1 = floor(2^320 b2 / r), g2 = floor(2^320 (-b1) / r)
static const uint64_t G1[3] = {0x6eb9c714773a6ef2ull, 0xd91d232ec7e0b3d7ull, 0x2ull};
static const uint64_t G2[4] = {0xa5e38cfb5eaa26d9ull, 0x7a7bd9d4391eb18dull, 0x4ccef014a773d2cfull, 0x2ull};
static const uint64_t LAMBDA[4] = {0x8b17ea66b99c90ddull, 0x5bfc41088d8daaa7ull, 0xb3c4d79d41a91758ull, 0x0ull};
static const uint64_t BETA[4] = {0x5763473177fffffeull, 0xd4f263f1acdb5c4full, 0x59e26bcea0d48bacull, 0x0ull};

// out[0 .. na + nb) = a * b
static inline void mul_wide(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out) {
  for (int i = 0; i < na + nb; i++) out[i] = 0;
  for (int i = 0; i < na; i++) {
    u128 c = 0;
    for (int j = 0; j < nb; j++) {
      c += (u128)a[i] * b[j] + out[i + j];
      out[i + j] = (uint64_t)c;
      c >>= 64;
    }
    out[i + nb] = (uint64_t)c;
  }
}
// x -= a * b (mod 2^256); a: na limbs, b: nb limbs
static inline void submul256(uint64_t x[4], const uint64_t* a, int na, const uint64_t* b, int nb) {
  uint64_t p[8];
  mul_wide(a, na, b, nb, p);
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    const uint64_t pi = i < na + nb ? p[i] : 0;
    const u128 d = (u128)x[i] - pi - br;
    x[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void neg256(uint64_t x[4]) {
  u128 c = 1;
  for (int i = 0; i < 4; i++) {
    c += (uint64_t)~x[i];
    x[i] = (uint64_t)c;
    c >>= 64;
  }
}
static inline Fr fr_from_u64(const uint64_t w[4]) {
  uint32_t v[8];
  for (int i = 0; i < 4; i++) { v[2 * i] = (uint32_t)w[i]; v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
  return fe_to_mont(fe_from_words<FrParams>(v));
}

// k: 32 B little-endian, canonical (< r).  out: |k1| (16 B LE, bit 127 = sign) || |k2| (16 B LE, bit 127 = sign).
// Returns false if k is not canonical or the decomposition does not verify (the caller then uses the plain path).
static inline bool decompose(const uint8_t k_bytes[32], uint8_t out[32]) {
  uint64_t k[4];
  memcpy(k, k_bytes, 32);
  uint32_t kw[8];
  memcpy(kw, k_bytes, 32);
  const Fr kraw = fe_from_words<FrParams>(kw);
  if (!fe_lt_modulus(kraw)) return false;
  uint64_t wide[8], c1[3], c2[3];
  mul_wide(k, 4, G1, 3, wide);  // c1 = (k g1) >> 320
  c1[0] = wide[5]; c1[1] = wide[6]; c1[2] = 0;
  mul_wide(k, 4, G2, 4, wide);  // c2 = (k g2) >> 320
  c2[0] = wide[5]; c2[1] = wide[6]; c2[2] = wide[7];
  // k1 = k - c1 a1 - c2 a2;  k2 = c1 (-b1) - c2 b2     (mod 2^256, two's complement)
  uint64_t k1[4] = {k[0], k[1], k[2], k[3]}, k2[4] = {0, 0, 0, 0};
  submul256(k1, c1, 2, &A1, 1);
  submul256(k1, c2, 3, A2, 2);
  submul256(k2, c1, 2, B1N, 2);  // k2 = -c1 (-b1)
  neg256(k2);                    // k2 =  c1 (-b1)
  submul256(k2, c2, 3, &B2, 1);
  bool neg1 = (k1[3] >> 63) != 0, neg2 = (k2[3] >> 63) != 0;
  if (neg1) neg256(k1);
  if (neg2) neg256(k2);
  if (k1[2] | k1[3] | k2[2] | k2[3] | (k1[1] >> 63) | (k2[1] >> 63)) return false;  // a half is not below 2^127
  // verify in the field layer: k1 + lambda k2 == k (mod r)
  Fr f1 = fr_from_u64(k1), f2 = fr_from_u64(k2);
  if (neg1) f1 = fe_neg(f1);
  if (neg2) f2 = fe_neg(f2);
  const Fr sum = fe_add(f1, fe_mul(fr_from_u64(LAMBDA), f2));
  if (!(fe_from_mont(sum) == fe_canon(kraw))) return false;
  k1[1] |= (uint64_t)neg1 << 63;
  k2[1] |= (uint64_t)neg2 << 63;
  memcpy(out, k1, 16);
  memcpy(out + 16, k2, 16);
  return true;
}

}  // namespace glv
}  // namespace og
