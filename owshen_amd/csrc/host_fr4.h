// HOST-side Fr arithmetic for the serial chains a host may take over (og_set_host_chains: the Merkle walk of witness.hip, the
// one-leaf append of mimc7.hip).  Host code only; the device layer is field.hip.h.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "field.hip.h"
#include "mimc7.hip.h"

namespace og {

// The host's arithmetic is NOT the 9 x 29-bit layer (53 ns per product on a 2.1 GHz Xeon: that layout exists for v_mad_u64_u32) but
// four 64-bit limbs with R' = 2^256 and unsigned __int128 products (CIOS; every value fully reduced, < r): ~3x faster on a core
// with a 64 x 64 multiplier.  Its constants are DERIVED at first use from FrParams::N -- -N^-1 mod 2^64 by Newton's iteration,
// R' mod N and R'^2 mod N by doubling -- nothing is typed in.  The wires go up in THAT Montgomery form and the GPU takes them out
// of it with the product it already runs on every core wire: stored * 2^5 * 2^-261 = stored / 2^256 (k_wires_from_mont, `mult`).
struct H4 { uint64_t v[4]; };
struct H4Field {
  H4 n, one, r2;   // N, R' mod N, R'^2 mod N
  uint64_t ninv;   // -N^-1 mod 2^64
};
typedef unsigned __int128 u128;
static inline bool h4_geq(const H4& a, const H4& b) {
  for (int i = 3; i >= 0; i--) if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  return true;
}
static inline H4 h4_sub_raw(const H4& a, const H4& b) {
  H4 r; u128 br = 0;
  for (int i = 0; i < 4; i++) { const u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)br; r.v[i] = (uint64_t)d; br = (d >> 64) & 1; }
  return r;
}
static inline H4 h4_add(const H4Field& f, const H4& a, const H4& b) {  // a, b < N -> (a + b) mod N   (N < 2^254: no carry out)
  H4 r; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
  return h4_geq(r, f.n) ? h4_sub_raw(r, f.n) : r;
}
static inline H4 h4_mul(const H4Field& f, const H4& a, const H4& b) {  // a b / R' mod N, operands and result < N
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.v[i] * b.v[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * f.ninv;
    c = (u128)m * f.n.v[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * f.n.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  H4 r = {{t[0], t[1], t[2], t[3]}};
  return (t[4] || h4_geq(r, f.n)) ? h4_sub_raw(r, f.n) : r;
}
static inline const H4Field& h4_field() {
  static const H4Field f = [] {
    H4Field g;
    uint32_t w[8];
    Fr n;
    for (int i = 0; i < 9; i++) n.l[i] = FrParams::N[i];
    fe_to_words(w, n);
    for (int i = 0; i < 4; i++) g.n.v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    uint64_t x = 1;  // Newton: x <- x (2 - n0 x) doubles the correct low bits; n0 is odd
    for (int k = 0; k < 6; k++) x *= 2 - g.n.v[0] * x;
    g.ninv = (uint64_t)0 - x;
    H4 v = {{1, 0, 0, 0}};
    for (int k = 0; k < 512; k++) {  // v = 2^k mod N
      if (k == 256) g.one = v;
      v = h4_add(g, v, v);
    }
    g.r2 = v;
    return g;
  }();
  return f;
}
static inline H4 h4_load(const uint8_t* p) { H4 r; memcpy(r.v, p, 32); return r; }   // 32 B little-endian, a value < N
static inline H4 h4_to_mont(const H4Field& f, const H4& a) { return h4_mul(f, a, f.r2); }

// the 91 round constants in the host's Montgomery form (per ctx: canonical constants -> x R' mod N)
static inline std::vector<H4> h4_round_constants(const uint8_t* canon) {
  const H4Field& f = h4_field();
  std::vector<H4> c(MIMC7_ROUNDS);
  for (int i = 0; i < MIMC7_ROUNDS; i++) c[i] = h4_to_mont(f, h4_load(canon + i * 32));
  return c;
}

// MultiMiMC7([l, r], key 0) on values in the host's Montgomery form (mimc7_hash2 of mimc7.hip.h, the same chaining)
static inline H4 h4_mimc7_hash2(const H4Field& f, const H4* rc, const H4& l, const H4& r) {
  const H4 zero = {{0, 0, 0, 0}};
  H4 k = zero, x = l, k1 = zero;
  for (int p = 0; p < 2; p++) {
    for (int i = 0; i < MIMC7_ROUNDS; i++) {
      const H4 t = h4_add(f, h4_add(f, x, k), rc[i]);
      const H4 t2 = h4_mul(f, t, t), t4 = h4_mul(f, t2, t2);
      x = h4_mul(f, h4_mul(f, t4, t2), t);
    }
    if (p == 0) { k1 = h4_add(f, l, x); k = k1; x = r; }
  }
  return h4_add(f, h4_add(f, h4_add(f, k1, k1), r), x);
}
static inline H4 h4_from_mont(const H4Field& f, const H4& a) { const H4 one = {{1, 0, 0, 0}}; return h4_mul(f, a, one); }

}  // namespace og
