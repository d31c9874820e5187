// Host Keccak-256 (Ethereum padding 0x01) used once at og_init to derive the
// circomlib MiMC7 round constants c_i = keccak256^(i+1)("mimc") mod r.
// Public standard; the reference has no MiMC7.
#include "ctx.h"
#include <string.h>

namespace og {

static inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static void keccak_f(uint64_t s[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // rho offsets indexed [x + 5*y]
  static const int RHO[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                              25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};
  for (int rnd = 0; rnd < 24; rnd++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) s[i] ^= d[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(s[x + 5 * y], RHO[x + 5 * y]);
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++)
        s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    s[0] ^= RC[rnd];
  }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t s[25];
  memset(s, 0, sizeof(s));
  uint8_t blk[136];
  size_t off = 0;
  bool done = false;
  while (!done) {
    size_t take = len - off < rate ? len - off : rate;
    memset(blk, 0, rate);
    memcpy(blk, data + off, take);
    off += take;
    if (take < rate) {
      blk[take] ^= 0x01;
      blk[rate - 1] ^= 0x80;
      done = true;
    }
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w = 0;
      for (int k = 7; k >= 0; k--) w = (w << 8) | blk[8 * i + k];
      s[i] ^= w;
    }
    keccak_f(s);
  }
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(s[i] >> (8 * k));
}

}  // namespace og
