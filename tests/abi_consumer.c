/* A compiled consumer of the C ABI: plain C99, nothing but include/owshen_gpu.h and
 * libowshen_gpu.so -- the calls a Rust host makes through ffi/owshen_gpu.rs, in the same order:
 *   og_init -> og_withdraw_r1cs -> og_setup -> og_pk_load -> og_malloc / og_memcpy_h2d -> og_withdraw_prove_batch_d
 *   -> og_withdraw_witness_d (to read the public inputs back) -> og_verify (CPU) -> teardown.
 * Prints the proofs as hex (tests/test_gpu_abi_consumer.py compares them with the same request made through ctypes) and
 * "verify: ok".  usage: abi_consumer <depth> <n_pad3> <n_pad2> <n_proofs> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "owshen_gpu.h"

#define CHECK(expr)                                                              \
  do {                                                                           \
    int rc_ = (expr);                                                            \
    if (rc_ != OG_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, og_last_error());           \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

/* deterministic byte stream (xorshift64*) so that the Python side can reproduce the request */
static unsigned long long st = 0x9E3779B97F4A7C15ull;
static unsigned char next_byte(void) {
  st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
  return (unsigned char)((st * 2685821657736338717ull) >> 56);
}

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: abi_consumer depth n_pad3 n_pad2 n_proofs\n"); return 2; }
  const int depth = atoi(argv[1]);
  const uint64_t n_pad3 = strtoull(argv[2], 0, 10), n_pad2 = strtoull(argv[3], 0, 10);
  const size_t n = (size_t)atoi(argv[4]);
  og_ctx* ctx = NULL;
  CHECK(og_init(0, &ctx));
  og_r1cs* r1cs = NULL;
  CHECK(og_withdraw_r1cs(ctx, depth, n_pad3, n_pad2, 0, &r1cs));
  uint64_t info[6];
  CHECK(og_r1cs_info(r1cs, info));
  uint8_t toxic[160];
  memset(toxic, 0, sizeof toxic);
  for (int k = 0; k < 5; k++) toxic[32 * k] = (uint8_t)(101 + k);   /* tau, alpha, beta, gamma, delta = 101..105 */
  uint8_t *pk_blob = NULL, *vk_blob = NULL;
  size_t pk_len = 0, vk_len = 0;
  CHECK(og_setup(ctx, r1cs, toxic, &pk_blob, &pk_len, &vk_blob, &vk_len));
  og_r1cs_free(r1cs);
  og_pk* pk = NULL;
  CHECK(og_pk_load(ctx, pk_blob, pk_len, &pk));
  const size_t rec = (size_t)(8 + depth) * 32, m = (size_t)info[0];
  uint8_t* inputs = (uint8_t*)malloc(n * rec);
  uint8_t* rs = (uint8_t*)malloc(n * 64);
  uint8_t* proofs = (uint8_t*)malloc(n * 256);
  uint8_t* wit = (uint8_t*)malloc(n * m * 32);
  uint8_t* pub_all = (uint8_t*)malloc(n * 6 * 32);                                         /* the six public inputs per proof */
  if (!inputs || !rs || !proofs || !wit || !pub_all) return 3;
  for (size_t i = 0; i < n * rec; i++) inputs[i] = next_byte();
  for (size_t g = 0; g < n; g++) {
    for (int f = 0; f < 8 + depth; f++) inputs[g * rec + f * 32 + 31] &= 0x1F;            /* < 2^253 < r */
    memset(inputs + g * rec + 6 * 32 + 20, 0, 12);                                         /* token: a 160-bit address */
    memset(inputs + g * rec + 7 * 32 + 8, 0, 24);                                          /* chain id: u64 */
    memset(inputs + g * rec + 5 * 32 + 8, 0, 24);                                          /* index: u64 */
    if (depth < 64) {
      uint64_t idx;
      memcpy(&idx, inputs + g * rec + 5 * 32, 8);
      idx &= ((uint64_t)1 << depth) - 1;
      memcpy(inputs + g * rec + 5 * 32, &idx, 8);
    }
  }
  for (size_t i = 0; i < n * 64; i++) rs[i] = next_byte();
  for (size_t g = 0; g < n; g++) { rs[g * 64 + 31] &= 0x1F; rs[g * 64 + 63] &= 0x1F; }
  void *inputs_d = NULL, *wit_d = NULL;
  CHECK(og_malloc(ctx, n * rec, &inputs_d));
  CHECK(og_malloc(ctx, n * m * 32, &wit_d));
  CHECK(og_memcpy_h2d(ctx, inputs_d, inputs, n * rec));
  CHECK(og_withdraw_prove_batch_d(ctx, pk, depth, n_pad3, n_pad2, (const uint8_t*)inputs_d, n, rs, proofs, pub_all));
  CHECK(og_withdraw_witness_d(ctx, depth, n_pad3, n_pad2, (const uint8_t*)inputs_d, n, (uint8_t*)wit_d));
  CHECK(og_memcpy_d2h(ctx, wit, wit_d, n * m * 32));
  for (size_t g = 0; g < n; g++) {
    int ok = 0;
    if (memcmp(pub_all + g * 192, wit + g * m * 32 + 32, 192) != 0) {                       /* wires 1..6 = the public inputs */
      fprintf(stderr, "proof %zu: public_out differs from the witness's public wires\n", g);
      return 6;
    }
    CHECK(og_verify(vk_blob, vk_len, pub_all + g * 192, 6, proofs + g * 256, &ok));
    if (!ok) { fprintf(stderr, "proof %zu does not verify\n", g); return 4; }
    printf("proof %zu ", g);
    for (int i = 0; i < 256; i++) printf("%02x", proofs[g * 256 + i]);
    printf("\n");
    int bad = 1;
    uint8_t pub[192];
    for (int slot = 2; slot < 6; slot++) {                                                  /* someone else's recipient, another */
      memcpy(pub, pub_all + g * 192, 192);                                                  /* amount, another token, another chain */
      pub[slot * 32] ^= 1;
      CHECK(og_verify(vk_blob, vk_len, pub, 6, proofs + g * 256, &bad));
      if (bad) { fprintf(stderr, "proof %zu verifies with public input %d changed\n", g, slot); return 5; }
    }
  }
  printf("verify: ok\n");
  /* the boundary refuses malformed records (include/owshen_gpu.h): a field >= r -- here all ones -- is OG_ERR_INVALID, and
   * og_last_error names the record and the field; the well-formed batch still proves afterwards */
  {
    uint8_t* bad_in = (uint8_t*)malloc(n * rec);
    uint8_t* scratch = (uint8_t*)malloc(n * 256);
    if (!bad_in || !scratch) return 3;
    memcpy(bad_in, inputs, n * rec);
    memset(bad_in + (n - 1) * rec + 1 * 32, 0xff, 32);                                      /* the last record's `secret` */
    CHECK(og_memcpy_h2d(ctx, inputs_d, bad_in, n * rec));
    int rc = og_withdraw_prove_batch_d(ctx, pk, depth, n_pad3, n_pad2, (const uint8_t*)inputs_d, n, rs, scratch, NULL);
    if (rc != OG_ERR_INVALID || !strstr(og_last_error(), "secret")) {
      fprintf(stderr, "malformed record: rc %d (%s), expected OG_ERR_INVALID naming the field\n", rc, og_last_error());
      return 7;
    }
    rc = og_withdraw_witness_d(ctx, depth, n_pad3, n_pad2, (const uint8_t*)inputs_d, n, (uint8_t*)wit_d);
    if (rc != OG_ERR_INVALID) { fprintf(stderr, "og_withdraw_witness_d accepted a malformed record\n"); return 7; }
    CHECK(og_memcpy_h2d(ctx, inputs_d, inputs, n * rec));
    CHECK(og_withdraw_prove_batch_d(ctx, pk, depth, n_pad3, n_pad2, (const uint8_t*)inputs_d, n, rs, scratch, NULL));
    if (memcmp(scratch, proofs, n * 256) != 0) { fprintf(stderr, "proofs changed after a refused call\n"); return 7; }
    free(bad_in);
    free(scratch);
    printf("boundary: ok\n");
  }
  CHECK(og_free(ctx, inputs_d));
  CHECK(og_free(ctx, wit_d));
  og_pk_free(pk);
  og_blob_free(pk_blob);
  og_blob_free(vk_blob);
  og_shutdown(ctx);
  free(inputs); free(rs); free(proofs); free(wit); free(pub_all);
  return 0;
}
