// Groth16 prover for BN254 on gfx950: proving-key residency, batched proving.
//
// No reference counterpart.  The entry points are shaped for the reference's
// withdraw seam: `withdraw_handler` (/root/reference/src/services/api_services/withdraw.rs:27-71)
// would call og_prove between the ECDSA recover (:34) and the sequencer re-sign (:56).
//
// Sub-batches of up to 256 proofs go through a two-stage pipeline on the ctx's two streams (prep: witness, sparse
// products, digit sorts; math: quotient, bucket accumulation / reduction, assembly -- see prove_batch_impl), with scratch
// per sub-batch parity, so the memory-bound stages of sub-batch k + 1 run under the VALU-bound ones of k; per sub-batch, in HBM:
//   k_withdraw_*     (og_withdraw_prove_batch_d only) the sub-batch's witnesses
//   k_spmv x3        a = A z, b = B z, c = C z over the QAP rows              (CSR, one lane per row)
//   h_poly_device    3 iNTT + 3 coset NTT + pointwise + coset iNTT            (ntt.hip)
//   msm_digit_sort   signed 16-bit digits of z, counting-sorted once per DENSITY MAP: the A query, the B query
//                    (its G1 and G2 copies share the sort) and the L query each keep only the wires whose
//                    base is not the point at infinity (bellman's "query density")
//   msm_run x4       bucket accumulate + reduce over the compacted, precomputed window tables
//   msm_digit_sort + msm_run   the H query over the quotient coefficients
//   k_assemble_*     r/s blinding (s delta2 from a fixed-base table), final sums, affine conversion, 256 B proofs
// A call that is a single small sub-batch (one withdraw request) instead splits ONE proof across the lanes: the B
// query on lane 1, everything else on lane 0.
// (r, s) are explicit inputs: proofs are reproducible and bit-comparable with the oracle.
#include "ctx.h"
#include "msm.hip.h"
#include "field.hip.h"
#include "glv.h"
#include "ec.hip.h"
#include <string.h>
#include <algorithm>
#include <iterator>
#include <atomic>
#include <mutex>
#include <thread>

struct og_pk {
  uint64_t m = 0, n_pub = 0, log_d = 0, n_rows = 0;
  size_t d = 0;
  uint64_t nnz[3] = {0, 0, 0};
  uint32_t* ptr[3] = {nullptr, nullptr, nullptr};
  uint32_t* col[3] = {nullptr, nullptr, nullptr};
  uint8_t* val[3] = {nullptr, nullptr, nullptr};  // Fr, Montgomery form
  uint32_t* long_rows[3] = {nullptr, nullptr, nullptr};  // rows with more than SPMV_LONG entries (a workgroup each)
  uint32_t n_long[3] = {0, 0, 0};
  og_bases *a = nullptr, *b1 = nullptr, *b2 = nullptr, *l = nullptr, *h = nullptr;
  // density compaction: query q holds only its non-infinity bases; map[q][k] = wire of compact base k.
  // q: 0 = A, 1 = B (shared by the G1 and G2 copies), 2 = L
  uint32_t* map[3] = {nullptr, nullptr, nullptr};
  size_t n_dense[3] = {0, 0, 0};  // entries of the compact wire list (what the digit sort and the table hold)
  size_t n_real[3] = {0, 0, 0};   // ... of which bases that are not the point at infinity (og_pk_density)
  int sort_src[3] = {0, 1, 2};    // sort_src[q] = p < q: query q's wire list is query p's, and it reuses p's digit sort
  bool merge_lh = false;          // the L and H queries share one bucket set per proof (same window bits): C = sum z L + sum h H is ONE MSM
  bool c_is_ab = false;           // header flag 1 (an imported snarkjs key, zkey.hip): no C matrix, C z := (A z) o (B z) on the domain
  uint8_t* consts1 = nullptr;  // alpha1 | beta1 | delta1, affine Montgomery (3 x 64 B)
  uint8_t* consts2 = nullptr;  // beta2 | delta2, affine Montgomery (2 x 128 B)
  uint8_t* fb_delta2 = nullptr;  // fixed-base table of delta2: 64 windows x 16 digits x 128 B
  int device = 0;
  // host copies for the proof assembly on the host (og_set_host_chains): the constants and delta2's table as they sit on the
  // device, and delta1's fixed-base table (64 x 16 XYZZ points, built on first use)
  uint8_t consts1_h[192] = {}, consts2_h[256] = {};
  std::vector<uint8_t> fb_delta2_h;
  mutable std::vector<uint8_t> fb_delta1_h;
  mutable std::once_flag fb_delta1_once;
};

namespace og {

int scalar_mul_fixed_g1(og_ctx*, const uint8_t*, const uint8_t*, size_t, uint8_t*);
int scalar_mul_fixed_g2(og_ctx*, const uint8_t*, const uint8_t*, size_t, uint8_t*);
int fixed_table_g1(og_ctx*, const uint8_t*, uint8_t*);
int import_points_g1(og_ctx*, const uint8_t*, uint8_t*, size_t);
int import_points_g2(og_ctx*, const uint8_t*, uint8_t*, size_t);
int assemble_g1(og_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, size_t,
                uint8_t*, uint8_t*, const uint8_t*);
int assemble_g2(og_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, bool wave_per_proof = false);
int assemble_g1_early(og_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint8_t*, const uint8_t*, hipEvent_t);
int assemble_g1_late(og_ctx*, const uint8_t*, const uint8_t*, size_t, const uint8_t*, uint8_t*, bool);
int fixed_table_g2(og_ctx*, const uint8_t*, uint8_t*);
int ntt_domain_consts(og_ctx* ctx, int log_n, uint8_t** consts_d);
int withdraw_witness(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d, size_t n, uint8_t* out_d);
int withdraw_shape_query(int depth, uint64_t n_pad3, uint64_t n_pad2, uint64_t out[3]);
int withdraw_check_records(og_ctx* ctx, int depth, const uint8_t* inputs_d, size_t n, uint32_t* bad_d);
const char* withdraw_field_name(uint32_t f);

// ---- sparse matrix x witness ---------------------------------------------------------
// out[g][row] = sum_k val[k] * x[g][col[k]] for row < n_rows, 0 for n_rows <= row < n_out.
// val_mont: values are in Montgomery form (product of a Montgomery value and a canonical x is
// canonical); out_mont: convert the row sum to Montgomery form before storing.
__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ col,
                                             const uint8_t* __restrict__ val, size_t n_rows, size_t n_out,
                                             const uint8_t* __restrict__ x, size_t x_stride, uint8_t* __restrict__ out,
                                             size_t out_stride, int val_mont, int out_mont, uint32_t long_row) {
  OG_FILLER_PRIO();
  size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_out) return;
  const int g = blockIdx.y;
  const uint8_t* xg = x + (size_t)g * x_stride;
  Fr acc = Fr::zero();
  if (row < n_rows) {
    if (ptr[row + 1] - ptr[row] > long_row) return;  // a workgroup takes this row (k_spmv_long)
    for (uint32_t k = ptr[row]; k < ptr[row + 1]; k++) {
      Fr v = fe_load<FrParams>(val + (size_t)k * 32);
      if (!val_mont) v = fe_to_mont(v);
      acc = fe_add(acc, fe_mul(v, fe_load<FrParams>(xg + (size_t)col[k] * 32)));
    }
    acc = out_mont ? fe_to_mont(acc) : fe_canon(acc);
  }
  fe_store(out + (size_t)g * out_stride + row * 32, acc);
}

// rows with more than SPMV_LONG entries (e.g. the density rows: one entry per wire): one workgroup per (row, proof),
// strided partial sums, LDS tree
constexpr uint32_t SPMV_LONG = 2048;
__global__ void __launch_bounds__(256) k_spmv_long(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ ptr,
                                                  const uint32_t* __restrict__ col, const uint8_t* __restrict__ val,
                                                  const uint8_t* __restrict__ x, size_t x_stride, uint8_t* __restrict__ out,
                                                  size_t out_stride, int val_mont, int out_mont) {
  OG_FILLER_PRIO();
  __shared__ __align__(16) uint32_t part[256 * 8];
  const uint32_t row = rows[blockIdx.x];
  const int g = blockIdx.y;
  const uint8_t* xg = x + (size_t)g * x_stride;
  Fr acc = Fr::zero();
  for (uint32_t k = ptr[row] + threadIdx.x; k < ptr[row + 1]; k += 256) {
    Fr v = fe_load<FrParams>(val + (size_t)k * 32);
    if (!val_mont) v = fe_to_mont(v);
    acc = fe_add(acc, fe_mul(v, fe_load<FrParams>(xg + (size_t)col[k] * 32)));
  }
  fe_store(&part[threadIdx.x * 8], acc);
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d)
      fe_store(&part[threadIdx.x * 8], fe_add(fe_load<FrParams>(&part[threadIdx.x * 8]), fe_load<FrParams>(&part[(threadIdx.x + d) * 8])));
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    acc = fe_load<FrParams>(&part[0]);
    fe_store(out + (size_t)g * out_stride + (size_t)row * 32, out_mont ? fe_to_mont(acc) : fe_canon(acc));
  }
}

__global__ void __launch_bounds__(256) k_fr_to_mont(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store(out + i * 32, fe_to_mont(fe_load<FrParams>(in + i * 32)));
}

// c = a o b, row by row (Montgomery form): the C z of a key without a C matrix (og_pk::c_is_ab)
__global__ void __launch_bounds__(256) k_mul_rows(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ c, size_t d) {
  OG_FILLER_PRIO();
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= d) return;
  const size_t o = ((size_t)blockIdx.y * d + row) * 32;
  fe_store(c + o, fe_mul(fe_load<FrParams>(a + o), fe_load<FrParams>(b + o)));
}

// Exact satisfiability check on the QAP row products (Montgomery form): flags[g] |= 1 if some row has a_i b_i != c_i,
// or if wire 0 of the witness is not the constant 1 (the input-consistency rows cannot see that).
__global__ void __launch_bounds__(256) k_check_rows(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, const uint8_t* __restrict__ c,
                                                   size_t n_rows, size_t d, const uint8_t* __restrict__ z, size_t z_stride,
                                                   uint32_t* __restrict__ flags) {
  OG_FILLER_PRIO();
  size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (row >= n_rows) return;
  const size_t o = ((size_t)g * d + row) * 32;
  bool bad = !fe_sub(fe_mul(fe_load<FrParams>(a + o), fe_load<FrParams>(b + o)), fe_load<FrParams>(c + o)).is_zero();
  if (row == 0) {
    Fr z0 = fe_load<FrParams>(z + (size_t)g * z_stride);
    z0.l[0] ^= 1u;  // canonical 1 -> all-zero limbs
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) nz |= z0.l[i];
    bad = bad || nz != 0;
  }
  if (bad) atomicOr(&flags[g], 1u);
}

// Boundary check of caller-supplied witnesses: every wire must be the canonical encoding of an Fr element (< r).  A value
// >= r would be reduced silently by the Montgomery entry -- the proof would be one for a DIFFERENT byte string than the caller
// holds.  bad[g] = lowest offending wire of witness g (atomicMin; initialised to 0xffffffff by the caller).
__global__ void __launch_bounds__(256) k_check_canonical(const uint8_t* __restrict__ x, size_t stride, size_t count, uint32_t* __restrict__ bad) {
  OG_FILLER_PRIO();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int g = blockIdx.y;
  if (!fe_lt_modulus(fe_load<FrParams>(x + (size_t)g * stride + i * 32))) atomicMin(&bad[g], (uint32_t)i);
}

// Lagrange basis of the size-2^log_d domain at tau: out[k] = Z(tau)/d * w^k / (tau - w^k), canonical
__global__ void __launch_bounds__(256) k_lagrange(const uint8_t* __restrict__ consts, const uint8_t* __restrict__ tau_c, int log_d,
                                                 uint8_t* __restrict__ out) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t d = (size_t)1 << log_d;
  if (k >= d) return;
  Fr tau = fe_to_mont(fe_load<FrParams>(tau_c));
  Fr zt = tau;
  for (int i = 0; i < log_d; i++) zt = fe_sqr(zt);
  zt = fe_sub(zt, Fr::one());
  Fr b = fe_load<FrParams>(consts + 0 * 32), wk = Fr::one();
  for (size_t e = k; e; e >>= 1) {
    if (e & 1) wk = fe_mul(wk, b);
    b = fe_sqr(b);
  }
  Fr num = fe_mul(fe_mul(zt, fe_load<FrParams>(consts + 4 * 32)), wk);
  Fr den = fe_sub(tau, wk);
  fe_store(out + k * 32, fe_from_mont(fe_mul(num, fe_inv(den))));
}

int spmv_canonical(og_ctx* ctx, const uint32_t* ptr_d, const uint32_t* col_d, const uint8_t* val_d, size_t n_rows,
                   const uint8_t* x_d, uint8_t* out_d) {
  if (n_rows == 0) return OG_OK;
  hipLaunchKernelGGL(k_spmv, dim3(grid_for(n_rows, 256), 1), dim3(256), 0, ctx->stream, ptr_d, col_d, val_d, n_rows, n_rows, x_d,
                     (size_t)0, out_d, (size_t)0, 0, 0, 0xffffffffu);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int lagrange_evals(og_ctx* ctx, int log_d, const uint8_t tau[32], uint8_t* out_d) {
  uint8_t* consts = nullptr;
  OG_TRY(ntt_domain_consts(ctx, log_d, &consts));
  uint8_t* tau_d = nullptr;
  OG_TRY(arena_get(ctx, "g16.tau", 32, (void**)&tau_d));
  OG_HIP(hipMemcpyAsync(tau_d, tau, 32, hipMemcpyHostToDevice, ctx->stream));
  const size_t d = (size_t)1 << log_d;
  hipLaunchKernelGGL(k_lagrange, dim3(grid_for(d, 256)), dim3(256), 0, ctx->stream, consts, tau_d, log_d, out_d);
  OG_HIP(hipGetLastError());
  OG_HIP(hipStreamSynchronize(ctx->stream));  // tau_d / the caller's tau are released after this
  return OG_OK;
}

int scalar_mul_fixed(og_ctx* ctx, int is_g2, const uint8_t* base_host, const uint8_t* scalars_d, size_t n, uint8_t* out_d) {
  const size_t pb = is_g2 ? 128 : 64;
  const int g = is_g2 ? 1 : 0;
  uint8_t *raw = nullptr, *mont = nullptr;
  OG_TRY(arena_get(ctx, "g16.base.raw", 128, (void**)&raw));
  OG_TRY(arena_get(ctx, "g16.base.mont", 128, (void**)&mont));
  if (!ctx->fb_tab[g]) {  // fixed-base table (ecmul_impl.hip.h), kept for the life of the context
    OG_HIP(hipMalloc((void**)&ctx->fb_tab[g], 64 * 16 * pb));
    ctx->owned.push_back(ctx->fb_tab[g]);
  }
  uint8_t* tab = ctx->fb_tab[g];
  if (!ctx->fb_valid[g] || memcmp(ctx->fb_base[g], base_host, pb) != 0) {
    ctx->fb_valid[g] = false;
    OG_HIP(hipMemcpyAsync(raw, base_host, pb, hipMemcpyHostToDevice, ctx->stream));
    OG_TRY(is_g2 ? import_points_g2(ctx, raw, mont, 1) : import_points_g1(ctx, raw, mont, 1));
    OG_TRY(is_g2 ? fixed_table_g2(ctx, mont, tab) : fixed_table_g1(ctx, mont, tab));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(ctx->fb_base[g], base_host, pb);
    ctx->fb_valid[g] = true;
  }
  OG_TRY(is_g2 ? scalar_mul_fixed_g2(ctx, tab, scalars_d, n, out_d) : scalar_mul_fixed_g1(ctx, tab, scalars_d, n, out_d));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  return OG_OK;
}

// ---- proving key -----------------------------------------------------------------------
// Serialized key ("OWPK0001"), all little-endian, every section padded to a multiple of 32 B:
//   u64 x 10 : magic, n_wires, n_pub, log_d, n_rows, nnz_a, nnz_b, nnz_c, flags, 0
//              flags bit 0: the key carries no C matrix (nnz_c = 0) and the prover takes C z = (A z) o (B z) row by row -- what
//              snarkjs' prover does, whose .zkey stores A and B only (og_zkey_import); such a key cannot tell a witness that
//              violates a constraint (the proof simply does not verify), only wire 0 != 1
//   alpha_g1 (64) beta_g1 (64) delta_g1 (64) pad (64) | beta_g2 (128) delta_g2 (128)
//   for M in A, B, C: ptr (n_rows+1 u32) | col (nnz u32) | val (nnz x 32 B canonical)
//   a_query (m x 64) | b_g1_query (m x 64) | b_g2_query (m x 128) | l_query ((m-n_pub-1) x 64) | h_query ((d-1) x 64)
static const uint64_t PK_MAGIC = 0x313030304b50574full;  // "OWPK0001"

static inline size_t pad32(size_t n) { return (n + 31) / 32 * 32; }

void pk_destroy(og_pk* pk) {
  if (!pk) return;
  (void)hipSetDevice(pk->device);
  (void)hipDeviceSynchronize();  // a submitted call (og_withdraw_prove_batch_submit_d) may still be reading the key on any stream
  for (int k = 0; k < 3; k++) {
    if (pk->ptr[k]) (void)hipFree(pk->ptr[k]);
    if (pk->col[k]) (void)hipFree(pk->col[k]);
    if (pk->val[k]) (void)hipFree(pk->val[k]);
    if (pk->long_rows[k]) (void)hipFree(pk->long_rows[k]);
  }
  bases_destroy(pk->a); bases_destroy(pk->b1); bases_destroy(pk->b2); bases_destroy(pk->l); bases_destroy(pk->h);
  for (int k = 0; k < 3; k++)
    if (pk->map[k]) (void)hipFree(pk->map[k]);
  if (pk->consts1) (void)hipFree(pk->consts1);
  if (pk->consts2) (void)hipFree(pk->consts2);
  if (pk->fb_delta2) (void)hipFree(pk->fb_delta2);
  delete pk;
}

static int pk_load_impl(og_ctx* ctx, const uint8_t* blob, size_t len, og_pk* pk) {
  OG_REQUIRE(len >= 80 + 256 + 256, "og_pk_load: blob too short");
  uint64_t hd[10];
  memcpy(hd, blob, 80);
  OG_REQUIRE(hd[0] == PK_MAGIC, "og_pk_load: bad magic (want OWPK0001)");
  pk->m = hd[1]; pk->n_pub = hd[2]; pk->log_d = hd[3]; pk->n_rows = hd[4];
  pk->nnz[0] = hd[5]; pk->nnz[1] = hd[6]; pk->nnz[2] = hd[7];
  OG_REQUIRE(hd[8] <= 1 && hd[9] == 0, "og_pk_load: unknown header flags");
  pk->c_is_ab = (hd[8] & 1) != 0;
  OG_REQUIRE(!pk->c_is_ab || pk->nnz[2] == 0, "og_pk_load: a key with the C = A o B flag carries no C matrix");
  OG_REQUIRE(pk->log_d >= 1 && pk->log_d <= 28, "og_pk_load: log_d must be 1..28");
  pk->d = (size_t)1 << pk->log_d;
  OG_REQUIRE(pk->m >= 1 && pk->m < (1ull << 31) && pk->n_pub < pk->m, "og_pk_load: bad wire counts");
  OG_REQUIRE(pk->n_rows <= pk->d, "og_pk_load: more QAP rows than the domain holds");
  for (int k = 0; k < 3; k++) OG_REQUIRE(pk->nnz[k] < (1ull << 32), "og_pk_load: nnz too large");
  const size_t m = pk->m, nl = m - pk->n_pub - 1, nh = pk->d - 1;
  size_t off = 80;
  const uint8_t* c1 = blob + off; off += 256;
  const uint8_t* c2 = blob + off; off += 256;
  const uint8_t *ptr_h[3], *col_h[3], *val_h[3];
  for (int k = 0; k < 3; k++) {
    ptr_h[k] = blob + off; off += pad32((pk->n_rows + 1) * 4);
    col_h[k] = blob + off; off += pad32(pk->nnz[k] * 4);
    val_h[k] = blob + off; off += pad32(pk->nnz[k] * 32);
    OG_REQUIRE(off <= len, "og_pk_load: truncated R1CS section");
  }
  const uint8_t* q_h[5];
  const size_t q_n[5] = {m, m, m, nl, nh};
  const size_t q_pb[5] = {64, 64, 128, 64, 64};
  for (int k = 0; k < 5; k++) {
    q_h[k] = blob + off;
    off += pad32(q_n[k] * q_pb[k]);
  }
  OG_REQUIRE(off == len, "og_pk_load: blob length does not match its header");
  // validate the CSR on the host (cheap, and a malformed key must never index out of bounds on the GPU)
  for (int k = 0; k < 3; k++) {
    const uint32_t* p = (const uint32_t*)ptr_h[k];
    const uint32_t* c = (const uint32_t*)col_h[k];
    OG_REQUIRE(p[0] == 0 && p[pk->n_rows] == pk->nnz[k], "og_pk_load: CSR row pointers inconsistent");
    for (size_t r = 0; r < pk->n_rows; r++) OG_REQUIRE(p[r] <= p[r + 1], "og_pk_load: CSR row pointers not monotone");
    for (size_t i = 0; i < pk->nnz[k]; i++) OG_REQUIRE(c[i] < m, "og_pk_load: CSR column out of range");
  }
  for (int k = 0; k < 3; k++) {
    OG_HIP(hipMalloc((void**)&pk->ptr[k], (pk->n_rows + 1) * 4));
    OG_HIP(hipMalloc((void**)&pk->col[k], pk->nnz[k] * 4 + 4));
    OG_HIP(hipMalloc((void**)&pk->val[k], pk->nnz[k] * 32 + 32));
    OG_HIP(hipMemcpyAsync(pk->ptr[k], ptr_h[k], (pk->n_rows + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    OG_HIP(hipMemcpyAsync(pk->col[k], col_h[k], pk->nnz[k] * 4, hipMemcpyHostToDevice, ctx->stream));
    OG_HIP(hipMemcpyAsync(pk->val[k], val_h[k], pk->nnz[k] * 32, hipMemcpyHostToDevice, ctx->stream));
    {
      std::vector<uint32_t> lr;
      const uint32_t* p = (const uint32_t*)ptr_h[k];
      for (size_t r = 0; r < pk->n_rows; r++)
        if (p[r + 1] - p[r] > SPMV_LONG) lr.push_back((uint32_t)r);
      pk->n_long[k] = (uint32_t)lr.size();
      if (!lr.empty()) {
        OG_HIP(hipMalloc((void**)&pk->long_rows[k], lr.size() * 4));
        OG_HIP(hipMemcpy(pk->long_rows[k], lr.data(), lr.size() * 4, hipMemcpyHostToDevice));
      }
    }
    if (pk->nnz[k]) {
      hipLaunchKernelGGL(k_fr_to_mont, dim3(grid_for(pk->nnz[k], 256)), dim3(256), 0, ctx->stream, pk->val[k], pk->val[k],
                         (size_t)pk->nnz[k]);
      OG_HIP(hipGetLastError());
    }
  }
  // constants -> affine Montgomery
  uint8_t* stage = nullptr;
  const size_t stage_bytes = std::max<size_t>(std::max<size_t>(512, m * 128), nh * 64);
  OG_HIP(hipMalloc((void**)&stage, stage_bytes));
  struct Guard { uint8_t* p; ~Guard() { if (p) (void)hipFree(p); } } guard{stage};
  OG_HIP(hipMalloc((void**)&pk->consts1, 256));
  OG_HIP(hipMalloc((void**)&pk->consts2, 256));
  OG_HIP(hipMemcpyAsync(stage, c1, 256, hipMemcpyHostToDevice, ctx->stream));
  OG_TRY(import_points_g1(ctx, stage, pk->consts1, 3));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  OG_HIP(hipMemcpyAsync(stage, c2, 256, hipMemcpyHostToDevice, ctx->stream));
  OG_TRY(import_points_g2(ctx, stage, pk->consts2, 2));
  OG_HIP(hipMalloc((void**)&pk->fb_delta2, 64 * 16 * 128));
  OG_TRY(fixed_table_g2(ctx, pk->consts2 + 128, pk->fb_delta2));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  pk->fb_delta2_h.resize(64 * 16 * 128);  // (140 KB: what a host-side assembly reads, og_set_host_chains)
  OG_HIP(hipMemcpy(pk->consts1_h, pk->consts1, 192, hipMemcpyDeviceToHost));
  OG_HIP(hipMemcpy(pk->consts2_h, pk->consts2, 256, hipMemcpyDeviceToHost));
  OG_HIP(hipMemcpy(pk->fb_delta2_h.data(), pk->fb_delta2, pk->fb_delta2_h.size(), hipMemcpyDeviceToHost));
  // queries -> compacted, precomputed window tables.  A wire whose base is the point at infinity (its
  // polynomial is zero at tau: the wire never occurs in that matrix) contributes nothing; drop it from the
  // table and from the digit sort.  B1 / B2 are the same polynomial in two groups, so they share one map.
  int ch = (int)msm_pick_query_c(nh);
  auto is_inf = [](const uint8_t* p, size_t nb) {
    for (size_t i = 0; i < nb; i++)
      if (p[i]) return false;
    return true;
  };
  std::vector<uint32_t> wire[3];
  for (size_t i = 0; i < m; i++) {
    if (!is_inf(q_h[0] + i * 64, 64)) wire[0].push_back((uint32_t)i);
    if (!is_inf(q_h[1] + i * 64, 64) || !is_inf(q_h[2] + i * 128, 128)) wire[1].push_back((uint32_t)i);
  }
  for (size_t i = 0; i < nl; i++)
    if (!is_inf(q_h[3] + i * 64, 64)) wire[2].push_back((uint32_t)(i + pk->n_pub + 1));
  // Queries whose wire lists coincide up to a few wires share ONE list -- the union; a wire a query does not have keeps its
  // zero bytes in that query's table, the point at infinity, which the accumulation skips -- and with it ONE digit sort per
  // sub-batch in the stage pipeline instead of one each.  The benchmark's dense padding: A and B hold every wire, L all but
  // the 7 public ones -> one sort instead of three (the three read and decompose the same witness).  The padding as built
  // (131 k | 117 k | 262 k wires) shares nothing; the natural statement shares A with L.
  for (int k = 0; k < 3; k++) pk->n_real[k] = wire[k].size();
  for (int q = 1; q < 3; q++)
    for (int p0 = 0; p0 < q; p0++) {
      if (pk->sort_src[p0] != p0) continue;
      std::vector<uint32_t> u;
      std::set_union(wire[p0].begin(), wire[p0].end(), wire[q].begin(), wire[q].end(), std::back_inserter(u));
      const size_t least = std::min(wire[p0].size(), wire[q].size());
      if (u.size() - least > 64 || msm_pick_query_c(u.size()) != msm_pick_query_c(least)) continue;
      for (int k = 0; k < q; k++)
        if (pk->sort_src[k] == p0) wire[k] = u;  // (everyone already sharing p0's list moves to the union)
      wire[q] = u;
      pk->sort_src[q] = p0;
      break;
    }
  for (int k = 0; k < 3; k++) {
    pk->n_dense[k] = wire[k].size();
    OG_HIP(hipMalloc((void**)&pk->map[k], wire[k].size() * 4 + 4));
    OG_HIP(hipMemcpyAsync(pk->map[k], wire[k].data(), wire[k].size() * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  OG_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<uint8_t> host(std::max<size_t>(std::max<size_t>(1, m * 128), nh * 64));
  struct QSpec { int src, mapk, is_g2; og_bases** dst; };
  const QSpec qs[4] = {{0, 0, 0, &pk->a}, {1, 1, 0, &pk->b1}, {2, 1, 1, &pk->b2}, {3, 2, 0, &pk->l}};
  for (const QSpec& q : qs) {
    const size_t pb = q.is_g2 ? 128 : 64;
    const std::vector<uint32_t>& w = wire[q.mapk];
    const size_t shift = q.src == 3 ? pk->n_pub + 1 : 0;  // l_query is indexed from wire n_pub + 1
    const size_t q_len = q.src == 3 ? nl : m;
    for (size_t k = 0; k < w.size(); k++) {
      if (w[k] >= shift && w[k] - shift < q_len) memcpy(host.data() + k * pb, q_h[q.src] + (w[k] - shift) * pb, pb);
      else memset(host.data() + k * pb, 0, pb);  // a wire of the shared list this query has no base for: infinity
    }
    OG_HIP(hipMemcpyAsync(stage, host.data(), w.size() * pb, hipMemcpyHostToDevice, ctx->stream));
    OG_TRY(bases_create(ctx, q.is_g2, stage, w.size(), (int)msm_pick_query_c(w.size()), 1, q.dst));  // synchronises the stream
  }
  // the quotient leaves h_poly_device in bit-reversed order (ntt.hip): store the H query in that order.  Position
  // d - 1 is its own reversal, so the d - 1 bases stay the first d - 1 positions.
  for (size_t k = 0; k < nh; k++) {
    size_t r = 0;
    for (uint64_t bit = 0; bit < pk->log_d; bit++) r |= ((k >> bit) & 1) << (pk->log_d - 1 - bit);
    memcpy(host.data() + k * 64, q_h[4] + r * 64, 64);
  }
  OG_HIP(hipMemcpyAsync(stage, host.data(), nh * 64, hipMemcpyHostToDevice, ctx->stream));
  // C = sum z_i L_i + sum h_j H_j is ONE multi-scalar multiplication: when the H query can take the L query's window size, the
  // two share a bucket set per proof (msm_run_phase) -- one bucket reduction, one heavy-bucket tail and one window combine per
  // proof instead of two, and at 17 bits the H query's 131 071 points cost 15 additions each instead of 16.  (17 bits needs
  // n x 15 < 2^23 entries for the two-level radix sort, msm.hip.)  Hooks builds: OG_MERGE_LH=0 keeps the queries apart (A/B).
  const int cl = pk->l->c;
  const bool h_takes_cl = cl == ch || (cl == 17 && (double)nh * 15 < (double)(1u << 23) && nh >= 512) || (cl == 16 && nh >= 512);
  if (OG_HOOK_INT("OG_MERGE_LH", 1) && h_takes_cl && (cl == 16 || cl == 17 || cl == ch)) {
    ch = cl;
    pk->merge_lh = true;
  }
  OG_TRY(bases_create(ctx, 0, stage, nh, ch, 1, &pk->h));
  return OG_OK;
}

// out[0..3] = window bits of the A, B (G1 and G2 copies), L and H queries' precomputed tables (msm_pick_query_c)
void pk_windows(const og_pk* pk, uint64_t out[4]) {
  out[0] = pk->a->c;
  out[1] = pk->b1->c;
  out[2] = pk->l->c;
  out[3] = pk->h->c;
}

// out[0..2] = bases kept by the A, B (G1 and G2 copies) and L queries after density compaction; out[3] = the H query's d - 1
void pk_density(const og_pk* pk, uint64_t out[4]) {
  for (int k = 0; k < 3; k++) out[k] = pk->n_real[k];
  out[3] = pk->d - 1;
}

// HBM bytes of the resident key: CSR matrices, the five queries' per-window tables, wire maps, constants
uint64_t pk_bytes(const og_pk* pk) {
  uint64_t b = 0;
  for (int k = 0; k < 3; k++) b += (pk->n_rows + 1) * 4 + pk->nnz[k] * 36 + 36 + (uint64_t)pk->n_long[k] * 4 + pk->n_dense[k] * 4 + 4;
  for (const og_bases* q : {pk->a, pk->b1, pk->b2, pk->l, pk->h})
    if (q) b += (uint64_t)(q->n ? q->n : 1) * (q->precomp ? q->nwin : 1) * (q->is_g2 ? 128 : 64);
  return b + 256 + 256 + 64 * 16 * 128;
}

int pk_load(og_ctx* ctx, const uint8_t* blob, size_t len, og_pk** out) {
  og_pk* pk = new og_pk();
  pk->device = ctx->device;
  int r = pk_load_impl(ctx, blob, len, pk);
  if (r != OG_OK) {
    (void)hipStreamSynchronize(ctx->stream);
    pk_destroy(pk);
    return r;
  }
  *out = pk;
  return OG_OK;
}

// ---- proving ---------------------------------------------------------------------------
// glv.h's (lambda, beta) must be a PAIR: phi(x, y) = (beta x, y) has to be multiplication by lambda -- the other primitive cube
// root of unity of Fq goes with lambda^2, and a mismatched pair would assemble valid-looking but WRONG A / C for every call of
// <= 64 proofs while every scalar decomposition still verifies.  Checked once per process with the group law of ec.hip.h on the
// host: [lambda] G == (beta x_G, y_G) for G = (1, 2).  On failure the GLV path is off (the plain 254-bit chains are used).
bool glv_pair_ok() {
  static const bool ok = [] {
    const Fq one = Fq::one();
    const Fq two = fe_add(one, one);
    const Affine<Fq> g{one, two};
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (int bit = 255; bit >= 0; bit--) {
      acc = xyzz_dbl(acc);
      if ((glv::LAMBDA[bit >> 6] >> (bit & 63)) & 1) acc = xyzz_madd(acc, g);
    }
    const Affine<Fq> got = xyzz_to_affine(acc);
    uint32_t bw[8];
    for (int i = 0; i < 4; i++) { bw[2 * i] = (uint32_t)glv::BETA[i]; bw[2 * i + 1] = (uint32_t)(glv::BETA[i] >> 32); }
    const Fq want_x = fe_mul(one, fe_to_mont(fe_from_words<FqParams>(bw)));  // beta * x_G, x_G = 1
    return fe_canon(got.x) == fe_canon(want_x) && fe_canon(got.y) == fe_canon(two);
  }();
  return ok;
}

// Scratch slots of the stage pipeline.  Round 3 went from two to three because the preparation of sub-batch k waited for the
// ASSEMBLY of the slot's previous user -- the very end of the tail stream's chain, and the tail kernels are the ones that find
// room last.  But what the preparation overwrites (witnesses, A z / B z / C z, the sorted digit entries) is last read by the
// previous user's heavy-bucket kernels; the bucket sets and reduction levels the late tail kernels work on are not touched
// before the MATH stage.  So since round 6 a slot is released in two steps -- event [11] behind the last heavy-bucket kernels
// gates the preparation, event [6] (assembly done) gates the first accumulation -- and TWO slots do what three did:
// same box, interleaved (profiles/r06b_ab_pipe_slots.txt).  One slot less is ~35 GB of HBM at batch 1024.
static int pipe_slots() {
  static const int n = std::max(2, std::min((int)og_ctx::PIPE_SLOTS, (int)OG_HOOK_INT("OG_PIPE_SLOTS", 2)));
  return n;
}

static int choose_sub_batch(og_ctx* ctx, const og_pk* pk, size_t n) {
  // Large sub-batches amortise the latency-bound tails (reduction levels, scans: a few hundred microseconds each
  // whatever the batch).  Scratch per proof and per scratch slot: the digit entries of the sorts (4 B x nwin x the compacted
  // query sizes -- a shared wire list counted once -- twice: partition + final order), five d x 32 B polynomial buffers, the
  // witness, bucket sets and reduction levels; bounded to ~64 GiB per scratch slot (three slots in the stage pipeline) of
  // the 288 GB -- the 2^18-wire circuit with 17-bit windows (2^16 buckets per set) is ~230 MB per proof, and 256 proofs per
  // sub-batch still fit -- and to what the device has left: (free memory + what this context's arena already holds) x 0.85
  // over the three slots, so a GPU shared with other work gets smaller sub-batches instead of a failed allocation.
  size_t pts = pk->d;
  for (int q = 0; q < 3; q++)
    if (pk->sort_src[q] == q) pts += pk->n_dense[q];
  const size_t per = (size_t)pk->l->nwin * pts * 4 * 2 + pk->d * 32 * 5 + pk->m * 32 + ((size_t)1 << (pk->l->c - 1)) * (4 * 128 + 256) * 2;
  size_t budget = (size_t)64 << 30, free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
    size_t mine = 0;
    for (const auto& kv : ctx->arena) mine += kv.second.second;
    budget = std::min(budget, std::max<size_t>((size_t)1 << 30, (size_t)((double)(free_b + mine) * 0.85) / pipe_slots()));
  }
  // og_set_scratch_budget: the operator's bound on what the sub-batch slots may reserve together (a GPU shared with other work)
  if (ctx->scratch_budget) budget = std::min(budget, std::max<size_t>(per, ctx->scratch_budget / pipe_slots()));
  size_t sb = budget / (per ? per : 1);
  if (const char* e = getenv("OG_SUB_BATCH")) sb = (size_t)atoi(e);
  // At most 256 proofs per sub-batch for the 2^18-wire circuit (~230 MB of scratch per proof: three slots of 256 are 177 GB),
  // more for smaller statements -- up to 1024 where a proof needs <= 40 MB: the natural depth-32 statement (26 k wires, 38 MB per
  // proof) in sub-batches of 252 launches kernels too short to fill the chip and pays 17 latency-bound tails per 4096 proofs.
  // Same box, batch 4096 (hooks builds: OG_SUB_CAP): cap 256 -> 4 297 / 4 280 proofs/s, 512 -> 4 642 / 4 641, 1024 -> 4 817,
  // 2048 -> 4 816 (profiles/r05_ab_sub_cap.txt).
  const size_t cap_by_size = std::max<size_t>(256, std::min<size_t>(1024, ((size_t)40 << 30) / (per ? per : 1)));
  sb = std::max<size_t>(1, std::min<size_t>(sb, (size_t)OG_HOOK_INT("OG_SUB_CAP", (long long)cap_by_size)));
  return (int)std::min(sb, n);
}

// Optional in-lane witness generation for the withdraw circuit: the sub-batch's witnesses are produced by the
// lane that proves them (into lane-private scratch), so the latency-bound MiMC7 walk overlaps the other lane.
struct WithdrawGen {
  int depth;
  uint64_t n_pad3, n_pad2;
  const uint8_t* inputs_d;  // n records of (8 + depth) x 32 B
};

// Window-sharded PROVING (round 6; north_star: "MSM windows shard naturally across GPUs", BASELINE.json configs[3]).  A rank
// of a `world`-rank group runs the whole front of the prover on its own copy of the inputs -- witness, sparse products,
// quotient: replicated, they are latency chains or a few per cent of a proof -- but sorts and accumulates only the windows
// k = rank (mod world) of every query.  The key's tables are per window (tab[k][i] = 2^(c k) P_i: ONE bucket set per query), so
// a rank's share of a query is ONE point per proof and the shares simply add: no Horner step, no per-window slots.  The call
// leaves them in `partials_d` -- five arrays, A | B1 | L | H (n x 128 B each, XYZZ over Fq) | B2 (n x 256 B, XYZZ over Fq2),
// PARTIAL_BYTES per proof -- instead of assembling proofs; the host's all-gather (RCCL inside og_multi_*, torch.distributed in
// owshen_amd/shard.py) and prove_from_partials finish the call.  No reference line: the caller it serves is the
// one-request-per-call site /root/reference/src/services/api_services/withdraw.rs:27-71.
struct WinShard {
  int rank = 0, world = 1;
  uint8_t* partials_d = nullptr;
};
constexpr size_t PARTIAL_BYTES = 4 * 128 + 256;

// How a call of n proofs is cut into sub-batches and which of the three schedules runs them (prove_enqueue; og_prove_plan
// reports it, so that a bench line can say what the step actually was).
struct ProvePlan {
  int sb_max = 1;
  bool split = false, sym = false, pipe = false;
  std::vector<int> plan;
};

static int make_plan(og_ctx* ctx, const og_pk* pk, size_t n, ProvePlan* out) {
  static const bool env_one_lane = OG_HOOK_INT("OG_ONE_LANE", 0) != 0;
  const bool two_lanes = !env_one_lane && ctx->n_lanes >= 2;
  int sb_max = choose_sub_batch(ctx, pk, n);
  const size_t split_max = (size_t)OG_HOOK_INT("OG_SPLIT_MAX", 1024);  // the largest call that fans its queries out (below) instead of halving (hooks builds: OG_SPLIT_MAX=1 restores round 5)
  if (two_lanes && (size_t)sb_max * 2 > n && n >= 2 && n > split_max) sb_max = (int)((n + 1) / 2);
  // A call that is one small sub-batch (the single-withdraw case) has nothing to pipeline across sub-batches; instead the
  // B query (sort + its G1 and G2 MSMs) runs on lane 1 while lane 0 does the quotient and the A, L, H queries: at this
  // size no launch fills the chip, so the two streams genuinely run side by side.
  // Round 6: the same for EVERY call that fits one sub-batch (up to 1024 requests).  Until then such a call was cut in two halves
  // run side by side (`sym` below: each half walks its five MSMs -- five latency-bound tails -- one after the other) or, from 128
  // requests on, sent through the stage pipeline, which has nothing to overlap when there is one sub-batch or two.  Fanned out,
  // the call waits for its longest query instead of their sum.  Same box, interleaved (profiles/r06i_ab_split_max.txt), natural
  // statement: 2 requests 15.7 -> 12.2 ms, 8: 17.5 -> 14.2, 16: 20.3 -> 17.3, 64: 30.0 -> 26.4, 128: 47.7 -> 39.6, 256: 76.0 ->
  // 64.1, 512: 124 -> 116, 1024: level; with the host walking the chains 2: 5.5 -> 2.9, 8: 6.8 -> 4.8, 64: 22.2 -> 19.5;
  // 2^18-wire shape 2: 19.1 -> 15.0, 16: 39.6 -> 36.4, 64: 95.7 -> 87.6, 256: 321 -> 313.
  const bool split = two_lanes && n <= (size_t)sb_max && n <= std::max<size_t>(split_max, 1);
  // Sub-batches too small to fill the chip (a handful of requests) are latency-bound end to end: there, whole sub-batches
  // run side by side on the two streams (`sym`), which is worth ~1.5x (batch 8: 46 vs 68 ms); from 64 proofs per
  // sub-batch on, the stage pipeline (`pipe`) takes over.
  const int pipe_min = (int)OG_HOOK_INT("OG_PIPE_MIN", 64);  // test hook: reach the pipeline at toy sizes
  const bool sym = two_lanes && !split && sb_max < pipe_min;
  const bool pipe = two_lanes && !split && !sym;
  // Sub-batch plan.  A call starts cold: nothing can run on the math stream until the first sub-batch's witnesses, sparse
  // products and sorts exist, and that preparation takes as long as the sub-batch is big.  So the pipelined path RAMPS: a
  // small first sub-batch (its preparation is short, its math covers the preparation of a larger second one), then full
  // size, and the remainder is spread so that no straggler sub-batch is left at the end (1024 proofs at sb_max = 245 used
  // to end on a sub-batch of 44).  OG_SUB_PLAN="64,192,256" overrides (sizes are clamped to sb_max; the last repeats).
  std::vector<int>& plan = out->plan;
  plan.clear();
  {
    size_t left = n;
    if (const char* e = pipe ? OG_HOOK_STR("OG_SUB_PLAN") : nullptr) {
      int last = sb_max;
      for (const char* q = e; *q && left;) {
        OG_REQUIRE(atoi(q) >= 1, "OG_SUB_PLAN: every sub-batch size must be >= 1 (empty or non-numeric entry)");
        last = std::min(sb_max, atoi(q));
        plan.push_back((int)std::min<size_t>(last, left));
        left -= plan.back();
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
      }
      while (left) {
        plan.push_back((int)std::min<size_t>(last, left));
        left -= plan.back();
      }
    } else if (pipe && n > (size_t)sb_max) {
      const int first = std::max(1, std::max(pipe_min, sb_max / 4));  // (OG_PIPE_MIN=0 with sb_max < 4 must not plan an empty sub-batch)
      plan.push_back(first);
      left -= first;
      const size_t parts = (left + sb_max - 1) / sb_max;
      for (size_t k = 0; k < parts; k++) {
        const size_t sz = (left + (parts - k) - 1) / (parts - k);
        plan.push_back((int)sz);
        left -= sz;
      }
    } else {
      while (left) {
        plan.push_back((int)std::min<size_t>(sb_max, left));
        left -= plan.back();
      }
    }
  }
  out->sb_max = sb_max; out->split = split; out->sym = sym; out->pipe = pipe;
  return OG_OK;
}

// sizes_out[0 .. *count_out) = the sub-batches a call of n proofs is cut into; *mode_out = 0 one stream, serial; 1 a call of <= 16 requests, its queries
// fanned out over the streams; 2 whole sub-batches side by side on two streams; 3 the stage pipeline
int prove_plan(og_ctx* ctx, const og_pk* pk, size_t n, uint32_t* sizes_out, size_t cap, size_t* count_out, int* mode_out) {
  ProvePlan pp;
  OG_TRY(make_plan(ctx, pk, n, &pp));
  if (count_out) *count_out = pp.plan.size();
  if (mode_out) *mode_out = pp.pipe ? 3 : pp.sym ? 2 : pp.split ? 1 : 0;
  for (size_t k = 0; k < pp.plan.size() && k < cap; k++) sizes_out[k] = (uint32_t)pp.plan[k];
  return OG_OK;
}

// ---- proof assembly on the HOST (og_set_host_chains) ---------------------------------------------------------------------------
// What is left of a request once its MSMs are done is five scalar multiplications and three inversions: chains of ~3 400 / ~380
// dependent field products, which a lone wave walks at 0.42 us per product (k_assemble_g1_muls_glv + finish + k_assemble_g2:
// 2.2-2.5 ms of a 3.7 ms request whose witness the host already walks) and a server core at 20-50 ns.  So in the same opt-in
// mode the five query results (640 B per proof) come down instead of the proof and the host runs the formulas of
// ecmul_impl.hip.h with the library's own group law (ec.hip.h is host code too: og_verify uses it):
//   A = alpha + Am + r delta1        B = beta2 + B2m + s delta2        C = L + H + s (alpha + Am) + r (beta1 + B1m) + (r s) delta1
// r delta1, (r s) delta1 and s delta2 through fixed-base tables (64 additions each), the other two by 4-bit windows.  The proofs
// are affine and canonical, so the bytes are the kernels'.
template <class T>
static XYZZ<T> host_mul_window4(const XYZZ<T>& p, const uint32_t k[8]) {
  XYZZ<T> tab[16];
  tab[1] = p;
  for (int d = 2; d < 16; d++) tab[d] = d == 2 ? xyzz_dbl(p) : xyzz_add(tab[d - 1], p);
  XYZZ<T> acc = XYZZ<T>::inf();
  for (int w = 63; w >= 0; w--) {
    if (w != 63) for (int e = 0; e < 4; e++) acc = xyzz_dbl(acc);
    const uint32_t dgt = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (dgt) acc = xyzz_add(acc, tab[dgt]);
  }
  return acc;
}
static const std::vector<uint8_t>& host_fb_delta1(const og_pk* pk) {  // tab[w * 16 + d] = d 16^w delta1 (XYZZ; d = 0 unused)
  std::call_once(pk->fb_delta1_once, [&]() {
    pk->fb_delta1_h.resize(64 * 16 * G1XYZZ::BYTES);
    G1XYZZ pw = G1XYZZ::from_affine(G1Affine::load(pk->consts1_h + 128));
    for (int w = 0; w < 64; w++) {
      G1XYZZ q = pw;
      for (int d = 1; d < 16; d++) {
        q.store(pk->fb_delta1_h.data() + (size_t)(w * 16 + d) * G1XYZZ::BYTES);
        q = xyzz_add(q, pw);
      }
      pw = q;  // 16 * (16^w delta1)
    }
  });
  return pk->fb_delta1_h;
}
// The terms that need only (r, s) -- r delta1, (r s) delta1, beta2 + s delta2 -- are formed while the GPU still runs the call's MSMs
// (host_assemble_fixed, at the end of prove_enqueue: 128 + 128 + 256 B per proof, XYZZ); what waits for the results is the rest.
constexpr size_t HOST_FIXED_BYTES = 2 * G1XYZZ::BYTES + G2XYZZ::BYTES;
static void host_assemble_fixed(const og_pk* pk, const uint8_t* rs, uint8_t* out) {
  uint32_t r[8], s[8], p[8];
  memcpy(r, rs, 32);
  memcpy(s, rs + 32, 32);
  fe_to_words(p, fe_from_mont(fe_mul(fe_to_mont(fe_from_words<FrParams>(r)), fe_to_mont(fe_from_words<FrParams>(s)))));  // r s mod the group order
  const uint8_t* fb1 = host_fb_delta1(pk).data();
  auto fixed1 = [&](const uint32_t k[8]) {
    G1XYZZ acc = G1XYZZ::inf();
    for (int w = 0; w < 64; w++) {
      const uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
      if (d) acc = xyzz_add(acc, G1XYZZ::load(fb1 + (size_t)(w * 16 + d) * G1XYZZ::BYTES));
    }
    return acc;
  };
  fixed1(r).store(out);
  fixed1(p).store(out + G1XYZZ::BYTES);
  G2XYZZ B = G2XYZZ::from_affine(G2Affine::load(pk->consts2_h));  // beta2 + s delta2 (k_assemble_g2's table, any order: the sum is the sum)
  for (int w = 0; w < 64; w++) {
    const uint32_t d = (s[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (d) B = xyzz_madd(B, G2Affine::load(pk->fb_delta2_h.data() + (size_t)(w * 16 + d) * G2Affine::BYTES));
  }
  B.store(out + 2 * G1XYZZ::BYTES);
}
static void host_assemble_one(const og_pk* pk, const uint8_t* rs, const uint8_t* fixed, const uint8_t* ra, const uint8_t* rb1,
                              const uint8_t* rb2, const uint8_t* rl, const uint8_t* rh, uint8_t* proof, bool helper_thread) {
  uint32_t r[8], s[8];
  memcpy(r, rs, 32);
  memcpy(s, rs + 32, 32);
  const G1Affine alpha = G1Affine::load(pk->consts1_h), beta1 = G1Affine::load(pk->consts1_h + 64);
  const G1XYZZ a_full = xyzz_madd(G1XYZZ::load(ra), alpha), b1_full = xyzz_madd(G1XYZZ::load(rb1), beta1);
  const G1XYZZ A = xyzz_add(a_full, G1XYZZ::load(fixed));
  G1XYZZ C = xyzz_add(G1XYZZ::load(rl), G1XYZZ::load(rh));
  G1XYZZ prod[2];
  auto mul = [&](size_t k) { prod[k] = k == 0 ? host_mul_window4(a_full, s) : host_mul_window4(b1_full, r); };
  if (helper_thread) host_parallel_for(2, mul);  // one request: its two variable-base products side by side (a call of several has a thread per proof already)
  else { mul(0); mul(1); }
  C = xyzz_add(xyzz_add(C, prod[0]), prod[1]);
  C = xyzz_add(C, G1XYZZ::load(fixed + G1XYZZ::BYTES));
  const G2XYZZ B = xyzz_add(G2XYZZ::load(rb2), G2XYZZ::load(fixed + 2 * G1XYZZ::BYTES));
  G1Affine a = xyzz_to_affine(A), c = xyzz_to_affine(C);
  a.x = fe_from_mont(a.x); a.y = fe_from_mont(a.y);
  c.x = fe_from_mont(c.x); c.y = fe_from_mont(c.y);
  G2Affine b = xyzz_to_affine(B);
  b.x = FieldIO<Fq2>::from_mont(b.x); b.y = FieldIO<Fq2>::from_mont(b.y);
  a.store(proof);
  b.store(proof + 64);
  c.store(proof + 192);
}
static void assemble_fixed_on_host(const og_pk* pk, const uint8_t* rs, size_t n, std::vector<uint8_t>& fixed) {
  fixed.resize(n * HOST_FIXED_BYTES);
  (void)host_fb_delta1(pk);  // (built once, before the threads)
  host_parallel_for(n, [&](size_t g) { host_assemble_fixed(pk, rs + g * 64, fixed.data() + g * HOST_FIXED_BYTES); });
}
// res: the five result arrays of the call on the host, [A n x 128 | B1 n x 128 | B2 n x 256 | L n x 128 | H n x 128]
static void assemble_on_host(const og_pk* pk, const uint8_t* rs, const uint8_t* fixed, const uint8_t* res, size_t n, uint8_t* proofs) {
  const uint8_t *ra = res, *rb1 = ra + n * 128, *rb2 = rb1 + n * 128, *rl = rb2 + n * 256, *rh = rl + n * 128;
  host_parallel_for(n, [&](size_t g) {
    host_assemble_one(pk, rs + g * 64, fixed + g * HOST_FIXED_BYTES, ra + g * 128, rb1 + g * 128, rb2 + g * 256, rl + g * 128, rh + g * 128,
                      proofs + g * 256, n == 1);
  });
}

// serial numbers of jobs (og_job::id)
static uint64_t next_job_id() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

// Latency-bound calls (a handful of requests) hand the assembly the GLV halves of its four scalars r, r s, s, r (glv.h): eight
// half-length chains per proof instead of four of 254 bits.  A scalar whose decomposition does not verify (never seen; a
// non-canonical r or s would do it) sends the whole call down the plain path (out stays empty).  OG_GLV=0 turns it off (A/B).
static void glv_halves(const uint8_t* rs, size_t n, std::vector<uint8_t>& out) {
  out.clear();
  const size_t glv_max = !glv_pair_ok() ? 0 : OG_HOOK_SET("OG_GLV") ? (OG_HOOK_INT("OG_GLV", 1) ? (size_t)1 << 30 : 0) : 64;  // (read per call: tests run both forms)
  if (n > glv_max) return;
  out.resize(n * 128);
  for (size_t g = 0; g < n && !out.empty(); g++) {
    const uint8_t *rb = rs + g * 64, *sb = rs + g * 64 + 32;
    uint32_t rw[8], sw[8], pw[8];
    memcpy(rw, rb, 32);
    memcpy(sw, sb, 32);
    fe_to_words(pw, fe_from_mont(fe_mul(fe_to_mont(fe_from_words<FrParams>(rw)), fe_to_mont(fe_from_words<FrParams>(sw)))));  // r s mod the group order
    uint8_t* o = out.data() + g * 128;
    if (!glv::decompose(rb, o) || !glv::decompose(reinterpret_cast<const uint8_t*>(pw), o + 32) || !glv::decompose(sb, o + 64)) out.clear();
    else memcpy(o + 96, o, 32);  // the fourth product is r (beta + B1m): r again
  }
}

// witnesses_d: n x m x 32 B canonical, device (or null with `gen`).  rs: n x 64 B host.  proofs: n x 256 B host.
//
// Scheduling (n_lanes = 2): a two-stage software pipeline over sub-batches, on two streams with per-parity scratch:
//   PREP stream (lanes[1])   witness generation, the three sparse products + the satisfiability check, the digit sorts of
//                            the A | B | L queries, and -- once the quotient of the same sub-batch exists -- the digit sort
//                            of h: memory- / latency-bound kernels with small register footprints
//   MATH stream (lanes[0])   quotient (NTT) and the five bucket accumulations: VALU-bound
//   TAIL stream              each MSM's heavy buckets / reduction / combine, then the sub-batch's proof assembly
// prep(k + 1) runs under math(k).  All VALU-heavy kernels sit on ONE stream, in order.  (Round 2's first scheme alternated
// whole sub-batches between two symmetric lanes; once bucket accumulation moved to one-wave workgroups, a 300-register
// reduction kernel of one lane could wait for the whole length of the other lane's accumulation kernel -- up to 129 ms in
// the rocprof trace -- because every slot a finishing wave freed was refilled at once by a smaller-footprint wave.)
// Events: e[p][0] sparse products ready, [1..3] sort A / B / L ready, [4] quotient ready, [5] sort h ready, [6] math done.
// pub_out (optional, host): n x n_pub x 32 B, the public wires 1..n_pub of every witness -- what the caller hands the verifier
// with the proof (withdraw: root, nullifier_hash, ...), so that it does not have to generate the witness a second time.
//
// prove_enqueue issues ALL the work of the call and returns an og_job; prove_finish waits for it and copies the results out.
static int prove_finish(og_job* job, size_t* first_bad);

static int prove_enqueue(og_ctx* ctx, const og_pk* pk, const uint8_t* z_d, size_t n, const uint8_t* rs, uint8_t* proofs,
                         const WithdrawGen* gen, uint8_t* pub_out, og_job** job_out, const uint8_t* z_host = nullptr,
                         bool trusted_z = false, const WinShard* sh = nullptr) {
  *job_out = nullptr;
  const int wr = sh ? sh->rank : 0, ww = sh ? sh->world : 1;  // this rank's windows: k = wr (mod ww)
  auto dsort = [&](int slot, const uint8_t* sc, size_t stride, size_t cnt, const uint32_t* map, int batch, int c, DigitSort* out) -> int {
    return msm_digit_sort_windows(ctx, slot, sc, stride, cnt, map, batch, c, 1, wr, ww, out);
  };
  // the first FREE call slot (not a toggle: a blocking call between a submit and its wait would flip a toggle back onto the
  // slot that is still occupied, and the next submit would be refused although only one call is in flight)
  const int call_slot = ctx->jobs[0] == nullptr ? 0 : 1;
  OG_REQUIRE(ctx->jobs[call_slot] == nullptr, "og_prove: two calls are already in flight on this context (og_job_wait or og_job_abandon one of them first)");
  const std::string cs = "#" + std::to_string(call_slot);  // call-level buffers exist once per call slot
  const size_t m = pk->m, d = pk->d;
  ProvePlan pp;
  OG_TRY(make_plan(ctx, pk, n, &pp));
  const int sb_max = pp.sb_max;
  const bool split = pp.split, sym = pp.sym, pipe = pp.pipe;
  const std::vector<int>& plan = pp.plan;
  uint8_t *res[5], *rs_d, *proofs_d, *asm_tmp;
  uint32_t* flags;
  const char* evn[3] = {"g16.eva", "g16.evb", "g16.evc"};
  struct LaneGuard {  // an error return in the middle of the pipeline leaves no work of this call in flight (the next call
    og_ctx* c;        // reuses the scratch); on every path the ctx is back on lane 0
    bool ok = false;  // set once everything is enqueued: then the streams are left running
    ~LaneGuard() {
      if (!ok) (void)drain_streams(c);
      c->lane = 0;
      c->stream = c->lanes[0];
      c->tail_stream = nullptr;
      c->sort_beside_acc = false;
    }
  } lane_guard{ctx};
  struct CallSize {  // the witness generator's host-chains bound looks at the call, not at the sub-batch it is handed
    og_ctx* c;
    ~CallSize() { c->call_requests = 0; }
  } call_size{ctx};
  ctx->call_requests = n;
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  const char* resn[5] = {"g16.res.a", "g16.res.b1", "g16.res.b2", "g16.res.l", "g16.res.h"};
  if (sh) {  // the queries' results ARE the call's output: this rank's partial sums, array by array
    res[0] = sh->partials_d; res[1] = res[0] + n * 128; res[3] = res[1] + n * 128; res[4] = res[3] + n * 128; res[2] = res[4] + n * 128;
  } else {
    for (int k = 0; k < 5; k++) OG_TRY(arena_get(ctx, (resn[k] + cs).c_str(), n * (k == 2 ? 256 : 128), (void**)&res[k]));
  }
  OG_TRY(arena_get(ctx, ("g16.rs" + cs).c_str(), n * 64, (void**)&rs_d));
  OG_TRY(arena_get(ctx, ("g16.proofs" + cs).c_str(), n * 256, (void**)&proofs_d));
  // og_set_host_chains: a call of a handful of requests leaves its assembly to the host (assemble_on_host, in prove_finish)
  const bool host_asm = !sh && ctx->host_chains_max > 0 && n <= (size_t)ctx->host_chains_max;
  std::vector<uint8_t> glv_h;
  if (!sh && !host_asm) glv_halves(rs, n, glv_h);
  const size_t asm_lanes = glv_h.empty() ? 4 : 8;
  OG_TRY(arena_get(ctx, ("g16.asm" + cs).c_str(), n * asm_lanes * 128 * 17, (void**)&asm_tmp));  // results + window tables of 16 points per lane
  uint8_t* glv_d = nullptr;
  if (!glv_h.empty()) OG_TRY(arena_get(ctx, ("g16.glv" + cs).c_str(), n * 128, (void**)&glv_d));
  OG_TRY(arena_get(ctx, ("g16.flags" + cs).c_str(), n * 8, (void**)&flags));  // [n] unsatisfied flags | [n] first non-canonical wire / field
  uint32_t* bad = flags + n;
  uint8_t* pub_d = nullptr;
  if (pub_out && pk->n_pub) OG_TRY(arena_get(ctx, ("g16.pub" + cs).c_str(), n * pk->n_pub * 32, (void**)&pub_d));
  // (r, s) go in on the copy stream, which never holds compute: the copy does not queue behind a previous call's kernels,
  // and every stream of this call may read rs_d once the host has seen it complete
  if (!sh) OG_HIP(hipMemcpyAsync(rs_d, rs, n * 64, hipMemcpyHostToDevice, ctx->copy_lane));  // (a sharded front assembles nothing: no blinding yet)
  if (glv_d) OG_HIP(hipMemcpyAsync(glv_d, glv_h.data(), n * 128, hipMemcpyHostToDevice, ctx->copy_lane));
  OG_HIP(hipStreamSynchronize(ctx->copy_lane));
  if (ctx->pipe_ev[0][0] == nullptr)
    for (int p = 0; p < og_ctx::PIPE_SLOTS; p++)
      for (int e = 0; e < og_ctx::PIPE_EVENTS; e++) OG_HIP(hipEventCreateWithFlags(&ctx->pipe_ev[p][e], hipEventDisableTiming));
  ctx->sort_beside_acc = pipe;
  // A call may be enqueued while the previous one is still running (og_withdraw_prove_batch_submit_d).  Between two calls
  // of the stage pipeline the scratch slots are guarded by their events; any other combination shares scratch without such
  // guards, so the streams are drained first (a no-op for the blocking entry points, which left them idle).
  if (!(pipe && ctx->last_call_piped)) OG_HIP(drain_streams(ctx));
  ctx->last_call_piped = pipe;
  hipStream_t math = ctx->lanes[0], prep = pipe ? ctx->lanes[1] : ctx->lanes[0];
  auto on = [&](hipStream_t st) { ctx->stream = st; };
  auto rec = [&](hipEvent_t e) -> int {
    if (pipe) OG_HIP(hipEventRecord(e, ctx->stream));
    return OG_OK;
  };
  auto wait = [&](hipEvent_t e) -> int {
    if (pipe) OG_HIP(hipStreamWaitEvent(ctx->stream, e, 0));
    return OG_OK;
  };
  hipEvent_t* prev_ev = nullptr;  // the events of the previous sub-batch of THIS call (stage pipeline): gates of the next quotient
  size_t g0 = 0;
  for (size_t sub_index = 0; sub_index < plan.size(); g0 += plan[sub_index], sub_index++) {
    const int sb = plan[sub_index];
    // Scratch slot of this sub-batch.  The stage pipeline uses THREE slots: the preparation of sub-batch k + 1 must not wait
    // for the assembly of k - 1 -- the end of the tail stream's chain, and the tail kernels (92..152 registers) are the ones
    // that find room last beside the accumulation and the sorts (round 3 trace: every sub-batch boundary cost the math stream
    // ~65 ms waiting for exactly that) -- but only for k - 2, which is long done.  Symmetric lanes keep two.
    const int n_slots = pipe_slots();
    static const bool early_release = OG_HOOK_INT("OG_PIPE_EARLY_RELEASE", 1) != 0;  // (A/B hook: 0 = the preparation waits for the assembly, as before round 6)
    // (the pipeline's slot index runs on across calls: the next call's first sub-batch must not take the slot this call's
    // last one is still using, and the slot's "free" event is the one its previous user recorded, whichever call that was)
    const int par = pipe ? (int)(ctx->pipe_counter++ % n_slots) : (sym ? (int)(sub_index & 1) : 0);
    if (sym) math = prep = ctx->lanes[par];
    hipEvent_t* ev_ = ctx->pipe_ev[par];
    ctx->lane = par;  // scratch namespace of this sub-batch (both stages)
    uint8_t *ev[3], *tmp, *h;
    bool asm_on_tail = false;
    // a call whose queries fan out assembles in two parts (ecmul_impl.hip.h k_assemble_g1_early / _late; OG_ASM_EARLY=0: A/B, the one-part form)
    const bool asm_early = split && !sh && !host_asm && OG_HOOK_INT("OG_ASM_EARLY", 1) != 0;
    // ---------------- PREP ----------------
    on(prep);
    // the preparation-side scratch of this slot is free once its previous user's last heavy-bucket kernels have run ([11]; the
    // stage pipeline); symmetric lanes and OG_PIPE_EARLY_RELEASE=0 wait for the previous user's assembly ([6])
    OG_TRY(wait(pipe && early_release ? ev_[11] : ev_[6]));
    for (int k = 0; k < 3; k++) OG_TRY(arena_get(ctx, evn[k], (size_t)sb_max * d * 32, (void**)&ev[k]));
    OG_TRY(arena_get(ctx, "g16.tmp", 32, (void**)&tmp));
    OG_TRY(arena_get(ctx, "g16.h", (size_t)sb_max * d * 32, (void**)&h));
    const uint8_t* zs = z_d ? z_d + g0 * m * 32 : nullptr;
    OG_HIP(hipMemsetAsync(bad + g0, 0xff, (size_t)sb * 4, ctx->stream));
    if (gen) {
      OG_TRY(withdraw_check_records(ctx, gen->depth, gen->inputs_d + g0 * (size_t)(8 + gen->depth) * 32, (size_t)sb, bad + g0));
      uint8_t* zbuf = nullptr;
      OG_TRY(arena_get(ctx, "g16.zgen", (size_t)sb_max * m * 32, (void**)&zbuf));
      OG_TRY(withdraw_witness(ctx, gen->depth, gen->n_pad3, gen->n_pad2, gen->inputs_d + g0 * (size_t)(8 + gen->depth) * 32, (size_t)sb,
                              zbuf));
      zs = zbuf;
    } else if (z_host) {
      // witnesses in HOST memory (og_prove_batch): the sub-batch crosses PCIe into its slot's staging buffer on the prep
      // stream -- under the math stage of the previous sub-batch, like everything else the prep stream does.  (Pageable
      // memory: the call holds the host until the copy is done; the math work of sub-batch k is already enqueued by then.)
      uint8_t* zbuf = nullptr;
      OG_TRY(arena_get(ctx, "g16.zgen", (size_t)sb_max * m * 32, (void**)&zbuf));
      OG_HIP(hipMemcpyAsync(zbuf, z_host + g0 * m * 32, (size_t)sb * m * 32, hipMemcpyHostToDevice, ctx->stream));
      zs = zbuf;
    }
    if (!gen && !trusted_z) {  // caller-supplied witnesses: every wire must be canonical (one more read of the witness)
      hipLaunchKernelGGL(k_check_canonical, dim3(grid_for(m, 256), sb), dim3(256), 0, ctx->stream, zs, m * 32, m, bad + g0);
      OG_HIP(hipGetLastError());
    }
    if (pub_d)  // wires 1..n_pub of each witness of the sub-batch (a strided device-to-device copy)
      OG_HIP(hipMemcpy2DAsync(pub_d + g0 * pk->n_pub * 32, pk->n_pub * 32, zs + 32, m * 32, pk->n_pub * 32, (size_t)sb,
                              hipMemcpyDeviceToDevice, ctx->stream));
    if (split) {
      // One request: nothing here fills the chip, and every MSM ends in a chain of ~140 dependent additions (its bucket
      // reduction: 0.9 ms in G1, 2.5 ms in G2).  So the four witness queries fan out -- B's sort and G2 MSM on stream 1, its G1
      // MSM (same sorted entries) on the copy stream, A on the tail stream, L on the aux stream -- while stream 0 goes on to
      // the quotient and the H query; the assembly joins them.  (Round 2 ran A, L, H one after the other on stream 0.)
      OG_HIP(hipEventRecord(ctx->ev0, ctx->lanes[0]));  // the witness is complete
      hipEvent_t* sev = ctx->pipe_ev[0];                // (the stage pipeline's events are idle in this mode)
      auto side = [&](int lane_id, hipStream_t st, hipEvent_t after) -> int {
        ctx->lane = lane_id;  // scratch namespace
        ctx->stream = st;
        OG_HIP(hipStreamWaitEvent(st, after, 0));
        return OG_OK;
      };
      DigitSort dsb, dsa, dsl;
      OG_TRY(side(1, ctx->lanes[1], ctx->ev0));
      OG_TRY(dsort(1, zs, m * 32, pk->n_dense[1], pk->map[1], sb, pk->b1->c, &dsb));
      OG_HIP(hipEventRecord(sev[1], ctx->lanes[1]));
      OG_TRY(msm_run(ctx, pk->b2, dsb, res[2] + g0 * 256));
      {  // B's half of the proof is assembled right here, on the stream that produced B2: the G2 query is the longest chain of
         // a request (its bucket reduction: 2.5 ms), and its 1.2 ms of assembly (s delta2 from the fixed-base table, one
         // inversion) used to queue behind the G1 half on stream 0 instead of running beside it
        ProfScope ps_asm(ctx, PROF_ASSEMBLE, 0.0);  // (0 items: the G1 half below counts the sub-batch's proofs, og_profile_read must not see them twice)
        if (!sh && !host_asm)  // (a wave per proof, the sum as a tree: OG_ASM_G2_TREE=0 is the lane per proof it replaced here)
          OG_TRY(assemble_g2(ctx, pk->consts2, pk->fb_delta2, rs_d + g0 * 64, res[2] + g0 * 256, (size_t)sb, proofs_d + g0 * 256,
                             OG_HOOK_INT("OG_ASM_G2_TREE", 1) != 0));
      }
      OG_HIP(hipEventRecord(ctx->ev1, ctx->lanes[1]));
      hipStream_t s_b1 = ctx->copy_lane ? ctx->copy_lane : ctx->lanes[1];
      OG_TRY(side(4, s_b1, sev[1]));
      OG_TRY(msm_run(ctx, pk->b1, dsb, res[1] + g0 * 128));
      OG_HIP(hipEventRecord(sev[2], s_b1));
      hipStream_t s_a = ctx->tail_lane ? ctx->tail_lane : ctx->lanes[1];
      OG_TRY(side(2, s_a, ctx->ev0));
      OG_TRY(dsort(1, zs, m * 32, pk->n_dense[0], pk->map[0], sb, pk->a->c, &dsa));
      OG_TRY(msm_run(ctx, pk->a, dsa, res[0] + g0 * 128));
      if (asm_early) {  // the part of the G1 assembly that needs A and B1 only: here, beside the quotient and the H query on stream 0
        ProfScope ps_asm(ctx, PROF_ASSEMBLE, 0.0);  // (0 items: as for the G2 half above)
        OG_HIP(hipStreamWaitEvent(s_a, sev[2], 0));  // B1
        OG_TRY(assemble_g1_early(ctx, pk->consts1, rs_d + g0 * 64, res[0] + g0 * 128, res[1] + g0 * 128, (size_t)sb,
                                 asm_tmp + g0 * asm_lanes * 128 * 17, proofs_d + g0 * 256, glv_d ? glv_d + g0 * 128 : nullptr, sev[5]));
      }
      OG_HIP(hipEventRecord(sev[3], s_a));
      hipStream_t s_l = ctx->aux_lane ? ctx->aux_lane : ctx->lanes[1];
      OG_TRY(side(3, s_l, ctx->ev0));
      OG_TRY(dsort(1, zs, m * 32, pk->n_dense[2], pk->map[2], sb, pk->l->c, &dsl));
      OG_TRY(msm_run(ctx, pk->l, dsl, res[3] + g0 * 128));
      OG_HIP(hipEventRecord(sev[4], s_l));
      ctx->lane = 0;
      ctx->stream = ctx->lanes[0];
    }
    for (int k = 0; k < 3; k++) {
      ProfScope ps(ctx, PROF_SPMV, (double)pk->nnz[k] * sb);
      if (k == 2 && pk->c_is_ab) {
        hipLaunchKernelGGL(k_mul_rows, dim3(grid_for(d, 256), sb), dim3(256), 0, ctx->stream, ev[0], ev[1], ev[2], d);
        OG_HIP(hipGetLastError());
        continue;
      }
      hipLaunchKernelGGL(k_spmv, dim3(grid_for(d, 256), sb), dim3(256), 0, ctx->stream, pk->ptr[k], pk->col[k], pk->val[k],
                         (size_t)pk->n_rows, d, zs, m * 32, ev[k], d * 32, 1, 1, SPMV_LONG);
      OG_HIP(hipGetLastError());
      if (pk->n_long[k]) {
        hipLaunchKernelGGL(k_spmv_long, dim3(pk->n_long[k], sb), dim3(256), 0, ctx->stream, pk->long_rows[k], pk->ptr[k], pk->col[k],
                           pk->val[k], zs, m * 32, ev[k], d * 32, 1, 1);
        OG_HIP(hipGetLastError());
      }
    }
    OG_HIP(hipMemsetAsync(flags + g0, 0, (size_t)sb * 4, ctx->stream));
    if (pk->n_rows) {
      hipLaunchKernelGGL(k_check_rows, dim3(grid_for(pk->n_rows, 256), sb), dim3(256), 0, ctx->stream, ev[0], ev[1], ev[2],
                         (size_t)pk->n_rows, d, zs, m * 32, flags + g0);
      OG_HIP(hipGetLastError());
    }
    OG_STEP(ctx, "g16.spmv");
    OG_TRY(rec(ev_[0]));
    // one digit sort per density map: A | B (G1 and G2 copies) | L, each over its compacted wire list.  Pipelined, every
    // query keeps its own sorted entries (slots 1..3) so that all three can be ready before the math stage needs them;
    // serial, they share slot 1 and are interleaved with the MSMs as before (a third of the scratch).
    DigitSort ds_a, ds_b, ds_l, dh;
    if (pipe) {
      OG_TRY(dsort(1, zs, m * 32, pk->n_dense[0], pk->map[0], sb, pk->a->c, &ds_a));
      OG_TRY(rec(ev_[1]));
      if (pk->sort_src[1] == 0) ds_b = ds_a;  // same wire list (pk_load): the sorted entries serve both
      else OG_TRY(dsort(3, zs, m * 32, pk->n_dense[1], pk->map[1], sb, pk->b1->c, &ds_b));
      OG_TRY(rec(ev_[2]));
      if (pk->sort_src[2] == 0) ds_l = ds_a;
      else if (pk->sort_src[2] == 1) ds_l = ds_b;
      else OG_TRY(dsort(4, zs, m * 32, pk->n_dense[2], pk->map[2], sb, pk->l->c, &ds_l));
      OG_TRY(rec(ev_[3]));
    }
    // ---------------- QUOTIENT ----------------
    // On the math stream, in front of the sub-batch's accumulations.  Round 4 measured the alternative -- the quotient of
    // sub-batch k + 1 on the aux stream BESIDE the accumulations of sub-batch k (OG_HPOLY_ASIDE=1), its 11 passes gated so
    // that they start beside a G1 accumulation and not in front of the G2 one, whose accumulators fill the LDS
    // (OG_HPOLY_GATED) -- and it loses, same box, interleaved: 583.2 / 584.1 (here) against 573.8 / 575.3 (aside, ungated),
    // 591.0 / 591.5 against 580.4 / 580.6 (aside, gated).  The NTT passes are not the half-idle kernels their 0.43 of an
    // idealised butterfly count suggested: beside the accumulation they take from it what they cost alone (accumulate
    // 1490 -> 1571 ms per 1024 proofs for 95 ms of quotient), and the ~37 ms per sub-batch in which the math stream runs the
    // quotient are the only windows where the tails' big-register kernels (reduction: 184 / 308 registers) find room;
    // without them every tail kernel waits for an accumulation-kernel boundary, the assembly of sub-batch k - 3 is late, the
    // preparation of k stalls on its scratch slot, and the math stream idles 170 - 190 ms instead of 35 - 60.
    static const bool hpoly_aside = OG_HOOK_INT("OG_HPOLY_ASIDE", 0) != 0;
    const bool quot_aside = pipe && hpoly_aside && ctx->aux_lane != nullptr;
    static const bool hpoly_gated = OG_HOOK_INT("OG_HPOLY_GATED", 1) != 0;
    on(quot_aside ? ctx->aux_lane : math);
    OG_TRY(wait(ev_[0]));
    {
      ProfScope ps(ctx, PROF_HPOLY, (double)d * sb);
      OG_TRY(h_poly_device(ctx, ev[0], ev[1], ev[2], tmp, h, (int)pk->log_d, sb, quot_aside && hpoly_gated && prev_ev ? prev_ev + 7 : nullptr));
    }
    OG_STEP(ctx, "g16.hpoly");
    OG_TRY(rec(ev_[4]));
    // ---------------- MATH ----------------
    if (pipe) {
      // the H query's sort needs THIS sub-batch's quotient and is needed by its last accumulation: on a stream of its own it
      // does not queue behind the next sub-batch's preparation (which the prep stream was given first)
      on(ctx->aux_lane ? ctx->aux_lane : prep);
      OG_TRY(wait(ev_[4]));
      OG_TRY(dsort(2, h, d * 32, d - 1, nullptr, sb, pk->h->c, &dh));
      OG_TRY(rec(ev_[5]));
      on(math);
      // the five accumulation kernels run back to back on the math stream; each MSM's tail (heavy buckets, reduction,
      // combine) goes to the tail stream, where it fills the ramp-down of the following accumulation
      static const bool no_tail = OG_HOOK_INT("OG_NO_TAIL", 0) != 0;
      struct TailGuard {
        og_ctx* c;
        ~TailGuard() { c->tail_stream = nullptr; c->msm_tag = 0; }
      } tail_guard{ctx};
      ctx->tail_stream = no_tail ? nullptr : ctx->tail_lane;
      if (early_release) OG_TRY(wait(ev_[6]));  // the bucket sets / reduction levels of this slot: free once its previous user is assembled
      OG_TRY(wait(ev_[1]));
      OG_TRY(rec(ev_[7]));  // "about to launch accumulation A": the gates of the next sub-batch's quotient passes
      ctx->msm_tag = 0;
      OG_TRY(msm_run(ctx, pk->a, ds_a, res[0] + g0 * 128));
      OG_TRY(wait(ev_[2]));
      OG_TRY(rec(ev_[8]));
      ctx->msm_tag = 1;
      OG_TRY(msm_run(ctx, pk->b1, ds_b, res[1] + g0 * 128));
      ctx->msm_tag = 2;
      OG_TRY(msm_run(ctx, pk->b2, ds_b, res[2] + g0 * 256));
      OG_TRY(wait(ev_[3]));
      OG_TRY(rec(ev_[9]));
      ctx->msm_tag = 3;
      if (pk->merge_lh) {  // L and H into ONE bucket set (pk_load): the L half accumulates and stops, its result slot is the point at infinity
        OG_HIP(hipMemsetAsync(res[3] + g0 * 128, 0, (size_t)sb * 128, ctx->stream));
        OG_TRY(msm_run_phase(ctx, pk->l, ds_l, nullptr, MSM_FIRST));
      } else {
        OG_TRY(msm_run(ctx, pk->l, ds_l, res[3] + g0 * 128));
      }
      OG_TRY(wait(ev_[5]));
      OG_TRY(rec(ev_[10]));
      {
        struct AfterHeavy {  // the H query is the sub-batch's last MSM: behind its heavy buckets nothing reads the slot's digit sorts
          og_ctx* c;
          ~AfterHeavy() { c->after_heavy_ev = nullptr; }
        } after_heavy{ctx};
        ctx->after_heavy_ev = ev_[11];
        if (pk->merge_lh) {
          OG_TRY(msm_run_phase(ctx, pk->h, dh, res[4] + g0 * 128, MSM_SECOND));  // (same msm_tag: the same buckets)
        } else {
          ctx->msm_tag = 4;
          OG_TRY(msm_run(ctx, pk->h, dh, res[4] + g0 * 128));
        }
      }
      prev_ev = ev_;
      // Assembly needs every tail, and it is latency-bound (a few waves of scalar multiplications): it is queued on the
      // tail stream behind the last tail, so the math stream goes straight on to the next sub-batch's quotient instead of
      // idling through the H query's reduction and the assembly.  (Everything the math stream did for this sub-batch
      // precedes one of the tails, so "assembly done" on the tail stream is also "math done".)
      static const bool asm_on_math = OG_HOOK_INT("OG_ASM_ON_MATH", 0) != 0;  // A/B hook: the old order
      if (ctx->tail_stream && asm_on_math) {
        OG_HIP(hipEventRecord(ctx->ev1, ctx->tail_stream));
        OG_HIP(hipStreamWaitEvent(math, ctx->ev1, 0));
      }
      asm_on_tail = ctx->tail_stream != nullptr && !asm_on_math;
      ctx->tail_stream = nullptr;
    } else {
      DigitSort ds;
      if (!split) {  // (one request: the A, B and L queries are already under way on their own streams)
        // A, then the queries that share A's wire list (their sort is A's: pk_load), then the rest; `held` = whose list the
        // one set of sorted entries (slot 1) currently holds
        int order[3] = {0, 1, 2}, held = -1;
        if (pk->sort_src[1] != 0 && pk->sort_src[2] == 0) std::swap(order[1], order[2]);
        for (int q : order) {
          if (pk->sort_src[q] != held) {
            OG_TRY(dsort(1, zs, m * 32, pk->n_dense[q], pk->map[q], sb, (q == 0 ? pk->a : q == 1 ? pk->b1 : pk->l)->c, &ds));
            held = pk->sort_src[q];
          }
          if (q == 0) OG_TRY(msm_run(ctx, pk->a, ds, res[0] + g0 * 128));
          if (q == 1) {
            OG_TRY(msm_run(ctx, pk->b1, ds, res[1] + g0 * 128));
            OG_TRY(msm_run(ctx, pk->b2, ds, res[2] + g0 * 256));
          }
          if (q == 2) {
            if (pk->merge_lh) {
              OG_HIP(hipMemsetAsync(res[3] + g0 * 128, 0, (size_t)sb * 128, ctx->stream));
              OG_TRY(msm_run_phase(ctx, pk->l, ds, nullptr, MSM_FIRST));
            } else {
              OG_TRY(msm_run(ctx, pk->l, ds, res[3] + g0 * 128));
            }
          }
        }
      }
      OG_TRY(dsort(2, h, d * 32, d - 1, nullptr, sb, pk->h->c, &dh));
      // (one request fanned out over the streams keeps L and H apart: there the two run SIDE BY SIDE, which is worth more)
      OG_TRY(msm_run_phase(ctx, pk->h, dh, res[4] + g0 * 128, pk->merge_lh && !split ? MSM_SECOND : MSM_FULL));
    }
    OG_STEP(ctx, "g16.msm");
    if (split) {  // the side streams' G1 results (A, B1, L); the G2 half joins after the G1 assembly below
      // (two-part assembly: C's sum waits for the four PRODUCTS [5] and L [4]; A's own sum and inversion [3] join at the end)
      for (int k = asm_early ? 4 : 2; k <= (asm_early ? 5 : 4); k++) OG_HIP(hipStreamWaitEvent(ctx->lanes[0], ctx->pipe_ev[0][k], 0));
    }
    if (asm_on_tail) on(ctx->tail_lane);
    if (asm_early) {  // the products and A are the side stream's (above): C = L + H + three of the products is left
      ProfScope ps_asm(ctx, PROF_ASSEMBLE, (double)sb);
      OG_TRY(assemble_g1_late(ctx, res[3] + g0 * 128, res[4] + g0 * 128, (size_t)sb, asm_tmp + g0 * asm_lanes * 128 * 17, proofs_d + g0 * 256,
                              glv_d != nullptr));
      OG_STEP(ctx, "g16.assemble");
    } else if (!sh && !host_asm) {  // assemble this sub-batch's proofs (latency-bound scalar multiplications)
      ProfScope ps_asm(ctx, PROF_ASSEMBLE, (double)sb);
      OG_TRY(assemble_g1(ctx, pk->consts1, rs_d + g0 * 64, res[0] + g0 * 128, res[1] + g0 * 128, res[3] + g0 * 128, res[4] + g0 * 128,
                         (size_t)sb, asm_tmp + g0 * asm_lanes * 128 * 17, proofs_d + g0 * 256,  // (a sub-batch's products and tables: its own region)
                         glv_d ? glv_d + g0 * 128 : nullptr));
      if (!split) OG_TRY(assemble_g2(ctx, pk->consts2, pk->fb_delta2, rs_d + g0 * 64, res[2] + g0 * 256, (size_t)sb, proofs_d + g0 * 256));
      OG_STEP(ctx, "g16.assemble");
    }
    if (asm_early) OG_HIP(hipStreamWaitEvent(ctx->lanes[0], ctx->pipe_ev[0][3], 0));  // A's half of the proof
    if (split) OG_HIP(hipStreamWaitEvent(ctx->lanes[0], ctx->ev1, 0));  // stream 0 ends after B's half too (scratch reuse by the next call)
    OG_TRY(rec(ev_[6]));
    if (asm_on_tail) on(math);
  }
  // everything is enqueued: one event per stream marks the end of this call's work there
  og_job* job = new og_job();
  job->id = next_job_id();
  job->ctx = ctx; job->call_slot = call_slot; job->n = n; job->n_pub = pub_d ? pk->n_pub : 0;
  job->proofs = proofs; job->pub_out = pub_out; job->proofs_d = proofs_d; job->pub_d = pub_d; job->flags_d = flags;
  job->bad_kind = gen ? 2 : (trusted_z ? 0 : 1);
  if (host_asm) {
    job->host_asm_pk = pk;
    for (int k = 0; k < 5; k++) job->res_d[k] = res[k];
    job->rs_h.assign(rs, rs + n * 64);
  }
  hipStream_t all[4] = {ctx->lanes[0], ctx->lanes[1], ctx->tail_lane, ctx->aux_lane};
  for (hipStream_t st : all) {
    if (!st) continue;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, st) != hipSuccess) {
      for (int k = 0; k < job->n_done; k++) (void)hipEventDestroy(job->done[k]);
      if (e) (void)hipEventDestroy(e);
      delete job;
      set_error("og_prove: could not record the completion events");
      return OG_ERR_HIP;  // (the guard drains the streams)
    }
    job->done[job->n_done++] = e;
  }
  lane_guard.ok = true;
  ctx->jobs[call_slot] = job;
  *job_out = job;
  if (host_asm) assemble_fixed_on_host(pk, job->rs_h.data(), n, job->host_fixed);  // (the GPU is busy with the call's MSMs meanwhile)
  return OG_OK;
}

// waits for the job's last kernels, copies proofs / flags / public inputs to the caller's buffers, frees the job
static int prove_finish(og_job* job, size_t* first_bad) {
  og_ctx* ctx = job->ctx;
  const size_t n = job->n;
  if (job->call_slot < 0) {  // completed inside the submit call
    auto& dj = ctx->done_jobs;
    dj.erase(std::remove(dj.begin(), dj.end(), job), dj.end());
    delete job;
    return OG_OK;
  }
  std::vector<uint32_t> fl(2 * n);
  struct Release {
    og_job* j;
    ~Release() {
      for (int k = 0; k < j->n_done; k++) (void)hipEventDestroy(j->done[k]);
      if (j->ctx->jobs[j->call_slot] == j) j->ctx->jobs[j->call_slot] = nullptr;
      delete j;
    }
  } release{job};
  for (int k = 0; k < job->n_done; k++) OG_HIP(hipStreamWaitEvent(ctx->copy_lane, job->done[k], 0));
  std::vector<uint8_t> res_h;
  if (job->host_asm_pk) {  // og_set_host_chains: the five query results come down, the host assembles (below)
    res_h.resize(n * 768);
    const size_t off[5] = {0, n * 128, n * 256, n * 512, n * 640}, len[5] = {128, 128, 256, 128, 128};
    for (int k = 0; k < 5; k++) OG_HIP(hipMemcpyAsync(res_h.data() + off[k], job->res_d[k], n * len[k], hipMemcpyDeviceToHost, ctx->copy_lane));
  } else if (job->proofs) OG_HIP(hipMemcpyAsync(job->proofs, job->proofs_d, n * 256, hipMemcpyDeviceToHost, ctx->copy_lane));  // (null: a sharded front)
  OG_HIP(hipMemcpyAsync(fl.data(), job->flags_d, n * 8, hipMemcpyDeviceToHost, ctx->copy_lane));
  if (job->pub_d) OG_HIP(hipMemcpyAsync(job->pub_out, job->pub_d, n * job->n_pub * 32, hipMemcpyDeviceToHost, ctx->copy_lane));
  OG_HIP(hipStreamSynchronize(ctx->copy_lane));
  if (job->host_asm_pk && job->proofs) assemble_on_host(job->host_asm_pk, job->rs_h.data(), job->host_fixed.data(), res_h.data(), n, job->proofs);
  // a malformed input (a non-canonical encoding) comes before "does not satisfy": OG_ERR_INVALID, naming the first offender
  if (job->bad_kind)
    for (size_t g = 0; g < n; g++) {
      const uint32_t b = fl[n + g];
      if (b == 0xffffffffu) continue;
      if (first_bad) *first_bad = g;
      if (job->bad_kind == 2)
        set_error("og_withdraw_prove: input record " + std::to_string(g) + ": field " + std::to_string(b) + " (" + withdraw_field_name(b) +
                  (b >= 8 ? " " + std::to_string(b - 8) : std::string()) + ") is not a canonical value (>= r, or an index outside the tree)");
      else
        set_error("og_prove: witness " + std::to_string(g) + ": wire " + std::to_string(b) + " is not a canonical Fr element (>= r)");
      return OG_ERR_INVALID;
    }
  for (size_t g = 0; g < n; g++)
    if (fl[g]) {
      if (first_bad) *first_bad = g;
      set_error("og_prove: witness " + std::to_string(g) + " does not satisfy the circuit (a row has a*b != c, or wire 0 is not 1)");
      return OG_ERR_UNSATISFIED;
    }
  return OG_OK;
}

// blocking form: enqueue + finish
static int prove_batch_impl(og_ctx* ctx, const og_pk* pk, const uint8_t* z_d, size_t n, const uint8_t* rs, uint8_t* proofs,
                            size_t* first_bad, const WithdrawGen* gen, uint8_t* pub_out, bool trusted_z = false) {
  if (n == 0) return OG_OK;
  og_job* job = nullptr;
  OG_TRY(prove_enqueue(ctx, pk, z_d, n, rs, proofs, gen, pub_out, &job, nullptr, trusted_z));
  return prove_finish(job, first_bad);
}

int prove_batch_device(og_ctx* ctx, const og_pk* pk, const uint8_t* z_d, size_t n, const uint8_t* rs, uint8_t* proofs,
                       size_t* first_bad) {
  return prove_batch_impl(ctx, pk, z_d, n, rs, proofs, first_bad, nullptr, nullptr);
}

// host witnesses: staged through a device buffer one sub-batch-sized slab at a time
int prove_batch_host(og_ctx* ctx, const og_pk* pk, const uint8_t* z, size_t n, const uint8_t* rs, uint8_t* proofs) {
  if (n == 0) return OG_OK;
  // ONE call: every sub-batch's witnesses are copied into its scratch slot by the prep stream (prove_enqueue, z_host), so
  // the PCIe traffic of sub-batch k + 1 (8.4 MB per 2^18-wire witness) runs under the accumulations of sub-batch k.
  // (Rounds 1-2 copied a 4 GiB slab, then proved it, then copied the next: -7 % against resident witnesses,
  // profiles/r03_host_boundary.json.)
  og_job* job = nullptr;
  OG_TRY(prove_enqueue(ctx, pk, nullptr, n, rs, proofs, nullptr, nullptr, &job, z));
  size_t bad = 0;
  const int r = prove_finish(job, &bad);
  if (r == OG_ERR_UNSATISFIED)
    set_error("og_prove: witness " + std::to_string(bad) + " does not satisfy the circuit (a row has a*b != c, or wire 0 is not 1)");
  return r;
}

// blocking form of the record check (og_withdraw_witness_d, the whole-slab path of small circuits): OG_ERR_INVALID names the
// first malformed record (`base` = index of record 0 in the caller's batch)
int withdraw_records_ok(og_ctx* ctx, int depth, const uint8_t* inputs_d, size_t n, size_t base) {
  if (n == 0) return OG_OK;
  uint32_t* bad_d = nullptr;
  OG_TRY(arena_get(ctx, "wd.bad", n * 4, (void**)&bad_d));
  OG_HIP(hipMemsetAsync(bad_d, 0xff, n * 4, ctx->stream));
  OG_TRY(withdraw_check_records(ctx, depth, inputs_d, n, bad_d));
  std::vector<uint32_t> b(n);
  OG_HIP(hipMemcpyAsync(b.data(), bad_d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t g = 0; g < n; g++)
    if (b[g] != 0xffffffffu) {
      set_error("og_withdraw: input record " + std::to_string(base + g) + ": field " + std::to_string(b[g]) + " (" + withdraw_field_name(b[g]) +
                (b[g] >= 8 ? " " + std::to_string(b[g] - 8) : std::string()) + ") is not a canonical value (>= r, or an index outside the tree)");
      return OG_ERR_INVALID;
    }
  return OG_OK;
}

// inputs (withdraw circuit records) -> proofs: witness generation fused into the lanes
int withdraw_prove_batch(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, const uint8_t*, uint8_t*, uint8_t*);
// witnesses are generated inside the pipeline from this many wire values per sub-batch on (OG_GEN_MIN: test hook)
static size_t gen_threshold() { return (size_t)OG_HOOK_INT("OG_GEN_MIN", (long long)1 << 26); }
// Witnesses inside prove_enqueue (on stream 0, the queries' streams waiting for an event) instead of a slab up front with the host
// waiting for it: for big statements (a sub-batch proves for hundreds of ms) -- and for every call that fits ONE sub-batch, whose
// schedule is the fan-out: there the host's wait was the walk's whole length (3.4 ms of one request), and the ~80 launches of the
// queries were issued only after it -- 0.2-0.4 ms of launch latency in front of every stream.  Inside, they are queued while the
// walk runs.  (OG_GEN_ONE_SUB=0: the slab form, hooks builds.)
static bool gen_inside(const og_ctx* ctx, const og_pk* pk, size_t n, size_t sb) {
  (void)ctx;
  return sb * pk->m >= gen_threshold() || (n <= sb && OG_HOOK_INT("OG_GEN_ONE_SUB", 1) != 0);
}

int job_wait(og_job* job) { return prove_finish(job, nullptr); }

// the events that mark the end of the job's work on each stream (none for a job that completed inside its submit call)
int job_done_events(og_job* job, hipEvent_t* out) {
  if (job->call_slot < 0) return 0;
  for (int k = 0; k < job->n_done; k++) out[k] = job->done[k];
  return job->n_done;
}

// is `job` a handle this context handed out and has not yet consumed?  (pointer comparison only: never dereferences it)
bool job_is_live(og_ctx* ctx, og_job* job) {
  if (job == nullptr) return false;
  if (ctx->jobs[0] == job || ctx->jobs[1] == job) return true;
  return std::find(ctx->done_jobs.begin(), ctx->done_jobs.end(), job) != ctx->done_jobs.end();
}

// one waiter per job (capi.hip og_job_wait): set / clear the flag; returns the PREVIOUS state; *id_out = the job's serial number
bool job_mark_waiting(og_job* job, bool on, uint64_t* id_out) {
  const bool was = job->waiting;
  if (!(on && was)) job->waiting = on;
  if (id_out) *id_out = job->id;
  return was;
}
bool job_same(og_job* job, uint64_t id) { return job->id == id; }
bool job_is_waited_for(og_job* job) { return job->waiting; }
bool ctx_has_waiters(og_ctx* ctx) {
  for (og_job* j : {ctx->jobs[0], ctx->jobs[1]})
    if (j && j->waiting) return true;
  for (og_job* j : ctx->done_jobs)
    if (j && j->waiting) return true;
  return false;
}

// og_job_abandon: the caller no longer wants the results (its buffers may be gone).  Waits until the job's last kernels are
// done -- they write device scratch the next call reuses -- copies nothing out, frees the call slot and the handle.
int job_abandon(og_job* job) {
  og_ctx* ctx = job->ctx;
  if (job->call_slot < 0) {
    auto& dj = ctx->done_jobs;
    dj.erase(std::remove(dj.begin(), dj.end(), job), dj.end());
    delete job;
    return OG_OK;
  }
  hipError_t first = hipSuccess;
  for (int k = 0; k < job->n_done; k++) {
    const hipError_t e = hipEventSynchronize(job->done[k]);
    if (e != hipSuccess && first == hipSuccess) first = e;
    (void)hipEventDestroy(job->done[k]);
  }
  if (ctx->jobs[job->call_slot] == job) ctx->jobs[job->call_slot] = nullptr;
  delete job;
  OG_HIP(first);
  return OG_OK;
}

// Enqueue-only form of withdraw_prove_batch: returns a job whose results og_job_wait delivers.  Circuits whose witnesses are
// generated inside the pipeline (the large ones) are really left running; for small circuits (whole-slab witness
// generation up front, shared staging) the call completes here and the job only carries the status.
int withdraw_prove_batch_submit(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                                size_t n, const uint8_t* rs, uint8_t* proofs, uint8_t* pub_out, og_job** job_out) {
  *job_out = nullptr;
  uint64_t shp[3];
  OG_TRY(withdraw_shape_query(depth, n_pad3, n_pad2, shp));
  OG_REQUIRE(shp[0] == pk->m && shp[2] == pk->n_pub, "og_withdraw_prove_batch_submit_d: the key is not for this withdraw-circuit shape");
  OG_REQUIRE(n >= 1, "og_withdraw_prove_batch_submit_d: empty batch");
  const size_t sb = (size_t)choose_sub_batch(ctx, pk, n);
  if (gen_inside(ctx, pk, n, sb)) {
    WithdrawGen gen{depth, n_pad3, n_pad2, inputs_d};
    return prove_enqueue(ctx, pk, nullptr, n, rs, proofs, &gen, pub_out, job_out);
  }
  OG_TRY(withdraw_prove_batch(ctx, pk, depth, n_pad3, n_pad2, inputs_d, n, rs, proofs, pub_out));
  og_job* job = new og_job();  // nothing left to wait for
  job->id = next_job_id();
  job->ctx = ctx;
  job->call_slot = -1;
  ctx->done_jobs.push_back(job);
  *job_out = job;
  return OG_OK;
}

int withdraw_prove_batch(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d, size_t n,
                         const uint8_t* rs, uint8_t* proofs, uint8_t* pub_out) {
  uint64_t shp[3];
  OG_TRY(withdraw_shape_query(depth, n_pad3, n_pad2, shp));
  OG_REQUIRE(shp[0] == pk->m && shp[2] == pk->n_pub, "og_withdraw_prove_batch_d: the key is not for this withdraw-circuit shape");
  // The witness generator is latency-bound (one lane walks a proof's 35 hashes: ~30 ms whatever the launch size).
  // Big circuits hide that inside the lanes (a sub-batch proves for hundreds of ms); for small circuits a
  // sub-batch proves in about the same time, so generate whole slabs of witnesses in ONE launch up front instead.
  const size_t sb = (size_t)choose_sub_batch(ctx, pk, n);
  if (gen_inside(ctx, pk, n, sb)) {
    WithdrawGen gen{depth, n_pad3, n_pad2, inputs_d};
    return prove_batch_impl(ctx, pk, nullptr, n, rs, proofs, nullptr, &gen, pub_out);
  }
  const size_t slab = std::max<size_t>(1, std::min<size_t>(n, ((size_t)16 << 30) / (pk->m * 32)));
  uint8_t* z_d = nullptr;
  OG_TRY(arena_get(ctx, "g16.zall", std::min(slab, (size_t)65535) * pk->m * 32, (void**)&z_d));
  for (size_t g0 = 0; g0 < n;) {
    const size_t cnt = std::min(std::min(slab, (size_t)65535), n - g0);
    OG_TRY(withdraw_records_ok(ctx, depth, inputs_d + g0 * (size_t)(8 + depth) * 32, cnt, g0));
    OG_TRY(withdraw_witness(ctx, depth, n_pad3, n_pad2, inputs_d + g0 * (size_t)(8 + depth) * 32, cnt, z_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));  // both lanes read the slab
    size_t bad = 0;
    int r = prove_batch_impl(ctx, pk, z_d, cnt, rs + g0 * 64, proofs + g0 * 256, &bad, nullptr,
                             pub_out ? pub_out + g0 * pk->n_pub * 32 : nullptr, true);  // (our own generator's wires are canonical)
    if (r == OG_ERR_UNSATISFIED)
      set_error("og_prove: witness " + std::to_string(g0 + bad) + " does not satisfy the circuit (a row has a*b != c, or wire 0 is not 1)");
    if (r != OG_OK) return r;
    g0 += cnt;
  }
  return OG_OK;
}

// deposit records (nullifier | secret | depositor) -> proofs of the deposit statement (witness.hip, oracle/py/deposit.py): a small
// circuit, so whole slabs of witnesses in one launch, then the ordinary batched prover
int deposit_shape_query(uint64_t out[3]);
int deposit_records_ok(og_ctx*, const uint8_t*, size_t, size_t);
int deposit_witness(og_ctx*, const uint8_t*, size_t, uint8_t*);
int deposit_prove_batch(og_ctx* ctx, const og_pk* pk, const uint8_t* inputs_d, size_t n, const uint8_t* rs, uint8_t* proofs, uint8_t* pub_out) {
  uint64_t shp[3];
  OG_TRY(deposit_shape_query(shp));
  OG_REQUIRE(shp[0] == pk->m && shp[2] == pk->n_pub, "og_deposit_prove_batch_d: the key is not for the deposit statement");
  uint8_t* z_d = nullptr;
  const size_t slab = std::min<size_t>(n, 65535);
  OG_TRY(arena_get(ctx, "g16.zall", slab * pk->m * 32, (void**)&z_d));
  for (size_t g0 = 0; g0 < n;) {
    const size_t cnt = std::min(slab, n - g0);
    OG_TRY(deposit_records_ok(ctx, inputs_d + g0 * 96, cnt, g0));
    OG_TRY(deposit_witness(ctx, inputs_d + g0 * 96, cnt, z_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));  // both lanes read the slab
    size_t bad = 0;
    int r = prove_batch_impl(ctx, pk, z_d, cnt, rs + g0 * 64, proofs + g0 * 256, &bad, nullptr, pub_out ? pub_out + g0 * pk->n_pub * 32 : nullptr, true);
    if (r == OG_ERR_UNSATISFIED)
      set_error("og_prove: witness " + std::to_string(g0 + bad) + " does not satisfy the circuit (a row has a*b != c, or wire 0 is not 1)");
    if (r != OG_OK) return r;
    g0 += cnt;
  }
  return OG_OK;
}

// ---- window-sharded proving: the two halves (WinShard above) -------------------------------------------------------------------
// Front half, enqueue only: this rank's partial sums of the five queries of n proofs into partials_d (n x PARTIAL_BYTES, device,
// the caller's), everything left running on the context's streams; the job carries the boundary flags and the public inputs
// (og_job_wait semantics: prove_finish reports a malformed record / an unsatisfied witness exactly as the unsharded call does).
int withdraw_prove_partials_enqueue(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                                    size_t n, int win_rank, int win_world, uint8_t* partials_d, uint8_t* pub_out, og_job** job_out) {
  *job_out = nullptr;
  uint64_t shp[3];
  OG_TRY(withdraw_shape_query(depth, n_pad3, n_pad2, shp));
  OG_REQUIRE(shp[0] == pk->m && shp[2] == pk->n_pub, "og_withdraw_prove_partials: the key is not for this withdraw-circuit shape");
  OG_REQUIRE(n >= 1, "og_withdraw_prove_partials: empty batch");
  OG_REQUIRE(win_world >= 1 && win_world <= pk->a->nwin && win_rank >= 0 && win_rank < win_world,
             "og_withdraw_prove_partials: bad window shard (rank, world): at most one rank per window");
  const WinShard sh{win_rank, win_world, partials_d};
  const size_t sb = (size_t)choose_sub_batch(ctx, pk, n);
  if (gen_inside(ctx, pk, n, sb)) {
    WithdrawGen gen{depth, n_pad3, n_pad2, inputs_d};
    return prove_enqueue(ctx, pk, nullptr, n, nullptr, nullptr, &gen, pub_out, job_out, nullptr, false, &sh);
  }
  // small statements: the whole call's witnesses in one launch up front (withdraw_prove_batch); a sharded call is one slab
  const size_t slab = std::min<size_t>(65535, std::max<size_t>(1, ((size_t)16 << 30) / (pk->m * 32)));
  OG_REQUIRE(n <= slab, "og_withdraw_prove_partials: at most " + std::to_string(slab) + " proofs of this statement per window-sharded call");
  uint8_t* z_d = nullptr;
  OG_TRY(arena_get(ctx, "g16.zall", n * pk->m * 32, (void**)&z_d));
  OG_TRY(withdraw_records_ok(ctx, depth, inputs_d, n, 0));
  OG_TRY(withdraw_witness(ctx, depth, n_pad3, n_pad2, inputs_d, n, z_d));
  OG_HIP(hipStreamSynchronize(ctx->stream));  // every stream of the call reads the slab
  return prove_enqueue(ctx, pk, z_d, n, nullptr, nullptr, nullptr, pub_out, job_out, nullptr, true, &sh);
}

// the same for caller-supplied witnesses (device, n x m x 32 B canonical)
int prove_partials_enqueue(og_ctx* ctx, const og_pk* pk, const uint8_t* z_d, size_t n, int win_rank, int win_world, uint8_t* partials_d,
                           og_job** job_out) {
  *job_out = nullptr;
  OG_REQUIRE(n >= 1, "og_prove_partials: empty batch");
  OG_REQUIRE(win_world >= 1 && win_world <= pk->a->nwin && win_rank >= 0 && win_rank < win_world,
             "og_prove_partials: bad window shard (rank, world): at most one rank per window");
  const WinShard sh{win_rank, win_world, partials_d};
  return prove_enqueue(ctx, pk, z_d, n, nullptr, nullptr, nullptr, nullptr, job_out, nullptr, false, &sh);
}

// every stream of the job's call -> `st` waits for it (the in-library all-gather is issued behind the front half, stream-ordered)
int job_join_stream(og_job* job, hipStream_t st) {
  if (job->call_slot < 0) return OG_OK;
  for (int k = 0; k < job->n_done; k++) OG_HIP(hipStreamWaitEvent(st, job->done[k], 0));
  return OG_OK;
}

// Back half: gathered_d = `world` blocks of n x PARTIAL_BYTES (rank-major, every rank's partials_d as the all-gather left them).
// Adds the ranks' shares query by query, assembles the n proofs with the blinding (r, s) and copies them out.  Enqueued on
// lanes[0] behind whatever the caller ordered there (the all-gather); blocking.
int prove_from_partials(og_ctx* ctx, const og_pk* pk, const uint8_t* gathered_d, int world, size_t n, const uint8_t* rs, uint8_t* proofs) {
  OG_REQUIRE(world >= 1 && n >= 1 && n <= 65535, "og_prove_from_partials: bad world / batch");
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  uint8_t *res[5], *rs_d, *proofs_d, *asm_tmp, *glv_d = nullptr;
  const size_t rank_stride = n * PARTIAL_BYTES;
  const size_t off[5] = {0, n * 128, 4 * n * 128, 2 * n * 128, 3 * n * 128};  // A | B1 | (B2 last) | L | H: the order of `res`
  if (world == 1) {
    for (int k = 0; k < 5; k++) res[k] = const_cast<uint8_t*>(gathered_d) + off[k];
  } else {
    ProfScope ps(ctx, PROF_REDUCE_G1, 0.0);
    for (int k = 0; k < 5; k++) {
      OG_TRY(arena_get(ctx, ("g16.sh.res" + std::to_string(k)).c_str(), n * (k == 2 ? 256 : 128), (void**)&res[k]));
      OG_TRY(msm_sum_ranks(ctx, k == 2, gathered_d + off[k], rank_stride, world, (int)n, res[k]));
    }
  }
  std::vector<uint8_t> glv_h;
  glv_halves(rs, n, glv_h);
  const size_t asm_lanes = glv_h.empty() ? 4 : 8;
  OG_TRY(arena_get(ctx, "g16.sh.rs", n * 64, (void**)&rs_d));
  OG_TRY(arena_get(ctx, "g16.sh.proofs", n * 256, (void**)&proofs_d));
  OG_TRY(arena_get(ctx, "g16.sh.asm", n * asm_lanes * 128 * 17, (void**)&asm_tmp));
  if (!glv_h.empty()) OG_TRY(arena_get(ctx, "g16.sh.glv", n * 128, (void**)&glv_d));
  OG_HIP(hipMemcpyAsync(rs_d, rs, n * 64, hipMemcpyHostToDevice, ctx->stream));
  if (glv_d) OG_HIP(hipMemcpyAsync(glv_d, glv_h.data(), n * 128, hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps_asm(ctx, PROF_ASSEMBLE, (double)n);
    OG_TRY(assemble_g1(ctx, pk->consts1, rs_d, res[0], res[1], res[3], res[4], n, asm_tmp, proofs_d, glv_d));
    OG_TRY(assemble_g2(ctx, pk->consts2, pk->fb_delta2, rs_d, res[2], n, proofs_d, n <= 1024));  // (a waited-for tail: a wave per proof)
  }
  OG_HIP(hipMemcpyAsync(proofs, proofs_d, n * 256, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));  // (glv_h, pageable rs: the host buffers of the async copies outlive them)
  return OG_OK;
}

}  // namespace og
