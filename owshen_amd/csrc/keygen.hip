// Key material through the C ABI: the withdraw circuit as an R1CS, and Groth16 key generation from
// explicit toxic waste, so that a Rust host needs nothing but libowshen_gpu.so to go from "no key" to proofs.
//
// No reference counterpart (the snapshot's withdraw carries no circuit,
// /root/reference/src/services/api_services/withdraw.rs:27-71).  The statement, its wire order and its row order are
// specified by oracle/py/withdraw.py; owshen_amd/circuit.py holds the same builder on the Python side and the tests
// check all three against each other row by row.  Wire layout is a contract with witness.hip (which fills the wires).
//
//   og_withdraw_r1cs   constraint rows of the depth-D MiMC7 Merkle withdraw circuit (+ synthetic padding gates, + the two
//                      optional density rows), CSR, canonical 32-byte coefficients
//   og_r1cs_from_csr   any R1CS the caller built
//   og_setup           Groth16 key generation from (tau, alpha, beta, gamma, delta): Lagrange evaluations, the three
//                      transposed sparse products, the query scalars and ~3 m + d fixed-base multiplications, all on the
//                      GPU; returns the "OWPK0001" / "OWVK0001" blobs of include/owshen_gpu.h
#include "ctx.h"
#include "msm.hip.h"
#include "field.hip.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <memory>


namespace og {

int lagrange_evals(og_ctx*, int, const uint8_t*, uint8_t*);
int spmv_canonical(og_ctx*, const uint32_t*, const uint32_t*, const uint8_t*, size_t, const uint8_t*, uint8_t*);
int scalar_mul_fixed(og_ctx*, int, const uint8_t*, const uint8_t*, size_t, uint8_t*);

// ---- tiny host-side Fr (only what building linear combinations needs: add, small constants, -1) -----------------
struct HFr {
  uint64_t v[4];
};
static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static HFr hfr_u64(uint64_t x) { return {{x, 0, 0, 0}}; }
static bool hfr_is_zero(const HFr& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static bool hfr_ge_mod(const HFr& a) {
  for (int w = 3; w >= 0; w--)
    if (a.v[w] != FR_MOD[w]) return a.v[w] > FR_MOD[w];
  return true;
}
static HFr hfr_add(const HFr& a, const HFr& b) {  // a, b < r
  HFr r;
  unsigned __int128 c = 0;
  for (int w = 0; w < 4; w++) {
    c += (unsigned __int128)a.v[w] + b.v[w];
    r.v[w] = (uint64_t)c;
    c >>= 64;
  }
  if (c || hfr_ge_mod(r)) {
    unsigned __int128 bw = 0;
    for (int w = 0; w < 4; w++) {
      unsigned __int128 d = (unsigned __int128)r.v[w] - FR_MOD[w] - (uint64_t)bw;
      r.v[w] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }
  return r;
}
static HFr hfr_neg_one() { return {{FR_MOD[0] - 1, FR_MOD[1], FR_MOD[2], FR_MOD[3]}}; }
static HFr hfr_from_bytes(const uint8_t* p) {
  HFr r;
  memcpy(r.v, p, 32);
  return r;
}

typedef std::vector<std::pair<uint32_t, HFr>> LC;  // (wire, coefficient), wires unique, coefficients non-zero

static LC lc_merge(std::initializer_list<const LC*> parts) {
  LC out;
  for (const LC* p : parts)
    for (const auto& t : *p) {
      auto it = std::find_if(out.begin(), out.end(), [&](const std::pair<uint32_t, HFr>& x) { return x.first == t.first; });
      if (it == out.end()) out.push_back(t); else it->second = hfr_add(it->second, t.second);
    }
  out.erase(std::remove_if(out.begin(), out.end(), [](const std::pair<uint32_t, HFr>& x) { return hfr_is_zero(x.second); }), out.end());
  return out;
}

struct R1csBuilder {
  og_r1cs* r;
  uint32_t next = 0;
  const uint8_t* consts;  // 91 x 32 B canonical
  uint32_t alloc(uint32_t n = 1) {
    uint32_t w = next;
    next += n;
    return w;
  }
  void row(int k, const LC& lc) {
    for (const auto& t : lc) {
      r->col[k].push_back(t.first);
      const uint8_t* b = reinterpret_cast<const uint8_t*>(t.second.v);
      r->val[k].insert(r->val[k].end(), b, b + 32);
    }
    r->ptr[k].push_back((uint32_t)r->col[k].size());
  }
  void enforce(const LC& a, const LC& b, const LC& c) {
    row(0, a);
    row(1, b);
    row(2, c);
    r->n_constraints++;
  }
  static LC one(uint32_t w) { return LC{{w, hfr_u64(1)}}; }
  // 91 rounds of x <- (x + k + c_i)^7, four constraints / four wires (t^2, t^4, t^6, t^7) per round
  LC perm(const LC& x_lc, const LC& k_lc) {
    LC cur = x_lc;
    for (int i = 0; i < 91; i++) {
      const LC ci{{0u, hfr_from_bytes(consts + 32 * i)}};
      const LC t = lc_merge({&cur, &k_lc, &ci});
      const uint32_t t2 = alloc(4), t4 = t2 + 1, t6 = t2 + 2, t7 = t2 + 3;
      enforce(t, t, one(t2));
      enforce(one(t2), one(t2), one(t4));
      enforce(one(t4), one(t2), one(t6));
      enforce(one(t6), t, one(t7));
      cur = one(t7);
    }
    return cur;
  }
  // MultiMiMC7([l, r], key 0): k1 = l + E_0(l); out = 2 k1 + r + E_k1(r) ... as two linear rows
  uint32_t hash2(const LC& l_lc, const LC& r_lc, int out_wire = -1) {
    const LC x91 = perm(l_lc, LC{});
    const uint32_t k1 = alloc();
    enforce(lc_merge({&l_lc, &x91}), one(0), one(k1));
    const LC y91 = perm(r_lc, one(k1));
    const uint32_t out = out_wire >= 0 ? (uint32_t)out_wire : alloc();
    const LC k2{{k1, hfr_u64(2)}};
    enforce(lc_merge({&k2, &r_lc, &y91}), one(0), one(out));
    return out;
  }
};

constexpr uint32_t KW_ROOT = 1, KW_NH = 2, KW_RECIPIENT = 3, KW_AMOUNT = 4, KW_TOKEN = 5, KW_CHAIN = 6, KW_NULLIFIER = 7, KW_SECRET = 8;
constexpr uint32_t K_PAD_SEGMENT = 64;

int withdraw_shape_query(int depth, uint64_t n_pad3, uint64_t n_pad2, uint64_t out[3]);

int withdraw_r1cs_build(const uint8_t* mimc_consts, int depth, uint64_t n_pad3, uint64_t n_pad2, int dense, og_r1cs* r) {
  uint64_t shp[3];
  OG_TRY(withdraw_shape_query(depth, n_pad3, n_pad2, shp));
  OG_REQUIRE(n_pad3 < (1ull << 30) && n_pad2 < (1ull << 30), "og_withdraw_r1cs: too many padding gates");
  r->n_wires = shp[0];
  r->n_pub = shp[2];
  for (int k = 0; k < 3; k++) r->ptr[k].assign(1, 0u);
  R1csBuilder b{r, 0, mimc_consts};
  const HFr m1 = hfr_neg_one();
  OG_REQUIRE(r->n_pub == 6, "og_withdraw_r1cs: the statement has six public inputs");
  b.alloc(1 + 6 + 2);
  const uint32_t w_sib = b.alloc(depth), w_bit = b.alloc(depth), w_rsq = b.alloc(), w_csq = b.alloc();
  b.enforce(R1csBuilder::one(KW_RECIPIENT), R1csBuilder::one(KW_RECIPIENT), R1csBuilder::one(w_rsq));
  b.enforce(R1csBuilder::one(KW_CHAIN), R1csBuilder::one(KW_CHAIN), R1csBuilder::one(w_csq));  // binds chain_id to the proof
  const uint32_t inner = b.hash2(R1csBuilder::one(KW_NULLIFIER), R1csBuilder::one(KW_SECRET));
  const uint32_t asset = b.hash2(R1csBuilder::one(KW_AMOUNT), R1csBuilder::one(KW_TOKEN));
  uint32_t cur = b.hash2(R1csBuilder::one(inner), R1csBuilder::one(asset));
  b.hash2(R1csBuilder::one(KW_NULLIFIER), LC{}, (int)KW_NH);
  for (int l = 0; l < depth; l++) {
    const uint32_t bit = w_bit + l, s = w_sib + l;
    b.enforce(R1csBuilder::one(bit), LC{{bit, hfr_u64(1)}, {0u, m1}}, LC{});              // bit (bit - 1) = 0
    const uint32_t left = b.alloc();
    b.enforce(R1csBuilder::one(bit), LC{{s, hfr_u64(1)}, {cur, m1}}, LC{{left, hfr_u64(1)}, {cur, m1}});  // left = cur + bit (s - cur)
    const LC right{{s, hfr_u64(1)}, {cur, hfr_u64(1)}, {left, m1}};
    cur = b.hash2(R1csBuilder::one(left), right, l == depth - 1 ? (int)KW_ROOT : -1);
  }
  const uint32_t pad_base = b.next;
  OG_REQUIRE((uint64_t)pad_base + 3 * n_pad3 + 2 * n_pad2 == r->n_wires, "og_withdraw_r1cs: internal wire count mismatch");
  for (uint64_t g = 0; g < n_pad3; g++) {  // (p + 1)(q + 2) = w
    const uint32_t p = pad_base + 3 * (uint32_t)g;
    b.enforce(LC{{p, hfr_u64(1)}, {0u, hfr_u64(1)}}, LC{{p + 1, hfr_u64(1)}, {0u, hfr_u64(2)}}, R1csBuilder::one(p + 2));
  }
  const uint32_t base2 = pad_base + 3 * (uint32_t)n_pad3;
  for (uint64_t g = 0; g < n_pad2; g++) {  // (p + 1)(prev + 2) = w; prev = the previous gate's w, or the constant 1 at a segment start
    const uint32_t p = base2 + 2 * (uint32_t)g, w = p + 1;
    const LC bb = (g % K_PAD_SEGMENT == 0) ? LC{{0u, hfr_u64(3)}} : LC{{w - 2, hfr_u64(1)}, {0u, hfr_u64(2)}};
    b.enforce(LC{{p, hfr_u64(1)}, {0u, hfr_u64(1)}}, bb, R1csBuilder::one(w));
  }
  if (dense) {  // (sum of all wires) * 0 = 0 and 0 * (sum of all wires) = 0: every wire gets an A and a B base
    LC all;
    all.reserve(r->n_wires);
    for (uint32_t w = 0; w < r->n_wires; w++) all.push_back({w, hfr_u64(1)});
    b.enforce(all, LC{}, LC{});
    b.enforce(LC{}, all, LC{});
  }
  OG_REQUIRE(r->n_constraints == shp[1] + (dense ? 2 : 0), "og_withdraw_r1cs: internal constraint count mismatch");
  return OG_OK;
}

// the deposit statement (oracle/py/deposit.py; witness.hip k_deposit_witness): 735 wires, 731 rows
int deposit_shape_query(uint64_t out[3]);
int deposit_r1cs_build(const uint8_t* mimc_consts, og_r1cs* r) {
  uint64_t shp[3];
  OG_TRY(deposit_shape_query(shp));
  r->n_wires = shp[0];
  r->n_pub = shp[2];
  for (int k = 0; k < 3; k++) r->ptr[k].assign(1, 0u);
  R1csBuilder b{r, 0, mimc_consts};
  constexpr uint32_t DW_COMMITMENT = 1, DW_DEPOSITOR = 2, DW_NULLIFIER = 3, DW_SECRET = 4;
  b.alloc(1 + 2 + 2);
  const uint32_t w_dsq = b.alloc();
  b.enforce(R1csBuilder::one(DW_DEPOSITOR), R1csBuilder::one(DW_DEPOSITOR), R1csBuilder::one(w_dsq));  // binds the depositor to the proof
  b.hash2(R1csBuilder::one(DW_NULLIFIER), R1csBuilder::one(DW_SECRET), (int)DW_COMMITMENT);
  OG_REQUIRE(b.next == r->n_wires && r->n_constraints == shp[1], "og_deposit_r1cs: internal shape mismatch");
  return OG_OK;
}

// ---- setup kernels ---------------------------------------------------------------------------------------------
// toxic: tau | alpha | beta | gamma | delta (canonical).  out: 0 zt = tau^d - 1, 1 1/gamma, 2 1/delta, 3 zt/delta
__global__ void k_setup_scalars(const uint8_t* __restrict__ toxic, int log_d, uint8_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Fr tau = fe_to_mont(fe_load<FrParams>(toxic));
  Fr zt = tau;
  for (int i = 0; i < log_d; i++) zt = fe_sqr(zt);
  zt = fe_sub(zt, Fr::one());
  const Fr ginv = fe_inv(fe_to_mont(fe_load<FrParams>(toxic + 96)));
  const Fr dinv = fe_inv(fe_to_mont(fe_load<FrParams>(toxic + 128)));
  fe_store(out, fe_from_mont(zt));
  fe_store(out + 32, fe_from_mont(ginv));
  fe_store(out + 64, fe_from_mont(dinv));
  fe_store(out + 96, fe_from_mont(fe_mul(zt, dinv)));
}

// kk_i = beta a_i + alpha b_i + c_i;  i <= n_pub: ic[i] = kk_i / gamma, else l[i - n_pub - 1] = kk_i / delta  (all canonical)
__global__ void __launch_bounds__(256) k_setup_kk(const uint8_t* __restrict__ at_a, const uint8_t* __restrict__ at_b, const uint8_t* __restrict__ at_c,
                                                 const uint8_t* __restrict__ toxic, const uint8_t* __restrict__ derived, size_t m, size_t n_pub,
                                                 uint8_t* __restrict__ ic, uint8_t* __restrict__ l) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Fr alpha = fe_to_mont(fe_load<FrParams>(toxic + 32)), beta = fe_to_mont(fe_load<FrParams>(toxic + 64));
  const Fr a = fe_to_mont(fe_load<FrParams>(at_a + i * 32)), b = fe_to_mont(fe_load<FrParams>(at_b + i * 32));
  const Fr c = fe_to_mont(fe_load<FrParams>(at_c + i * 32));
  const Fr kk = fe_add(fe_add(fe_mul(beta, a), fe_mul(alpha, b)), c);
  if (i <= n_pub)
    fe_store(ic + i * 32, fe_from_mont(fe_mul(kk, fe_to_mont(fe_load<FrParams>(derived + 32)))));
  else
    fe_store(l + (i - n_pub - 1) * 32, fe_from_mont(fe_mul(kk, fe_to_mont(fe_load<FrParams>(derived + 64)))));
}

// out[j] = (zt / delta) tau^j, canonical
__global__ void __launch_bounds__(256) k_setup_h(const uint8_t* __restrict__ toxic, const uint8_t* __restrict__ derived, size_t n,
                                                uint8_t* __restrict__ out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  Fr b = fe_to_mont(fe_load<FrParams>(toxic));
  Fr acc = fe_to_mont(fe_load<FrParams>(derived + 96));
  for (size_t e = j; e; e >>= 1) {
    if (e & 1) acc = fe_mul(acc, b);
    b = fe_sqr(b);
  }
  fe_store(out + j * 32, fe_from_mont(acc));
}

static const uint8_t G1_GEN_BYTES[64] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// the EIP-197 G2 generator: x.c0 | x.c1 | y.c0 | y.c1, 32-byte little-endian each
static const uint8_t G2_GEN_BYTES[128] = {
    0xed, 0xf6, 0x92, 0xd9, 0x5c, 0xbd, 0xde, 0x46, 0xdd, 0xda, 0x5e, 0xf7, 0xd4, 0x22, 0x43, 0x67, 0x79, 0x44, 0x5c, 0x5e, 0x66, 0x00,
    0x6a, 0x42, 0x76, 0x1e, 0x1f, 0x12, 0xef, 0xde, 0x00, 0x18, 0xc2, 0x12, 0xf3, 0xae, 0xb7, 0x85, 0xe4, 0x97, 0x12, 0xe7, 0xa9, 0x35,
    0x33, 0x49, 0xaa, 0xf1, 0x25, 0x5d, 0xfb, 0x31, 0xb7, 0xbf, 0x60, 0x72, 0x3a, 0x48, 0x0d, 0x92, 0x93, 0x93, 0x8e, 0x19, 0xaa, 0x7d,
    0xfa, 0x66, 0x01, 0xcc, 0xe6, 0x4c, 0x7b, 0xd3, 0x43, 0x0c, 0x69, 0xe7, 0xd1, 0xe3, 0x8f, 0x40, 0xcb, 0x8d, 0x80, 0x71, 0xab, 0x4a,
    0xeb, 0x6d, 0x8c, 0xdb, 0xa5, 0x5e, 0xc8, 0x12, 0x5b, 0x97, 0x22, 0xd1, 0xdc, 0xda, 0xac, 0x55, 0xf3, 0x8e, 0xb3, 0x70, 0x33, 0x31,
    0x4b, 0xbc, 0x95, 0x33, 0x0c, 0x69, 0xad, 0x99, 0x9e, 0xec, 0x75, 0xf0, 0x5f, 0x58, 0xd0, 0x89, 0x06, 0x09};

static inline size_t pad32(size_t n) { return (n + 31) / 32 * 32; }
static void put_padded(std::vector<uint8_t>& out, const void* p, size_t n) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  out.insert(out.end(), b, b + n);
  out.resize(out.size() + (pad32(n) - n), 0);  // every section is padded to a multiple of 32 B on its own
}

struct DevBuf {  // hipMalloc'd scratch released on every exit path
  std::vector<void*> ptrs;
  ~DevBuf() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  int get(size_t bytes, uint8_t** out) {
    void* p = nullptr;
    OG_HIP(hipMalloc(&p, bytes ? bytes : 32));
    ptrs.push_back(p);
    *out = static_cast<uint8_t*>(p);
    return OG_OK;
  }
};

int keygen_setup(og_ctx* ctx, const og_r1cs* r, const uint8_t toxic[160], std::vector<uint8_t>& pk, std::vector<uint8_t>& vk) {
  const size_t m = r->n_wires, l = r->n_pub, nc = r->n_constraints, n_rows = nc + l + 1;
  OG_REQUIRE(m >= 1 && l < m && m < (1ull << 31), "og_setup: bad wire counts");
  int log_d = 1;
  while (((size_t)1 << log_d) < n_rows) log_d++;
  OG_REQUIRE(log_d <= 28, "og_setup: too many constraints for the 2^28 domain");
  const size_t d = (size_t)1 << log_d, nl = m - l - 1, nh = d - 1;
  for (int k = 0; k < 5; k++) {
    const HFr t = hfr_from_bytes(toxic + 32 * k);
    OG_REQUIRE(!hfr_is_zero(t) && !hfr_ge_mod(t), "og_setup: toxic values must be canonical and non-zero");
  }
  for (int k = 0; k < 3; k++) {
    OG_REQUIRE(r->ptr[k].size() == nc + 1 && r->ptr[k][0] == 0 && r->ptr[k][nc] == r->col[k].size() && r->val[k].size() == r->col[k].size() * 32,
               "og_setup: malformed R1CS");
    for (uint32_t c : r->col[k]) OG_REQUIRE(c < m, "og_setup: R1CS column out of range");
  }
  DevBuf dev;
  uint8_t *toxic_d, *derived_d, *lag_d, *at_d[3], *ic_s, *l_s, *h_s;
  OG_TRY(dev.get(160, &toxic_d));
  OG_TRY(dev.get(128, &derived_d));
  OG_TRY(dev.get(d * 32, &lag_d));
  for (int k = 0; k < 3; k++) OG_TRY(dev.get(m * 32, &at_d[k]));
  OG_TRY(dev.get((l + 1) * 32, &ic_s));
  OG_TRY(dev.get(nl * 32, &l_s));
  OG_TRY(dev.get(nh * 32, &h_s));
  OG_HIP(hipMemcpyAsync(toxic_d, toxic, 160, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_setup_scalars, dim3(1), dim3(64), 0, ctx->stream, toxic_d, log_d, derived_d);
  OG_HIP(hipGetLastError());
  uint8_t derived[128];
  OG_HIP(hipMemcpyAsync(derived, derived_d, 128, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  OG_REQUIRE(!hfr_is_zero(hfr_from_bytes(derived)), "og_setup: tau lies in the evaluation domain");
  OG_TRY(lagrange_evals(ctx, log_d, toxic, lag_d));
  // a_i(tau) = sum_rows M[row][i] L_row(tau): the transposed sparse product (QAP rows = constraints, then the
  // input-consistency rows A = wire i for i <= n_pub)
  std::vector<uint32_t> full_ptr[3];
  std::vector<uint32_t> full_col[3];
  std::vector<uint8_t> full_val[3];
  uint8_t one32[32] = {1};
  for (int k = 0; k < 3; k++) {
    full_ptr[k] = r->ptr[k];
    full_col[k] = r->col[k];
    full_val[k] = r->val[k];
    for (size_t i = 0; i <= l; i++) {
      if (k == 0) {
        full_col[k].push_back((uint32_t)i);
        full_val[k].insert(full_val[k].end(), one32, one32 + 32);
      }
      full_ptr[k].push_back((uint32_t)full_col[k].size());
    }
    const size_t nnz = full_col[k].size();
    OG_REQUIRE(nnz < (1ull << 32), "og_setup: nnz too large");
    std::vector<uint32_t> tptr(m + 1, 0), tcol(nnz);
    std::vector<uint8_t> tval(nnz * 32);
    for (uint32_t c : full_col[k]) tptr[c + 1]++;
    for (size_t i = 0; i < m; i++) tptr[i + 1] += tptr[i];
    std::vector<uint32_t> cursor(tptr.begin(), tptr.end() - 1);
    for (size_t row = 0; row < n_rows; row++)
      for (uint32_t e = full_ptr[k][row]; e < full_ptr[k][row + 1]; e++) {
        const uint32_t pos = cursor[full_col[k][e]]++;
        tcol[pos] = (uint32_t)row;
        memcpy(&tval[(size_t)pos * 32], &full_val[k][(size_t)e * 32], 32);
      }
    uint8_t *p_d, *c_d, *v_d;
    OG_TRY(dev.get((m + 1) * 4, &p_d));
    OG_TRY(dev.get(nnz * 4, &c_d));
    OG_TRY(dev.get(nnz * 32, &v_d));
    OG_HIP(hipMemcpyAsync(p_d, tptr.data(), (m + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    OG_HIP(hipMemcpyAsync(c_d, tcol.data(), nnz * 4, hipMemcpyHostToDevice, ctx->stream));
    OG_HIP(hipMemcpyAsync(v_d, tval.data(), nnz * 32, hipMemcpyHostToDevice, ctx->stream));
    OG_TRY(spmv_canonical(ctx, (const uint32_t*)p_d, (const uint32_t*)c_d, v_d, m, lag_d, at_d[k]));
    OG_HIP(hipStreamSynchronize(ctx->stream));  // the host staging vectors die at the end of this iteration
  }
  hipLaunchKernelGGL(k_setup_kk, dim3(grid_for(m, 256)), dim3(256), 0, ctx->stream, at_d[0], at_d[1], at_d[2], toxic_d, derived_d, m, l, ic_s, l_s);
  OG_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_setup_h, dim3(grid_for(nh, 256)), dim3(256), 0, ctx->stream, toxic_d, derived_d, nh, h_s);
  OG_HIP(hipGetLastError());
  // fixed-base multiplications: the five queries, IC, and the constants (alpha, beta, gamma, delta) in both groups
  struct Job { int is_g2; const uint8_t* sc; size_t n; std::vector<uint8_t> out; };
  Job jobs[8] = {{0, at_d[0], m, {}}, {0, at_d[1], m, {}}, {1, at_d[1], m, {}}, {0, l_s, nl, {}}, {0, h_s, nh, {}},
                 {0, ic_s, l + 1, {}}, {0, toxic_d + 32, 4, {}}, {1, toxic_d + 32, 4, {}}};
  uint8_t* pts_d;
  OG_TRY(dev.get(std::max<size_t>(std::max<size_t>(m * 128, nh * 64), 512), &pts_d));
  for (Job& j : jobs) {
    const size_t pb = j.is_g2 ? 128 : 64;
    j.out.resize(j.n * pb);
    if (j.n == 0) continue;
    OG_TRY(scalar_mul_fixed(ctx, j.is_g2, j.is_g2 ? G2_GEN_BYTES : G1_GEN_BYTES, j.sc, j.n, pts_d));  // synchronises
    OG_HIP(hipMemcpyAsync(j.out.data(), pts_d, j.n * pb, hipMemcpyDeviceToHost, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
  }
  const std::vector<uint8_t>&c1 = jobs[6].out, &c2 = jobs[7].out;  // index 0 alpha, 1 beta, 2 gamma, 3 delta
  // ---- "OWPK0001"
  pk.clear();
  const uint64_t head[10] = {0x313030304b50574full, m, l, (uint64_t)log_d, n_rows, full_col[0].size(), full_col[1].size(), full_col[2].size(), 0, 0};
  pk.insert(pk.end(), (const uint8_t*)head, (const uint8_t*)head + 80);  // the 80-byte header is not padded
  put_padded(pk, &c1[0], 64);        // alpha_g1
  put_padded(pk, &c1[64], 64);       // beta_g1
  put_padded(pk, &c1[192], 64);      // delta_g1
  pk.resize(pk.size() + 64, 0);
  put_padded(pk, &c2[128], 128);     // beta_g2
  put_padded(pk, &c2[384], 128);     // delta_g2
  for (int k = 0; k < 3; k++) {
    put_padded(pk, full_ptr[k].data(), full_ptr[k].size() * 4);
    put_padded(pk, full_col[k].data(), full_col[k].size() * 4);
    put_padded(pk, full_val[k].data(), full_val[k].size());
  }
  for (int q = 0; q < 5; q++) put_padded(pk, jobs[q].out.data(), jobs[q].out.size());
  // ---- "OWVK0001" | n_pub | alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | IC
  vk.clear();
  vk.insert(vk.end(), (const uint8_t*)"OWVK0001", (const uint8_t*)"OWVK0001" + 8);
  const uint64_t npub = l;
  vk.insert(vk.end(), (const uint8_t*)&npub, (const uint8_t*)&npub + 8);
  vk.insert(vk.end(), &c1[0], &c1[0] + 64);
  vk.insert(vk.end(), &c2[128], &c2[128] + 128);
  vk.insert(vk.end(), &c2[256], &c2[256] + 128);
  vk.insert(vk.end(), &c2[384], &c2[384] + 128);
  vk.insert(vk.end(), jobs[5].out.begin(), jobs[5].out.end());
  return OG_OK;
}

}  // namespace og

using namespace og;

extern "C" {

int og_withdraw_r1cs(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, int dense, og_r1cs** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(ctx != nullptr && out != nullptr, "og_withdraw_r1cs: null argument");
    *out = nullptr;
    og_r1cs* r = new og_r1cs();
    int rc = withdraw_r1cs_build(ctx->mimc_consts_canon, depth, n_pad3, n_pad2, dense, r);
    if (rc != OG_OK) {
      delete r;
      return rc;
    }
    *out = r;
    return OG_OK;
  });
}

int og_deposit_r1cs(og_ctx* ctx, og_r1cs** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(ctx != nullptr && out != nullptr, "og_deposit_r1cs: null argument");
    *out = nullptr;
    og_r1cs* r = new og_r1cs();
    int rc = deposit_r1cs_build(ctx->mimc_consts_canon, r);
    if (rc != OG_OK) {
      delete r;
      return rc;
    }
    *out = r;
    return OG_OK;
  });
}

int og_r1cs_from_csr(uint64_t n_wires, uint64_t n_pub, uint64_t n_constraints, const uint32_t* const ptr[3], const uint32_t* const col[3],
                     const uint8_t* const val[3], og_r1cs** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(out != nullptr && ptr && col && val, "og_r1cs_from_csr: null argument");
    *out = nullptr;
    OG_REQUIRE(n_wires >= 1 && n_pub < n_wires && n_wires < (1ull << 31) && n_constraints < (1ull << 31), "og_r1cs_from_csr: bad sizes");
    std::unique_ptr<og_r1cs> r(new og_r1cs());
    r->n_wires = n_wires; r->n_pub = n_pub; r->n_constraints = n_constraints;
    for (int k = 0; k < 3; k++) {
      OG_REQUIRE(ptr[k] != nullptr, "og_r1cs_from_csr: null row pointers");
      OG_REQUIRE(ptr[k][0] == 0, "og_r1cs_from_csr: row pointers must start at 0");
      for (uint64_t i = 0; i < n_constraints; i++) OG_REQUIRE(ptr[k][i] <= ptr[k][i + 1], "og_r1cs_from_csr: row pointers not monotone");
      const size_t nnz = ptr[k][n_constraints];
      OG_REQUIRE(nnz == 0 || (col[k] && val[k]), "og_r1cs_from_csr: null columns / values");
      r->ptr[k].assign(ptr[k], ptr[k] + n_constraints + 1);
      r->col[k].assign(col[k], col[k] + nnz);
      r->val[k].assign(val[k], val[k] + nnz * 32);
      for (uint32_t c : r->col[k]) OG_REQUIRE(c < n_wires, "og_r1cs_from_csr: column out of range");
    }
    *out = r.release();
    return OG_OK;
  });
}

void og_r1cs_free(og_r1cs* r) { delete r; }

int og_r1cs_info(const og_r1cs* r, uint64_t info[6]) {
  return guarded([&]() -> int {
    OG_REQUIRE(r != nullptr && info != nullptr, "og_r1cs_info: null argument");
    info[0] = r->n_wires; info[1] = r->n_pub; info[2] = r->n_constraints;
    for (int k = 0; k < 3; k++) info[3 + k] = r->col[k].size();
    return OG_OK;
  });
}

int og_r1cs_export(const og_r1cs* r, int matrix, uint32_t* ptr_out, uint32_t* col_out, uint8_t* val_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(r != nullptr && matrix >= 0 && matrix < 3 && ptr_out != nullptr, "og_r1cs_export: bad arguments");
    memcpy(ptr_out, r->ptr[matrix].data(), r->ptr[matrix].size() * 4);
    if (!r->col[matrix].empty()) {
      OG_REQUIRE(col_out != nullptr && val_out != nullptr, "og_r1cs_export: null output");
      memcpy(col_out, r->col[matrix].data(), r->col[matrix].size() * 4);
      memcpy(val_out, r->val[matrix].data(), r->val[matrix].size());
    }
    return OG_OK;
  });
}

int og_setup(og_ctx* ctx, const og_r1cs* r1cs, const uint8_t toxic[160], uint8_t** pk_out, size_t* pk_len, uint8_t** vk_out, size_t* vk_len) {
  return guarded([&]() -> int {
    OG_REQUIRE(ctx && r1cs && toxic && pk_out && pk_len && vk_out && vk_len, "og_setup: null argument");
    *pk_out = *vk_out = nullptr;
    *pk_len = *vk_len = 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    OG_HIP(hipSetDevice(ctx->device));
    std::vector<uint8_t> pk, vk;
    OG_TRY(keygen_setup(ctx, r1cs, toxic, pk, vk));
    uint8_t* a = static_cast<uint8_t*>(malloc(pk.size() ? pk.size() : 1));
    uint8_t* b = static_cast<uint8_t*>(malloc(vk.size() ? vk.size() : 1));
    if (!a || !b) {
      free(a);
      free(b);
      set_error("og_setup: out of host memory");
      return OG_ERR_INVALID;
    }
    memcpy(a, pk.data(), pk.size());
    memcpy(b, vk.data(), vk.size());
    *pk_out = a; *pk_len = pk.size();
    *vk_out = b; *vk_len = vk.size();
    return OG_OK;
  });
}

void og_blob_free(uint8_t* blob) { free(blob); }

}  // extern "C"
