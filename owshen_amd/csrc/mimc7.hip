// Batched MiMC7 kernels: 2-to-1 hash, Merkle paths, full-tree build.
// One lane per hash; purely VALU-bound (728 mulmod per 2-to-1 hash vs 96 B of traffic).
#include "ctx.h"
#include "mimc7.hip.h"
#include "host_fr4.h"
#include <string.h>
#include <vector>

namespace og {

__global__ void __launch_bounds__(256) k_to_mont_fr(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store(out + i * 32, fe_to_mont(fe_load<FrParams>(in + i * 32)));
}

// PAIR: two lanes per hash (mimc7.hip.h: the latency-bound form); lanes 2i and 2i + 1 of a wave stay or leave together
template <bool PAIR>
__global__ void __launch_bounds__(256) k_mimc7_hash2(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ left,
                                                    const uint8_t* __restrict__ right, uint8_t* __restrict__ out, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = PAIR ? t >> 1 : t;
  const bool odd = PAIR && (threadIdx.x & 1);
  if (i >= n) return;
  Fr l = fe_to_mont(fe_load<FrParams>(left + i * 32));
  Fr r = fe_to_mont(fe_load<FrParams>(right + i * 32));
  const Fr h = fe_from_mont(mimc7_hash2<PAIR>(consts, l, r, odd));
  if (!odd) fe_store(out + i * 32, h);
}

// one tree level: out[i] = H(in[2i], in[2i+1]); canonical in/out
template <bool PAIR>
__global__ void __launch_bounds__(256) k_mimc7_tree_level(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ in,
                                                         uint8_t* __restrict__ out, size_t n_out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = PAIR ? t >> 1 : t;
  const bool odd = PAIR && (threadIdx.x & 1);
  if (i >= n_out) return;
  Fr l = fe_to_mont(fe_load<FrParams>(in + (2 * i) * 32));
  Fr r = fe_to_mont(fe_load<FrParams>(in + (2 * i + 1) * 32));
  const Fr h = fe_from_mont(mimc7_hash2<PAIR>(consts, l, r, odd));
  if (!odd) fe_store(out + i * 32, h);
}

// the wave-wide forms (mimc7.hip.h w9_mimc7_hash2): ONE hash per wave, for launches that leave most of the chip idle.
// ROWS: rounds three products deep over two rows of the wave -- what the launches use; the four-deep single-row form is
// instantiated in the hooks build only (OG_W9_ROWS=0: A/B, tests)
template <int FORM>
__global__ void __launch_bounds__(64) k_mimc7_hash2_w9(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ left,
                                                      const uint8_t* __restrict__ right, uint8_t* __restrict__ out, size_t n) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const Fr h = fe_from_mont(w9_mimc7_hash2<FORM>(consts9, fe_to_mont(fe_load<FrParams>(left + i * 32)), fe_to_mont(fe_load<FrParams>(right + i * 32)), threadIdx.x));
  if (threadIdx.x == 0) fe_store(out + i * 32, h);
}
template <int FORM>
__global__ void __launch_bounds__(64) k_mimc7_tree_level_w9(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ in,
                                                           uint8_t* __restrict__ out, size_t n_out) {
  const size_t i = blockIdx.x;
  if (i >= n_out) return;
  const Fr h = fe_from_mont(w9_mimc7_hash2<FORM>(consts9, fe_to_mont(fe_load<FrParams>(in + (2 * i) * 32)), fe_to_mont(fe_load<FrParams>(in + (2 * i + 1) * 32)), threadIdx.x));
  if (threadIdx.x == 0) fe_store(out + i * 32, h);
}
template <int FORM>
__global__ void __launch_bounds__(64) k_mimc7_merkle_paths_w9(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ leaves,
                                                             const uint64_t* __restrict__ indices, const uint8_t* __restrict__ siblings,
                                                             int depth, uint8_t* __restrict__ nodes, size_t n) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const bool first = threadIdx.x == 0;
  const uint64_t idx = indices[i];
  uint8_t* o = nodes + i * (size_t)(depth + 1) * 32;
  Fr cur = fe_load<FrParams>(leaves + i * 32);
  if (first) fe_store(o, cur);
  cur = fe_to_mont(cur);
#pragma unroll 1
  for (int l = 0; l < depth; l++) {
    const Fr sib = fe_to_mont(fe_load<FrParams>(siblings + (i * (size_t)depth + l) * 32));
    const bool right = (idx >> l) & 1;
    cur = w9_mimc7_hash2<FORM>(consts9, right ? sib : cur, right ? cur : sib, threadIdx.x);
    if (first) fe_store(o + (size_t)(l + 1) * 32, fe_from_mont(cur));
  }
}

template <bool PAIR>
__global__ void __launch_bounds__(64) k_mimc7_merkle_paths(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ leaves,
                                                          const uint64_t* __restrict__ indices, const uint8_t* __restrict__ siblings,
                                                          int depth, uint8_t* __restrict__ nodes, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i = PAIR ? t >> 1 : t;
  const bool odd = PAIR && (threadIdx.x & 1);
  if (i >= n) return;
  uint64_t idx = indices[i];
  uint8_t* o = nodes + i * (size_t)(depth + 1) * 32;
  Fr cur = fe_load<FrParams>(leaves + i * 32);
  if (!odd) fe_store(o, cur);
  cur = fe_to_mont(cur);
  for (int l = 0; l < depth; l++) {
    Fr sib = fe_to_mont(fe_load<FrParams>(siblings + (i * (size_t)depth + l) * 32));
    bool right = (idx >> l) & 1;
    Fr a = right ? sib : cur;
    Fr b = right ? cur : sib;
    cur = mimc7_hash2<PAIR>(consts, a, b, odd);
    if (!odd) fe_store(o + (size_t)(l + 1) * 32, fe_from_mont(cur));
  }
}

// zeros[h] = root of an all-zero subtree of height h (zeros[0] = the zero leaf): one lane pair, 64 sequential hashes
__global__ void k_mimc7_zero_hashes(const uint32_t* __restrict__ consts, uint8_t* __restrict__ zeros) {
  if (blockIdx.x != 0 || threadIdx.x >= 2) return;
  const bool odd = threadIdx.x & 1;
  Fr cur = Fr::zero();
  if (!odd) fe_store(zeros, cur);
#pragma unroll 1
  for (int h = 1; h <= 64; h++) {
    cur = mimc7_hash2<true>(consts, cur, cur, odd);
    if (!odd) fe_store(zeros + (size_t)h * 32, fe_from_mont(cur));
  }
}

// One level of a batched append to an incremental tree (the commitment tree `mint_tx` would feed; spec:
// oracle/py/mimc7.py IncrementalTree).  `run` holds the level-`lvl` nodes at positions [a, b); thread t builds the
// parent at position (a >> 1) + t from its children, taking the left child from the frontier when it precedes the
// run (a odd) and the right child from the zero-subtree table when it lies beyond it.  Thread 0 also records the
// new frontier entry of this level: the completed node at position (n_total >> lvl) - 1, when that bit of
// n_total is set (other entries are never read before being rewritten).
__global__ void __launch_bounds__(64) k_mimc7_append_level(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ run,
                                                          uint64_t a, uint64_t b, int lvl, const uint8_t* __restrict__ frontier_in,
                                                          const uint8_t* __restrict__ zeros, uint64_t n_total,
                                                          uint8_t* __restrict__ frontier_out, uint8_t* __restrict__ out) {
  const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, t = lane >> 1;  // a lane pair per parent
  const bool odd = lane & 1;
  const uint64_t p0 = a >> 1, n_par = ((b - 1) >> 1) - p0 + 1;
  if (lane == 0) {
    Fr f = fe_load<FrParams>(frontier_in + (size_t)lvl * 32);
    if ((n_total >> lvl) & 1) {
      const uint64_t q = (n_total >> lvl) - 1;
      if (q >= a) f = fe_load<FrParams>(run + (size_t)(q - a) * 32);
    }
    fe_store(frontier_out + (size_t)lvl * 32, f);
  }
  if (t >= n_par) return;
  const uint64_t p = p0 + t, lc = 2 * p, rc = 2 * p + 1;
  const Fr l = fe_to_mont(fe_load<FrParams>(lc >= a ? run + (size_t)(lc - a) * 32 : frontier_in + (size_t)lvl * 32));
  const Fr r = fe_to_mont(fe_load<FrParams>(rc < b ? run + (size_t)(rc - a) * 32 : zeros + (size_t)lvl * 32));
  const Fr h = fe_from_mont(mimc7_hash2<true>(consts, l, r, odd));
  if (!odd) fe_store(out + (size_t)t * 32, h);
}

// The same append on the HOST (og_set_host_chains: calls of a handful of leaves).  One leaf is a chain of `depth` dependent
// two-to-one hashes -- 32 x 728 products, 8.3 ms as 32 launches of a lone lane pair, ~0.4 ms on a core (host_fr4.h) -- and one
// leaf per call is what `mint_tx` (/root/reference/src/blockchain/tx/mint_tx.rs:11-49) appends.  Level by level exactly as
// k_mimc7_append_level: same children (frontier on the left, zero subtree on the right), same frontier entries, same root.
static int mimc7_append_host(og_ctx* ctx, int depth, const uint8_t* frontier_in_d, uint64_t next_index, const uint8_t* leaves_d, size_t k,
                             uint8_t* frontier_out_d, uint8_t* root_out_d) {
  const H4Field& f = h4_field();
  const std::vector<H4> rc = h4_round_constants(ctx->mimc_consts_canon);
  if (ctx->mimc_zeros_h.empty()) {  // roots of all-zero subtrees of height 0..64, the host's Montgomery form (zero is zero)
    ctx->mimc_zeros_h.assign(65 * 32, 0);
    H4 cur = {{0, 0, 0, 0}};
    for (int h = 1; h <= 64; h++) {
      cur = h4_mimc7_hash2(f, rc.data(), cur, cur);
      memcpy(ctx->mimc_zeros_h.data() + (size_t)h * 32, cur.v, 32);
    }
  }
  std::vector<uint8_t> fin((size_t)depth * 32), fout((size_t)depth * 32), lv(k * 32);
  OG_HIP(hipMemcpyAsync(fin.data(), frontier_in_d, fin.size(), hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipMemcpyAsync(lv.data(), leaves_d, lv.size(), hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<H4> run(k), nxt;
  for (size_t i = 0; i < k; i++) run[i] = h4_to_mont(f, h4_load(lv.data() + i * 32));
  uint64_t a = next_index, b = next_index + k;
  const uint64_t n_total = next_index + k;
  H4 root = {{0, 0, 0, 0}};
  for (int lvl = 0; lvl < depth; lvl++) {
    // the new frontier entry of this level: the completed node at position (n_total >> lvl) - 1, when that bit of n_total is set
    const uint8_t* fi = fin.data() + (size_t)lvl * 32;
    memcpy(fout.data() + (size_t)lvl * 32, fi, 32);
    if ((n_total >> lvl) & 1) {
      const uint64_t q = (n_total >> lvl) - 1;
      if (q >= a) { const H4 c = h4_from_mont(f, run[q - a]); memcpy(fout.data() + (size_t)lvl * 32, c.v, 32); }
    }
    const uint64_t p0 = a >> 1, n_par = ((b - 1) >> 1) - p0 + 1;
    nxt.resize(n_par);
    H4 zr;
    memcpy(zr.v, ctx->mimc_zeros_h.data() + (size_t)lvl * 32, 32);
    host_parallel_for((size_t)n_par, [&](size_t t) {
      const uint64_t p = p0 + t, lc = 2 * p, rc_ = 2 * p + 1;
      const H4 l = lc >= a ? run[lc - a] : h4_to_mont(f, h4_load(fi));
      const H4 r = rc_ < b ? run[rc_ - a] : zr;
      nxt[t] = h4_mimc7_hash2(f, rc.data(), l, r);
    });
    run.swap(nxt);
    b = ((b - 1) >> 1) + 1;
    a >>= 1;
    if (lvl == depth - 1) root = h4_from_mont(f, run[0]);
  }
  OG_HIP(hipMemcpyAsync(frontier_out_d, fout.data(), fout.size(), hipMemcpyHostToDevice, ctx->stream));
  OG_HIP(hipMemcpyAsync(root_out_d, root.v, 32, hipMemcpyHostToDevice, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));  // (the pageable host buffers of the two copies)
  return OG_OK;
}

// the same level with a WAVE per parent (appends of a few leaves: the chain of `depth` hashes is the whole call)
template <int FORM>
__global__ void __launch_bounds__(64) k_mimc7_append_level_w9(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ run,
                                                             uint64_t a, uint64_t b, int lvl, const uint8_t* __restrict__ frontier_in,
                                                             const uint8_t* __restrict__ zeros, uint64_t n_total,
                                                             uint8_t* __restrict__ frontier_out, uint8_t* __restrict__ out) {
  const uint64_t t = blockIdx.x;
  const uint64_t p0 = a >> 1, n_par = ((b - 1) >> 1) - p0 + 1;
  if (t == 0 && threadIdx.x == 0) {
    Fr f = fe_load<FrParams>(frontier_in + (size_t)lvl * 32);
    if ((n_total >> lvl) & 1) {
      const uint64_t q = (n_total >> lvl) - 1;
      if (q >= a) f = fe_load<FrParams>(run + (size_t)(q - a) * 32);
    }
    fe_store(frontier_out + (size_t)lvl * 32, f);
  }
  if (t >= n_par) return;
  const uint64_t p = p0 + t, lc = 2 * p, rc = 2 * p + 1;
  const Fr l = fe_to_mont(fe_load<FrParams>(lc >= a ? run + (size_t)(lc - a) * 32 : frontier_in + (size_t)lvl * 32));
  const Fr r = fe_to_mont(fe_load<FrParams>(rc < b ? run + (size_t)(rc - a) * 32 : zeros + (size_t)lvl * 32));
  const Fr h = fe_from_mont(w9_mimc7_hash2<FORM>(consts9, l, r, threadIdx.x));
  if (threadIdx.x == 0) fe_store(out + (size_t)t * 32, h);
}

// a wave per hash while the launch stays at two waves per SIMD (OG_MIMC_W9 = 0 | 1 forces either way in hooks builds; the crossover,
// 2^20-leaf tree: at most 1024 / 2048 / 4096 / 8192 hashes -> 7.35 / 7.27 / 7.35 / 7.73 ms, profiles/r06l_ab_w9_rows.txt)
static bool wave_per_hash(const og_ctx* ctx, size_t n_hashes) {
  if (const char* e = OG_HOOK_STR("OG_MIMC_W9")) return atoi(e) != 0;
  return n_hashes <= (size_t)OG_HOOK_INT("OG_MIMC_W9_MAX", ctx->n_cu * 8);  // (A/B: the crossover)
}

int mimc7_append(og_ctx* ctx, int depth, const uint8_t* frontier_in, uint64_t next_index, const uint8_t* leaves, size_t k,
                 uint8_t* frontier_out, uint8_t* root_out) {
  OG_REQUIRE(depth >= 1 && depth <= 63, "og_mimc7_append_d: depth must be 1..63");
  OG_REQUIRE(k >= 1, "og_mimc7_append_d: at least one leaf");
  OG_REQUIRE(next_index + k >= next_index && next_index + k <= ((uint64_t)1 << depth), "og_mimc7_append_d: the tree is full");
  if (ctx->host_chains_max > 0 && k <= (size_t)ctx->host_chains_max)
    return mimc7_append_host(ctx, depth, frontier_in, next_index, leaves, k, frontier_out, root_out);
  if (!ctx->mimc_zeros_d) {
    OG_HIP(hipMalloc((void**)&ctx->mimc_zeros_d, 65 * 32));
    hipLaunchKernelGGL(k_mimc7_zero_hashes, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, ctx->mimc_zeros_d);
    OG_HIP(hipGetLastError());
  }
  uint8_t* buf[2];
  const size_t cap = (k / 2 + 2) * 32;
  OG_HIP(hipMalloc((void**)&buf[0], cap));
  if (hipMalloc((void**)&buf[1], cap) != hipSuccess) {
    (void)hipFree(buf[0]);
    set_error("og_mimc7_append_d: out of device memory");
    return OG_ERR_HIP;
  }
  const uint8_t* run = leaves;
  uint64_t a = next_index, b = next_index + k;
  const uint64_t n_total = next_index + k;
  int rc = OG_OK;
  for (int lvl = 0; lvl < depth; lvl++) {
    const uint64_t n_par = ((b - 1) >> 1) - (a >> 1) + 1;
    uint8_t* out = lvl == depth - 1 ? root_out : buf[lvl & 1];
    if (wave_per_hash(ctx, n_par))
      OG_W9_LAUNCH(k_mimc7_append_level_w9, w9_rows(), dim3((unsigned)n_par), dim3(64), 0, ctx->stream,
                         (const uint32_t*)ctx->mimc_consts9_d, run, a, b, lvl, frontier_in, ctx->mimc_zeros_d, n_total, frontier_out, out);
    else
      hipLaunchKernelGGL(k_mimc7_append_level, dim3(grid_for(2 * n_par, 64)), dim3(64), 0, ctx->stream,
                         (const uint32_t*)ctx->mimc_consts_d, run, a, b, lvl, frontier_in, ctx->mimc_zeros_d, n_total, frontier_out, out);
    if (hipGetLastError() != hipSuccess) { rc = OG_ERR_HIP; set_error("og_mimc7_append_d: launch failed"); break; }
    run = out;
    b = ((b - 1) >> 1) + 1;
    a >>= 1;
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(buf[0]);
  (void)hipFree(buf[1]);
  if (rc == OG_OK && e != hipSuccess) { set_error(std::string("og_mimc7_append_d: ") + hipGetErrorString(e)); rc = OG_ERR_HIP; }
  return rc;
}

int mimc7_init(og_ctx* ctx) {
  // c_0 = 0; c_i = keccak256^(i+1)("mimc") (big-endian integer) mod r
  static const uint64_t RMOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                   0x30644e72e131a029ULL};
  uint8_t h[32];
  keccak256((const uint8_t*)"mimc", 4, h);
  memset(ctx->mimc_consts_canon, 0, sizeof(ctx->mimc_consts_canon));
  for (int i = 1; i < MIMC7_ROUNDS; i++) {
    uint8_t h2[32];
    keccak256(h, 32, h2);
    memcpy(h, h2, 32);
    uint64_t v[4];
    for (int w = 0; w < 4; w++) {  // big-endian bytes -> LE limbs
      uint64_t x = 0;
      for (int k = 0; k < 8; k++) x = (x << 8) | h[(3 - w) * 8 + k];
      v[w] = x;
    }
    for (;;) {  // v mod r by repeated subtraction (2^256 / r < 6)
      bool ge = true;
      for (int w = 3; w >= 0; w--) {
        if (v[w] != RMOD[w]) { ge = v[w] > RMOD[w]; break; }
      }
      if (!ge) break;
      unsigned __int128 bw = 0;
      for (int w = 0; w < 4; w++) {
        unsigned __int128 d = (unsigned __int128)v[w] - RMOD[w] - (uint64_t)bw;
        v[w] = (uint64_t)d;
        bw = (d >> 64) & 1;
      }
    }
    for (int w = 0; w < 4; w++)
      for (int k = 0; k < 8; k++) ctx->mimc_consts_canon[i * 32 + w * 8 + k] = (uint8_t)(v[w] >> (8 * k));
  }
  {  // limb form for the wave-wide kernels (the field layer is host code too: the same Montgomery limbs k_to_mont_fr produces)
    std::vector<uint32_t> c9((size_t)MIMC7_ROUNDS * 16, 0u);
    for (int i = 0; i < MIMC7_ROUNDS; i++) {
      const Fr c = fe_to_mont(fe_load<FrParams>(ctx->mimc_consts_canon + i * 32));
      for (int k = 0; k < 9; k++) c9[(size_t)i * 16 + k] = c.l[k];
    }
    OG_HIP(hipMalloc((void**)&ctx->mimc_consts9_d, c9.size() * 4));
    OG_HIP(hipMemcpy(ctx->mimc_consts9_d, c9.data(), c9.size() * 4, hipMemcpyHostToDevice));
  }
  OG_HIP(hipMalloc((void**)&ctx->mimc_consts_d, MIMC7_ROUNDS * 32));
  uint8_t* tmp = nullptr;
  OG_HIP(hipMalloc((void**)&tmp, MIMC7_ROUNDS * 32));
  OG_HIP(hipMemcpyAsync(tmp, ctx->mimc_consts_canon, MIMC7_ROUNDS * 32, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_to_mont_fr, dim3(1), dim3(256), 0, ctx->stream, tmp, ctx->mimc_consts_d, (size_t)MIMC7_ROUNDS);
  OG_HIP(hipGetLastError());
  OG_HIP(hipStreamSynchronize(ctx->stream));
  OG_HIP(hipFree(tmp));
  return OG_OK;
}

// two lanes per hash while the launch stays under half a wave per SIMD: then the chain's latency is what is being waited
// for, and the pair form shortens the chain (mimc7.hip.h).  Crossover measured on the 2^20-leaf tree (levels of 2^19 .. 1
// hashes): pairs up to 2^12 / 2^14 / 2^15 / 2^16 / 2^17 hashes -> 8.93 / 8.80 / 8.93 / 9.06 / 9.42 ms, never: 9.69 ms.
// OG_MIMC_PAIR = 0 | 1 forces either form (tests), OG_MIMC_PAIR_MAX moves the crossover (A/B).
static bool pair_lanes(const og_ctx* ctx, size_t n_hashes) {
  if (const char* e = OG_HOOK_STR("OG_MIMC_PAIR")) return atoi(e) != 0;
  if (const char* e = OG_HOOK_STR("OG_MIMC_PAIR_MAX")) return n_hashes <= (size_t)atoll(e);  // (A/B: the crossover)
  return 4 * n_hashes <= (size_t)ctx->n_cu * 4 * 64;
}

int mimc7_hash2(og_ctx* ctx, const uint8_t* l, const uint8_t* r, uint8_t* out, size_t n) {
  if (n == 0) return OG_OK;
  if (wave_per_hash(ctx, n))
    OG_W9_LAUNCH(k_mimc7_hash2_w9, w9_rows(), dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts9_d, l, r, out, n);
  else if (pair_lanes(ctx, n))
    hipLaunchKernelGGL(k_mimc7_hash2<true>, dim3(grid_for(2 * n, 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->mimc_consts_d, l, r, out, n);
  else
    hipLaunchKernelGGL(k_mimc7_hash2<false>, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)ctx->mimc_consts_d, l, r, out, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int mimc7_merkle_paths(og_ctx* ctx, const uint8_t* leaves, const uint64_t* idx, const uint8_t* sib, int depth,
                       uint8_t* nodes, size_t n) {
  if (n == 0) return OG_OK;
  if (wave_per_hash(ctx, n))
    OG_W9_LAUNCH(k_mimc7_merkle_paths_w9, w9_rows(), dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts9_d, leaves, idx, sib,
                       depth, nodes, n);
  else if (pair_lanes(ctx, n))
    hipLaunchKernelGGL(k_mimc7_merkle_paths<true>, dim3(grid_for(2 * n, 64)), dim3(64), 0, ctx->stream,
                       (const uint32_t*)ctx->mimc_consts_d, leaves, idx, sib, depth, nodes, n);
  else
    hipLaunchKernelGGL(k_mimc7_merkle_paths<false>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream,
                       (const uint32_t*)ctx->mimc_consts_d, leaves, idx, sib, depth, nodes, n);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int mimc7_tree_build(og_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t* nodes) {
  OG_HIP(hipMemcpyAsync(nodes, leaves, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
  size_t off = 0;
  for (size_t w = n; w > 1; w >>= 1) {
    size_t n_out = w >> 1;
    // small blocks near the root keep every CU busy a little longer
    unsigned block = n_out >= 65536 ? 256 : 64;
    if (wave_per_hash(ctx, n_out))
      OG_W9_LAUNCH(k_mimc7_tree_level_w9, w9_rows(), dim3((unsigned)n_out), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts9_d,
                         nodes + off * 32, nodes + (off + w) * 32, n_out);
    else if (pair_lanes(ctx, n_out))
      hipLaunchKernelGGL(k_mimc7_tree_level<true>, dim3(grid_for(2 * n_out, block)), dim3(block), 0, ctx->stream,
                         (const uint32_t*)ctx->mimc_consts_d, nodes + off * 32, nodes + (off + w) * 32, n_out);
    else
      hipLaunchKernelGGL(k_mimc7_tree_level<false>, dim3(grid_for(n_out, block)), dim3(block), 0, ctx->stream,
                         (const uint32_t*)ctx->mimc_consts_d, nodes + off * 32, nodes + (off + w) * 32, n_out);
    OG_HIP(hipGetLastError());
    off += w;
  }
  return OG_OK;
}

}  // namespace og
