// snarkjs key and witness files at the boundary: `.zkey` -> this library's "OWPK0001" / "OWVK0001" blobs (og_zkey_import),
// the way back (og_zkey_export), `.wtns` <-> canonical witness bytes (og_wtns_read / og_wtns_write).
//
// No reference counterpart: the snapshot holds no prover and no key; the lineage BASELINE.json's north_star
// names proved with circom / snarkjs, and VERDICT r5 asks for "an import path for keys this repository did not make".  The
// container and the Groth16 sections are restated in oracle/py/zkey.py (the oracle of this file and the tests' writer), with
// the upstream function each fact comes from; nothing here has met a file snarkjs itself wrote -- there is none in this
// environment -- so the importer checks what it CAN check on any file: both primes, every section
// length against the header, every coordinate < q, every point on its curve (a mis-read encoding leaves the curve), every
// coefficient < r and inside the matrix; and a proof made with an imported key is judged by og_verify against the imported
// verifying key.
//
// What an import has to bridge (oracle/py/zkey.py has the derivations):
//   * points: affine, little-endian MONTGOMERY numbers with R = 2^256            -> canonical (k_lem_import; R here is 2^261)
//   * coefficients: value x R^2 mod r as plain integers, matrices A and B only     -> canonical CSR; the prover takes
//     C z = (A z) o (B z) for such a key (OWPK header flag 1; groth16.hip k_mul_rows)
//   * constraint c sits at w^c with ffjavascript's root w = 5^((r-1)/d); this library's row i at 7^((r-1)/d)^i (ntt.hip,
//     the reference's `Fp` generator, /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:9): the same points
//     in another order                                                               -> a permutation of the rows
//   * the H section holds L_{2i+1}(tau) / delta . G1 over the size-2d domain (the ODD points), this library's H query
//     tau^j Z(tau) / delta . G1: since x^j Z(x) vanishes on the domain and equals -2 psi^j w^(i j) at psi^(2i+1),
//         H[j] = -2 psi^j  sum_i w^(i j) H'[i]
//     -- a DFT over group elements, run here as a radix-2 decimation-in-frequency FFT whose butterflies multiply a POINT
//     by a twiddle (k_ecntt_stage: (d / 2) log d scalar multiplications, ~45 ms for d = 2^17).  The prover itself is untouched:
//     an imported key proves through exactly the kernels every other key does, and the proof is the one snarkjs' own
//     algorithm gives for the same (r, s) (tests: oracle/py/zkey.snarkjs_prove, byte for byte).
#include "ctx.h"
#include "msm.hip.h"
#include "field.hip.h"
#include "ec.hip.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <memory>
#include <vector>

namespace og {

// ---- kernels ----------------------------------------------------------------------------------------------------------------
// a file coordinate v = x 2^256 mod q -> x in this library's Montgomery form: (v R)(2^-256 R) / R
OG_HD Fq lem_fix(const Fq& raw, const Fq& k) { return fe_mul(fe_to_mont(raw), k); }
OG_HD Fq2 lem_fix(const Fq2& raw, const Fq& k) { return {lem_fix(raw.c0, k), lem_fix(raw.c1, k)}; }
OG_HD bool lem_lt(const Fq& a) { return fe_lt_modulus(a); }
OG_HD bool lem_lt(const Fq2& a) { return fe_lt_modulus(a.c0) && fe_lt_modulus(a.c1); }
OG_HD uint32_t lem_or(const Fq& a) {
  uint32_t z = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) z |= a.l[i];
  return z;
}
OG_HD uint32_t lem_or(const Fq2& a) { return lem_or(a.c0) | lem_or(a.c1); }

// consts: K = 2^-256 (import) or 2^256 (export) in Montgomery form (32 B) | the curve's b in Montgomery form (T)
// flags[0] |= 1: a coordinate >= q;  |= 2: a point off the curve.  The point at infinity is all zeros on both sides.
// to_file = 0: file -> canonical (out_canon) and / or Montgomery (out_mont);  1: canonical -> file (out_canon)
template <class T>
__global__ void __launch_bounds__(256) k_lem_import(const uint8_t* __restrict__ in, size_t n, const uint8_t* __restrict__ consts,
                                                   uint8_t* __restrict__ out_canon, uint8_t* __restrict__ out_mont, uint32_t* __restrict__ flags,
                                                   int to_file) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<T> p = Affine<T>::load(in + i * Affine<T>::BYTES);
  Affine<T> c = Affine<T>::inf(), m = Affine<T>::inf();
  if (lem_or(p.x) | lem_or(p.y)) {
    if (!lem_lt(p.x) || !lem_lt(p.y)) {
      atomicOr(flags, 1u);
    } else {
      const Fq k = fe_load<FqParams>(consts);
      if (to_file) {  // x -> x 2^256 mod q, as a plain number
        m = {FieldIO<T>::to_mont(p.x), FieldIO<T>::to_mont(p.y)};
        c = {FieldIO<T>::from_mont(lem_fix(p.x, k)), FieldIO<T>::from_mont(lem_fix(p.y, k))};
      } else {
        m = {lem_fix(p.x, k), lem_fix(p.y, k)};
        c = {FieldIO<T>::from_mont(m.x), FieldIO<T>::from_mont(m.y)};
      }
      const T b = FieldIO<T>::load(consts + 32);
      if (!(f_sqr(m.y) == f_add(f_mul(f_sqr(m.x), m.x), b))) atomicOr(flags, 2u);
    }
  }
  if (out_canon) c.store(out_canon + i * Affine<T>::BYTES);
  if (out_mont) m.store(out_mont + i * Affine<T>::BYTES);
}

struct K256 {
  uint32_t l[8];
};
__device__ __forceinline__ K256 k256_load(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  return {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
// k P, k < 2^254: double-and-add from the top bit.  One site each of the doubling and the addition (ec.hip.h: inlined).
__device__ __forceinline__ G1XYZZ ecntt_smul(const G1XYZZ& p, const K256& k) {
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int b = 253; b >= 0; b--) {
    acc = xyzz_dbl(acc);
    if ((k.l[b >> 5] >> (b & 31)) & 1) acc = xyzz_add(acc, p);
  }
  return acc;
}

// x[i] = scale[i] . P_i (or P_i), affine Montgomery -> XYZZ
__global__ void __launch_bounds__(64) k_ecntt_load(const uint8_t* __restrict__ aff, size_t n, const uint8_t* __restrict__ scale, uint8_t* __restrict__ x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1XYZZ p = G1XYZZ::from_affine(G1Affine::load(aff + i * 64));
  if (scale) p = ecntt_smul(p, k256_load(scale + i * 32));
  p.store(x + i * 128);
}

// one decimation-in-frequency stage over points: (u, v) = (x[i], x[i + 2^s]) -> (u + v, tw[..] (u - v)); natural order in,
// bit-reversed order out after stages log_n - 1 .. 0.  tw[k] = w^k, canonical, k < n / 2.
__global__ void __launch_bounds__(64) k_ecntt_stage(uint8_t* __restrict__ x, int log_n, int s, const uint8_t* __restrict__ tw) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ((size_t)1 << (log_n - 1))) return;
  const size_t half = (size_t)1 << s, lo = t & (half - 1);
  const size_t i = ((t >> s) << (s + 1)) | lo, j = i + half;
  const G1XYZZ u = G1XYZZ::load(x + i * 128), v = G1XYZZ::load(x + j * 128);
  G1XYZZ r[2];
#pragma unroll 1
  for (int op = 0; op < 2; op++) r[op] = xyzz_add(u, op ? xyzz_neg(v) : v);
  const size_t k = lo << (log_n - 1 - s);
  if (k) r[1] = ecntt_smul(r[1], k256_load(tw + k * 32));
  r[0].store(x + i * 128);
  r[1].store(x + j * 128);
}

// position p holds X[rev(p)]: out[j] = scale[j] . X[j] (or X[j]) as canonical affine bytes, j = rev(p) < n_out
__global__ void __launch_bounds__(64) k_ecntt_finish(const uint8_t* __restrict__ x, int log_n, size_t n_out, const uint8_t* __restrict__ scale,
                                                    uint8_t* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ((size_t)1 << log_n)) return;
  const size_t j = log_n ? (size_t)(__brevll((unsigned long long)p) >> (64 - log_n)) : 0;
  if (j >= n_out) return;
  G1XYZZ v = G1XYZZ::load(x + p * 128);
  if (scale) v = ecntt_smul(v, k256_load(scale + j * 32));
  G1Affine a = xyzz_to_affine(v);
  a.x = fe_from_mont(a.x);
  a.y = fe_from_mont(a.y);
  a.store(out + j * 64);
}

// ---- host arithmetic (the field layer's own routines, on the host) ----------------------------------------------------------
static Fr zfr_load(const uint8_t* p) {
  uint32_t w[8];
  memcpy(w, p, 32);
  return fe_from_words<FrParams>(w);
}
static void zfr_store(uint8_t* p, const Fr& v) {  // v canonical
  uint32_t w[8];
  fe_to_words(w, v);
  memcpy(p, w, 32);
}
static Fr zfr_pow_words(const Fr& a, const uint32_t e[8]) {  // a in Montgomery form
  Fr r = Fr::one();
  for (int b = 255; b >= 0; b--) {
    r = fe_sqr(r);
    if ((e[b >> 5] >> (b & 31)) & 1) r = fe_mul(r, a);
  }
  return r;
}
static Fr zfr_pow2k(Fr a, int k) {  // a^(2^k)
  for (int i = 0; i < k; i++) a = fe_sqr(a);
  return a;
}
static bool zfr_is_one(const Fr& a) { return a == Fr::one(); }

// (r - 1) / 2^28, the odd cofactor (ntt.hip carries the same words)
static const uint32_t FR_T[8] = {0x3e1f593fu, 0x9b970914u, 0x833e8487u, 0x181585d2u, 0x85045b68u, 0x131a029bu, 0x0644e72eu, 0x00000003u};

struct FfRoots {
  bool ok = false;
  Fr w28_ff, w28_own;  // primitive 2^28-th roots, Montgomery: ffjavascript's nqr^T and this library's 7^T
};
// ffjavascript's F1Field takes the SMALLEST quadratic non-residue, counting up from 2, as the base of its roots of unity.
// For BN254's r that is 5 -- checked here, not remembered: g^T has order exactly 2^28 iff g is a non-residue.
static const FfRoots& ff_roots() {
  static const FfRoots roots = [] {
    FfRoots f;
    auto order_is_2_28 = [](uint32_t g, Fr* out) {
      const Fr w = zfr_pow_words(fe_to_mont(fe_from_u32<FrParams>(g)), FR_T);
      if (out) *out = w;
      return !zfr_is_one(zfr_pow2k(w, 27));
    };
    f.ok = !order_is_2_28(2, nullptr) && !order_is_2_28(3, nullptr) && order_is_2_28(5, &f.w28_ff) && order_is_2_28(7, &f.w28_own);
    return f;
  }();
  return roots;
}

// k (mod 2^log_n) with base^k = target, both of exact order 2^log_n
static uint64_t zfr_dlog_pow2(const Fr& base, const Fr& target, int log_n) {
  const Fr binv = fe_inv(base);
  uint64_t k = 0;
  Fr bk = Fr::one();  // binv^k
  Fr step = binv;     // binv^(2^b)
  for (int b = 0; b < log_n; b++) {
    const Fr t = fe_mul(target, bk);
    if (!zfr_is_one(zfr_pow2k(t, log_n - 1 - b))) {
      k |= 1ull << b;
      bk = fe_mul(bk, step);
    }
    step = fe_sqr(step);
  }
  return k;
}

static uint64_t inv_mod_pow2(uint64_t a, int log_n) {  // a odd
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - a * x;  // Newton: doubles the correct low bits
  return log_n >= 64 ? x : x & ((1ull << log_n) - 1);
}

// ---- containers ---------------------------------------------------------------------------------------------------------------
struct BinFile {
  uint32_t version = 0;
  std::map<uint32_t, std::pair<const uint8_t*, uint64_t>> sec;  // the first occurrence of every section id
};
static uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static int binfile_parse(const uint8_t* p, size_t len, const char* magic, uint32_t max_version, const std::string& who, BinFile* out) {
  OG_REQUIRE(len >= 12 && memcmp(p, magic, 4) == 0, who + ": not a " + magic + " file");
  out->version = rd32(p + 4);
  OG_REQUIRE(out->version >= 1 && out->version <= max_version, who + ": unsupported version " + std::to_string(out->version));
  const uint32_t n_sec = rd32(p + 8);
  size_t off = 12;
  for (uint32_t k = 0; k < n_sec; k++) {
    OG_REQUIRE(off + 12 <= len, who + ": truncated section header");
    const uint32_t id = rd32(p + off);
    const uint64_t size = rd64(p + off + 4);
    off += 12;
    OG_REQUIRE(size <= len - off, who + ": section " + std::to_string(id) + " runs past the end of the file");
    out->sec.insert({id, {p + off, size}});
    off += size;
  }
  return OG_OK;
}

static const uint8_t FQ_BYTES[32] = {0x47, 0xfd, 0x7c, 0xd8, 0x16, 0x8c, 0x20, 0x3c, 0x8d, 0xca, 0x71, 0x68, 0x91, 0x6a, 0x81, 0x97,
                                     0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
static const uint8_t FR_BYTES[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                     0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

struct ZDev {  // hipMalloc'd scratch released on every exit path
  std::vector<void*> ptrs;
  ~ZDev() {
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

// K = 2^(+-256) mod q in Montgomery form | b of the curve in Montgomery form, as the kernels load them
static void lem_consts(bool g2, bool to_file, uint8_t out[32 + 64]) {
  Fq k = Fq::one();
  for (int i = 0; i < 256; i++) k = fe_dbl(k);
  if (!to_file) k = fe_inv(k);
  fe_store(out, k);
  const Fq three = fe_to_mont(fe_from_u32<FqParams>(3));
  if (!g2) {
    fe_store(out + 32, three);
    memset(out + 64, 0, 32);
  } else {  // b' = 3 / (9 + u)
    const Fq2 xi = {fe_to_mont(fe_from_u32<FqParams>(9)), Fq::one()};
    const Fq2 b = f_mul(Fq2{three, Fq::zero()}, f_inv(xi));
    fe_store(out + 32, b.c0);
    fe_store(out + 64, b.c1);
  }
}

// file <-> canonical for `n` points of one group (host buffers in and out; `mont_d`, optional: the Montgomery copy stays on the device)
static int lem_convert(og_ctx* ctx, ZDev& dev, bool g2, bool to_file, const uint8_t* in, size_t n, uint8_t* out, uint8_t* mont_d,
                       const std::string& who) {
  if (n == 0) return OG_OK;
  const size_t pb = g2 ? 128 : 64;
  uint8_t *in_d, *out_d, *c_d, *f_d;
  OG_TRY(dev.get(n * pb, &in_d));
  OG_TRY(dev.get(n * pb, &out_d));
  OG_TRY(dev.get(96, &c_d));
  OG_TRY(dev.get(4, &f_d));
  alignas(16) uint8_t consts[96];
  lem_consts(g2, to_file, consts);
  OG_HIP(hipMemcpyAsync(in_d, in, n * pb, hipMemcpyHostToDevice, ctx->stream));
  OG_HIP(hipMemcpyAsync(c_d, consts, 96, hipMemcpyHostToDevice, ctx->stream));
  OG_HIP(hipMemsetAsync(f_d, 0, 4, ctx->stream));
  if (g2)
    hipLaunchKernelGGL(k_lem_import<Fq2>, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in_d, n, c_d, out_d, mont_d, (uint32_t*)f_d, to_file ? 1 : 0);
  else
    hipLaunchKernelGGL(k_lem_import<Fq>, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in_d, n, c_d, out_d, mont_d, (uint32_t*)f_d, to_file ? 1 : 0);
  OG_HIP(hipGetLastError());
  uint32_t flags = 0;
  OG_HIP(hipMemcpyAsync(out, out_d, n * pb, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipMemcpyAsync(&flags, f_d, 4, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  OG_REQUIRE(!(flags & 1), who + ": a point coordinate is not below the base-field modulus");
  OG_REQUIRE(!(flags & 2), who + ": a point is not on the curve");
  return OG_OK;
}

// DFT over G1 points (see the head of this file).  aff_mont_d: n = 2^log_n affine Montgomery points; tw: n / 2 canonical
// twiddles w^k; pre / post: optional n canonical scalars applied before / after (post indexed by OUTPUT position);
// out: n_out canonical affine points (host)
static int ecntt_g1(og_ctx* ctx, ZDev& dev, const uint8_t* aff_mont_d, int log_n, const std::vector<uint8_t>& tw, const std::vector<uint8_t>* pre,
                    const std::vector<uint8_t>* post, size_t n_out, uint8_t* out) {
  const size_t n = (size_t)1 << log_n;
  uint8_t *x_d, *tw_d, *pre_d = nullptr, *post_d = nullptr, *out_d;
  OG_TRY(dev.get(n * 128, &x_d));
  OG_TRY(dev.get(tw.size(), &tw_d));
  OG_TRY(dev.get(n * 64, &out_d));
  if (!tw.empty()) OG_HIP(hipMemcpyAsync(tw_d, tw.data(), tw.size(), hipMemcpyHostToDevice, ctx->stream));
  if (pre) {
    OG_TRY(dev.get(n * 32, &pre_d));
    OG_HIP(hipMemcpyAsync(pre_d, pre->data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  if (post) {
    OG_TRY(dev.get(n * 32, &post_d));
    OG_HIP(hipMemcpyAsync(post_d, post->data(), n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(k_ecntt_load, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, aff_mont_d, n, pre_d, x_d);
  OG_HIP(hipGetLastError());
  for (int s = log_n - 1; s >= 0; s--) {
    hipLaunchKernelGGL(k_ecntt_stage, dim3(grid_for(n / 2, 64)), dim3(64), 0, ctx->stream, x_d, log_n, s, tw_d);
    OG_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_ecntt_finish, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, x_d, log_n, n_out, post_d, out_d);
  OG_HIP(hipGetLastError());
  OG_HIP(hipMemcpyAsync(out, out_d, n_out * 64, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));  // (tw / pre / post: the host vectors outlive the copies)
  return OG_OK;
}

// powers of a Montgomery value as canonical bytes: out[k] = c a^k, k < count
static void pow_table(const Fr& a, const Fr& c, size_t count, std::vector<uint8_t>& out) {
  out.resize(count * 32);
  Fr t = c;
  for (size_t k = 0; k < count; k++) {
    zfr_store(&out[k * 32], fe_from_mont(t));
    t = fe_mul(t, a);
  }
}

static inline size_t zpad32(size_t n) { return (n + 31) / 32 * 32; }
static void zput(std::vector<uint8_t>& out, const void* p, size_t n) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  out.insert(out.end(), b, b + n);
  out.resize(out.size() + (zpad32(n) - n), 0);
}

struct ZHeader {
  uint64_t m = 0, l = 0, d = 0;
  int power = 0;
};

// the domain's roots as ffjavascript numbers them, and the row <-> constraint permutation
struct ZDomain {
  Fr w, psi;     // primitive d-th and 2d-th roots (ffjavascript's), Montgomery
  uint64_t k;    // this library's row i <-> constraint i k mod d
};
static int zdomain(int power, const std::string& who, ZDomain* out) {
  const FfRoots& f = ff_roots();
  OG_REQUIRE(f.ok, who + ": internal: 5 is not the smallest quadratic non-residue of the scalar field");
  OG_REQUIRE(power >= 1 && power <= 27, who + ": domain must hold 2 .. 2^27 points");
  out->w = zfr_pow2k(f.w28_ff, 28 - power);
  out->psi = zfr_pow2k(f.w28_ff, 27 - power);
  out->k = zfr_dlog_pow2(out->w, zfr_pow2k(f.w28_own, 28 - power), power);
  return OG_OK;
}

// (row, signal, value) entries of one matrix -> CSR over d rows: columns ascending, duplicates summed, zeros dropped, values canonical
struct ZEnt {
  uint32_t row, sig;
  Fr v;  // Montgomery
};
static void zcsr(std::vector<ZEnt>& ent, uint64_t d, std::vector<uint32_t>& ptr, std::vector<uint32_t>& col, std::vector<uint8_t>& val) {
  std::stable_sort(ent.begin(), ent.end(), [](const ZEnt& a, const ZEnt& b) { return a.row != b.row ? a.row < b.row : a.sig < b.sig; });
  ptr.assign(d + 1, 0);
  col.clear();
  val.clear();
  for (size_t i = 0; i < ent.size();) {
    size_t j = i;
    Fr acc = Fr::zero();
    for (; j < ent.size() && ent[j].row == ent[i].row && ent[j].sig == ent[i].sig; j++) acc = fe_add(acc, ent[j].v);
    if (!acc.is_zero()) {
      col.push_back(ent[i].sig);
      val.resize(val.size() + 32);
      zfr_store(&val[val.size() - 32], fe_from_mont(acc));
      ptr[ent[i].row + 1]++;
    }
    i = j;
  }
  for (uint64_t r = 0; r < d; r++) ptr[r + 1] += ptr[r];
}

// r1cs (optional): the circuit the key was made for (og_r1cs_read of circom's .r1cs).  Its A and B matrices must be the key's
// coefficient section, row for row; its C matrix then rides in the imported key (header flag 0), and the prover refuses a
// witness that violates a constraint exactly as it does for a key of og_setup.
int zkey_import(og_ctx* ctx, const uint8_t* data, size_t len, const og_r1cs* r1cs, std::vector<uint8_t>& pk, std::vector<uint8_t>& vk) {
  const std::string who = "og_zkey_import";
  BinFile bf;
  OG_TRY(binfile_parse(data, len, "zkey", 1, who, &bf));
  for (uint32_t id = 1; id <= 9; id++) OG_REQUIRE(bf.sec.count(id), who + ": section " + std::to_string(id) + " missing");
  OG_REQUIRE(bf.sec[1].second >= 4 && rd32(bf.sec[1].first) == 1, who + ": not a Groth16 key (protocol id != 1)");
  const uint8_t* h = bf.sec[2].first;
  OG_REQUIRE(bf.sec[2].second == 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128, who + ": header section has the wrong length for BN254");
  OG_REQUIRE(rd32(h) == 32 && memcmp(h + 4, FQ_BYTES, 32) == 0, who + ": the base field is not BN254's");
  OG_REQUIRE(rd32(h + 36) == 32 && memcmp(h + 40, FR_BYTES, 32) == 0, who + ": the scalar field is not BN254's");
  ZHeader z;
  z.m = rd32(h + 72); z.l = rd32(h + 76); z.d = rd32(h + 80);
  OG_REQUIRE(z.d >= 2 && (z.d & (z.d - 1)) == 0, who + ": domain size is not a power of two >= 2");
  while (((uint64_t)1 << z.power) < z.d) z.power++;
  OG_REQUIRE(z.m >= 1 && z.l < z.m && z.m < (1ull << 31), who + ": bad wire counts");
  const uint64_t m = z.m, l = z.l, d = z.d, nl = m - l - 1;
  const uint64_t want[10] = {0, 0, 0, (l + 1) * 64, 0, m * 64, m * 64, m * 128, nl * 64, d * 64};
  for (uint32_t id : {3u, 5u, 6u, 7u, 8u, 9u})
    OG_REQUIRE(bf.sec[id].second == want[id], who + ": section " + std::to_string(id) + " has " + std::to_string(bf.sec[id].second) +
                                                  " bytes, the header asks for " + std::to_string(want[id]));
  ZDomain dom;
  OG_TRY(zdomain(z.power, who, &dom));
  // ---- coefficients -> CSR rows (A, B), canonical
  const uint8_t* cs = bf.sec[4].first;
  OG_REQUIRE(bf.sec[4].second >= 4, who + ": coefficient section truncated");
  const uint64_t n_coef = rd32(cs);
  OG_REQUIRE(bf.sec[4].second == 4 + n_coef * 44, who + ": coefficient section length does not match its count");
  Fr ri2 = Fr::one();  // 2^-512 mod r, Montgomery
  for (int i = 0; i < 512; i++) ri2 = fe_dbl(ri2);
  ri2 = fe_inv(ri2);
  const uint64_t kinv = inv_mod_pow2(dom.k, z.power);
  std::vector<ZEnt> ent[2];
  for (uint64_t i = 0; i < n_coef; i++) {
    const uint8_t* e = cs + 4 + i * 44;
    const uint32_t mt = rd32(e), c = rd32(e + 4), sg = rd32(e + 8);
    const Fr raw = zfr_load(e + 12);
    OG_REQUIRE(mt <= 1 && c < d && sg < m && fe_lt_modulus(raw), who + ": coefficient " + std::to_string(i) + " out of range");
    ent[mt].push_back({(uint32_t)((c * kinv) & (d - 1)), sg, fe_mul(fe_to_mont(raw), ri2)});
  }
  std::vector<uint32_t> ptr[3], col[3];
  std::vector<uint8_t> val[3];
  for (int k = 0; k < 2; k++) zcsr(ent[k], d, ptr[k], col[k], val[k]);
  ptr[2].assign(d + 1, 0);
  uint64_t flags = 1;
  if (r1cs) {
    const uint64_t nc = r1cs->n_constraints;
    OG_REQUIRE(r1cs->n_wires == m && r1cs->n_pub == l, who + ": the .r1cs has " + std::to_string(r1cs->n_wires) + " wires / " + std::to_string(r1cs->n_pub) +
                                                           " public, the .zkey " + std::to_string(m) + " / " + std::to_string(l));
    OG_REQUIRE(nc + l + 1 <= d, who + ": the .r1cs has more constraints than the key's domain holds");
    std::vector<ZEnt> re[3];
    for (int k = 0; k < 3; k++) {
      OG_REQUIRE(r1cs->ptr[k].size() == nc + 1 && r1cs->val[k].size() == r1cs->col[k].size() * 32, who + ": malformed R1CS");
      for (uint64_t c = 0; c < nc; c++)
        for (uint32_t e = r1cs->ptr[k][c]; e < r1cs->ptr[k][c + 1]; e++) {
          const Fr v = zfr_load(&r1cs->val[k][(size_t)e * 32]);
          OG_REQUIRE(r1cs->col[k][e] < m && fe_lt_modulus(v), who + ": R1CS entry out of range");
          re[k].push_back({(uint32_t)((c * kinv) & (d - 1)), r1cs->col[k][e], fe_to_mont(v)});
        }
    }
    for (uint64_t s = 0; s <= l; s++) re[0].push_back({(uint32_t)(((nc + s) * kinv) & (d - 1)), (uint32_t)s, Fr::one()});  // the public-input rows
    std::vector<uint32_t> rp, rc;
    std::vector<uint8_t> rv;
    for (int k = 0; k < 2; k++) {
      zcsr(re[k], d, rp, rc, rv);
      OG_REQUIRE(rp == ptr[k] && rc == col[k] && rv == val[k], who + std::string(": the .r1cs is not this key's circuit (matrix ") + (k ? "B" : "A") + " differs)");
    }
    zcsr(re[2], d, ptr[2], col[2], val[2]);
    flags = 0;
  }
  // ---- points
  std::lock_guard<std::mutex> lk(ctx->mu);
  OG_HIP(hipSetDevice(ctx->device));
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  ZDev dev;
  const size_t n1 = 3 + (l + 1) + m + m + nl, n2 = 3 + m;
  std::vector<uint8_t> g1_in(n1 * 64), g1(n1 * 64), g2_in(n2 * 128), g2(n2 * 128), hq((d - 1) * 64);
  {
    uint8_t* o = g1_in.data();
    memcpy(o, h + 84, 64); o += 64;                      // alpha1
    memcpy(o, h + 84 + 64, 64); o += 64;                 // beta1
    memcpy(o, h + 84 + 64 + 64 + 128 + 128, 64); o += 64;  // delta1
    for (uint32_t id : {3u, 5u, 6u, 8u}) {
      memcpy(o, bf.sec[id].first, bf.sec[id].second);
      o += bf.sec[id].second;
    }
    uint8_t* q = g2_in.data();
    memcpy(q, h + 84 + 128, 128); q += 128;        // beta2
    memcpy(q, h + 84 + 128 + 128, 128); q += 128;  // gamma2
    memcpy(q, h + 84 + 128 + 256 + 64, 128); q += 128;  // delta2
    memcpy(q, bf.sec[7].first, bf.sec[7].second);
  }
  OG_TRY(lem_convert(ctx, dev, false, false, g1_in.data(), n1, g1.data(), nullptr, who));
  OG_TRY(lem_convert(ctx, dev, true, false, g2_in.data(), n2, g2.data(), nullptr, who));
  {  // the H section -> the coefficient-basis H query
    uint8_t* hm_d;
    OG_TRY(dev.get(d * 64, &hm_d));
    std::vector<uint8_t> h_in(bf.sec[9].first, bf.sec[9].first + d * 64), h_canon(d * 64), tw, post;
    OG_TRY(lem_convert(ctx, dev, false, false, h_in.data(), d, h_canon.data(), hm_d, who));
    pow_table(dom.w, Fr::one(), d / 2, tw);
    const Fr two = fe_dbl(Fr::one());
    pow_table(dom.psi, fe_neg(two), d, post);
    OG_TRY(ecntt_g1(ctx, dev, hm_d, z.power, tw, nullptr, &post, d - 1, hq.data()));
  }
  // ---- "OWPK0001" (groth16.hip pk_load_impl); header flag 1 (no .r1cs given): C z = (A z) o (B z)
  pk.clear();
  const uint64_t head[10] = {0x313030304b50574full, m, l, (uint64_t)z.power, d, col[0].size(), col[1].size(), col[2].size(), flags, 0};
  pk.insert(pk.end(), (const uint8_t*)head, (const uint8_t*)head + 80);
  const uint8_t *alpha1 = &g1[0], *beta1 = &g1[64], *delta1 = &g1[128], *ic = &g1[192];
  const uint8_t *aq = ic + (l + 1) * 64, *b1q = aq + m * 64, *cq = b1q + m * 64;
  const uint8_t *beta2 = &g2[0], *gamma2 = &g2[128], *delta2 = &g2[256], *b2q = &g2[384];
  zput(pk, alpha1, 64);
  zput(pk, beta1, 64);
  zput(pk, delta1, 64);
  pk.resize(pk.size() + 64, 0);
  zput(pk, beta2, 128);
  zput(pk, delta2, 128);
  for (int k = 0; k < 3; k++) {
    zput(pk, ptr[k].data(), ptr[k].size() * 4);
    zput(pk, col[k].data(), col[k].size() * 4);
    zput(pk, val[k].data(), val[k].size());
  }
  zput(pk, aq, m * 64);
  zput(pk, b1q, m * 64);
  zput(pk, b2q, m * 128);
  zput(pk, cq, nl * 64);
  zput(pk, hq.data(), hq.size());
  // ---- "OWVK0001" | n_pub | alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | IC
  vk.clear();
  vk.insert(vk.end(), (const uint8_t*)"OWVK0001", (const uint8_t*)"OWVK0001" + 8);
  vk.insert(vk.end(), (const uint8_t*)&l, (const uint8_t*)&l + 8);
  vk.insert(vk.end(), alpha1, alpha1 + 64);
  vk.insert(vk.end(), beta2, beta2 + 128);
  vk.insert(vk.end(), gamma2, gamma2 + 128);
  vk.insert(vk.end(), delta2, delta2 + 128);
  vk.insert(vk.end(), ic, ic + (l + 1) * 64);
  return OG_OK;
}

// The way back: this library's key as a .zkey snarkjs can prove with.  The C matrix stays behind (the format has no room for it:
// snarkjs' prover takes C z = (A z) o (B z), which is C z for every satisfying witness); the H section is the inverse transform
//   H'[i] = 1/d sum_j w^(-i j) (-2 psi^j)^-1 H[j],   H[d - 1] := the point at infinity
// (the quotient has degree <= d - 2, so its d-th coefficient never meets a base); section 10 says "no contributions".
int zkey_export(og_ctx* ctx, const uint8_t* pkb, size_t pk_len, const uint8_t* vkb, size_t vk_len, std::vector<uint8_t>& out) {
  const std::string who = "og_zkey_export";
  OG_REQUIRE(pk_len >= 80 + 512 && rd64(pkb) == 0x313030304b50574full, who + ": not an OWPK0001 blob");
  uint64_t hd[10];
  memcpy(hd, pkb, 80);
  const uint64_t m = hd[1], l = hd[2], power = hd[3], n_rows = hd[4];
  OG_REQUIRE(power >= 1 && power <= 27 && m >= 1 && l < m && m < (1ull << 31), who + ": bad key header");
  const uint64_t d = 1ull << power, nl = m - l - 1, nh = d - 1;
  OG_REQUIRE(n_rows <= d, who + ": more rows than the domain holds");
  OG_REQUIRE(vk_len == 16 + 64 + 3 * 128 + (l + 1) * 64 && memcmp(vkb, "OWVK0001", 8) == 0 && rd64(vkb + 8) == l,
             who + ": the verifying key does not belong to this proving key");
  size_t off = 80 + 512;
  const uint8_t *ptr_h[3], *col_h[3], *val_h[3];
  for (int k = 0; k < 3; k++) {
    OG_REQUIRE(hd[5 + k] < (1ull << 32), who + ": nnz too large");
    ptr_h[k] = pkb + off; off += zpad32((n_rows + 1) * 4);
    col_h[k] = pkb + off; off += zpad32(hd[5 + k] * 4);
    val_h[k] = pkb + off; off += zpad32(hd[5 + k] * 32);
    OG_REQUIRE(off <= pk_len, who + ": truncated key");
  }
  const uint8_t* q_h[5];
  const size_t q_n[5] = {m, m, m, nl, nh};
  const size_t q_pb[5] = {64, 64, 128, 64, 64};
  for (int k = 0; k < 5; k++) {
    q_h[k] = pkb + off;
    off += zpad32(q_n[k] * q_pb[k]);
  }
  OG_REQUIRE(off == pk_len, who + ": key length does not match its header");
  ZDomain dom;
  OG_TRY(zdomain((int)power, who, &dom));
  // ---- coefficients: row i -> constraint i k mod d, value x 2^512
  Fr r2 = Fr::one();
  for (int i = 0; i < 512; i++) r2 = fe_dbl(r2);
  std::vector<uint8_t> cs(4);
  uint64_t n_coef = 0;
  for (int k = 0; k < 2; k++) {
    const uint32_t* p = (const uint32_t*)ptr_h[k];
    const uint32_t* c = (const uint32_t*)col_h[k];
    OG_REQUIRE(p[0] == 0 && p[n_rows] == hd[5 + k], who + ": CSR row pointers inconsistent");
    for (uint64_t row = 0; row < n_rows; row++) {
      OG_REQUIRE(p[row] <= p[row + 1] && p[row + 1] <= hd[5 + k], who + ": CSR row pointers not monotone");
      for (uint32_t e = p[row]; e < p[row + 1]; e++) {
        OG_REQUIRE(c[e] < m, who + ": CSR column out of range");
        const Fr v = zfr_load(val_h[k] + (size_t)e * 32);
        OG_REQUIRE(fe_lt_modulus(v), who + ": coefficient not canonical");
        const uint32_t rec[3] = {(uint32_t)k, (uint32_t)((row * dom.k) & (d - 1)), c[e]};
        cs.insert(cs.end(), (const uint8_t*)rec, (const uint8_t*)rec + 12);
        cs.resize(cs.size() + 32);
        zfr_store(&cs[cs.size() - 32], fe_from_mont(fe_mul(fe_to_mont(v), r2)));
        n_coef++;
      }
    }
  }
  OG_REQUIRE(n_coef < (1ull << 32), who + ": too many coefficients");
  const uint32_t n_coef32 = (uint32_t)n_coef;
  memcpy(cs.data(), &n_coef32, 4);
  // ---- points
  std::lock_guard<std::mutex> lk(ctx->mu);
  OG_HIP(hipSetDevice(ctx->device));
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  ZDev dev;
  const size_t n1 = 3 + (l + 1) + m + m + nl, n2 = 3 + m;
  std::vector<uint8_t> g1_in(n1 * 64), g1(n1 * 64), g2_in(n2 * 128), g2(n2 * 128), hsec(d * 64);
  {
    uint8_t* o = g1_in.data();
    memcpy(o, pkb + 80, 64); o += 64;        // alpha1
    memcpy(o, pkb + 80 + 64, 64); o += 64;   // beta1
    memcpy(o, pkb + 80 + 128, 64); o += 64;  // delta1
    memcpy(o, vkb + 16 + 64 + 384, (l + 1) * 64); o += (l + 1) * 64;
    memcpy(o, q_h[0], m * 64); o += m * 64;
    memcpy(o, q_h[1], m * 64); o += m * 64;
    memcpy(o, q_h[3], nl * 64);
    uint8_t* q = g2_in.data();
    memcpy(q, pkb + 80 + 256, 128); q += 128;      // beta2
    memcpy(q, vkb + 16 + 64 + 128, 128); q += 128;  // gamma2
    memcpy(q, pkb + 80 + 384, 128); q += 128;      // delta2
    memcpy(q, q_h[2], m * 128);
  }
  OG_REQUIRE(memcmp(vkb + 16, pkb + 80, 64) == 0 && memcmp(vkb + 16 + 64, pkb + 80 + 256, 128) == 0 && memcmp(vkb + 16 + 64 + 256, pkb + 80 + 384, 128) == 0,
             who + ": the verifying key's alpha / beta / delta differ from the proving key's");
  OG_TRY(lem_convert(ctx, dev, false, true, g1_in.data(), n1, g1.data(), nullptr, who));
  OG_TRY(lem_convert(ctx, dev, true, true, g2_in.data(), n2, g2.data(), nullptr, who));
  {
    uint8_t* hm_d;
    OG_TRY(dev.get(d * 64, &hm_d));
    std::vector<uint8_t> h_in(d * 64, 0), h_file(d * 64), h_canon(d * 64), tw, pre;
    memcpy(h_in.data(), q_h[4], nh * 64);  // H[d - 1] = infinity
    OG_TRY(lem_convert(ctx, dev, false, true, h_in.data(), d, h_file.data(), hm_d, who));  // (only the Montgomery copy on the device is used)
    pow_table(fe_inv(dom.w), Fr::one(), d / 2, tw);
    Fr nn = Fr::one();  // d as a field element: 2^power
    for (uint64_t i = 0; i < power; i++) nn = fe_dbl(nn);
    const Fr two = fe_dbl(Fr::one());
    pow_table(fe_inv(dom.psi), fe_inv(fe_mul(fe_neg(two), nn)), d, pre);  // (-2 psi^j)^-1 / d
    OG_TRY(ecntt_g1(ctx, dev, hm_d, (int)power, tw, &pre, nullptr, d, h_canon.data()));
    OG_TRY(lem_convert(ctx, dev, false, true, h_canon.data(), d, hsec.data(), nullptr, who));
  }
  // ---- the file
  out.clear();
  auto put32 = [&](uint32_t v) { out.insert(out.end(), (const uint8_t*)&v, (const uint8_t*)&v + 4); };
  auto put64 = [&](uint64_t v) { out.insert(out.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8); };
  auto section = [&](uint32_t id, const std::vector<uint8_t>& body) {
    put32(id);
    put64(body.size());
    out.insert(out.end(), body.begin(), body.end());
  };
  out.insert(out.end(), (const uint8_t*)"zkey", (const uint8_t*)"zkey" + 4);
  put32(1);
  put32(10);
  const uint8_t *alpha1 = &g1[0], *beta1 = &g1[64], *delta1 = &g1[128], *ic = &g1[192];
  const uint8_t *aq = ic + (l + 1) * 64, *b1q = aq + m * 64, *cq = b1q + m * 64;
  const uint8_t *beta2 = &g2[0], *gamma2 = &g2[128], *delta2 = &g2[256], *b2q = &g2[384];
  section(1, {1, 0, 0, 0});
  {
    std::vector<uint8_t> hdr;
    auto h32 = [&](uint32_t v) { hdr.insert(hdr.end(), (const uint8_t*)&v, (const uint8_t*)&v + 4); };
    auto hb = [&](const uint8_t* p, size_t n) { hdr.insert(hdr.end(), p, p + n); };
    h32(32); hb(FQ_BYTES, 32); h32(32); hb(FR_BYTES, 32);
    h32((uint32_t)m); h32((uint32_t)l); h32((uint32_t)d);
    hb(alpha1, 64); hb(beta1, 64); hb(beta2, 128); hb(gamma2, 128); hb(delta1, 64); hb(delta2, 128);
    section(2, hdr);
  }
  section(3, std::vector<uint8_t>(ic, ic + (l + 1) * 64));
  section(4, cs);
  section(5, std::vector<uint8_t>(aq, aq + m * 64));
  section(6, std::vector<uint8_t>(b1q, b1q + m * 64));
  section(7, std::vector<uint8_t>(b2q, b2q + m * 128));
  section(8, std::vector<uint8_t>(cq, cq + nl * 64));
  section(9, hsec);
  section(10, std::vector<uint8_t>(68, 0));  // circuit hash (zeros: not snarkjs' hash of a .r1cs) | 0 contributions
  return OG_OK;
}

}  // namespace og

using namespace og;

static int blob_out(const std::vector<uint8_t>& v, uint8_t** out, size_t* len, const char* who) {
  uint8_t* a = static_cast<uint8_t*>(malloc(v.size() ? v.size() : 1));
  if (!a) {
    set_error(std::string(who) + ": out of host memory");
    return OG_ERR_INVALID;
  }
  memcpy(a, v.data(), v.size());
  *out = a;
  *len = v.size();
  return OG_OK;
}

extern "C" {

int og_zkey_import(og_ctx* ctx, const uint8_t* zkey, size_t zkey_len, const og_r1cs* r1cs, uint8_t** pk_out, size_t* pk_len, uint8_t** vk_out,
                   size_t* vk_len) {
  return guarded([&]() -> int {
    OG_REQUIRE(ctx && zkey && pk_out && pk_len && vk_out && vk_len, "og_zkey_import: null argument");
    *pk_out = *vk_out = nullptr;
    *pk_len = *vk_len = 0;
    std::vector<uint8_t> pk, vk;
    OG_TRY(zkey_import(ctx, zkey, zkey_len, r1cs, pk, vk));
    uint8_t *a = nullptr, *b = nullptr;
    size_t al = 0, bl = 0;
    OG_TRY(blob_out(pk, &a, &al, "og_zkey_import"));
    if (blob_out(vk, &b, &bl, "og_zkey_import") != OG_OK) {
      free(a);
      return OG_ERR_INVALID;
    }
    *pk_out = a; *pk_len = al;
    *vk_out = b; *vk_len = bl;
    return OG_OK;
  });
}

int og_zkey_export(og_ctx* ctx, const uint8_t* pk, size_t pk_len, const uint8_t* vk, size_t vk_len, uint8_t** zkey_out, size_t* zkey_len) {
  return guarded([&]() -> int {
    OG_REQUIRE(ctx && pk && vk && zkey_out && zkey_len, "og_zkey_export: null argument");
    *zkey_out = nullptr;
    *zkey_len = 0;
    std::vector<uint8_t> z;
    OG_TRY(zkey_export(ctx, pk, pk_len, vk, vk_len, z));
    return blob_out(z, zkey_out, zkey_len, "og_zkey_export");
  });
}

// circom's .r1cs (iden3 r1csfile): magic "r1cs", version 1; section 1 = u32 n8 | prime | u32 nWires | u32 nPubOut | u32 nPubIn |
// u32 nPrvIn | u64 nLabels | u32 mConstraints; section 2 = per constraint three linear combinations A, B, C, each u32 n followed
// by n x (u32 wire | n8-byte little-endian NORMAL-form coefficient); section 3 (wire -> label) is not needed here.
// Wires: 0 = the constant 1, then the public outputs, the public inputs (n_pub = nPubOut + nPubIn), the rest.
int og_r1cs_read(const uint8_t* file, size_t len, og_r1cs** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(file && out, "og_r1cs_read: null argument");
    *out = nullptr;
    BinFile bf;
    OG_TRY(binfile_parse(file, len, "r1cs", 1, "og_r1cs_read", &bf));
    OG_REQUIRE(bf.sec.count(1) && bf.sec.count(2), "og_r1cs_read: section missing");
    const uint8_t* h = bf.sec[1].first;
    OG_REQUIRE(bf.sec[1].second == 4 + 32 + 16 + 8 + 4 && rd32(h) == 32 && memcmp(h + 4, FR_BYTES, 32) == 0, "og_r1cs_read: not BN254's scalar field");
    const uint64_t n_wires = rd32(h + 36), n_pub = (uint64_t)rd32(h + 40) + rd32(h + 44), nc = rd32(h + 60);
    OG_REQUIRE(n_wires >= 1 && n_pub < n_wires && n_wires < (1ull << 31) && nc < (1ull << 31), "og_r1cs_read: bad sizes");
    std::unique_ptr<og_r1cs> r(new og_r1cs());
    r->n_wires = n_wires; r->n_pub = n_pub; r->n_constraints = nc;
    const uint8_t* p = bf.sec[2].first;
    const uint8_t* end = p + bf.sec[2].second;
    for (int k = 0; k < 3; k++) r->ptr[k].push_back(0);
    for (uint64_t c = 0; c < nc; c++)
      for (int k = 0; k < 3; k++) {
        OG_REQUIRE(end - p >= 4, "og_r1cs_read: constraint section truncated");
        const uint64_t n = rd32(p);
        p += 4;
        OG_REQUIRE((uint64_t)(end - p) >= n * 36, "og_r1cs_read: constraint section truncated");
        for (uint64_t i = 0; i < n; i++, p += 36) {
          const uint32_t w = rd32(p);
          OG_REQUIRE(w < n_wires && fe_lt_modulus(zfr_load(p + 4)), "og_r1cs_read: constraint " + std::to_string(c) + " holds a wire or a coefficient out of range");
          r->col[k].push_back(w);
          r->val[k].insert(r->val[k].end(), p + 4, p + 36);
        }
        OG_REQUIRE(r->col[k].size() < (1ull << 32), "og_r1cs_read: too many coefficients");
        r->ptr[k].push_back((uint32_t)r->col[k].size());
      }
    OG_REQUIRE(p == end, "og_r1cs_read: bytes left over after the last constraint");
    *out = r.release();
    return OG_OK;
  });
}

int og_r1cs_write(const og_r1cs* r, uint8_t** file_out, size_t* file_len) {
  return guarded([&]() -> int {
    OG_REQUIRE(r && file_out && file_len, "og_r1cs_write: null argument");
    *file_out = nullptr;
    *file_len = 0;
    std::vector<uint8_t> out, body;
    auto put32 = [](std::vector<uint8_t>& v, uint32_t x) { v.insert(v.end(), (const uint8_t*)&x, (const uint8_t*)&x + 4); };
    auto put64 = [](std::vector<uint8_t>& v, uint64_t x) { v.insert(v.end(), (const uint8_t*)&x, (const uint8_t*)&x + 8); };
    for (uint64_t c = 0; c < r->n_constraints; c++)
      for (int k = 0; k < 3; k++) {
        put32(body, r->ptr[k][c + 1] - r->ptr[k][c]);
        for (uint32_t e = r->ptr[k][c]; e < r->ptr[k][c + 1]; e++) {
          put32(body, r->col[k][e]);
          body.insert(body.end(), &r->val[k][(size_t)e * 32], &r->val[k][(size_t)e * 32] + 32);
        }
      }
    out.insert(out.end(), (const uint8_t*)"r1cs", (const uint8_t*)"r1cs" + 4);
    put32(out, 1); put32(out, 3);
    put32(out, 1); put64(out, 64);
    put32(out, 32);
    out.insert(out.end(), FR_BYTES, FR_BYTES + 32);
    put32(out, (uint32_t)r->n_wires); put32(out, 0); put32(out, (uint32_t)r->n_pub); put32(out, (uint32_t)(r->n_wires - r->n_pub - 1));  // (every public wire as an input)
    put64(out, r->n_wires); put32(out, (uint32_t)r->n_constraints);
    put32(out, 2); put64(out, body.size());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, 3); put64(out, r->n_wires * 8);
    for (uint64_t w = 0; w < r->n_wires; w++) put64(out, w);  // wire -> label: the identity
    return blob_out(out, file_out, file_len, "og_r1cs_write");
  });
}

int og_wtns_read(const uint8_t* wtns, size_t len, uint8_t* values_out, size_t capacity, uint64_t* n_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(wtns && n_out, "og_wtns_read: null argument");
    *n_out = 0;
    BinFile bf;
    OG_TRY(binfile_parse(wtns, len, "wtns", 2, "og_wtns_read", &bf));
    OG_REQUIRE(bf.sec.count(1) && bf.sec.count(2), "og_wtns_read: section missing");
    const uint8_t* h = bf.sec[1].first;
    OG_REQUIRE(bf.sec[1].second == 40 && rd32(h) == 32 && memcmp(h + 4, FR_BYTES, 32) == 0, "og_wtns_read: not BN254's scalar field");
    const uint64_t n = rd32(h + 36);
    OG_REQUIRE(bf.sec[2].second == n * 32, "og_wtns_read: witness section length does not match its count");
    *n_out = n;
    if (!values_out) return OG_OK;  // (a size query)
    OG_REQUIRE(capacity >= n, "og_wtns_read: the output holds " + std::to_string(capacity) + " values, the file " + std::to_string(n));
    for (uint64_t i = 0; i < n; i++)
      OG_REQUIRE(fe_lt_modulus(zfr_load(bf.sec[2].first + i * 32)), "og_wtns_read: value " + std::to_string(i) + " is not below the modulus");
    memcpy(values_out, bf.sec[2].first, n * 32);
    return OG_OK;
  });
}

int og_wtns_write(const uint8_t* values, uint64_t n, uint8_t** wtns_out, size_t* wtns_len) {
  return guarded([&]() -> int {
    OG_REQUIRE((values || n == 0) && wtns_out && wtns_len && n < (1ull << 32), "og_wtns_write: bad argument");
    *wtns_out = nullptr;
    *wtns_len = 0;
    for (uint64_t i = 0; i < n; i++) OG_REQUIRE(fe_lt_modulus(zfr_load(values + i * 32)), "og_wtns_write: value " + std::to_string(i) + " is not canonical");
    std::vector<uint8_t> out;
    auto put32 = [&](uint32_t v) { out.insert(out.end(), (const uint8_t*)&v, (const uint8_t*)&v + 4); };
    auto put64 = [&](uint64_t v) { out.insert(out.end(), (const uint8_t*)&v, (const uint8_t*)&v + 8); };
    out.insert(out.end(), (const uint8_t*)"wtns", (const uint8_t*)"wtns" + 4);
    put32(2); put32(2);
    put32(1); put64(40);
    put32(32);
    out.insert(out.end(), FR_BYTES, FR_BYTES + 32);
    put32((uint32_t)n);
    put32(2); put64(n * 32);
    out.insert(out.end(), values, values + n * 32);
    return blob_out(out, wtns_out, wtns_len, "og_wtns_write");
  });
}

}  // extern "C"
