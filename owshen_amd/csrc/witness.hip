// Batched witness generation for the withdraw circuit ("batched MiMC7 Merkle-path
// hashing for witness generation").  No reference counterpart: the snapshot's withdraw carries no
// circuit (/root/reference/src/services/api_services/withdraw.rs:27-71 is an ECDSA-authorised burn).
// The statement and its wire order are defined by oracle/py/withdraw.py (the spec) and mirrored by
// owshen_amd/circuit.py (the R1CS); this file fills the wires:
//
//   k_withdraw_core  one lane per proof: public inputs, the (4 + depth) MultiMiMC7 gadgets with every
//                    intermediate power (t^2, t^4, t^6, t^7 per round), the Merkle selectors
//   k_withdraw_pad   one lane per padding unit (a 3-wire gate or a 64-gate chained segment)
//
// Input record per proof, (8 + depth) x 32 B canonical LE:
//   nullifier | secret | amount | recipient | pad_seed | index (u64 in the low bytes) | token | chain_id | siblings[depth]
// (token and chain_id: the rest of what the reference's gate signs, /root/reference/contracts/src/Owshen.sol:69)
// Output: n_wires x 32 B canonical per proof, wire order as documented in oracle/py/withdraw.py.
#include "ctx.h"
#include "mimc7.hip.h"
#include "host_fr4.h"
#include "field_w9.hip.h"
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>

namespace og {

__device__ __forceinline__ void lds_put9(uint32_t* p, const Fr& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) p[i] = v.l[i];
}
__device__ __forceinline__ Fr lds_get9(const uint32_t* p) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p[i];
  return r;
}

constexpr int W_PUB = 6;
constexpr int W_REC = 8;  // fields of an input record before the siblings
constexpr int PAD_SEGMENT = 64;

struct WithdrawShape {
  uint64_t n_wires, n_constraints, first_gadget_wire, pad_base;
};

static WithdrawShape withdraw_shape(int depth, uint64_t n_pad3, uint64_t n_pad2) {
  WithdrawShape s;
  const uint64_t hashes = 4 + (uint64_t)depth;
  s.first_gadget_wire = 1 + W_PUB + 2 + 2 * (uint64_t)depth + 2;
  s.pad_base = s.first_gadget_wire + depth + hashes * 730 - 2;
  s.n_wires = s.pad_base + 3 * n_pad3 + 2 * n_pad2;
  s.n_constraints = 2 + 2 * (uint64_t)depth + hashes * 730 + n_pad3 + n_pad2;
  return s;
}

// CANON = true converts every wire out of Montgomery form as it is stored (the padding kernel: parallel, throughput
// bound); CANON = false stores the Montgomery value (the core kernel: one lane walks a proof's whole MiMC7 chain, so
// every multiplication taken off that chain is latency saved -- k_wires_from_mont converts afterwards, in parallel).
template <bool CANON>
struct WireWriterT {
  uint8_t* z;
  uint32_t w;
  __device__ __forceinline__ void put(uint32_t wire, const Fr& mont) { fe_store(z + (size_t)wire * 32, CANON ? fe_from_mont(mont) : mont); }
  __device__ __forceinline__ void push(const Fr& mont) { put(w++, mont); }
  // the same, by the lane of a pair that owns the wire (both lanes count)
  __device__ __forceinline__ void put_if(bool mine, uint32_t wire, const Fr& mont) { if (mine) put(wire, mont); }
  __device__ __forceinline__ void push_if(bool mine, const Fr& mont) { put_if(mine, w++, mont); }
};
typedef WireWriterT<true> WireWriter;

// One lane per proof.  The (4 + depth) MultiMiMC7 gadgets run through ONE inlined permutation body (rolled
// loops over gadgets, the two permutations of a gadget, and the 91 rounds): no device-function calls.
//
// PAIR: lanes 2g and 2g + 1 walk proof g together (mimc7.hip.h, the latency-bound form): per round both square t, the even lane
// forms t^4 and the odd lane t^3, they swap, then the even lane forms t^7 = t^4 t^3 -- the value the chain waits for -- while
// the odd lane forms the wire t^6 = t^4 t^2 beside it; the even lane stores t^2 and t^7, the odd lane t^4 and t^6.  Three
// multiplications deep instead of four, and half the stores on the chain.  One request's walk: 12 -> 9 ms.
template <bool PAIR>
__global__ void __launch_bounds__(64) k_withdraw_core(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ inputs,
                                                     int depth, size_t n_wires, uint32_t first_gadget_wire, size_t n,
                                                     uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x, g = PAIR ? lane >> 1 : lane;
  const bool odd = PAIR && (threadIdx.x & 1), even = !odd;
  if (g >= n) return;
  const uint8_t* in = inputs + g * (size_t)(W_REC + depth) * 32;
  WireWriterT<false> ww{out + g * n_wires * 32, first_gadget_wire};
  const Fr nullifier = fe_to_mont(fe_load<FrParams>(in));
  const Fr secret = fe_to_mont(fe_load<FrParams>(in + 32));
  const Fr amount = fe_to_mont(fe_load<FrParams>(in + 64));
  const Fr recipient = fe_to_mont(fe_load<FrParams>(in + 96));
  const uint64_t index = *reinterpret_cast<const uint64_t*>(in + 160);
  const Fr token = fe_to_mont(fe_load<FrParams>(in + 192));
  const Fr chain_id = fe_to_mont(fe_load<FrParams>(in + 224));
  ww.put_if(even, 0, Fr::one());
  ww.put_if(even, 3, recipient);
  ww.put_if(even, 4, amount);
  ww.put_if(even, 5, token);
  ww.put_if(even, 6, chain_id);
  ww.put_if(even, 7, nullifier);
  ww.put_if(even, 8, secret);
  for (int l = 0; l < depth; l++) {
    ww.put_if(even, 9 + l, fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + l) * 32)));
    ww.put_if(even, 9 + depth + l, ((index >> l) & 1) ? Fr::one() : Fr::zero());
  }
  ww.put_if(even, 9 + 2 * depth, fe_sqr(recipient));
  ww.put_if(even, 10 + 2 * depth, fe_sqr(chain_id));
  // gadget 0: inner = H(nullifier, secret); 1: asset = H(amount, token); 2: leaf = H(inner, asset);
  // 3: nullifier_hash = H(nullifier, 0) -> wire 2; gadget 4 + l: level l of the path, output -> next cur (wire 1 = root for
  // the last level)
  Fr cur = Fr::zero(), inner = Fr::zero();
#pragma unroll 1
  for (int h = 0; h < 4 + depth; h++) {
    Fr l_in, r_in;
    int out_wire = -1;
    if (h == 0) {
      l_in = nullifier; r_in = secret;
    } else if (h == 1) {
      l_in = amount; r_in = token;
    } else if (h == 2) {
      l_in = inner; r_in = cur;
    } else if (h == 3) {
      l_in = nullifier; r_in = Fr::zero(); out_wire = 2;
    } else {
      const int lvl = h - 4;
      const Fr sib = fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + lvl) * 32));
      const bool right_child = (index >> lvl) & 1;
      l_in = right_child ? sib : cur;
      r_in = right_child ? cur : sib;
      ww.push_if(even, l_in);  // the `left` selector wire
      if (lvl == depth - 1) out_wire = 1;
    }
    // MultiMiMC7([l, r], key 0): k1 = l + E_0(l); out = k1 + r + E_k1(r), with E_k(x) = x_91 + k
    Fr k = Fr::zero(), x = l_in, k1 = Fr::zero();
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
#pragma unroll 1
      for (int i = 0; i < MIMC7_ROUNDS; i++) {
        Fr t = fe_add3_weak(x, k, mimc7_const(consts, i));  // < 5N, only ever multiplied
        Fr t2 = PAIR ? OG_MIMC_LAT_SQR(t) : fe_sqr(t);
        if constexpr (PAIR) {  // (the latency forms of the products, field.hip.h: a request's walk is one wave waiting for itself)
          const Fr u = OG_MIMC_LAT_MUL(t2, pair_select(odd, t, t2));                    // even: t^4        odd: t^3
          const Fr v = pair_swap(u);                                           // even: t^3        odd: t^4
          const Fr y = OG_MIMC_LAT_MUL(pair_select(odd, v, u), pair_select(odd, t2, v));  // even: t^4 t^3    odd: t^4 t^2 = t^6
          x = pair_select(odd, pair_swap(y), y);                               // t^7 in both
          ww.put(ww.w + (odd ? 1 : 0), pair_select(odd, v, t2));               // t^2 | t^4
          ww.put(ww.w + (odd ? 2 : 3), y);                                     // t^7 | t^6
          ww.w += 4;
        } else {
          Fr t4 = fe_sqr(t2);
          Fr t6 = fe_mul(t4, t2);
          x = fe_mul(t6, t);
          ww.push(t2);
          ww.push(t4);
          ww.push(t6);
          ww.push(x);
        }
      }
      if (p == 0) {
        k1 = fe_add(l_in, x);
        ww.push_if(even, k1);
        k = k1;
        x = r_in;
      }
    }
    const Fr hout = fe_add(fe_add(fe_dbl(k1), r_in), x);
    if (out_wire < 0) ww.push_if(even, hout); else ww.put_if(even, (uint32_t)out_wire, hout);
    if (h == 0) inner = hout;
    if (h != 3) cur = hout;
  }
}

// ---- the walk of ONE request: a wave per proof (round 4) ------------------------------------------------------------------
// k_withdraw_core<true> gives a proof two lanes and walks its 36 hashes = 72 permutations one after the other: ~9.5 ms, the
// largest part of a single request.  The permutations are not all dependent:
//   * inner = H(nullifier, secret), asset = H(amount, token) and nullifier_hash = H(nullifier, 0) are independent of one
//     another (and the first permutation of the last, E_0(nullifier), IS the first of the first);
//   * MultiMiMC7([l, r]) = k1 + r + E_k1(r) with k1 = l + E_0(l): on a level where the path node is the RIGHT input, the first
//     permutation E_0(sibling) does not depend on the path at all.
// So a proof gets a whole wave, 32 lane pairs.  Phase A: pair 0 walks E_0(nullifier), pair 1 E_0(amount), pairs 2.. the
// E_0(sibling) of the levels whose path node is the right input; phase B: pairs 0 / 1 / 2 the second permutations of inner /
// asset / nullifier_hash, the remaining pairs the sibling jobs that did not fit phase A.  Every pair runs exactly one
// permutation per phase with one instruction stream (a pair without a job computes on zeros and stores nothing), results meet
// in LDS, and pair 0 goes on alone: leaf (2 permutations), then 2 permutations per left level and ONE per right level.
// 4 + depth + (number of left levels) permutations on the chain instead of 8 + 2 depth: 52 instead of 72 for a random leaf of a
// depth-32 tree (68 for leaf 0).  Every lane stores the wires of the permutations it walks, in the same places.
constexpr int WLAT_PAIRS = 32;
constexpr int WLAT_JOBS_A = WLAT_PAIRS - 2, WLAT_JOBS_B = WLAT_PAIRS - 3;  // sibling jobs per phase: 30 + 29 >= depth 59

__global__ void __launch_bounds__(64) k_withdraw_core_lat(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ inputs, int depth,
                                                         size_t n_wires, uint32_t first_gadget_wire, size_t n, uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  __shared__ uint32_t xch[64 + 4][9];  // k1 of sibling job j at [j]; [64] inner, [65] asset, [66] k1 of inner
  const size_t g = blockIdx.x;
  if (g >= n) return;
  const int lane = threadIdx.x, pair = lane >> 1;
  const bool odd = lane & 1, even = !odd;
  const uint8_t* in = inputs + g * (size_t)(W_REC + depth) * 32;
  uint8_t* z = out + g * n_wires * 32;
  auto put = [&](bool mine, uint32_t wire, const Fr& v) { if (mine) fe_store(z + (size_t)wire * 32, v); };
  const uint64_t index = *reinterpret_cast<const uint64_t*>(in + 160);
  // wire layout of a gadget (oracle/py/withdraw.py): [selector (path levels only)] | 364 wires of E_0 | k1 | 364 wires of E_k1 | out
  // gadgets 0..2 take 730 wires, nullifier_hash 729 (its output is a public wire), a path level 731 (the selector in front)
  auto gadget_base = [&](int h) -> uint32_t { return first_gadget_wire + (h < 4 ? (uint32_t)h * 730u : 2919u + 731u * (uint32_t)(h - 4)); };
  // the j-th level (ascending) whose path node is the RIGHT input, or -1
  auto right_level = [&](int j) -> int {
    int seen = 0;
    for (int l = 0; l < depth; l++)
      if ((index >> l) & 1) {
        if (seen == j) return l;
        seen++;
      }
    return -1;
  };
  // one permutation by this pair: E_k(x) over 91 rounds, the round wires stored at wbase (and, dup != 0, at dup as well)
  auto permute = [&](Fr x, const Fr& k, bool active, uint32_t wbase, uint32_t dup) -> Fr {
#pragma unroll 1
    for (int i = 0; i < MIMC7_ROUNDS; i++) {
      const Fr t = fe_add3_weak(x, k, mimc7_const(consts, i));
      const Fr t2 = fe_sqr(t);
      const Fr u = fe_mul(t2, pair_select(odd, t, t2));                        // even: t^4        odd: t^3
      const Fr v = pair_swap(u);                                               // even: t^3        odd: t^4
      const Fr y = fe_mul(pair_select(odd, v, u), pair_select(odd, t2, v));    // even: t^7        odd: t^6
      x = pair_select(odd, pair_swap(y), y);
      if (active) {
        const uint32_t w = wbase + 4u * (uint32_t)i;
        fe_store(z + (size_t)(w + (odd ? 1 : 0)) * 32, pair_select(odd, v, t2));  // t^2 | t^4
        fe_store(z + (size_t)(w + (odd ? 2 : 3)) * 32, y);                        // t^7 | t^6
        if (dup) {
          const uint32_t w2 = dup + 4u * (uint32_t)i;
          fe_store(z + (size_t)(w2 + (odd ? 1 : 0)) * 32, pair_select(odd, v, t2));
          fe_store(z + (size_t)(w2 + (odd ? 2 : 3)) * 32, y);
        }
      }
    }
    return x;
  };
  const Fr zero = Fr::zero();
  const Fr nullifier = fe_to_mont(fe_load<FrParams>(in)), secret = fe_to_mont(fe_load<FrParams>(in + 32));
  const Fr amount = fe_to_mont(fe_load<FrParams>(in + 64)), recipient = fe_to_mont(fe_load<FrParams>(in + 96));
  const Fr token = fe_to_mont(fe_load<FrParams>(in + 192)), chain_id = fe_to_mont(fe_load<FrParams>(in + 224));
  if (pair == 0) {  // the wires that are inputs or squares of inputs
    put(even, 0, Fr::one()); put(even, 3, recipient); put(even, 4, amount); put(even, 5, token); put(even, 6, chain_id);
    put(even, 7, nullifier); put(even, 8, secret);
    put(odd, 9 + 2 * depth, fe_sqr(recipient)); put(odd, 10 + 2 * depth, fe_sqr(chain_id));
  }
  for (int l = pair; l < depth; l += WLAT_PAIRS) {
    put(even, 9 + l, fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + l) * 32)));
    put(odd, 9 + depth + l, ((index >> l) & 1) ? Fr::one() : Fr::zero());
  }
  // ---------------- phase A ----------------
  Fr xa = zero;
  bool act = false;
  uint32_t wb = 0, dup = 0;
  int job = -1, lvl = -1;
  if (pair == 0) { xa = nullifier; act = true; wb = gadget_base(0); dup = gadget_base(3); }
  else if (pair == 1) { xa = amount; act = true; wb = gadget_base(1); }
  else {
    job = pair - 2;
    lvl = right_level(job);
    if (lvl >= 0) {
      xa = fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + lvl) * 32));
      act = true;
      wb = gadget_base(4 + lvl) + 1;
      put(even, gadget_base(4 + lvl), xa);  // the level's `left` selector wire: the sibling
    }
  }
  Fr k1 = fe_add(xa, permute(xa, zero, act, wb, dup));  // l + E_0(l)
  if (act) {
    put(even, wb + 364, k1);
    if (dup) put(odd, dup + 364, k1);
    if (even) lds_put9(xch[pair == 0 ? 66 : (pair == 1 ? 67 : job)], k1);
  }
  __syncthreads();
  // ---------------- phase B ----------------
  Fr xb = zero, kb = zero, rin = zero;
  act = false; wb = 0;
  int out_slot = -1;
  if (pair == 0) { xb = secret; rin = secret; kb = k1; act = true; wb = gadget_base(0) + 365; out_slot = 64; }
  else if (pair == 1) { xb = token; rin = token; kb = k1; act = true; wb = gadget_base(1) + 365; out_slot = 65; }
  else if (pair == 2) { xb = zero; rin = zero; kb = lds_get9(xch[66]); act = true; wb = gadget_base(3) + 365; }
  else {
    job = WLAT_JOBS_A + pair - 3;
    lvl = right_level(job);
    if (lvl >= 0) {
      xb = fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + lvl) * 32));
      act = true;
      wb = gadget_base(4 + lvl) + 1;
      put(even, gadget_base(4 + lvl), xb);
    }
  }
  const Fr xo = permute(xb, kb, act, wb, 0);
  if (pair <= 2) {  // second permutations: out = 2 k1 + r + x_91
    const Fr hout = fe_add(fe_add(fe_dbl(kb), rin), xo);
    if (pair == 2) put(even, 2, hout);                    // nullifier_hash: a public wire
    else {
      put(even, wb + 364, hout);
      if (even) lds_put9(xch[out_slot], hout);
    }
  } else if (act) {  // a late sibling job: k1 = l + E_0(l)
    const Fr k1b = fe_add(xb, xo);
    put(even, wb + 364, k1b);
    if (even) lds_put9(xch[job], k1b);
  }
  __syncthreads();
  if (pair != 0) return;
  // ---------------- the chain: pair 0 alone ----------------
  auto hash_rest = [&](const Fr& l_in, const Fr& r_in, bool have_k1, const Fr& k1_in, uint32_t base, int out_wire) -> Fr {
    Fr kk = k1_in;
    if (!have_k1) {
      kk = fe_add(l_in, permute(l_in, zero, true, base, 0));
      put(even, base + 364, kk);
    }
    const Fr xr = permute(r_in, kk, true, base + 365, 0);
    const Fr h = fe_add(fe_add(fe_dbl(kk), r_in), xr);
    put(even, out_wire >= 0 ? (uint32_t)out_wire : base + 729, h);
    return h;
  };
  Fr cur = hash_rest(lds_get9(xch[64]), lds_get9(xch[65]), false, zero, gadget_base(2), -1);  // leaf = H(inner, asset)
  int rj = 0;
#pragma unroll 1
  for (int l = 0; l < depth; l++) {
    const uint32_t gb = gadget_base(4 + l);
    const int out_wire = l == depth - 1 ? 1 : -1;
    if ((index >> l) & 1) {  // the path node is the right input: E_0(sibling) and the selector wire are already there
      cur = hash_rest(zero, cur, true, lds_get9(xch[rj]), gb + 1, out_wire);
      rj++;
    } else {
      const Fr sib = fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + l) * 32));
      put(even, gb, cur);  // `left` selector wire: the path node
      cur = hash_rest(cur, sib, false, zero, gb + 1, out_wire);
    }
  }
}

// ---- the walk in the wave-wide form (round 6; calls of at most 512 requests): ONE permutation per wave, three launches -----------------------------------------------
// field_w9.hip.h: a Montgomery product with its nine limbs in nine lanes is 616 cycles on a lone wave against 904 for the lane-local
// one, and in that form a round's additions, constant and stores are ONE instruction each instead of nine: a MiMC7 round -- t = x +
// k + c, four products (t^2, t^4, t^6, t^7), four stores -- is ~2 600 cycles against the lane pair's ~3 400; and with a second row of
// the wave as the lane group for the pair's t^3 || t^4 (ROWS: mimc7.hip.h w9_mimc7_round, the form the launches use) it is three
// products deep, t^6 computed beside t^7.  A wave holds one permutation, so the independent permutations of a proof (what
// k_withdraw_core_lat gives its 32 lane pairs) become one-wave WORKGROUPS that land on different CUs, in three launches:
//   k_w9_first   grid n x (3 + depth): E_0(nullifier) (wires of gadgets 0 and 3), E_0(amount), E_0(sibling_l) of every level whose path
//                node is the RIGHT input, and one workgroup for the wires that are inputs or squares of inputs
//   k_w9_second  grid n x 3: the second permutations of inner = H(nullifier, secret), asset = H(amount, token), nullifier_hash
//   k_w9_chain   grid n: leaf = H(inner, asset), then per level two permutations (left) or one (right): 4 + depth + #left on the chain
// Values pass between the launches as nine lazy limbs (xch).  Wires are stored as nine limbs too (36 B, one store instruction per
// wire: wl) and k_wires_from_limbs takes them to the canonical 32 bytes afterwards, in parallel -- where k_wires_from_mont takes
// the other kernels' 32-byte Montgomery values.  Bounds (multiples of N; the strict products accept limbs < 2^31 and a b < 169 N^2):
// forms 0 and 1: x < 2, c < 2, key k1 = l + x_91 < 4  =>  t < 8, t^2 .. t^7 fine (t^6 t: 2 x 8); a hash's output 2 k1 + r + x_91 < 12
// is carried and multiplied by one (< 2 again, limbs < 2^29 + 32) before it is anybody's input.  Form 2 (the 32-bit digit: what runs):
// mimc7.hip.h w9_mimc7_round -- x < 8.5, k1 < 10.5, t < 21, a hash's output < 32, through the same STRICT product by one: < 2.
constexpr int W9_XCH = 8;  // xch slots per proof beyond the levels: [depth + 0] k1 of inner / nullifier_hash, [1] k1 of asset, [2] inner, [3] asset,
                           // [4 .. 7]: 36 words nobody reads -- where the lanes without a limb store (w9_permute `dump`)
// `tid` = threadIdx.x (row 0 stores); `lane` = the limb a lane holds (w9_row_limb(tid) with ROWS -- rows 0 and 1 load --, tid without)
__device__ __forceinline__ void w9_store(uint32_t* wl, uint32_t wire, uint32_t v, int tid) { if (tid < 9) wl[(size_t)wire * 9 + tid] = v; }
__device__ __forceinline__ uint32_t w9_load(const uint32_t* p, int lane) { return lane < 9 ? p[lane] : 0u; }
// E_k(x) without the final + k: 91 rounds, the round wires at wbase (and at dup, if non-zero)
// With rows, a round's four wires leave in TWO store instructions: row 0 holds t^2 and t^4, row 1 t^6 and t^7 (w9_mimc7_round), the
// four wires are 36 consecutive words, so lane (row, limb) points at word 18 row + limb of the round's block, stores once there and
// once nine words on (a vector store costs a lone wave its address pass whether nine lanes are active or eighteen).
// The stores are UNCONDITIONAL -- lanes without a limb write a dump word (`dump`: a spare xch slot of the proof) -- because a store
// behind a branch makes the compiler wait for ALL memory operations before the next round's constant is used (vmcnt(0): the
// count is unknown on the path around the branch), i.e. for the stores' own completion, every round; without the branch the
// wait is for the constant alone (asked for a round ahead: mimc7.hip.h w9_mimc7_rounds) and the stores drain under the products.
template <int FORM, bool DUP>
__device__ __forceinline__ uint32_t w9_permute(const uint32_t* __restrict__ consts9, uint32_t x, uint32_t k, uint32_t nj, int tid, int lane,
                                               uint32_t* __restrict__ wl, uint32_t wbase, uint32_t dup, uint32_t* __restrict__ dump) {
  const int cl = lane < 15 ? lane : 15;
  const bool row1 = (tid & 16) != 0;
  const bool stores = FORM ? lane < 9 : tid < 9;
  const uint32_t off = (uint32_t)(lane < 9 ? lane : 0) + (FORM && row1 ? 18u : 0u);
  uint32_t* p = stores ? wl + (size_t)wbase * 9 + off : dump;
  uint32_t* pd = stores ? wl + (size_t)dup * 9 + off : dump;
  const ptrdiff_t step = stores ? 36 : 0;
  uint32_t c = consts9[cl];
#pragma unroll 1
  for (int i = 0; i < MIMC7_ROUNDS; i++, p += step, pd += step) {
    const uint32_t c_next = consts9[(i + 1 < MIMC7_ROUNDS ? i + 1 : i) * 16 + cl];
    uint32_t t2, t4, t6, t6r0;
    w9_mimc7_round<FORM>(x + k + c, nj, row1, t2, t4, t6, t6r0, x);
    c = c_next;
    if (FORM) {
      const uint32_t s0 = row1 ? t6r0 : t2, s1 = row1 ? t6 : t4;  // row 0: wires w, w + 1 = t^2, t^4; row 1: w + 2, w + 3 = t^6, t^7
      p[0] = s0; p[9] = s1;
      if (DUP) { pd[0] = s0; pd[9] = s1; }
    } else {
      p[0] = t2; p[9] = t4; p[18] = t6; p[27] = x;
      if (DUP) { pd[0] = t2; pd[9] = t4; pd[18] = t6; pd[27] = x; }
    }
  }
  return x;
}
// a sum of a few elements (limbs < 2^32, value < 169 N) -> the same value mod N below 2 N with limbs < 2^29 + 32
__device__ __forceinline__ uint32_t w9_renorm(uint32_t v, uint32_t nj, int lane) {
  return w9_mul<FrParams>(w9_uniform(FrParams::ONE), w9_carry(v, lane), nj);
}
__device__ __forceinline__ uint32_t w9_gadget_base(uint32_t fgw, int h) { return fgw + (h < 4 ? (uint32_t)h * 730u : 2919u + 731u * (uint32_t)(h - 4)); }
__device__ __forceinline__ void w9_put_fe(uint32_t* wl, uint32_t wire, const Fr& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) wl[(size_t)wire * 9 + i] = v.l[i];
}

template <int FORM>
__global__ void __launch_bounds__(64) k_w9_first(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ inputs, int depth, size_t n_core,
                                                uint32_t fgw, uint32_t* __restrict__ wl_all, uint32_t* __restrict__ xch_all) {
  OG_FILLER_PRIO();
  const int jobs = 3 + depth, tid = threadIdx.x, lane = FORM ? w9_row_limb(tid) : tid;
  const size_t g = blockIdx.x / jobs;
  const int job = (int)(blockIdx.x % jobs);
  const uint8_t* in = inputs + g * (size_t)(W_REC + depth) * 32;
  uint32_t* wl = wl_all + g * n_core * 9;
  uint32_t* xch = xch_all + g * (size_t)(depth + W9_XCH) * 9;
  const uint64_t index = *reinterpret_cast<const uint64_t*>(in + 160);
  if (job == 2 + depth) {  // the wires that are inputs or squares of inputs: lane-local values, a lane per wire
    const int lane = tid;
    const Fr recipient = fe_to_mont(fe_load<FrParams>(in + 96)), chain_id = fe_to_mont(fe_load<FrParams>(in + 224));
    if (lane == 0) w9_put_fe(wl, 0, Fr::one());
    if (lane == 1) w9_put_fe(wl, 3, recipient);
    if (lane == 2) w9_put_fe(wl, 4, fe_to_mont(fe_load<FrParams>(in + 64)));
    if (lane == 3) w9_put_fe(wl, 5, fe_to_mont(fe_load<FrParams>(in + 192)));
    if (lane == 4) w9_put_fe(wl, 6, chain_id);
    if (lane == 5) w9_put_fe(wl, 7, fe_to_mont(fe_load<FrParams>(in)));
    if (lane == 6) w9_put_fe(wl, 8, fe_to_mont(fe_load<FrParams>(in + 32)));
    if (lane == 7) w9_put_fe(wl, 9 + 2 * depth, fe_sqr(recipient));
    if (lane == 8) w9_put_fe(wl, 10 + 2 * depth, fe_sqr(chain_id));
    for (int l = lane; l < depth; l += 64) {
      w9_put_fe(wl, 9 + l, fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + l) * 32)));
      w9_put_fe(wl, 9 + depth + l, ((index >> l) & 1) ? Fr::one() : Fr::zero());
    }
    return;
  }
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  const uint8_t* src = in;  // job 0: nullifier
  uint32_t wbase = w9_gadget_base(fgw, 0), dup = w9_gadget_base(fgw, 3);
  int slot = depth;
  if (job == 1) { src = in + 64; wbase = w9_gadget_base(fgw, 1); dup = 0; slot = depth + 1; }
  if (job >= 2) {
    const int lvl = job - 2;
    if (!((index >> lvl) & 1)) return;  // the path node is the LEFT input of this level: its first permutation is the chain's
    src = in + (size_t)(W_REC + lvl) * 32;
    wbase = w9_gadget_base(fgw, 4 + lvl) + 1;
    dup = 0;
    slot = lvl;
  }
  const uint32_t l_in = w9_spread(fe_to_mont(fe_load<FrParams>(src)), lane);
  if (job >= 2) w9_store(wl, wbase - 1, l_in, tid);  // the level's `left` selector wire: the sibling
  uint32_t* dump = xch + (size_t)(depth + 4) * 9;
  const uint32_t k1 = l_in + (dup ? w9_permute<FORM, true>(consts9, l_in, 0u, nj, tid, lane, wl, wbase, dup, dump)
                                  : w9_permute<FORM, false>(consts9, l_in, 0u, nj, tid, lane, wl, wbase, 0u, dump));  // l + E_0(l): < 4 N, lazy limbs
  w9_store(wl, wbase + 364, k1, tid);
  if (dup) w9_store(wl, dup + 364, k1, tid);
  if (tid < 9) xch[(size_t)slot * 9 + lane] = k1;
}

template <int FORM>
__global__ void __launch_bounds__(64) k_w9_second(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ inputs, int depth, size_t n_core,
                                                 uint32_t fgw, uint32_t* __restrict__ wl_all, uint32_t* __restrict__ xch_all) {
  OG_FILLER_PRIO();
  const int tid = threadIdx.x, lane = FORM ? w9_row_limb(tid) : tid;
  const size_t g = blockIdx.x / 3;
  const int job = (int)(blockIdx.x % 3);  // 0 inner, 1 asset, 2 nullifier_hash
  const uint8_t* in = inputs + g * (size_t)(W_REC + depth) * 32;
  uint32_t* wl = wl_all + g * n_core * 9;
  uint32_t* xch = xch_all + g * (size_t)(depth + W9_XCH) * 9;
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  const uint32_t k1 = w9_load(xch + (size_t)(depth + (job == 1 ? 1 : 0)) * 9, lane);
  uint32_t r_in = 0;
  if (job == 0) r_in = w9_spread(fe_to_mont(fe_load<FrParams>(in + 32)), lane);   // secret
  if (job == 1) r_in = w9_spread(fe_to_mont(fe_load<FrParams>(in + 192)), lane);  // token
  const uint32_t base = w9_gadget_base(fgw, job == 2 ? 3 : job);
  const uint32_t xr = w9_permute<FORM, false>(consts9, r_in, k1, nj, tid, lane, wl, base + 365, 0u, xch + (size_t)(depth + 4) * 9);
  const uint32_t hout = w9_renorm(2u * k1 + r_in + xr, nj, lane);
  if (job == 2) { w9_store(wl, 2, hout, tid); return; }  // nullifier_hash: a public wire
  w9_store(wl, base + 729, hout, tid);
  if (tid < 9) xch[(size_t)(depth + 2 + job) * 9 + lane] = hout;
}

template <int FORM>
__global__ void __launch_bounds__(64) k_w9_chain(const uint32_t* __restrict__ consts9, const uint8_t* __restrict__ inputs, int depth, size_t n_core,
                                                uint32_t fgw, uint32_t* __restrict__ wl_all, uint32_t* __restrict__ xch_all) {
  OG_FILLER_PRIO();
  const int tid = threadIdx.x, lane = FORM ? w9_row_limb(tid) : tid;
  const size_t g = blockIdx.x;
  const uint8_t* in = inputs + g * (size_t)(W_REC + depth) * 32;
  uint32_t* wl = wl_all + g * n_core * 9;
  uint32_t* xch = xch_all + g * (size_t)(depth + W9_XCH) * 9;
  uint32_t* dump = xch + (size_t)(depth + 4) * 9;
  const uint32_t nj = w9_modulus_limb<FrParams>(lane);
  const uint64_t index = *reinterpret_cast<const uint64_t*>(in + 160);
  // H(l, r) from its first permutation on (have_k1: that one is already there), wires at base; the output below 2 N
  auto hash_rest = [&](uint32_t l_in, uint32_t r_in, bool have_k1, uint32_t k1, uint32_t base, int out_wire) -> uint32_t {
    if (!have_k1) {
      k1 = l_in + w9_permute<FORM, false>(consts9, l_in, 0u, nj, tid, lane, wl, base, 0u, dump);
      w9_store(wl, base + 364, k1, tid);
    }
    const uint32_t xr = w9_permute<FORM, false>(consts9, r_in, k1, nj, tid, lane, wl, base + 365, 0u, dump);
    const uint32_t h = w9_renorm(2u * k1 + r_in + xr, nj, lane);
    w9_store(wl, out_wire >= 0 ? (uint32_t)out_wire : base + 729, h, tid);
    return h;
  };
  uint32_t cur = hash_rest(w9_load(xch + (size_t)(depth + 2) * 9, lane), w9_load(xch + (size_t)(depth + 3) * 9, lane), false, 0u,
                           w9_gadget_base(fgw, 2), -1);  // leaf = H(inner, asset)
#pragma unroll 1
  for (int l = 0; l < depth; l++) {
    const uint32_t gb = w9_gadget_base(fgw, 4 + l);
    const int out_wire = l == depth - 1 ? 1 : -1;
    if ((index >> l) & 1) {  // the path node is the right input: E_0(sibling), k1 and the selector wire are k_w9_first's
      cur = hash_rest(0u, cur, true, w9_load(xch + (size_t)l * 9, lane), gb + 1, out_wire);
    } else {
      w9_store(wl, gb, cur, tid);  // `left` selector wire: the path node
      cur = hash_rest(cur, w9_spread(fe_to_mont(fe_load<FrParams>(in + (size_t)(W_REC + l) * 32)), lane), false, 0u, gb + 1, out_wire);
    }
  }
}

// wires [0, n_core) of every proof: nine lazy Montgomery limbs (what the w9 kernels left in wl) -> canonical 32 bytes in z
__global__ void __launch_bounds__(256) k_wires_from_limbs(const uint32_t* __restrict__ wl, uint8_t* __restrict__ out, size_t n_wires, uint32_t n_core) {
  OG_FILLER_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_core) return;
  const uint32_t* p = wl + ((size_t)blockIdx.y * n_core + i) * 9;
  uint32_t t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = p[k];
  Fr o = Fr::zero();
  o.l[0] = 1;
  fe_store(out + ((size_t)blockIdx.y * n_wires + i) * 32, fe_canon(fe_mul(fe_from_lazy_limbs<FrParams>(t), o)));
}

// wires [0, n_core) of every proof: Montgomery -> canonical (what k_withdraw_core left behind)
// mult = 1: the kernels' form (x 2^261); mult = 32: the host walk's form (x 2^256: stored * 2^5 * 2^-261 = stored / 2^256)
__global__ void __launch_bounds__(256) k_wires_from_mont(uint8_t* __restrict__ out, size_t n_wires, uint32_t n_core, uint32_t mult) {
  OG_FILLER_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_core) return;
  uint8_t* p = out + ((size_t)blockIdx.y * n_wires + i) * 32;
  Fr o = Fr::zero();
  o.l[0] = mult;
  fe_store(p, fe_canon(fe_mul(fe_load<FrParams>(p), o)));
}

// x = seed + wire; v = x^5; boolean parity wire when wire % 5 == 0.  Returns Montgomery form.
__device__ __forceinline__ Fr pad_value(const Fr& seed_canon, uint32_t wire) {
  Fr x = fe_to_mont(fe_add(seed_canon, fe_from_u32<FrParams>(wire)));
  Fr x2 = fe_sqr(x);
  Fr v = fe_mul(fe_sqr(x2), x);
  if (wire % 5 == 0) {
    Fr c = fe_from_mont(v);
    return (c.l[0] & 1) ? Fr::one() : Fr::zero();
  }
  return v;
}

__global__ void __launch_bounds__(256) k_withdraw_pad(const uint8_t* __restrict__ inputs, int depth, size_t n_wires, uint32_t pad_base,
                                                     uint32_t n_pad3, uint32_t n_pad2, uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  const uint32_t n_seg = (n_pad2 + PAD_SEGMENT - 1) / PAD_SEGMENT;
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_pad3 + n_seg) return;
  const size_t g = blockIdx.y;
  const Fr seed = fe_load<FrParams>(inputs + g * (size_t)(W_REC + depth) * 32 + 128);
  WireWriter ww{out + g * n_wires * 32, 0};
  const Fr one = Fr::one(), two = fe_dbl(one);
  if (u < n_pad3) {
    ww.w = pad_base + 3 * u;
    Fr p = pad_value(seed, ww.w), q = pad_value(seed, ww.w + 1);
    ww.push(p);
    ww.push(q);
    ww.push(fe_mul(fe_add(p, one), fe_add(q, two)));
  } else {
    const uint32_t s = u - n_pad3;
    const uint32_t g0 = s * PAD_SEGMENT, g1 = g0 + PAD_SEGMENT < n_pad2 ? g0 + PAD_SEGMENT : n_pad2;
    ww.w = pad_base + 3 * n_pad3 + 2 * g0;
    Fr prev = one;  // the constant-one wire
    for (uint32_t k = g0; k < g1; k++) {
      Fr p = pad_value(seed, ww.w);
      ww.push(p);
      prev = fe_mul(fe_add(p, one), fe_add(prev, two));
      ww.push(prev);
    }
  }
}

// Boundary check of the input records (they arrive from HTTP: /root/reference/src/services/api_services/withdraw.rs:15-19):
// every field element must be the canonical encoding (< r) -- a value >= r would be reduced silently, i.e. two encodings of one
// nullifier -- and the index must fit the tree (u64 in the low bytes, < 2^depth, upper bytes zero: the circuit reads only its low
// `depth` bits).  That recipient and token are 160-bit addresses is the gate contract's check (contracts/OwshenWithdrawGate.sol):
// the statement itself takes any field element.  bad[g] = lowest offending field of record g (atomicMin;
// the caller initialises it to 0xffffffff): 0 nullifier, 1 secret, 2 amount, 3 recipient, 4 pad_seed, 5 index, 6 token,
// 7 chain_id, 8 + l sibling l.
__global__ void __launch_bounds__(64) k_check_records(const uint8_t* __restrict__ inputs, int depth, size_t n, uint32_t* __restrict__ bad) {
  OG_FILLER_PRIO();
  const size_t g = blockIdx.y;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n || f >= (uint32_t)(W_REC + depth)) return;
  const uint8_t* p = inputs + (g * (size_t)(W_REC + depth) + f) * 32;
  bool ok = fe_lt_modulus(fe_load<FrParams>(p));
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  if (f == 5) {  // index
    ok = ok && (w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0;
    const uint64_t idx = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    ok = ok && (depth >= 64 || (idx >> depth) == 0);
  }
  if (!ok) atomicMin(&bad[g], f);
}

const char* withdraw_field_name(uint32_t f) {
  static const char* names[W_REC] = {"nullifier", "secret", "amount", "recipient", "pad_seed", "index", "token", "chain_id"};
  return f < (uint32_t)W_REC ? names[f] : "sibling";
}

int withdraw_check_records(og_ctx* ctx, int depth, const uint8_t* inputs_d, size_t n, uint32_t* bad_d) {
  OG_REQUIRE(depth >= 1 && depth <= 64, "withdraw: depth must be 1..64");
  OG_REQUIRE(n <= 65535, "withdraw: at most 65535 records per call");  // (the record index is grid.y; checked HERE so that the caller gets this message, not a launch error)
  if (n == 0) return OG_OK;
  hipLaunchKernelGGL(k_check_records, dim3(grid_for(W_REC + depth, 64), (unsigned)n), dim3(64), 0, ctx->stream, inputs_d, depth, n, bad_d);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

int withdraw_shape_query(int depth, uint64_t n_pad3, uint64_t n_pad2, uint64_t out[3]) {
  OG_REQUIRE(depth >= 1 && depth <= 64, "withdraw: depth must be 1..64");
  WithdrawShape s = withdraw_shape(depth, n_pad3, n_pad2);
  OG_REQUIRE(s.n_wires < (1ull << 31), "withdraw: too many wires");
  out[0] = s.n_wires; out[1] = s.n_constraints; out[2] = W_PUB;
  return OG_OK;
}

// ---- the walk on the HOST (round 6; opt-in: og_set_host_chains) ---------------------------------------------------------------------
// One request's Merkle walk is a chain of ~19 000 dependent Montgomery products (52 of its 72 permutations, k_withdraw_core_lat),
// and a chain runs at the latency of ONE product: ~900 shader cycles = 0.42 us on a lone wave -- a wave issues an instruction
// every ~4.8 cycles whatever it depends on, and the product is 205 of them (the wave-wide form of field_w9.hip.h
// takes 616 cycles and the k_w9_* kernels above walk with it: 7.4 -> 5.7 ms for one request -- the GPU's own best).
// A server core with 64-bit multipliers runs the same product -- the SAME code: the field layer is OG_HD, og_verify already runs
// it on the host -- in 20-50 ns.  So for calls of a handful of requests (the reference's handler proves ONE per HTTP call,
// /root/reference/src/services/api_services/withdraw.rs:27-71) the host may walk the chains: records down (1.3 KB each), one
// thread per request fills the core wires in Montgomery form exactly as k_withdraw_core<false> does, wire for wire, the
// wires go up (0.84 MB per request) and everything after -- the conversion, the padding gates, the sparse products, the quotient,
// the MSMs -- is the GPU's as before.  Not a fallback (the call still needs the GPU, and a batch never takes this path) and off
// by default: og_set_host_chains(ctx, max_requests) turns it on for calls of at most that many requests.  Same bytes as the
// kernels (tests/withdraw_cases.py, interpreter and GPU).
// one proof's core wires, k_withdraw_core<false> wire for wire (same gadget order, same wire indices), values in the host's form
static void withdraw_core_host(const H4* rc, const uint8_t* in, int depth, uint32_t first_gadget_wire, uint8_t* z) {
  const H4Field& f = h4_field();
  auto put = [&](uint32_t wire, const H4& v) { memcpy(z + (size_t)wire * 32, v.v, 32); };
  auto ld = [&](const uint8_t* p) { return h4_to_mont(f, h4_load(p)); };
  const H4 zero = {{0, 0, 0, 0}};
  const H4 nullifier = ld(in), secret = ld(in + 32), amount = ld(in + 64), recipient = ld(in + 96);
  uint64_t index = 0;
  memcpy(&index, in + 160, 8);
  const H4 token = ld(in + 192), chain_id = ld(in + 224);
  put(0, f.one); put(3, recipient); put(4, amount); put(5, token); put(6, chain_id); put(7, nullifier); put(8, secret);
  for (int l = 0; l < depth; l++) {
    put(9 + l, ld(in + (size_t)(W_REC + l) * 32));
    put(9 + depth + l, ((index >> l) & 1) ? f.one : zero);
  }
  put(9 + 2 * depth, h4_mul(f, recipient, recipient));
  put(10 + 2 * depth, h4_mul(f, chain_id, chain_id));
  uint32_t w = first_gadget_wire;
  H4 cur = zero, inner = zero;
  for (int h = 0; h < 4 + depth; h++) {
    H4 l_in, r_in;
    int out_wire = -1;
    if (h == 0) { l_in = nullifier; r_in = secret; }
    else if (h == 1) { l_in = amount; r_in = token; }
    else if (h == 2) { l_in = inner; r_in = cur; }
    else if (h == 3) { l_in = nullifier; r_in = zero; out_wire = 2; }
    else {
      const int lvl = h - 4;
      const H4 sib = ld(in + (size_t)(W_REC + lvl) * 32);
      const bool right_child = (index >> lvl) & 1;
      l_in = right_child ? sib : cur;
      r_in = right_child ? cur : sib;
      put(w++, l_in);  // the `left` selector wire
      if (lvl == depth - 1) out_wire = 1;
    }
    H4 k = zero, x = l_in, k1 = zero;
    for (int p = 0; p < 2; p++) {
      for (int i = 0; i < MIMC7_ROUNDS; i++) {
        const H4 t = h4_add(f, h4_add(f, x, k), rc[i]);
        const H4 t2 = h4_mul(f, t, t), t4 = h4_mul(f, t2, t2), t6 = h4_mul(f, t4, t2);
        x = h4_mul(f, t6, t);
        put(w, t2); put(w + 1, t4); put(w + 2, t6); put(w + 3, x);
        w += 4;
      }
      if (p == 0) {
        k1 = h4_add(f, l_in, x);
        put(w++, k1);
        k = k1;
        x = r_in;
      }
    }
    const H4 hout = h4_add(f, h4_add(f, h4_add(f, k1, k1), r_in), x);
    if (out_wire < 0) put(w++, hout); else put((uint32_t)out_wire, hout);
    if (h == 0) inner = hout;
    if (h != 3) cur = hout;
  }
}

// records (device) -> the core wires of n proofs in out_d (device, Montgomery form, as the core kernels leave them), via the host
static int withdraw_walk_on_host(og_ctx* ctx, int depth, const WithdrawShape& s, const uint8_t* inputs_d, size_t n, uint8_t* out_d) {
  const size_t rec = (size_t)(W_REC + depth) * 32, core = (size_t)s.pad_base * 32, need = n * (rec + core);
  // the staging block is the context's: the upload of its previous user (another sub-batch of this call on the other lane, or a
  // submitted call still in flight) must have left it before the host writes into it again
  if (ctx->walk_ev) OG_HIP(hipEventSynchronize(ctx->walk_ev));
  else OG_HIP(hipEventCreateWithFlags(&ctx->walk_ev, hipEventDisableTiming));
  if (ctx->walk_stage_bytes < need) {
    if (ctx->walk_stage) (void)hipHostFree(ctx->walk_stage);
    ctx->walk_stage = nullptr;
    ctx->walk_stage_bytes = 0;
    OG_HIP(hipHostMalloc((void**)&ctx->walk_stage, need, 0));
    ctx->walk_stage_bytes = need;
  }
  uint8_t* recs = ctx->walk_stage + n * core;  // (the wires first: 16-byte aligned stores)
  OG_HIP(hipMemcpyAsync(recs, inputs_d, n * rec, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  const std::vector<H4> rc = h4_round_constants(ctx->mimc_consts_canon);  // (91 products: not worth a cache)
  const uint32_t fgw = (uint32_t)s.first_gadget_wire;
  auto walk = [&](size_t g) { withdraw_core_host(rc.data(), recs + g * rec, depth, fgw, ctx->walk_stage + g * core); };
  host_parallel_for(n, walk);  // a thread per request (a call that takes this path is a handful of requests)
  OG_HIP(hipMemcpy2DAsync(out_d, (size_t)s.n_wires * 32, ctx->walk_stage, core, core, n, hipMemcpyHostToDevice, ctx->stream));
  OG_HIP(hipEventRecord(ctx->walk_ev, ctx->stream));
  return OG_OK;
}

int arena_get(og_ctx* ctx, const char* name, size_t bytes, void** out);

int withdraw_witness(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d, size_t n, uint8_t* out_d) {
  OG_REQUIRE(depth >= 1 && depth <= 64, "withdraw: depth must be 1..64");
  WithdrawShape s = withdraw_shape(depth, n_pad3, n_pad2);
  OG_REQUIRE(s.n_wires < (1ull << 31) && n_pad3 < (1ull << 30) && n_pad2 < (1ull << 30), "withdraw: too many wires");
  OG_REQUIRE(n <= 65535, "withdraw: at most 65535 witnesses per call");
  if (n == 0) return OG_OK;
  ProfScope ps(ctx, PROF_WITNESS, (double)n);
  // two lanes per proof (the latency-bound form) unless OG_MIMC_PAIR=0: a sub-batch is at most 256 proofs = 8 waves
  const bool pair = OG_HOOK_INT("OG_MIMC_PAIR", 1) != 0;  // (read per call: tests run both forms)
  // a handful of requests: a wave per proof, the independent permutations side by side (k_withdraw_core_lat); OG_WITNESS_LAT=0 | 1
  // forces either form, OG_WITNESS_LAT_MAX moves the bound (tests, A/B)
  const size_t lat_max = (size_t)OG_HOOK_INT("OG_WITNESS_LAT_MAX", 16);  // (64 requests: the two-lane form is level or better)
  const bool lat = OG_HOOK_SET("OG_WITNESS_LAT") ? OG_HOOK_INT("OG_WITNESS_LAT", 0) != 0 : (pair && n <= lat_max);
  const bool on_host = ctx->host_chains_max > 0 && std::max(n, ctx->call_requests) <= (size_t)ctx->host_chains_max;  // (a sub-batch of a larger call stays on the GPU)
  // the wave-wide form (k_w9_*): one permutation per wave in three launches; OG_WITNESS_W9=0 | 1 forces either way, OG_WITNESS_W9_MAX moves the bound
  // -- for calls of at most 512 requests: there the walk is on the call's critical path (measured, same box, natural statement:
  // witness 7.4 -> 5.7 ms for one request, 9.3 -> 5.8 for 8, 9.3 -> 6.4 for 64, 9.4 -> 6.9 for 512; the call 10.9 -> 8.7, 14.7 -> 11.1,
  // 26.5 -> 23.3, 115 -> 113 ms: profiles/r06k_ab_witness_w9.txt); a wave per PERMUTATION is ~19 x the wave-instructions of the
  // lane-pair form (nine useful lanes of 64), which a throughput batch would pay out of its accumulations
  const bool w9 = !on_host && (OG_HOOK_SET("OG_WITNESS_W9") ? OG_HOOK_INT("OG_WITNESS_W9", 1) != 0
                                                            : std::max(n, ctx->call_requests) <= (size_t)OG_HOOK_INT("OG_WITNESS_W9_MAX", 512));
  if (w9) {
    uint32_t *wl = nullptr, *xch = nullptr;
    OG_TRY(arena_get(ctx, "wit.w9.limbs", n * (size_t)s.pad_base * 36, (void**)&wl));
    OG_TRY(arena_get(ctx, "wit.w9.xch", n * (size_t)(depth + W9_XCH) * 36, (void**)&xch));
    const uint32_t* c9 = (const uint32_t*)ctx->mimc_consts9_d;
    OG_W9_LAUNCH(k_w9_first, w9_rows(), dim3((unsigned)(n * (size_t)(3 + depth))), dim3(64), 0, ctx->stream, c9, inputs_d, depth, (size_t)s.pad_base,
                       (uint32_t)s.first_gadget_wire, wl, xch);
    OG_W9_LAUNCH(k_w9_second, w9_rows(), dim3((unsigned)(n * 3)), dim3(64), 0, ctx->stream, c9, inputs_d, depth, (size_t)s.pad_base,
                       (uint32_t)s.first_gadget_wire, wl, xch);
    OG_W9_LAUNCH(k_w9_chain, w9_rows(), dim3((unsigned)n), dim3(64), 0, ctx->stream, c9, inputs_d, depth, (size_t)s.pad_base,
                       (uint32_t)s.first_gadget_wire, wl, xch);
    OG_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_wires_from_limbs, dim3(grid_for(s.pad_base, 256), (unsigned)n), dim3(256), 0, ctx->stream, (const uint32_t*)wl, out_d,
                       (size_t)s.n_wires, (uint32_t)s.pad_base);
    OG_HIP(hipGetLastError());
  } else if (on_host)
    OG_TRY(withdraw_walk_on_host(ctx, depth, s, inputs_d, n, out_d));
  else if (lat && depth <= WLAT_JOBS_A + WLAT_JOBS_B)
    hipLaunchKernelGGL(k_withdraw_core_lat, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, inputs_d, depth,
                       (size_t)s.n_wires, (uint32_t)s.first_gadget_wire, n, out_d);
  else if (pair)
    hipLaunchKernelGGL(k_withdraw_core<true>, dim3(grid_for(2 * n, 64)), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, inputs_d,
                     depth, (size_t)s.n_wires, (uint32_t)s.first_gadget_wire, n, out_d);
  else
    hipLaunchKernelGGL(k_withdraw_core<false>, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, inputs_d,
                     depth, (size_t)s.n_wires, (uint32_t)s.first_gadget_wire, n, out_d);
  OG_HIP(hipGetLastError());
  if (!w9)
    hipLaunchKernelGGL(k_wires_from_mont, dim3(grid_for(s.pad_base, 256), (unsigned)n), dim3(256), 0, ctx->stream, out_d,
                       (size_t)s.n_wires, (uint32_t)s.pad_base, on_host ? 32u : 1u);
  OG_HIP(hipGetLastError());
  const uint64_t units = n_pad3 + (n_pad2 + PAD_SEGMENT - 1) / PAD_SEGMENT;
  if (units) {
    hipLaunchKernelGGL(k_withdraw_pad, dim3(grid_for(units, 256), (unsigned)n), dim3(256), 0, ctx->stream, inputs_d, depth,
                       (size_t)s.n_wires, (uint32_t)s.pad_base, (uint32_t)n_pad3, (uint32_t)n_pad2, out_d);
    OG_HIP(hipGetLastError());
  }
  return OG_OK;
}

// ---- the deposit statement (round 6; BASELINE.json north_star: "deposit/withdraw circuits") ----------------------------------------
// Spec: oracle/py/deposit.py.  public: commitment, depositor; private: nullifier, secret; commitment = H(nullifier, secret) -- ONE
// MultiMiMC7 gadget whose output is the public wire 1 -- and depositor bound by its square.  No reference counterpart: the
// snapshot's deposit (/root/reference/src/services/api_services/deposit.rs:32-154 -> /root/reference/src/blockchain/tx/mint_tx.rs:11-49)
// credits an account on the word of an L1 transaction hash and has no commitment at all.
// Input record per deposit, 3 x 32 B canonical LE: nullifier | secret | depositor.  Output: 735 wires x 32 B canonical:
//   0 one | 1 commitment | 2 depositor | 3 nullifier | 4 secret | 5 depositor^2 | 6.. perm0 (364) | k1 | perm1 (364)
constexpr int D_REC = 3, D_PUB = 2;
constexpr uint32_t D_WIRES = 6 + 729, D_CONSTRAINTS = 1 + 730;

// One lane per deposit (a batch is parallel across lanes; one request is two permutations = 0.33 ms of chain).  Montgomery
// values are stored and converted afterwards in parallel (k_wires_from_mont), as in k_withdraw_core.
__global__ void __launch_bounds__(64) k_deposit_witness(const uint32_t* __restrict__ consts, const uint8_t* __restrict__ inputs, size_t n,
                                                       uint8_t* __restrict__ out) {
  OG_FILLER_PRIO();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* in = inputs + g * (size_t)D_REC * 32;
  WireWriterT<false> ww{out + g * (size_t)D_WIRES * 32, 6};
  const Fr nullifier = fe_to_mont(fe_load<FrParams>(in));
  const Fr secret = fe_to_mont(fe_load<FrParams>(in + 32));
  const Fr depositor = fe_to_mont(fe_load<FrParams>(in + 64));
  ww.put(0, Fr::one());
  ww.put(2, depositor);
  ww.put(3, nullifier);
  ww.put(4, secret);
  ww.put(5, fe_sqr(depositor));
  // MultiMiMC7([l, r], key 0): k1 = l + E_0(l); out = 2 k1 + r + E_k1(r), with E_k(x) = x_91 + k (one rolled body, no calls)
  Fr k = Fr::zero(), x = nullifier, k1 = Fr::zero();
#pragma unroll 1
  for (int p = 0; p < 2; p++) {
#pragma unroll 1
    for (int i = 0; i < MIMC7_ROUNDS; i++) {
      const Fr t = fe_add3_weak(x, k, mimc7_const(consts, i));
      const Fr t2 = fe_sqr(t);
      const Fr t4 = fe_sqr(t2);
      const Fr t6 = fe_mul(t4, t2);
      x = fe_mul(t6, t);
      ww.push(t2);
      ww.push(t4);
      ww.push(t6);
      ww.push(x);
    }
    if (p == 0) {
      k1 = fe_add(nullifier, x);
      ww.push(k1);
      k = k1;
      x = secret;
    }
  }
  ww.put(1, fe_add(fe_add(fe_dbl(k1), secret), x));
}

// bad[g] = lowest field of record g that is not the canonical encoding of an Fr element (0 nullifier, 1 secret, 2 depositor)
__global__ void __launch_bounds__(64) k_check_deposit_records(const uint8_t* __restrict__ inputs, size_t n, uint32_t* __restrict__ bad) {
  OG_FILLER_PRIO();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D_REC) return;
  const size_t g = t / D_REC;
  const uint32_t f = (uint32_t)(t % D_REC);
  if (!fe_lt_modulus(fe_load<FrParams>(inputs + t * 32))) atomicMin(&bad[g], f);
}

int deposit_shape_query(uint64_t out[3]) {
  out[0] = D_WIRES; out[1] = D_CONSTRAINTS; out[2] = D_PUB;
  return OG_OK;
}

// OG_ERR_INVALID names the first malformed record (`base` = index of record 0 in the caller's batch); blocking
int deposit_records_ok(og_ctx* ctx, const uint8_t* inputs_d, size_t n, size_t base) {
  if (n == 0) return OG_OK;
  uint32_t* bad_d = nullptr;
  OG_TRY(arena_get(ctx, "dp.bad", n * 4, (void**)&bad_d));
  OG_HIP(hipMemsetAsync(bad_d, 0xff, n * 4, ctx->stream));
  hipLaunchKernelGGL(k_check_deposit_records, dim3(grid_for(n * D_REC, 64)), dim3(64), 0, ctx->stream, inputs_d, n, bad_d);
  OG_HIP(hipGetLastError());
  std::vector<uint32_t> b(n);
  OG_HIP(hipMemcpyAsync(b.data(), bad_d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  OG_HIP(hipStreamSynchronize(ctx->stream));
  static const char* names[D_REC] = {"nullifier", "secret", "depositor"};
  for (size_t g = 0; g < n; g++)
    if (b[g] != 0xffffffffu) {
      set_error("og_deposit: input record " + std::to_string(base + g) + ": field " + std::to_string(b[g]) + " (" + names[b[g] % D_REC] +
                ") is not a canonical value (>= r)");
      return OG_ERR_INVALID;
    }
  return OG_OK;
}

int deposit_witness(og_ctx* ctx, const uint8_t* inputs_d, size_t n, uint8_t* out_d) {
  OG_REQUIRE(n <= 65535, "deposit: at most 65535 witnesses per call");
  if (n == 0) return OG_OK;
  ProfScope ps(ctx, PROF_WITNESS, (double)n);
  hipLaunchKernelGGL(k_deposit_witness, dim3(grid_for(n, 64)), dim3(64), 0, ctx->stream, (const uint32_t*)ctx->mimc_consts_d, inputs_d, n, out_d);
  OG_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_wires_from_mont, dim3(grid_for(D_WIRES, 256), (unsigned)n), dim3(256), 0, ctx->stream, out_d, (size_t)D_WIRES, D_WIRES, 1u);
  OG_HIP(hipGetLastError());
  return OG_OK;
}

}  // namespace og
