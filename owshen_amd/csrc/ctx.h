// Internal context + error plumbing for libowshen_gpu.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <mutex>
#include <thread>
#include <vector>
#include <map>
#include "../../include/owshen_gpu.h"

struct og_ctx {
  int device = 0;
  hipStream_t stream = nullptr;   // the stream work is currently issued on (= lanes[lane])
  // Two streams + two scratch namespaces ("lanes"): the batched prover runs the VALU-bound stages of sub-batch k on
  // lanes[0] and the memory-bound preparation of sub-batch k + 1 on lanes[1]; `lane` selects the scratch namespace
  // (= sub-batch parity) the next arena_get uses.  A single small request splits one proof across the two streams instead.
  hipStream_t lanes[2] = {nullptr, nullptr};
  int lane = 0;
  int n_lanes = 2;               // og_set_lanes: 1 = strictly serial sub-batches (isolated kernel timings)
  std::mutex mu;                 // calls on one ctx are serialised
  uint8_t* mimc_consts_d = nullptr;  // 91 x 32 B, Fr Montgomery form
  uint8_t mimc_consts_canon[91 * 32];
  uint8_t* mimc_consts9_d = nullptr; // the same constants as 91 x 16 u32: nine Montgomery limbs + zeros (the wave-wide kernels: limb j in lane j)
  uint8_t* mimc_zeros_d = nullptr;   // roots of all-zero subtrees of height 0..64, canonical (built on first use)
  std::vector<uint8_t> mimc_zeros_h;  // the same roots on the host, in the host's Montgomery form (mimc7.hip: the host append; built on first use)
  int host_chains_max = 0;             // og_set_host_chains: withdraw calls of at most this many requests walk their MiMC7 chains on the host CPU (0 = never)
  uint8_t* walk_stage = nullptr;     // pinned staging for it (records down, core wires up), grown on demand
  size_t walk_stage_bytes = 0;
  hipEvent_t walk_ev = nullptr;      // behind the last upload out of walk_stage
  size_t call_requests = 0;          // set by prove_enqueue around its sub-batches: the size of the CALL (the host-chains bound is per call, not per sub-batch)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // msm_run: optional side stream for an MSM's tail (heavy buckets, reduction, window combine), see msm_impl.hip.h
  hipStream_t tail_stream = nullptr;   // set by the batched prover around its MSMs, null otherwise
  hipStream_t tail_lane = nullptr;     // the stream object (created at og_init)
  hipStream_t aux_lane = nullptr;      // prove_batch pipeline: the H query's digit sort (needs the quotient of the SAME sub-batch)
  hipEvent_t tail_ev[8] = {};
  unsigned tail_ev_next = 0;
  int msm_tag = 0;                     // which of the caller's MSMs this is (names the buffers a tail still reads)
  // prove_batch jobs (groth16.hip): a call is enqueued completely (every kernel of every sub-batch, on the four streams) and
  // finished separately (wait for its last kernels, copy the proofs out).  The blocking entry points do both at once;
  // og_withdraw_prove_batch_submit_d / og_job_wait let a caller keep ONE call ahead, so that the next call's cold start
  // (first witnesses, first sorts) runs under the current call's last accumulations.  Call-level buffers come in two sets.
  struct og_job* jobs[2] = {nullptr, nullptr};  // pending job of each call slot (a call takes the first free one)
  uint8_t fb_base[2][128] = {};                 // scalar_mul_fixed: the G1 / G2 point (canonical bytes) whose 64 x 16 fixed-base table
  uint8_t* fb_tab[2] = {nullptr, nullptr};      //   fb_tab holds (device, in `owned`; not arena scratch, which is per lane): key
  bool fb_valid[2] = {false, false};            //   generation multiplies the same generator once per query
  std::vector<struct og_job*> done_jobs;        // jobs that completed inside the submit call (small circuits): live handles
                                                // og_job_wait / og_job_abandon still have to see
  bool last_call_piped = false;        // the previous call went through the stage pipeline (its scratch is guarded by slot events)
  uint64_t pipe_counter = 0;           // sub-batches ever issued through the stage pipeline (scratch slot = counter mod 3)
  hipStream_t copy_lane = nullptr;     // (r, s) in, proofs / flags / public inputs out: never queues behind compute
  bool sort_beside_acc = false;        // set by the pipelined prover: digit sorts run BESIDE bucket accumulation (msm.hip picks the
                                       // small-footprint sort kernels, which fit the registers / LDS the accumulation leaves free)
  static constexpr int PIPE_SLOTS = 3;  // scratch slots the prove_batch pipeline can rotate over (events exist for three; TWO are used
                                        // since round 6 -- sub-batch k uses slot k mod 2 --, hooks builds: OG_PIPE_SLOTS=3)
  static constexpr int PIPE_EVENTS = 12;  // [0..6] stage hand-offs; [7..10] "the math stream is about to launch accumulation A | B1 | L | H";
                                          // [11] "the slot's PREPARATION-side scratch is free" (groth16.hip: two slots, released early)
  hipEvent_t after_heavy_ev = nullptr;  // set by the pipelined prover around a sub-batch's LAST MSM: msm_run records it behind that MSM's
                                        // heavy-bucket kernels, the last readers of the sub-batch's sorted digit entries
  size_t scratch_budget = 0;            // og_set_scratch_budget: bytes the prover may reserve for sub-batch scratch (0 = the default rule)
  hipEvent_t pipe_ev[PIPE_SLOTS][PIPE_EVENTS] = {};  // prove_batch pipeline (groth16.hip): per scratch slot, hand-offs between the streams
  int n_cu = 256;
  // scratch arena for MSM / NTT / prover (grown on demand, freed at shutdown)
  std::vector<void*> owned;
  std::map<std::string, std::pair<void*, size_t>> arena;
  // opt-in region timing with HIP events on `stream` (og_profile / og_profile_read)
  bool prof_on = false;
  struct ProfEntry { int kind; hipEvent_t a, b; double units; };
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> prof_pool;
};

// host-side R1CS, constraint rows only (keygen.hip builds them, zkey.hip reads / writes circom's .r1cs; opaque at the C ABI)
struct og_r1cs {
  uint64_t n_wires = 0, n_pub = 0, n_constraints = 0;
  std::vector<uint32_t> ptr[3], col[3];
  std::vector<uint8_t> val[3];  // nnz x 32 B canonical
};

// one enqueued prove_batch call (opaque at the C ABI)
struct og_job {
  og_ctx* ctx = nullptr;
  int call_slot = 0;
  size_t n = 0, n_pub = 0;
  uint8_t* proofs = nullptr;       // caller's host buffers, filled by og_job_wait
  uint8_t* pub_out = nullptr;
  uint8_t *proofs_d = nullptr, *pub_d = nullptr;
  uint32_t* flags_d = nullptr;     // [n] unsatisfied flags | [n] first non-canonical wire / record field (0xffffffff = none)
  int bad_kind = 0;                // what the second half indexes: 0 nothing checked, 1 witness wires, 2 withdraw input-record fields
  hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};  // one per stream, recorded behind the call's last work
  int n_done = 0;
  uint64_t id = 0;                 // per-process serial number: a handle's ADDRESS can be reused by a later job, its id cannot
  bool waiting = false;            // a thread is inside og_job_wait for this job, outside the context's lock
  // og_set_host_chains: the call's proofs are assembled on the HOST from the five query results (groth16.hip, assemble_on_host)
  const struct og_pk* host_asm_pk = nullptr;   // non-null: no assembly kernels were enqueued; proofs_d is unused
  const uint8_t* res_d[5] = {};                // A | B1 | B2 | L | H results of the call (XYZZ, Montgomery; device)
  std::vector<uint8_t> rs_h;                   // the call's blinding pairs (the caller's buffer need not outlive a submit)
  std::vector<uint8_t> host_fixed;             // per proof: r delta1 | (r s) delta1 | beta2 + s delta2 (XYZZ), formed while the GPU runs
};

// ---- A/B and test hooks -------------------------------------------------------------------------------------------------
// Rounds 1-4 grew ~50 environment switches: launch shapes, thresholds that make rare paths reachable at toy sizes, rejected
// variants kept for same-box A/Bs, failure injection for the multi-device layer.  A node process inherits its environment, so
// the SHIPPED library reads none of them: they exist only in builds with -DOG_AB_HOOKS (libowshen_gpu_hooks.so, the CPU
// interpreter build of tests/hipemu) -- in the default build every OG_HOOK_* below is its default, a compile-time constant:
// the variable names are not even in the binary, and the code they select is compiled out (the rejected kernels live behind
// the same macro).  The default library reads exactly two variables, with plain getenv: OG_SUB_BATCH (cap the sub-batch
// size: an operator's memory knob, groth16.hip) and OG_DEBUG_SYNC (synchronise and log after every stage: fault bisection).
#ifdef OG_AB_HOOKS
#include <stdlib.h>
namespace og {
static inline const char* hook_str_(const char* name) { return getenv(name); }
static inline long long hook_int_(const char* name, long long dflt) {
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
}
static inline double hook_dbl_(const char* name, double dflt) {
  const char* e = getenv(name);
  return e ? atof(e) : dflt;
}
}  // namespace og
#define OG_HOOK_STR(name) og::hook_str_(name)                 /* the variable's text, or nullptr */
#define OG_HOOK_SET(name) (og::hook_str_(name) != nullptr)    /* is it set at all */
#define OG_HOOK_INT(name, dflt) og::hook_int_(name, (dflt))   /* its integer value, or dflt */
#define OG_HOOK_DBL(name, dflt) og::hook_dbl_(name, (dflt))
#else
#define OG_HOOK_STR(name) ((const char*)nullptr)
#define OG_HOOK_SET(name) false
#define OG_HOOK_INT(name, dflt) ((long long)(dflt))
#define OG_HOOK_DBL(name, dflt) ((double)(dflt))
#endif
// the wave-wide MiMC7 kernels (template <int FORM>: mimc7.hip.h w9_mimc7_round): form 2 everywhere; the forms it replaced
// exist in hooks builds only, behind OG_W9_ROWS = 0 | 1
#ifdef OG_AB_HOOKS
#define OG_W9_LAUNCH(kern, form, ...)                                         \
  do {                                                                        \
    const int og_w9_form_ = (form);                                           \
    if (og_w9_form_ >= 2) hipLaunchKernelGGL(kern<2>, __VA_ARGS__);           \
    else if (og_w9_form_ == 1) hipLaunchKernelGGL(kern<1>, __VA_ARGS__);      \
    else hipLaunchKernelGGL(kern<0>, __VA_ARGS__);                            \
  } while (0)
#else
#define OG_W9_LAUNCH(kern, form, ...) hipLaunchKernelGGL(kern<2>, __VA_ARGS__)
#endif

namespace og {

static inline int w9_rows() { return (int)OG_HOOK_INT("OG_W9_ROWS", 2); }

void set_error(const std::string& msg);

// every stream of a context that can hold work: the two lanes, the tail / aux streams of the stage pipeline and the copy
// stream (which runs the B1 MSM of a single request).  Anything that frees or reuses scratch drains ALL of them.
static inline hipError_t drain_streams(og_ctx* c) {
  hipError_t first = hipSuccess;
  for (hipStream_t st : {c->lanes[0], c->lanes[1], c->tail_lane, c->aux_lane, c->copy_lane}) {
    if (!st) continue;
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  return first;
}

#define OG_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      og::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +      \
                    __FILE__ + ":" + std::to_string(__LINE__));                       \
      return OG_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

#define OG_REQUIRE(cond, msg)       \
  do {                              \
    if (!(cond)) {                  \
      og::set_error(msg);           \
      return OG_ERR_INVALID;        \
    }                               \
  } while (0)

#define OG_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != OG_OK) return _r; \
  } while (0)

// catch-all wrapper so nothing unwinds across the extern "C" boundary
template <class F>
static inline int guarded(F&& f) noexcept {
  try {
    return f();
  } catch (const std::exception& e) {
    set_error(std::string("exception: ") + e.what());
    return OG_ERR_INVALID;
  } catch (...) {
    set_error("unknown exception");
    return OG_ERR_INVALID;
  }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);

// OG_DEBUG_SYNC=1: synchronise and log after every launch site marked with OG_STEP (fault bisection)
bool debug_sync();
#define OG_STEP(ctx, name)                                         \
  do {                                                             \
    if (og::debug_sync()) {                                        \
      hipError_t _e = hipStreamSynchronize((ctx)->stream);         \
      fprintf(stderr, "[og] step %s -> %s\n", name, hipGetErrorString(_e)); \
      fflush(stderr);                                              \
    }                                                              \
  } while (0)

// dynamic LDS declaration (overridable: a HIP runtime header that has no `extern __shared__` may define its own)
#ifndef OG_DYN_LDS
#define OG_DYN_LDS(name) extern __shared__ __align__(16) uint8_t name[]
#endif

// Wave priority of the "filler" kernels: everything the batched prover queues BESIDE its bucket accumulation -- the next
// sub-batch's witnesses / sparse products / digit sorts, an MSM's heavy buckets / reduction / combine, proof assembly.
// The accumulation kernels are persistent (msm_impl.hip.h) and leave every SIMD a free wave slot, but a SIMD issues VALU
// instructions oldest wave first and the resident accumulation waves keep the multiply-add pipe ~95 % busy, so a young
// co-resident wave at the same priority is starved (round 3 trace: a 0.08 ms conversion kernel took 12.7 ms under an
// accumulation launch).  s_setprio 3 lets the fillers -- short, mostly memory- and latency-bound -- issue when they are
// ready; what they take from the accumulation is their own instruction count.
#ifndef OG_FILLER_PRIO
#define OG_FILLER_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// Claim a vector register without using it: raises the kernel's register allocation to at least n + 1 (occupancy control
// from inside the kernel, where __launch_bounds__ can only lower the register count)
#ifndef OG_CLAIM_VGPR
#define OG_CLAIM_VGPR(n) asm volatile("" ::: "v" #n)
#endif

// The value lane (lane ^ 1) holds: one v_mov_b32 with the DPP quad permutation [1, 0, 3, 2] (no LDS round trip, unlike the
// ds_bpermute __shfl_xor compiles to).  Both lanes of the pair must be active.
#ifndef OG_PAIR_SWAP32
#define OG_PAIR_SWAP32(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0xB1, 0xF, 0xF, true))
#endif

// this wave's shader clock (s_memtime ticks at the shader clock on gfx950); the interpreter has none
#ifndef OG_SHADER_CYCLES
#define OG_SHADER_CYCLES() ((unsigned long long)__builtin_amdgcn_s_memtime())
#endif

// arr[key]++ in LDS, returning the old value, with the lanes of the wave that hit the FIRST active lane's counter served by ONE
// atomic (ballot + mbcnt ranks).  A digit sort has hot spots -- every scalar that is 1 lands in the same bucket -- and 64 lanes
// on one LDS address serialise completely: the second sort level of a 2^26-point MSM spent 23 ms on the one bin that holds
// the 4 M "digit 1" entries.  Must be called convergently by the active lanes (a runtime header may supply its own form).
#ifndef OG_LDS_ATOMIC_INC_AGG
#define OG_LDS_ATOMIC_INC_AGG(arr, key) og_lds_inc_agg((arr), (key))
__device__ __forceinline__ uint32_t og_lds_inc_agg(uint32_t* arr, uint32_t key) {
  const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
  const unsigned long long m = __ballot(key == first);
  if (key == first) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if (rank == 0) base = atomicAdd(&arr[key], (uint32_t)__popcll(m));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + rank;  // (the first active lane in here is the rank-0 lane)
  }
  return atomicAdd(&arr[key], 1u);
}
#endif

// timed regions (kind indices are part of the C ABI: og_profile_read)
enum ProfKind { PROF_ACC_G1 = 0, PROF_ACC_G2 = 1, PROF_HPOLY = 2, PROF_SORT = 3, PROF_REDUCE_G1 = 4, PROF_REDUCE_G2 = 5,
                PROF_WITNESS = 6, PROF_SPMV = 7, PROF_ASSEMBLE = 8, PROF_HEAVY_G1 = 9, PROF_HEAVY_G2 = 10, PROF_NKINDS = 11 };

struct ProfScope {
  og_ctx* c;
  int idx = -1;
  ProfScope(og_ctx* ctx, int kind, double units) : c(ctx) {
    if (!c->prof_on) return;
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) {
      if (!c->prof_pool.empty()) {
        ev[i] = c->prof_pool.back();
        c->prof_pool.pop_back();
      } else if (hipEventCreate(&ev[i]) != hipSuccess) {
        return;
      }
    }
    (void)hipEventRecord(ev[0], c->stream);
    idx = (int)c->prof.size();
    c->prof.push_back({kind, ev[0], ev[1], units});
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(c->prof[idx].b, c->stream);
  }
};

// fn(g) for g in [0, n) on the host: the caller's thread plus up to min(n, 32) - 1 helpers, item g on thread g mod T (the host side
// of og_set_host_chains: a handful of independent requests).  A helper that cannot be started just leaves its items to the
// caller's thread; nothing is left joinable on any path.
template <class F>
static inline void host_parallel_for(size_t n, F&& fn) {
  if (n == 0) return;
  const size_t T = n < 32 ? n : 32;
  std::vector<std::thread> th;
  size_t started = 1;  // thread 0 is the caller
  for (; started < T; started++) {
    try {
      th.emplace_back([&fn, started, T, n]() { for (size_t g = started; g < n; g += T) fn(g); });
    } catch (...) {
      break;
    }
  }
  for (size_t g = 0; g < n; g += T) fn(g);
  for (size_t t = started; t < T; t++)  // (the helpers that never started)
    for (size_t g = t; g < n; g += T) fn(g);
  for (auto& t : th) t.join();
}

static inline unsigned grid_for(size_t n, unsigned block) {
  return (unsigned)((n + block - 1) / block);
}

}  // namespace og
