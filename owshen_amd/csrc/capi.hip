// extern "C" surface of libowshen_gpu.so (include/owshen_gpu.h).  Every entry point is
// noexcept-by-construction: bodies run inside og::guarded(), errors become codes + a
// thread-local message, mirroring the anyhow::Result convention of the reference's callers
// (/root/reference/src/utils.rs:5-20).
#include "ctx.h"
#include "msm.hip.h"
#include "glv.h"
#include <string.h>
#include <time.h>
#include <stdlib.h>

namespace og {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
std::string get_error() { return g_err; }
bool debug_sync() {
  static int v = -1;
  if (v < 0) v = getenv("OG_DEBUG_SYNC") ? 1 : 0;
  return v == 1;
}

int mimc7_init(og_ctx* ctx);
int mimc7_hash2(og_ctx*, const uint8_t*, const uint8_t*, uint8_t*, size_t);
int mimc7_merkle_paths(og_ctx*, const uint8_t*, const uint64_t*, const uint8_t*, int, uint8_t*, size_t);
int mimc7_tree_build(og_ctx*, const uint8_t*, size_t, uint8_t*);
int mimc7_append(og_ctx*, int, const uint8_t*, uint64_t, const uint8_t*, size_t, uint8_t*, uint8_t*);
int field_op(og_ctx*, int, int, const uint8_t*, const uint8_t*, uint8_t*, size_t);
int field_mulchain(og_ctx*, int, uint8_t*, const uint8_t*, size_t, int, float*);
int field_mulchain_lat(og_ctx*, int, int, uint8_t*, const uint8_t*, size_t, int, float*, uint64_t*);
int ubench(og_ctx*, int, int, int, float*, uint64_t*);
int ubench_coresidency(og_ctx*, int, int, int, int, int, int, int, int, int, float*);
int ntt_canonical(og_ctx*, const uint8_t*, uint8_t*, int, int, int, int);
int h_poly_canonical(og_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, int, int, uint8_t*);

int pk_load(og_ctx*, const uint8_t*, size_t, og_pk**);
void pk_destroy(og_pk*);
void pk_density(const og_pk*, uint64_t*);
void pk_windows(const og_pk*, uint64_t*);
uint64_t pk_bytes(const og_pk*);
int prove_plan(og_ctx*, const og_pk*, size_t, uint32_t*, size_t, size_t*, int*);
int prove_batch_device(og_ctx*, const og_pk*, const uint8_t*, size_t, const uint8_t*, uint8_t*, size_t*);
int prove_batch_host(og_ctx*, const og_pk*, const uint8_t*, size_t, const uint8_t*, uint8_t*);
int scalar_mul_fixed(og_ctx*, int, const uint8_t*, const uint8_t*, size_t, uint8_t*);
int lagrange_evals(og_ctx*, int, const uint8_t*, uint8_t*);
int withdraw_shape_query(int, uint64_t, uint64_t, uint64_t*);
int withdraw_witness(og_ctx*, int, uint64_t, uint64_t, const uint8_t*, size_t, uint8_t*);
int withdraw_records_ok(og_ctx*, int, const uint8_t*, size_t, size_t);
int verify_cpu(const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, int*);
int withdraw_prove_batch(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, const uint8_t*, uint8_t*, uint8_t*);
int withdraw_prove_batch_submit(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, const uint8_t*, uint8_t*, uint8_t*,
                                og_job**);
int deposit_shape_query(uint64_t*);
int deposit_records_ok(og_ctx*, const uint8_t*, size_t, size_t);
int deposit_witness(og_ctx*, const uint8_t*, size_t, uint8_t*);
int deposit_prove_batch(og_ctx*, const og_pk*, const uint8_t*, size_t, const uint8_t*, uint8_t*, uint8_t*);
int job_wait(og_job*);
int withdraw_prove_partials_enqueue(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, int, int, uint8_t*, uint8_t*, og_job**);
int prove_partials_enqueue(og_ctx*, const og_pk*, const uint8_t*, size_t, int, int, uint8_t*, og_job**);
int prove_from_partials(og_ctx*, const og_pk*, const uint8_t*, int, size_t, const uint8_t*, uint8_t*);
bool glv_pair_ok();
int job_done_events(og_job*, hipEvent_t*);
int job_abandon(og_job*);
bool job_is_live(og_ctx*, og_job*);
bool job_mark_waiting(og_job*, bool, uint64_t*);
bool job_same(og_job*, uint64_t);
bool job_is_waited_for(og_job*);
bool ctx_has_waiters(og_ctx*);
int spmv_canonical(og_ctx*, const uint32_t*, const uint32_t*, const uint8_t*, size_t, const uint8_t*, uint8_t*);
int eddsa_verify(og_ctx*, const uint8_t*, size_t, uint32_t*);

}  // namespace og

// mirrors the head of the definition in groth16.hip (og_pk_info reads only these fields)
struct og_pk_head {
  uint64_t m, n_pub, log_d, n_rows;
};

using namespace og;

// every ctx-taking entry point runs on the ctx's device: scratch (arena_get -> hipMalloc) and launches follow the calling
// thread's CURRENT device, which after og_multi_init / another ctx's call need not be this one
#define LOCKED(ctx)                              \
  std::lock_guard<std::mutex> _lk((ctx)->mu);    \
  OG_HIP(hipSetDevice((ctx)->device))
#define CTX_OK(ctx) OG_REQUIRE((ctx) != nullptr, "null og_ctx")

extern "C" {

const char* og_last_error(void) { return g_err.c_str(); }

int og_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int og_init(int device, og_ctx** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(out != nullptr, "og_init: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
      set_error("og_init: no HIP device visible (this library has no CPU fallback)");
      return OG_ERR_NO_DEVICE;
    }
    OG_REQUIRE(device >= 0 && device < n, "og_init: device index out of range");
    OG_HIP(hipSetDevice(device));
    og_ctx* ctx = new og_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    OG_HIP(hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    OG_HIP(hipStreamCreateWithFlags(&ctx->lanes[0], hipStreamNonBlocking));
    OG_HIP(hipStreamCreateWithFlags(&ctx->lanes[1], hipStreamNonBlocking));  // (a higher queue priority for this, the prep
    // stream of the prove pipeline, was measured: 734 vs 746 proofs/s -- no help)
    ctx->stream = ctx->lanes[0];
    OG_HIP(hipStreamCreateWithFlags(&ctx->tail_lane, hipStreamNonBlocking));
    if (!OG_HOOK_INT("OG_NO_AUX_LANE", 0)) OG_HIP(hipStreamCreateWithFlags(&ctx->aux_lane, hipStreamNonBlocking));
    OG_HIP(hipStreamCreateWithFlags(&ctx->copy_lane, hipStreamNonBlocking));
    for (int k = 0; k < 8; k++) OG_HIP(hipEventCreateWithFlags(&ctx->tail_ev[k], hipEventDisableTiming));
    OG_HIP(hipEventCreate(&ctx->ev0));
    OG_HIP(hipEventCreate(&ctx->ev1));
    int r = mimc7_init(ctx);
    if (r != OG_OK) {
      delete ctx;
      return r;
    }
    *out = ctx;
    return OG_OK;
  });
}

void og_shutdown(og_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)drain_streams(ctx);  // ALL five streams before anything is freed: an abandoned submitted call may still be running
  // a thread inside og_job_wait (outside the lock, in hipEventSynchronize) wakes now that the streams are idle: let it consume
  // its job before the context goes away (bounded: a waiter that never returns is the caller's bug, not a hang here)
  for (int spin = 0; spin < 2000; spin++) {
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      if (!ctx_has_waiters(ctx)) break;
    }
    struct timespec ts = {0, 5 * 1000 * 1000};
    nanosleep(&ts, nullptr);
  }
  for (auto& e : ctx->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
  for (void* p : ctx->owned) (void)hipFree(p);
  for (auto& kv : ctx->arena) (void)hipFree(kv.second.first);
  if (ctx->mimc_consts_d) (void)hipFree(ctx->mimc_consts_d);
  if (ctx->mimc_consts9_d) (void)hipFree(ctx->mimc_consts9_d);
  if (ctx->mimc_zeros_d) (void)hipFree(ctx->mimc_zeros_d);
  if (ctx->walk_stage) (void)hipHostFree(ctx->walk_stage);
  if (ctx->walk_ev) (void)hipEventDestroy(ctx->walk_ev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (int p = 0; p < og_ctx::PIPE_SLOTS; p++)
    for (int e = 0; e < og_ctx::PIPE_EVENTS; e++)
      if (ctx->pipe_ev[p][e]) (void)hipEventDestroy(ctx->pipe_ev[p][e]);
  for (int k = 0; k < 8; k++)
    if (ctx->tail_ev[k]) (void)hipEventDestroy(ctx->tail_ev[k]);
  if (ctx->tail_lane) {
    (void)hipStreamSynchronize(ctx->tail_lane);
    (void)hipStreamDestroy(ctx->tail_lane);
  }
  if (ctx->aux_lane) {
    (void)hipStreamSynchronize(ctx->aux_lane);
    (void)hipStreamDestroy(ctx->aux_lane);
  }
  if (ctx->copy_lane) {
    (void)hipStreamSynchronize(ctx->copy_lane);
    (void)hipStreamDestroy(ctx->copy_lane);
  }
  for (og_job*& j : ctx->jobs)  // abandoned jobs: their work has drained above, only the handles are left
    if (j) {
      for (int k = 0; k < j->n_done; k++) (void)hipEventDestroy(j->done[k]);
      delete j;
      j = nullptr;
    }
  for (og_job* j : ctx->done_jobs) delete j;
  ctx->done_jobs.clear();
  for (int k = 0; k < 2; k++)
    if (ctx->lanes[k]) (void)hipStreamDestroy(ctx->lanes[k]);
  delete ctx;
}

int og_sync(og_ctx* ctx) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

void* og_stream(og_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int og_malloc(og_ctx* ctx, size_t bytes, void** out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(out_d != nullptr, "og_malloc: out is null");
    OG_HIP(hipSetDevice(ctx->device));
    OG_HIP(hipMalloc(out_d, bytes ? bytes : 1));
    return OG_OK;
  });
}

int og_free(og_ctx* ctx, void* p) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    if (p) OG_HIP(hipFree(p));
    return OG_OK;
  });
}

int og_memcpy_h2d(og_ctx* ctx, void* dst_d, const void* src, size_t bytes) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_HIP(hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_memcpy_d2h(og_ctx* ctx, void* dst, const void* src_d, size_t bytes) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_HIP(hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_field_op_d(og_ctx* ctx, int field, int op, const uint8_t* a_d, const uint8_t* b_d, uint8_t* out_d, size_t n) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(field == 0 || field == 1, "og_field_op_d: field must be 0 (Fr) or 1 (Fq)");
    OG_REQUIRE(op >= 0 && op <= 3, "og_field_op_d: op must be 0..3");
    LOCKED(ctx);
    OG_TRY(field_op(ctx, field, op, a_d, b_d, out_d, n));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_field_mulchain_d(og_ctx* ctx, int field, uint8_t* x_d, const uint8_t* y_d, size_t n, int iters, float* ms_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(field == 0 || field == 1, "og_field_mulchain_d: field must be 0 or 1");
    OG_REQUIRE(ms_out != nullptr && n > 0 && iters > 0, "og_field_mulchain_d: bad arguments");
    LOCKED(ctx);
    return field_mulchain(ctx, field, x_d, y_d, n, iters, ms_out);
  });
}

int og_field_mulchain_lat_d(og_ctx* ctx, int field, int form, uint8_t* x_d, const uint8_t* y_d, size_t n, int iters, float* ms_out,
                            uint64_t* wave_cycles_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(field == 0 || field == 1, "og_field_mulchain_lat_d: field must be 0 or 1");
    OG_REQUIRE(form == 0 || form == 1, "og_field_mulchain_lat_d: form must be 0 (one lane per element) or 1 (one wave per element)");
    OG_REQUIRE(ms_out != nullptr && n > 0 && n <= 65535 && iters > 0, "og_field_mulchain_lat_d: bad arguments");
    LOCKED(ctx);
    return field_mulchain_lat(ctx, field, form, x_d, y_d, n, iters, ms_out, wave_cycles_out);
  });
}

int og_ubench(og_ctx* ctx, int kind, int iters, int blocks, float* ms_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(ms_out != nullptr && iters > 0 && blocks > 0, "og_ubench: bad arguments");
    LOCKED(ctx);
    return ubench(ctx, kind, iters, blocks, ms_out, nullptr);
  });
}

int og_ubench_cycles(og_ctx* ctx, int kind, int iters, int blocks, float* ms_out, uint64_t* wave_cycles_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(ms_out != nullptr && wave_cycles_out != nullptr && iters > 0 && blocks > 0, "og_ubench_cycles: bad arguments");
    LOCKED(ctx);
    return ubench(ctx, kind, iters, blocks, ms_out, wave_cycles_out);
  });
}

int og_ubench_coresidency(og_ctx* ctx, int wgs_per_cu, int kind, int iters, int filler_blocks, int filler_threads, int filler_lds,
                          int filler_prio, int filler_work, int delay_us, float out[3]) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(out != nullptr && wgs_per_cu >= 0 && wgs_per_cu <= 32 && iters > 0 && filler_blocks > 0, "og_ubench_coresidency: bad arguments");
    OG_REQUIRE(filler_threads >= 64 && filler_threads <= 1024 && filler_threads % 64 == 0 && filler_lds >= 0 && filler_lds <= 160 * 1024,
               "og_ubench_coresidency: bad filler shape");
    LOCKED(ctx);
    return ubench_coresidency(ctx, wgs_per_cu, kind, iters, filler_blocks, filler_threads, filler_lds, filler_prio, filler_work, delay_us, out);
  });
}

int og_mimc7_constants(og_ctx* ctx, uint8_t out[91 * 32]) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    memcpy(out, ctx->mimc_consts_canon, 91 * 32);
    return OG_OK;
  });
}

int og_mimc7_hash2_d(og_ctx* ctx, const uint8_t* l, const uint8_t* r, uint8_t* out, size_t n) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_TRY(mimc7_hash2(ctx, l, r, out, n));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_mimc7_merkle_paths_d(og_ctx* ctx, const uint8_t* leaves, const uint64_t* idx, const uint8_t* sib, int depth,
                            uint8_t* nodes, size_t n) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(depth >= 0 && depth <= 64, "og_mimc7_merkle_paths_d: depth must be 0..64");
    LOCKED(ctx);
    OG_TRY(mimc7_merkle_paths(ctx, leaves, idx, sib, depth, nodes, n));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_mimc7_tree_build_d(og_ctx* ctx, const uint8_t* leaves, size_t n, uint8_t* nodes) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(n > 0 && (n & (n - 1)) == 0, "og_mimc7_tree_build_d: n must be a power of two");
    LOCKED(ctx);
    OG_TRY(mimc7_tree_build(ctx, leaves, n, nodes));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_ntt_fr_d(og_ctx* ctx, const uint8_t* in_d, uint8_t* out_d, int log_n, int batch, int inverse, int coset) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(log_n >= 0 && log_n <= 28, "og_ntt_fr_d: log_n must be 0..28");
    OG_REQUIRE(batch >= 1 && batch <= 65535, "og_ntt_fr_d: batch out of range");
    OG_REQUIRE(in_d != out_d, "og_ntt_fr_d: in and out must not overlap");
    LOCKED(ctx);
    OG_TRY(ntt_canonical(ctx, in_d, out_d, log_n, batch, inverse, coset));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_h_poly_d(og_ctx* ctx, const uint8_t* a_d, const uint8_t* b_d, const uint8_t* c_d, int log_d, int batch,
                uint8_t* h_out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(log_d >= 0 && log_d <= 28, "og_h_poly_d: log_d must be 0..28");
    OG_REQUIRE(batch >= 1 && batch <= 65535, "og_h_poly_d: batch out of range");
    LOCKED(ctx);
    OG_TRY(h_poly_canonical(ctx, a_d, b_d, c_d, log_d, batch, h_out_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_bases_create_d(og_ctx* ctx, int group, const uint8_t* points_d, size_t n, int window_bits, int precompute,
                      og_bases** out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(out != nullptr, "og_bases_create_d: out is null");
    OG_REQUIRE(group == 1 || group == 2, "og_bases_create_d: group must be 1 (G1) or 2 (G2)");
    OG_REQUIRE(window_bits == 0 || window_bits == 8 || window_bits == 12 || window_bits == 15 || window_bits == 16 || window_bits == 17,
               "og_bases_create_d: window_bits must be 0, 8, 12, 15, 16 or 17");
    LOCKED(ctx);
    int c = window_bits ? window_bits : (int)msm_pick_c(n);
    return bases_create(ctx, group == 2, points_d, n, c, precompute, out);
  });
}

void og_bases_free(og_bases* b) { bases_destroy(b); }

// A lone big G1 MSM over plain bases as its TWO WINDOW HALVES, side by side on the context's two lanes (round 5).  Each half is
// the window-sharded form a rank of an N-GPU MSM runs (even windows / odd windows: own digit sort, position-major
// accumulation, heavy buckets, scan-shaped reduction -> one point per window), the halves meet in the Horner combine that the
// N-GPU form uses after its all-gather.  Issued together -- NOT staggered: kernels of one half do not get to run beside the
// other half's persistent accumulation (measured from two contexts, tools/msm26_halves_probe.py: a second half started 4 - 20 ms
// late simply runs after the first, 87.5 - 88.5 ms) -- the two sorts share the memory system, the two accumulations the VALU,
// and each half's latency-bound tail (heavy-bucket trees, reduction, 2 ms of dependent doublings) hides under the other's
// work: 85.0 -> 81.2 ms at 2^26 points on one box.  Costs a second pass over the scalars (each half extracts its own digits).
static int msm_lone_halves(og_ctx* ctx, const og_bases* bases, const uint8_t* scalars_d, size_t n, uint8_t* res) {
  const size_t PB = bases->is_g2 ? 256 : 128, slots = (size_t)msm_partial_slots(bases);
  // (hooks builds: OG_LONE_PARTS = 4 cuts the windows four ways over four streams -- measured)
  hipStream_t st[4] = {ctx->lanes[0], ctx->lanes[1], ctx->tail_lane, ctx->aux_lane};
  int P = (int)OG_HOOK_INT("OG_LONE_PARTS", 2);
  P = P >= 4 && st[2] && st[3] ? 4 : 2;
  uint8_t* parts = nullptr;
  struct LaneGuard {
    og_ctx* c;
    ~LaneGuard() { c->lane = 0; c->stream = c->lanes[0]; }
  } guard{ctx};
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  OG_TRY(arena_get(ctx, "msm.halves", (size_t)P * slots * PB, (void**)&parts));
  OG_HIP(hipEventRecord(ctx->ev0, ctx->lanes[0]));  // whatever this context still has queued on lane 0 (a table import) precedes every part
  for (int h = 1; h < P; h++) OG_HIP(hipStreamWaitEvent(st[h], ctx->ev0, 0));
  for (int h = 0; h < P; h++) {
    ctx->lane = h;  // scratch namespace and stream of this part
    ctx->stream = st[h];
    DigitSort ds;
    OG_TRY(msm_digit_sort_windows(ctx, 0, scalars_d, n * 32, n, nullptr, 1, bases->c, bases->precomp, h, P, &ds));
    OG_TRY(msm_run_partial(ctx, bases, ds, parts + (size_t)h * slots * PB));
  }
  for (int h = 1; h < P; h++) {
    hipEvent_t e = ctx->tail_ev[h];
    OG_HIP(hipEventRecord(e, st[h]));
    OG_HIP(hipStreamWaitEvent(ctx->lanes[0], e, 0));
  }
  ctx->lane = 0;
  ctx->stream = ctx->lanes[0];
  return msm_combine(ctx, bases, parts, P, 1, res);
}

int og_msm_d(og_ctx* ctx, const og_bases* bases, const uint8_t* scalars_d, size_t n, int batch, size_t stride_bytes,
             uint8_t* out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(bases != nullptr && out != nullptr, "og_msm_d: null argument");
    OG_REQUIRE(batch >= 1, "og_msm_d: batch must be >= 1");
    OG_REQUIRE(batch == 1 || stride_bytes >= n * 32, "og_msm_d: stride smaller than one scalar vector");
    LOCKED(ctx);
    // (the MSM's scratch -- lanes 0 and 1, untagged reduction buffers -- is also the scratch of a submitted prove call)
    OG_REQUIRE(ctx->jobs[0] == nullptr && ctx->jobs[1] == nullptr, "og_msm_d: a submitted prove call has not been waited for (og_job_wait) -- its scratch is in use");
    const size_t pb = bases->is_g2 ? 128 : 64;
    uint8_t *res = nullptr, *aff = nullptr;
    OG_TRY(arena_get(ctx, "msm.result", (size_t)batch * 2 * pb, (void**)&res));
    OG_TRY(arena_get(ctx, "msm.affine", (size_t)batch * pb, (void**)&aff));
    // a lone big G1 MSM over plain bases: its two window halves side by side (hooks builds: OG_LONE_HALVES=0 keeps the single
    // launch set, OG_LONE_HALVES_MIN moves the size bound -- the interpreter reaches the path at toy size)
    const bool halves = batch == 1 && !bases->precomp && !bases->is_g2 && bases->c == 16 && bases->n >= n && ctx->n_lanes >= 2 && OG_HOOK_INT("OG_LONE_HALVES", 1) &&
                        n >= (size_t)OG_HOOK_INT("OG_LONE_HALVES_MIN", (long long)1 << 22) && n <= ((size_t)1 << 26);
    if (halves) {
      OG_TRY(msm_lone_halves(ctx, bases, scalars_d, n, res));
    } else {
      DigitSort ds;
      OG_TRY(msm_digit_sort(ctx, 0, scalars_d, stride_bytes, n, nullptr, batch, bases->c, bases->precomp, &ds));
      OG_TRY(msm_run(ctx, bases, ds, res));
    }
    OG_TRY(xyzz_to_affine_bytes(ctx, bases->is_g2, res, aff, batch));
    OG_HIP(hipMemcpyAsync(out, aff, (size_t)batch * pb, hipMemcpyDeviceToHost, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_msm_partial_slots(const og_bases* bases) { return bases ? msm_partial_slots(bases) : 0; }

int og_msm_windows_d(og_ctx* ctx, const og_bases* bases, const uint8_t* scalars_d, size_t n, int win_rank, int win_world,
                     uint8_t* partial_out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(bases != nullptr && partial_out_d != nullptr, "og_msm_windows_d: null argument");
    LOCKED(ctx);
    OG_REQUIRE(ctx->jobs[0] == nullptr && ctx->jobs[1] == nullptr, "og_msm_windows_d: a submitted prove call has not been waited for (og_job_wait) -- its scratch is in use");
    DigitSort ds;
    OG_TRY(msm_digit_sort_windows(ctx, 0, scalars_d, n * 32, n, nullptr, 1, bases->c, bases->precomp, win_rank, win_world, &ds));
    OG_TRY(msm_run_partial(ctx, bases, ds, partial_out_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_msm_combine_d(og_ctx* ctx, const og_bases* bases, const uint8_t* gathered_d, int world, uint8_t* out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(bases != nullptr && gathered_d != nullptr && out != nullptr, "og_msm_combine_d: null argument");
    LOCKED(ctx);
    OG_REQUIRE(ctx->jobs[0] == nullptr && ctx->jobs[1] == nullptr, "og_msm_combine_d: a submitted prove call has not been waited for (og_job_wait) -- its scratch is in use");
    const size_t pb = bases->is_g2 ? 128 : 64;
    uint8_t *res = nullptr, *aff = nullptr;
    OG_TRY(arena_get(ctx, "msm.result", 2 * pb, (void**)&res));
    OG_TRY(arena_get(ctx, "msm.affine", pb, (void**)&aff));
    OG_TRY(msm_combine(ctx, bases, gathered_d, world, 1, res));
    OG_TRY(xyzz_to_affine_bytes(ctx, bases->is_g2, res, aff, 1));
    OG_HIP(hipMemcpyAsync(out, aff, pb, hipMemcpyDeviceToHost, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_pk_load(og_ctx* ctx, const uint8_t* blob, size_t len, og_pk** out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(blob != nullptr && out != nullptr, "og_pk_load: null argument");
    *out = nullptr;
    LOCKED(ctx);
    OG_HIP(hipSetDevice(ctx->device));
    return pk_load(ctx, blob, len, out);
  });
}

void og_pk_free(og_pk* pk) { pk_destroy(pk); }

int og_pk_info(const og_pk* pk, uint64_t info[4]) {
  return guarded([&]() -> int {
    OG_REQUIRE(pk != nullptr && info != nullptr, "og_pk_info: null argument");
    const og_pk_head* h = reinterpret_cast<const og_pk_head*>(pk);
    info[0] = h->m; info[1] = h->n_pub; info[2] = h->log_d; info[3] = h->n_rows;
    return OG_OK;
  });
}

int og_pk_density(const og_pk* pk, uint64_t out[4]) {
  return guarded([&]() -> int {
    OG_REQUIRE(pk != nullptr && out != nullptr, "og_pk_density: null argument");
    pk_density(pk, out);
    return OG_OK;
  });
}

int og_pk_windows(const og_pk* pk, uint64_t out[4]) {
  return guarded([&]() -> int {
    OG_REQUIRE(pk != nullptr && out != nullptr, "og_pk_windows: null argument");
    pk_windows(pk, out);
    return OG_OK;
  });
}

int og_prove_batch(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses, size_t n, const uint8_t* rs, uint8_t* proofs_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr, "og_prove_batch: null key");
    OG_REQUIRE(n == 0 || (witnesses && rs && proofs_out), "og_prove_batch: null argument");
    LOCKED(ctx);
    OG_HIP(hipSetDevice(ctx->device));
    return prove_batch_host(ctx, pk, witnesses, n, rs, proofs_out);
  });
}

int og_prove(og_ctx* ctx, const og_pk* pk, const uint8_t* witness, const uint8_t rs[64], uint8_t proof_out[256]) {
  return og_prove_batch(ctx, pk, witness, 1, rs, proof_out);
}

int og_prove_batch_d(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses_d, size_t n, const uint8_t* rs, uint8_t* proofs_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr, "og_prove_batch_d: null key");
    OG_REQUIRE(n == 0 || (witnesses_d && rs && proofs_out), "og_prove_batch_d: null argument");
    LOCKED(ctx);
    OG_HIP(hipSetDevice(ctx->device));
    return prove_batch_device(ctx, pk, witnesses_d, n, rs, proofs_out, nullptr);
  });
}

int og_scalar_mul_d(og_ctx* ctx, int group, const uint8_t* base, const uint8_t* scalars_d, size_t n, uint8_t* out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(group == 1 || group == 2, "og_scalar_mul_d: group must be 1 (G1) or 2 (G2)");
    OG_REQUIRE(base != nullptr, "og_scalar_mul_d: null base");
    LOCKED(ctx);
    return scalar_mul_fixed(ctx, group == 2, base, scalars_d, n, out_d);
  });
}

int og_lagrange_evals_d(og_ctx* ctx, int log_d, const uint8_t tau[32], uint8_t* out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(log_d >= 0 && log_d <= 28, "og_lagrange_evals_d: log_d must be 0..28");
    OG_REQUIRE(tau != nullptr, "og_lagrange_evals_d: null tau");
    LOCKED(ctx);
    return lagrange_evals(ctx, log_d, tau, out_d);
  });
}

int og_spmv_fr_d(og_ctx* ctx, const uint32_t* row_ptr_d, const uint32_t* col_d, const uint8_t* val_d, size_t n_rows,
                 const uint8_t* x_d, uint8_t* out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_TRY(spmv_canonical(ctx, row_ptr_d, col_d, val_d, n_rows, x_d, out_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_withdraw_shape(int depth, uint64_t n_pad3, uint64_t n_pad2, uint64_t shape[3]) {
  return guarded([&]() -> int {
    OG_REQUIRE(shape != nullptr, "og_withdraw_shape: null argument");
    return withdraw_shape_query(depth, n_pad3, n_pad2, shape);
  });
}

int og_withdraw_witness_d(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d, size_t n,
                          uint8_t* witness_out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(n <= 65535, "og_withdraw_witness_d: at most 65535 witnesses per call");  // (before any scratch is grown or any kernel launched)
    LOCKED(ctx);
    OG_TRY(withdraw_records_ok(ctx, depth, inputs_d, n, 0));
    OG_TRY(withdraw_witness(ctx, depth, n_pad3, n_pad2, inputs_d, n, witness_out_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_set_lanes(og_ctx* ctx, int n_lanes) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(n_lanes == 1 || n_lanes == 2, "og_set_lanes: 1 or 2");
    LOCKED(ctx);
    ctx->n_lanes = n_lanes;
    return OG_OK;
  });
}

int og_set_scratch_budget(og_ctx* ctx, uint64_t bytes) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    ctx->scratch_budget = (size_t)bytes;
    return OG_OK;
  });
}

int og_set_host_chains(og_ctx* ctx, int max_requests) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(max_requests >= 0 && max_requests <= 64, "og_set_host_chains: max_requests must be 0 (off) .. 64");
    LOCKED(ctx);
    ctx->host_chains_max = max_requests;
    return OG_OK;
  });
}

int og_release_scratch(og_ctx* ctx) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_HIP(hipSetDevice(ctx->device));
    OG_REQUIRE(ctx->jobs[0] == nullptr && ctx->jobs[1] == nullptr, "og_release_scratch: a submitted call has not been waited for (og_job_wait)");
    OG_HIP(drain_streams(ctx));
    for (auto& kv : ctx->arena) (void)hipFree(kv.second.first);
    ctx->arena.clear();
    return OG_OK;
  });
}

int og_mem_info(og_ctx* ctx, uint64_t out[4]) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(out != nullptr, "og_mem_info: null argument");
    LOCKED(ctx);
    uint64_t held = 0;
    for (const auto& kv : ctx->arena) held += kv.second.second;
    size_t free_b = 0, total_b = 0;
    OG_HIP(hipMemGetInfo(&free_b, &total_b));
    out[0] = held; out[1] = ctx->arena.size(); out[2] = free_b; out[3] = total_b;
    return OG_OK;
  });
}

int og_prove_plan(og_ctx* ctx, const og_pk* pk, size_t n, uint32_t* sizes_out, size_t cap, size_t* count_out, int* mode_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr && (cap == 0 || sizes_out != nullptr), "og_prove_plan: null argument");
    LOCKED(ctx);
    return prove_plan(ctx, pk, n, sizes_out, cap, count_out, mode_out);
  });
}

int og_glv_decompose(const uint8_t k[32], uint8_t out[32]) {
  return guarded([&]() -> int {
    OG_REQUIRE(k != nullptr && out != nullptr, "og_glv_decompose: null argument");
    // (the library uses the GLV halves only if lambda and beta are a pair: [lambda] G = (beta x_G, y_G), checked once on the host)
    OG_REQUIRE(glv_pair_ok(), "og_glv_decompose: the endomorphism constants are not a pair ([lambda] G != (beta x, y)): the GLV path is disabled");
    OG_REQUIRE(glv::decompose(k, out), "og_glv_decompose: the scalar is not canonical (>= r)");
    return OG_OK;
  });
}

int og_pk_bytes(const og_pk* pk, uint64_t* out) {
  return guarded([&]() -> int {
    OG_REQUIRE(pk != nullptr && out != nullptr, "og_pk_bytes: null argument");
    *out = pk_bytes(pk);
    return OG_OK;
  });
}

int og_profile(og_ctx* ctx, int enable) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    LOCKED(ctx);
    OG_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& e : ctx->prof) {
      ctx->prof_pool.push_back(e.a);
      ctx->prof_pool.push_back(e.b);
    }
    ctx->prof.clear();
    ctx->prof_on = enable != 0;
    return OG_OK;
  });
}

int og_profile_read(og_ctx* ctx, int kind, double out[3]) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(kind >= 0 && kind < PROF_NKINDS && out != nullptr, "og_profile_read: bad arguments");
    LOCKED(ctx);
    OG_HIP(hipStreamSynchronize(ctx->stream));
    out[0] = out[1] = out[2] = 0.0;
    for (auto& e : ctx->prof) {
      if (e.kind != kind) continue;
      float ms = 0.f;
      OG_HIP(hipEventElapsedTime(&ms, e.a, e.b));
      out[0] += ms;
      out[1] += 1.0;
      out[2] += e.units;
    }
    return OG_OK;
  });
}

int og_verify(const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs, size_t n_pub, const uint8_t proof[256], int* ok_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(vk != nullptr && proof != nullptr && ok_out != nullptr && (n_pub == 0 || public_inputs != nullptr),
               "og_verify: null argument");
    return verify_cpu(vk, vk_len, public_inputs, n_pub, proof, ok_out);
  });
}

int og_withdraw_prove_batch_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                              size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr, "og_withdraw_prove_batch_d: null key");
    OG_REQUIRE(n == 0 || (inputs_d && rs && proofs_out), "og_withdraw_prove_batch_d: null argument");
    LOCKED(ctx);
    OG_HIP(hipSetDevice(ctx->device));
    return withdraw_prove_batch(ctx, pk, depth, n_pad3, n_pad2, inputs_d, n, rs, proofs_out, public_out);
  });
}

int og_withdraw_prove_batch_submit_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                                     size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out, og_job** job_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr && job_out != nullptr, "og_withdraw_prove_batch_submit_d: null argument");
    OG_REQUIRE(n >= 1 && inputs_d && rs && proofs_out, "og_withdraw_prove_batch_submit_d: null argument or empty batch");
    *job_out = nullptr;
    LOCKED(ctx);
    return withdraw_prove_batch_submit(ctx, pk, depth, n_pad3, n_pad2, inputs_d, n, rs, proofs_out, public_out, job_out);
  });
}

int og_deposit_shape(uint64_t shape[3]) {
  return guarded([&]() -> int {
    OG_REQUIRE(shape != nullptr, "og_deposit_shape: null argument");
    return deposit_shape_query(shape);
  });
}

int og_deposit_witness_d(og_ctx* ctx, const uint8_t* inputs_d, size_t n, uint8_t* witness_out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(n == 0 || (inputs_d && witness_out_d), "og_deposit_witness_d: null argument");
    LOCKED(ctx);
    OG_TRY(deposit_records_ok(ctx, inputs_d, n, 0));
    OG_TRY(deposit_witness(ctx, inputs_d, n, witness_out_d));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_deposit_prove_batch_d(og_ctx* ctx, const og_pk* pk, const uint8_t* inputs_d, size_t n, const uint8_t* rs, uint8_t* proofs_out,
                             uint8_t* public_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr, "og_deposit_prove_batch_d: null key");
    OG_REQUIRE(n == 0 || (inputs_d && rs && proofs_out), "og_deposit_prove_batch_d: null argument");
    if (n == 0) return OG_OK;
    LOCKED(ctx);
    return deposit_prove_batch(ctx, pk, inputs_d, n, rs, proofs_out, public_out);
  });
}

int og_withdraw_prove_partials_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                                 size_t n, int win_rank, int win_world, uint8_t* partials_d, uint8_t* public_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr && inputs_d && partials_d && n >= 1, "og_withdraw_prove_partials_d: null argument or empty batch");
    LOCKED(ctx);
    og_job* job = nullptr;
    OG_TRY(withdraw_prove_partials_enqueue(ctx, pk, depth, n_pad3, n_pad2, inputs_d, n, win_rank, win_world, partials_d, public_out, &job));
    return job_wait(job);
  });
}

int og_prove_partials_d(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses_d, size_t n, int win_rank, int win_world, uint8_t* partials_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr && witnesses_d && partials_d && n >= 1, "og_prove_partials_d: null argument or empty batch");
    LOCKED(ctx);
    og_job* job = nullptr;
    OG_TRY(prove_partials_enqueue(ctx, pk, witnesses_d, n, win_rank, win_world, partials_d, &job));
    return job_wait(job);
  });
}

int og_prove_from_partials_d(og_ctx* ctx, const og_pk* pk, const uint8_t* gathered_d, int win_world, size_t n, const uint8_t* rs,
                             uint8_t* proofs_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(pk != nullptr && gathered_d && rs && proofs_out, "og_prove_from_partials_d: null argument");
    LOCKED(ctx);
    OG_REQUIRE(ctx->jobs[0] == nullptr && ctx->jobs[1] == nullptr, "og_prove_from_partials_d: a submitted call has not been waited for (og_job_wait)");
    return prove_from_partials(ctx, pk, gathered_d, win_world, n, rs, proofs_out);
  });
}

int og_job_wait(og_ctx* ctx, og_job* job) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(job != nullptr, "og_job_wait: null job");
    // The wait itself happens OUTSIDE the context's lock: a host keeps one call ahead from two threads -- one blocked here for
    // batch k, one submitting batch k + 1 (the coalescer of INTEGRATION.md: tokio's spawn_blocking threads) -- and a wait that
    // held the lock for the length of a batch would make the submit queue behind it.  Only the bookkeeping is locked.
    hipEvent_t done[4];
    int n_done = 0;
    uint64_t id = 0;
    {
      LOCKED(ctx);
      // the handle is checked against the context's own records BEFORE it is dereferenced: a second wait, a wait after
      // og_job_abandon, or another context's job is an error, not a use-after-free
      OG_REQUIRE(job_is_live(ctx, job), "og_job_wait: not a pending job of this context (already waited for, abandoned, or another context's)");
      // ONE waiter per job: while this thread sits in hipEventSynchronize below, og_job_abandon and a second og_job_wait are
      // refused (they would destroy the events under it), and og_shutdown waits for it.  The job's id (a per-process counter)
      // rather than its address says on re-entry that it is still the same job.
      OG_REQUIRE(!job_mark_waiting(job, true, &id), "og_job_wait: another thread is already waiting for this job");
      n_done = job_done_events(job, done);
    }
    hipError_t werr = hipSuccess;
    for (int k = 0; k < n_done && werr == hipSuccess; k++) werr = hipEventSynchronize(done[k]);
    LOCKED(ctx);
    OG_REQUIRE(job_is_live(ctx, job) && job_same(job, id), "og_job_wait: the job was consumed while this thread waited for it");
    (void)job_mark_waiting(job, false, nullptr);
    OG_HIP(werr);
    return job_wait(job);
  });
}

int og_job_poll(og_ctx* ctx, og_job* job, int* done_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(job != nullptr && done_out != nullptr, "og_job_poll: null argument");
    *done_out = 0;
    LOCKED(ctx);
    OG_REQUIRE(job_is_live(ctx, job), "og_job_poll: not a pending job of this context");
    hipEvent_t done[4];
    const int n_done = job_done_events(job, done);
    for (int k = 0; k < n_done; k++) {
      const hipError_t e = hipEventQuery(done[k]);
      if (e == hipErrorNotReady) return OG_OK;
      OG_HIP(e);
    }
    *done_out = 1;
    return OG_OK;
  });
}

int og_job_abandon(og_ctx* ctx, og_job* job) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(job != nullptr, "og_job_abandon: null job");
    LOCKED(ctx);
    OG_REQUIRE(job_is_live(ctx, job), "og_job_abandon: not a pending job of this context");
    OG_REQUIRE(!job_is_waited_for(job), "og_job_abandon: another thread is waiting for this job (og_job_wait): it will consume it");
    return job_abandon(job);
  });
}

int og_eddsa_verify_batch_d(og_ctx* ctx, const uint8_t* records_d, size_t n, uint32_t* ok_out) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(n == 0 || (records_d && ok_out), "og_eddsa_verify_batch_d: null argument");
    if (n == 0) return OG_OK;
    LOCKED(ctx);
    uint32_t* ok_d = nullptr;
    OG_TRY(arena_get(ctx, "eddsa.ok", n * 4, (void**)&ok_d));
    OG_TRY(eddsa_verify(ctx, records_d, n, ok_d));
    OG_HIP(hipMemcpyAsync(ok_out, ok_d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    OG_HIP(hipStreamSynchronize(ctx->stream));
    return OG_OK;
  });
}

int og_mimc7_append_d(og_ctx* ctx, int depth, const uint8_t* frontier_in_d, uint64_t next_index, const uint8_t* leaves_d, size_t k,
                      uint8_t* frontier_out_d, uint8_t* root_out_d) {
  return guarded([&]() -> int {
    CTX_OK(ctx);
    OG_REQUIRE(frontier_in_d && leaves_d && frontier_out_d && root_out_d, "og_mimc7_append_d: null argument");
    OG_REQUIRE(frontier_in_d != frontier_out_d, "og_mimc7_append_d: frontier_in and frontier_out must not alias");
    LOCKED(ctx);
    return mimc7_append(ctx, depth, frontier_in_d, next_index, leaves_d, k, frontier_out_d, root_out_d);
  });
}

}  // extern "C"
