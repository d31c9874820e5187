// In-library multi-GPU: ONE process drives every GPU of the node -- what the reference's node is, a
// single process holding one Context (/root/reference/src/cli/node.rs:71-76; withdraw_handler at
// /root/reference/src/services/api_services/withdraw.rs:27-71 runs inside it).
//
//   og_multi_init            one og_ctx per device + an RCCL communicator per device (ncclCommInitAll)
//   og_multi_pk_load         the proving key replicated on every device
//   og_multi_prove_batch     proofs sharded across devices (contiguous slices, one host thread per device); proofs are
//   og_multi_withdraw_...    independent units, so there is NO data-path collective -- results land in the caller's buffer
//   og_multi_bases_create    MSM bases replicated on every device
//   og_multi_msm             ONE big MSM, window-sharded: the scalars go to device 0 and are broadcast over xGMI
//                            (ncclBroadcast), rank g accumulates the windows k = g (mod G), the per-window points
//                            (<= 16 x 256 B per rank) are exchanged with ncclAllGather -- an all-gather, never an
//                            all-reduce: curve points do not add limb-wise -- and every rank runs the Horner combine.
//
// Host threads only issue work and wait; they never touch results that another device produced except through RCCL.
#include "ctx.h"
#include "msm.hip.h"
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <stdlib.h>
#include <thread>
#include <rccl/rccl.h>

// One host thread per device, alive from og_multi_init to og_multi_shutdown: it binds its device once (hipSetDevice is
// per-thread state) and then runs whatever for_each_device hands it -- a prove call no longer spawns and joins N threads.
struct og_worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> task;
  bool has_task = false, done = false, stop = false;
  void run(int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_task || stop; });
      if (stop) return;
      std::function<void()> t = std::move(task);
      has_task = false;
      lk.unlock();
      t();
      lk.lock();
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<void()> t) {
    std::lock_guard<std::mutex> lk(mu);
    task = std::move(t);
    has_task = true;
    done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
};

struct og_multi {
  int n = 0;
  std::vector<og_ctx*> ctx;
  std::vector<ncclComm_t> comm;  // empty when n == 1
  std::mutex mu;                 // one og_multi_* call at a time (the calls drive every device and the communicators)
  std::vector<std::unique_ptr<og_worker>> workers;  // empty in sequential mode
};

namespace og {

int pk_load(og_ctx*, const uint8_t*, size_t, og_pk**);
void pk_destroy(og_pk*);
int prove_batch_host(og_ctx*, const og_pk*, const uint8_t*, size_t, const uint8_t*, uint8_t*);
int withdraw_prove_batch(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, const uint8_t*, uint8_t*, uint8_t*);
int withdraw_prove_partials_enqueue(og_ctx*, const og_pk*, int, uint64_t, uint64_t, const uint8_t*, size_t, int, int, uint8_t*, uint8_t*, og_job**);
int prove_partials_enqueue(og_ctx*, const og_pk*, const uint8_t*, size_t, int, int, uint8_t*, og_job**);
int prove_from_partials(og_ctx*, const og_pk*, const uint8_t*, int, size_t, const uint8_t*, uint8_t*);
int job_join_stream(og_job*, hipStream_t);
int job_wait(og_job*);
int job_abandon(og_job*);
std::string get_error();

#define OG_NCCL(expr)                                                                                          \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess) {                                                                                   \
      og::set_error(std::string(#expr) + ": " + ncclGetErrorString(_r) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
      return OG_ERR_HIP;                                                                                       \
    }                                                                                                          \
  } while (0)

// contiguous balanced slice [lo, hi) of n items for rank r of w (the same split as owshen_amd/shard.py)
static inline void slice_of(size_t n, int w, int r, size_t* lo, size_t* hi) {
  const size_t base = n / w, extra = n % w;
  *lo = (size_t)r * base + std::min<size_t>(r, extra);
  *hi = *lo + base + ((size_t)r < extra ? 1 : 0);
}

// run f(rank) for every device, each on its own host thread (OG_MULTI_SEQUENTIAL=1: in turn on the calling thread, for
// debugging and for single-threaded HIP runtimes); returns the first failing rank's code and leaves its message in this
// thread's og_last_error
// The calling thread's current device is whatever it was before the call (saved / restored around the sequential loop).
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// Failure injection (hooks builds ONLY, -DOG_AB_HOOKS: the shipped library has no such switch): OG_MULTI_FAIL="<site>:<rank>"
// makes rank <rank>'s part of the named step fail -- the error paths of the N-device entry points (the error names the device,
// an RCCL group is closed, the next call works) cannot be reached on healthy hardware.  Sites: pk_load, prove, withdraw,
// bases, msm.scratch, msm.broadcast, msm.accumulate, msm.allgather, sharded.front, sharded.allgather, sharded.finish.
#ifdef OG_AB_HOOKS
static bool injected_failure(const char* site, int r) {
  const char* e = OG_HOOK_STR("OG_MULTI_FAIL");
  if (!e || !site) return false;
  const std::string want = std::string(site) + ":" + std::to_string(r);
  if (want != e) return false;
  set_error(std::string("injected failure at ") + site + " (OG_MULTI_FAIL)");
  return true;
}
#else
static inline bool injected_failure(const char*, int) { return false; }
#endif

template <class F>
static int for_each_device(og_multi* m, F&& f, const char* site = nullptr) {
  std::vector<int> rc(m->n, OG_OK);
  std::vector<std::string> msg(m->n);
  auto body = [&](int r) {
    rc[r] = guarded([&]() -> int {
      OG_HIP(hipSetDevice(m->ctx[r]->device));
      if (injected_failure(site, r)) return OG_ERR_HIP;
      return f(r);
    });
    if (rc[r] != OG_OK) msg[r] = get_error();
  };
  if (m->workers.empty()) {
    DeviceGuard restore;
    for (int r = 0; r < m->n; r++) body(r);
  } else {
    for (int r = 0; r < m->n; r++) m->workers[r]->post([&body, r] { body(r); });
    for (int r = 0; r < m->n; r++) m->workers[r]->wait();
  }
  for (int r = 0; r < m->n; r++)
    if (rc[r] != OG_OK) {
      set_error("device " + std::to_string(r) + ": " + msg[r]);
      return rc[r];
    }
  return OG_OK;
}

// A grouped RCCL exchange: `enqueue(r)` issues rank r's call between ncclGroupStart and ncclGroupEnd.  The group is closed
// on EVERY path -- an error return between the two would leave it open and the next collective on these communicators
// would hang -- and the caller's current device is restored.
template <class F>
static int nccl_grouped(og_multi* m, F&& enqueue, const char* site = nullptr) {
  DeviceGuard restore;
  OG_NCCL(ncclGroupStart());
  int rc = OG_OK;
  for (int r = 0; r < m->n && rc == OG_OK; r++) {
    if (hipSetDevice(m->ctx[r]->device) != hipSuccess) {
      set_error("device " + std::to_string(r) + ": hipSetDevice failed inside an RCCL group");
      rc = OG_ERR_HIP;
      break;
    }
    rc = injected_failure(site, r) ? OG_ERR_HIP : enqueue(r);
    if (rc != OG_OK) set_error("device " + std::to_string(r) + ": " + get_error());  // (the group is still closed below)
  }
  const ncclResult_t e = ncclGroupEnd();
  if (rc != OG_OK) return rc;
  if (e != ncclSuccess) {
    set_error(std::string("ncclGroupEnd: ") + ncclGetErrorString(e));
    return OG_ERR_HIP;
  }
  return OG_OK;
}

int multi_init(int n_devices, og_multi** out) {
  int avail = 0;
  if (hipGetDeviceCount(&avail) != hipSuccess || avail == 0) {
    set_error("og_multi_init: no HIP device visible (this library has no CPU fallback)");
    return OG_ERR_NO_DEVICE;
  }
  if (n_devices == 0) n_devices = avail;
  OG_REQUIRE(n_devices >= 1 && n_devices <= avail,
             "og_multi_init: asked for " + std::to_string(n_devices) + " devices, " + std::to_string(avail) + " visible");
  DeviceGuard restore;  // og_init binds each device in turn on this thread
  og_multi* m = new og_multi();
  m->n = n_devices;
  for (int r = 0; r < n_devices; r++) {
    og_ctx* c = nullptr;
    int rc = og_init(r, &c);
    if (rc != OG_OK) {
      for (og_ctx* x : m->ctx) og_shutdown(x);
      delete m;
      return rc;
    }
    m->ctx.push_back(c);
  }
  // one rank needs no exchange; OG_MULTI_RCCL=1 still routes it through RCCL (a 1-GPU box can then exercise the
  // broadcast / all-gather path of og_multi_msm end to end)
  if (n_devices > 1 || OG_HOOK_INT("OG_MULTI_RCCL", 0)) {
    std::vector<int> devs(n_devices);
    for (int r = 0; r < n_devices; r++) devs[r] = r;
    m->comm.resize(n_devices);
    ncclResult_t nr = ncclCommInitAll(m->comm.data(), n_devices, devs.data());
    if (nr != ncclSuccess) {
      set_error(std::string("og_multi_init: ncclCommInitAll: ") + ncclGetErrorString(nr));
      for (og_ctx* x : m->ctx) og_shutdown(x);
      delete m;
      return OG_ERR_HIP;
    }
  }
  const bool sequential = n_devices == 1 || OG_HOOK_INT("OG_MULTI_SEQUENTIAL", 0);
  if (!sequential)
    for (int r = 0; r < n_devices; r++) {
      m->workers.emplace_back(new og_worker());
      og_worker* w = m->workers.back().get();
      const int dev = m->ctx[r]->device;
      w->th = std::thread([w, dev] { w->run(dev); });
    }
  *out = m;
  return OG_OK;
}

void multi_shutdown(og_multi* m) {
  if (!m) return;
  for (auto& w : m->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
      w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();
  }
  DeviceGuard restore;
  for (size_t r = 0; r < m->comm.size(); r++) (void)ncclCommDestroy(m->comm[r]);
  for (og_ctx* c : m->ctx) og_shutdown(c);
  delete m;
}

int multi_pk_load(og_multi* m, const uint8_t* blob, size_t len, og_pk** pks_out) {
  for (int r = 0; r < m->n; r++) pks_out[r] = nullptr;
  int rc = for_each_device(m, [&](int r) -> int {
    std::lock_guard<std::mutex> lk(m->ctx[r]->mu);
    return pk_load(m->ctx[r], blob, len, &pks_out[r]);
  }, "pk_load");
  if (rc != OG_OK)
    for (int r = 0; r < m->n; r++) {
      if (pks_out[r]) pk_destroy(pks_out[r]);
      pks_out[r] = nullptr;
    }
  return rc;
}

int multi_prove_batch(og_multi* m, og_pk* const* pks, const uint8_t* witnesses, size_t wit_bytes, size_t n, const uint8_t* rs,
                      uint8_t* proofs_out) {
  return for_each_device(m, [&](int r) -> int {
    size_t lo, hi;
    slice_of(n, m->n, r, &lo, &hi);
    if (hi == lo) return OG_OK;
    std::lock_guard<std::mutex> lk(m->ctx[r]->mu);
    return prove_batch_host(m->ctx[r], pks[r], witnesses + lo * wit_bytes, hi - lo, rs + lo * 64, proofs_out + lo * 256);
  }, "prove");
}

int multi_withdraw_prove_batch(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs,
                               size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out) {
  uint64_t info[4];
  OG_TRY(og_pk_info(pks[0], info));
  const size_t pub_bytes = (size_t)info[1] * 32;
  const size_t rec = (size_t)(8 + depth) * 32;
  return for_each_device(m, [&](int r) -> int {
    size_t lo, hi;
    slice_of(n, m->n, r, &lo, &hi);
    if (hi == lo) return OG_OK;
    og_ctx* c = m->ctx[r];
    std::lock_guard<std::mutex> lk(c->mu);
    uint8_t* in_d = nullptr;
    OG_TRY(arena_get(c, "multi.inputs", (hi - lo) * rec, (void**)&in_d));
    OG_HIP(hipMemcpyAsync(in_d, inputs + lo * rec, (hi - lo) * rec, hipMemcpyHostToDevice, c->stream));
    OG_HIP(hipStreamSynchronize(c->stream));
    return withdraw_prove_batch(c, pks[r], depth, n_pad3, n_pad2, in_d, hi - lo, rs + lo * 64, proofs_out + lo * 256,
                                public_out ? public_out + lo * pub_bytes : nullptr);
  }, "withdraw");
}

// Window-sharded proving (include/owshen_gpu.h: og_multi_withdraw_prove_sharded / og_multi_prove_sharded; groth16.hip: WinShard).
// `inputs` (withdraw records, host) or `witnesses` (host, n x wit_bytes): exactly one is non-null.
//   1. every device: the inputs over PCIe, the front half ENQUEUED (witness, sparse products, quotient, the windows
//      k = r (mod G) of the five queries -> n x 768 B of partial points), lanes[0] ordered behind all of the call's streams;
//   2. ONE grouped ncclAllGather on the devices' lanes[0] (n x 768 B per rank: 12 KB for 16 requests -- latency, not bandwidth);
//   3. device 0: adds the shares, assembles, copies the proofs out; every device: the front half's verdict (boundary flags,
//      satisfiability -- all ranks saw the same inputs) and, on device 0, the public inputs.
// A failure at any step leaves no job pending on any device and no RCCL group open: the next call works.
int multi_prove_sharded(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs,
                        const uint8_t* witnesses, size_t wit_bytes, size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out) {
  const int G = m->n;
  const size_t pbytes = n * (size_t)OG_PARTIAL_BYTES;
  const size_t rec = (size_t)(8 + depth) * 32;
  std::vector<uint8_t*> part_d(G, nullptr), gath_d(G, nullptr);
  std::vector<og_job*> jobs(G, nullptr);
  auto abandon_all = [&]() {
    const std::string keep = get_error();  // (the clean-up must not replace the message of the failure it cleans up after)
    (void)for_each_device(m, [&](int r) -> int {
      if (!jobs[r]) return OG_OK;
      std::lock_guard<std::mutex> lk(m->ctx[r]->mu);
      (void)job_abandon(jobs[r]);
      jobs[r] = nullptr;
      return OG_OK;
    });
    set_error(keep);
  };
  int rc = for_each_device(m, [&](int r) -> int {
    og_ctx* c = m->ctx[r];
    std::lock_guard<std::mutex> lk(c->mu);
    uint8_t* in_d = nullptr;
    OG_TRY(arena_get(c, "multi.sh.part", pbytes, (void**)&part_d[r]));
    OG_TRY(arena_get(c, "multi.sh.gathered", pbytes * G, (void**)&gath_d[r]));
    const size_t in_bytes = inputs ? n * rec : n * wit_bytes;
    OG_TRY(arena_get(c, inputs ? "multi.inputs" : "multi.witnesses", in_bytes, (void**)&in_d));
    OG_HIP(hipMemcpyAsync(in_d, inputs ? inputs : witnesses, in_bytes, hipMemcpyHostToDevice, c->stream));
    OG_HIP(hipStreamSynchronize(c->stream));
    if (inputs)
      OG_TRY(withdraw_prove_partials_enqueue(c, pks[r], depth, n_pad3, n_pad2, in_d, n, r, G, part_d[r], r == 0 ? public_out : nullptr, &jobs[r]));
    else
      OG_TRY(prove_partials_enqueue(c, pks[r], in_d, n, r, G, part_d[r], &jobs[r]));
    return job_join_stream(jobs[r], c->lanes[0]);  // the exchange below is stream-ordered behind the whole front half
  }, "sharded.front");
  if (rc != OG_OK) {
    abandon_all();
    return rc;
  }
  const bool rccl = !m->comm.empty();
  if (rccl) {
    rc = nccl_grouped(m, [&](int r) -> int {
      OG_NCCL(ncclAllGather(part_d[r], gath_d[r], pbytes, ncclUint8, m->comm[r], m->ctx[r]->lanes[0]));
      return OG_OK;
    }, "sharded.allgather");
    if (rc != OG_OK) {
      abandon_all();
      return rc;
    }
  }
  rc = for_each_device(m, [&](int r) -> int {
    og_ctx* c = m->ctx[r];
    std::lock_guard<std::mutex> lk(c->mu);
    int back = OG_OK;
    std::string back_msg;
    if (r == 0) {
      back = prove_from_partials(c, pks[0], rccl ? gath_d[0] : part_d[0], rccl ? G : 1, n, rs, proofs_out);
      if (back != OG_OK) back_msg = get_error();
    } else {
      (void)hipStreamSynchronize(c->lanes[0]);  // this rank's part of the all-gather (it reads part_d, writes gath_d)
    }
    og_job* j = jobs[r];
    jobs[r] = nullptr;
    const int front = job_wait(j);  // consumes the job on every path; the front half's verdict comes first (a malformed record, an unsatisfied witness)
    if (front != OG_OK) return front;
    if (back != OG_OK) set_error(back_msg);
    return back;
  }, "sharded.finish");
  if (rc != OG_OK) abandon_all();  // (an injected failure may have skipped a rank's body: its job is still pending)
  return rc;
}

int multi_bases_create(og_multi* m, int is_g2, const uint8_t* points, size_t n, int c, int precomp, og_bases** out) {
  for (int r = 0; r < m->n; r++) out[r] = nullptr;
  const size_t pb = is_g2 ? 128 : 64;
  int rc = for_each_device(m, [&](int r) -> int {
    og_ctx* cx = m->ctx[r];
    std::lock_guard<std::mutex> lk(cx->mu);
    uint8_t* stage = nullptr;
    OG_HIP(hipMalloc((void**)&stage, (n ? n : 1) * pb));
    hipError_t e = hipMemcpyAsync(stage, points, n * pb, hipMemcpyHostToDevice, cx->stream);
    int rr = e == hipSuccess ? bases_create(cx, is_g2, stage, n, c, precomp, &out[r]) : OG_ERR_HIP;  // synchronises the stream
    if (e != hipSuccess) set_error(std::string("og_multi_bases_create: ") + hipGetErrorString(e));
    (void)hipFree(stage);
    return rr;
  }, "bases");
  if (rc != OG_OK)
    for (int r = 0; r < m->n; r++) {
      bases_destroy(out[r]);
      out[r] = nullptr;
    }
  return rc;
}

int multi_msm(og_multi* m, og_bases* const* bases, const uint8_t* scalars, size_t n, uint8_t* out) {
  const int G = m->n;
  const og_bases* b0 = bases[0];
  const size_t pb = b0->is_g2 ? 128 : 64, xb = 2 * pb;
  const int slots = msm_partial_slots(b0);
  const size_t part_bytes = (size_t)slots * xb;
  std::vector<uint8_t*> sc_d(G, nullptr), part_d(G, nullptr), gath_d(G, nullptr);
  // 1. scratch on every device; the scalars reach device 0 over PCIe ...
  OG_TRY(for_each_device(m, [&](int r) -> int {
    og_ctx* c = m->ctx[r];
    OG_REQUIRE(bases[r] && bases[r]->device == c->device && bases[r]->n == b0->n && bases[r]->c == b0->c &&
                   bases[r]->precomp == b0->precomp && bases[r]->is_g2 == b0->is_g2,
               "og_multi_msm: bases[r] must be the replica created for device r");
    std::lock_guard<std::mutex> lk(c->mu);  // the ctx's arena map is not thread-safe against a direct og_* call on og_multi_ctx(m, r)
    OG_TRY(arena_get(c, "multi.scalars", (n ? n : 1) * 32, (void**)&sc_d[r]));
    OG_TRY(arena_get(c, "multi.part", part_bytes, (void**)&part_d[r]));
    OG_TRY(arena_get(c, "multi.gathered", part_bytes * G, (void**)&gath_d[r]));
    if (r == 0) OG_HIP(hipMemcpyAsync(sc_d[0], scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    return OG_OK;
  }, "msm.scratch"));
  // 2. ... and every other device over xGMI (one grouped broadcast)
  const bool rccl = !m->comm.empty();
  if (rccl && n > 0)
    OG_TRY(nccl_grouped(m, [&](int r) -> int {
      OG_NCCL(ncclBroadcast(sc_d[r], sc_d[r], n * 32, ncclUint8, 0, m->comm[r], m->ctx[r]->stream));
      return OG_OK;
    }, "msm.broadcast"));
  // 3. every rank: digit sort + bucket accumulation + reduction over its own windows (stream order after the broadcast)
  OG_TRY(for_each_device(m, [&](int r) -> int {
    og_ctx* c = m->ctx[r];
    std::lock_guard<std::mutex> lk(c->mu);
    DigitSort ds;
    OG_TRY(msm_digit_sort_windows(c, 0, sc_d[r], n * 32, n, nullptr, 1, bases[r]->c, bases[r]->precomp, r, G, &ds));
    return msm_run_partial(c, bases[r], ds, part_d[r]);
  }, "msm.accumulate"));
  // 4. all-gather of the per-window points
  if (rccl)
    OG_TRY(nccl_grouped(m, [&](int r) -> int {
      OG_NCCL(ncclAllGather(part_d[r], gath_d[r], part_bytes, ncclUint8, m->comm[r], m->ctx[r]->stream));
      return OG_OK;
    }, "msm.allgather"));
  // 5. Horner combine (every rank holds the gathered points; device 0 reports)
  DeviceGuard restore;
  OG_HIP(hipSetDevice(m->ctx[0]->device));
  og_ctx* c0 = m->ctx[0];
  std::lock_guard<std::mutex> lk(c0->mu);
  uint8_t *res = nullptr, *aff = nullptr;
  OG_TRY(arena_get(c0, "msm.result", xb, (void**)&res));
  OG_TRY(arena_get(c0, "msm.affine", pb, (void**)&aff));
  OG_TRY(msm_combine(c0, bases[0], rccl ? gath_d[0] : part_d[0], G, 1, res));
  OG_TRY(xyzz_to_affine_bytes(c0, b0->is_g2, res, aff, 1));
  OG_HIP(hipMemcpyAsync(out, aff, pb, hipMemcpyDeviceToHost, c0->stream));
  OG_HIP(hipStreamSynchronize(c0->stream));
  for (int r = 1; r < G; r++) {
    OG_HIP(hipSetDevice(m->ctx[r]->device));
    OG_HIP(hipStreamSynchronize(m->ctx[r]->stream));
  }
  return OG_OK;
}

}  // namespace og

using namespace og;

extern "C" {

int og_multi_init(int n_devices, og_multi** out) {
  return guarded([&]() -> int {
    OG_REQUIRE(out != nullptr && n_devices >= 0, "og_multi_init: bad arguments");
    *out = nullptr;
    return multi_init(n_devices, out);
  });
}

void og_multi_shutdown(og_multi* m) { multi_shutdown(m); }

int og_multi_size(const og_multi* m) { return m ? m->n : 0; }

int og_multi_slice(const og_multi* m, size_t n, int rank, size_t out[2]) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && out && rank >= 0 && rank < m->n, "og_multi_slice: bad arguments");
    slice_of(n, m->n, rank, &out[0], &out[1]);
    return OG_OK;
  });
}

og_ctx* og_multi_ctx(og_multi* m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }

// what makes "N ranks on N GPUs" a reading: the HIP device ordinal and PCI address rank `rank` is bound to, and -- when the
// devices share an RCCL communicator (n > 1) -- what RCCL itself says about this rank's communicator: its size, this rank's
// number in it and the device it was created on
int og_multi_device_info(const og_multi* m, int rank, uint64_t out[4], char pci_out[32]) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && out && pci_out && rank >= 0 && rank < m->n, "og_multi_device_info: bad arguments");
    out[0] = (uint64_t)m->ctx[rank]->device;
    out[1] = out[2] = out[3] = 0;
    pci_out[0] = 0;
    (void)hipDeviceGetPCIBusId(pci_out, 32, m->ctx[rank]->device);
    if (rank < (int)m->comm.size()) {
      int cnt = 0, ur = 0, dev = 0;
      OG_NCCL(ncclCommCount(m->comm[rank], &cnt));
      OG_NCCL(ncclCommUserRank(m->comm[rank], &ur));
      OG_NCCL(ncclCommCuDevice(m->comm[rank], &dev));
      out[1] = (uint64_t)cnt; out[2] = (uint64_t)ur; out[3] = (uint64_t)dev;
    }
    return OG_OK;
  });
}

int og_multi_pk_load(og_multi* m, const uint8_t* blob, size_t len, og_pk** pks_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && blob && pks_out, "og_multi_pk_load: null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    return multi_pk_load(m, blob, len, pks_out);
  });
}

void og_multi_pk_free(og_multi* m, og_pk** pks) {
  if (!m || !pks) return;
  for (int r = 0; r < m->n; r++) {
    if (pks[r]) og_pk_free(pks[r]);
    pks[r] = nullptr;
  }
}

int og_multi_prove_batch(og_multi* m, og_pk* const* pks, const uint8_t* witnesses, size_t n, const uint8_t* rs, uint8_t* proofs_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && pks, "og_multi_prove_batch: null argument");
    for (int r = 0; r < m->n; r++) OG_REQUIRE(pks[r] != nullptr, "og_multi_prove_batch: pks[r] is null (og_multi_pk_load fills one key per device)");
    OG_REQUIRE(n == 0 || (witnesses && rs && proofs_out), "og_multi_prove_batch: null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    uint64_t info[4];
    OG_TRY(og_pk_info(pks[0], info));
    return multi_prove_batch(m, pks, witnesses, (size_t)info[0] * 32, n, rs, proofs_out);
  });
}

int og_multi_withdraw_prove_batch(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs,
                                  size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && pks, "og_multi_withdraw_prove_batch: null argument");
    for (int r = 0; r < m->n; r++)
      OG_REQUIRE(pks[r] != nullptr, "og_multi_withdraw_prove_batch: pks[r] is null (og_multi_pk_load fills one key per device)");
    std::lock_guard<std::mutex> lk(m->mu);
    OG_REQUIRE(depth >= 1 && depth <= 64, "og_multi_withdraw_prove_batch: depth must be 1..64");
    OG_REQUIRE(n == 0 || (inputs && rs && proofs_out), "og_multi_withdraw_prove_batch: null argument");
    return multi_withdraw_prove_batch(m, pks, depth, n_pad3, n_pad2, inputs, n, rs, proofs_out, public_out);
  });
}

int og_multi_withdraw_prove_sharded(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs,
                                    size_t n, const uint8_t* rs, uint8_t* proofs_out, uint8_t* public_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && pks, "og_multi_withdraw_prove_sharded: null argument");
    for (int r = 0; r < m->n; r++)
      OG_REQUIRE(pks[r] != nullptr, "og_multi_withdraw_prove_sharded: pks[r] is null (og_multi_pk_load fills one key per device)");
    std::lock_guard<std::mutex> lk(m->mu);
    OG_REQUIRE(depth >= 1 && depth <= 64, "og_multi_withdraw_prove_sharded: depth must be 1..64");
    OG_REQUIRE(n == 0 || (inputs && rs && proofs_out), "og_multi_withdraw_prove_sharded: null argument");
    if (n == 0) return OG_OK;
    return multi_prove_sharded(m, pks, depth, n_pad3, n_pad2, inputs, nullptr, 0, n, rs, proofs_out, public_out);
  });
}

int og_multi_prove_sharded(og_multi* m, og_pk* const* pks, const uint8_t* witnesses, size_t n, const uint8_t* rs, uint8_t* proofs_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && pks, "og_multi_prove_sharded: null argument");
    for (int r = 0; r < m->n; r++) OG_REQUIRE(pks[r] != nullptr, "og_multi_prove_sharded: pks[r] is null (og_multi_pk_load fills one key per device)");
    OG_REQUIRE(n == 0 || (witnesses && rs && proofs_out), "og_multi_prove_sharded: null argument");
    if (n == 0) return OG_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    uint64_t info[4];
    OG_TRY(og_pk_info(pks[0], info));
    return multi_prove_sharded(m, pks, 0, 0, 0, nullptr, witnesses, (size_t)info[0] * 32, n, rs, proofs_out, nullptr);
  });
}

int og_multi_bases_create(og_multi* m, int group, const uint8_t* points, size_t n, int window_bits, int precompute, og_bases** bases_out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && bases_out && (n == 0 || points), "og_multi_bases_create: null argument");
    OG_REQUIRE(group == 1 || group == 2, "og_multi_bases_create: group must be 1 (G1) or 2 (G2)");
    OG_REQUIRE(window_bits == 0 || window_bits == 8 || window_bits == 12 || window_bits == 16,
               "og_multi_bases_create: window_bits must be 0, 8, 12 or 16");
    const int c = window_bits ? window_bits : (int)msm_pick_c(n);
    std::lock_guard<std::mutex> lk(m->mu);
    return multi_bases_create(m, group == 2, points, n, c, precompute, bases_out);
  });
}

void og_multi_bases_free(og_multi* m, og_bases** bases) {
  if (!m || !bases) return;
  for (int r = 0; r < m->n; r++) {
    if (bases[r]) og_bases_free(bases[r]);
    bases[r] = nullptr;
  }
}

int og_multi_msm(og_multi* m, og_bases* const* bases, const uint8_t* scalars, size_t n, uint8_t* out) {
  return guarded([&]() -> int {
    OG_REQUIRE(m && bases && out && (n == 0 || scalars), "og_multi_msm: null argument");
    for (int r = 0; r < m->n; r++) OG_REQUIRE(bases[r] != nullptr, "og_multi_msm: bases[r] is null (og_multi_bases_create fills one replica per device)");
    std::lock_guard<std::mutex> lk(m->mu);
    OG_REQUIRE(n <= bases[0]->n && (!bases[0]->precomp || n == bases[0]->n), "og_multi_msm: n does not fit the bases");
    return multi_msm(m, bases, scalars, n, out);
  });
}

}  // extern "C"
