/* coalescer -- the request coalescer of the node patch (ffi/burn_proof.patch, `prover::coalescer`), replayed from C.
 *
 * The metric is a BATCH of withdraw proofs, but the reference's `withdraw_handler`
 * (/root/reference/src/services/api_services/withdraw.rs:27-71) serves ONE request per HTTP call, next to a block producer
 * that wakes every 3 s (/root/reference/src/cli/node.rs:26-53).  Between the two sits a coalescer: requests queue on a
 * channel, a prover task drains up to N of them or waits at most T after the first, proves the batch on the GPU and answers
 * each request.  There is no rustc in this image, so the DESIGN is proven here, through nothing but include/owshen_gpu.h:
 * open-loop Poisson arrivals at a given rate are replayed against the same drain rule, and the latency of every request
 * (arrival -> its batch's proofs are on the host) is recorded.
 *
 *   mode 0  one request per call (what the handler does without a coalescer): og_withdraw_prove_batch_d(n = 1)
 *   mode 1  coalesced, one blocking call per batch
 *   mode 2  coalesced, one call kept ahead: og_withdraw_prove_batch_submit_d / og_job_poll / og_job_wait
 *
 * Arrival times are drawn up front (exponential gaps); the loop below is the prover task: "which requests have arrived by
 * now" is a comparison with the clock, so no second thread is needed to replay them faithfully -- a request's latency counts
 * from its drawn arrival time, however long the task was busy.
 *
 * usage: coalescer depth n_pad3 n_pad2 dense mode max_batch window_us duration_s seed rate [rate ...]
 * prints one JSON object per rate.  Built and swept by tools/coalescer.py. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "owshen_gpu.h"

#define CHECK(expr)                                                              \
  do {                                                                           \
    int rc_ = (expr);                                                            \
    if (rc_ != OG_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, og_last_error());           \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned long long rng_next(void) {
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return rng_state * 2685821657736338717ull;
}
static double rng_unit(void) { return ((double)(rng_next() >> 11) + 0.5) / 9007199254740992.0; }

static long long now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
static void sleep_until(long long t_ns) {
  struct timespec ts;
  if (t_ns - now_ns() < 20000) {  /* the last 20 us: spin (nanosleep's wake-up jitter is larger than that) */
    while (now_ns() < t_ns) {}
    return;
  }
  ts.tv_sec = (time_t)(t_ns / 1000000000ll);
  ts.tv_nsec = (long)(t_ns % 1000000000ll);
  clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL);
}
static int cmp_ll(const void* a, const void* b) {
  const long long x = *(const long long*)a, y = *(const long long*)b;
  return x < y ? -1 : x > y;
}

#define POOL 2048

typedef struct {
  og_ctx* ctx;
  og_pk* pk;
  int depth;
  uint64_t n_pad3, n_pad2;
  size_t rec;
  uint8_t *pool_recs, *pool_rs;       /* POOL request records / blinding pairs */
  /* two call slots */
  uint8_t *stage[2], *rs[2], *proofs[2], *pub[2];
  void* inputs_d[2];
  og_job* job[2];
  size_t first[2], count[2];          /* the requests a slot's batch holds: [first, first + count) */
} Driver;

static void gather(Driver* d, int slot, size_t first, size_t count) {
  for (size_t k = 0; k < count; k++) {
    const size_t p = (first + k) % POOL;
    memcpy(d->stage[slot] + k * d->rec, d->pool_recs + p * d->rec, d->rec);
    memcpy(d->rs[slot] + k * 64, d->pool_rs + p * 64, 64);
  }
  CHECK(og_memcpy_h2d(d->ctx, d->inputs_d[slot], d->stage[slot], count * d->rec));
  d->first[slot] = first;
  d->count[slot] = count;
}

int main(int argc, char** argv) {
  if (argc < 11) {
    fprintf(stderr, "usage: coalescer depth n_pad3 n_pad2 dense mode max_batch window_us duration_s seed rate [rate ...]\n");
    return 2;
  }
  Driver d;
  memset(&d, 0, sizeof d);
  d.depth = atoi(argv[1]);
  d.n_pad3 = strtoull(argv[2], 0, 10);
  d.n_pad2 = strtoull(argv[3], 0, 10);
  const int dense = atoi(argv[4]), mode = atoi(argv[5]);
  size_t max_batch = (size_t)atoi(argv[6]);
  const long long window_ns = (long long)atoi(argv[7]) * 1000ll;
  const double duration_s = atof(argv[8]);
  rng_state ^= strtoull(argv[9], 0, 10) * 0x2545F4914F6CDD1Dull;
  if (mode == 0) max_batch = 1;
  CHECK(og_init(0, &d.ctx));
  if (getenv("COALESCER_HOST_CHAINS"))  /* what a node proving per request would set once: og_set_host_chains(ctx, k) */
    CHECK(og_set_host_chains(d.ctx, atoi(getenv("COALESCER_HOST_CHAINS"))));
  og_r1cs* r1cs = NULL;
  CHECK(og_withdraw_r1cs(d.ctx, d.depth, d.n_pad3, d.n_pad2, dense, &r1cs));
  uint8_t toxic[160];
  memset(toxic, 0, sizeof toxic);
  for (int k = 0; k < 5; k++) toxic[32 * k] = (uint8_t)(101 + k);
  uint8_t *pk_blob = NULL, *vk_blob = NULL;
  size_t pk_len = 0, vk_len = 0;
  CHECK(og_setup(d.ctx, r1cs, toxic, &pk_blob, &pk_len, &vk_blob, &vk_len));
  og_r1cs_free(r1cs);
  CHECK(og_pk_load(d.ctx, pk_blob, pk_len, &d.pk));
  og_blob_free(pk_blob);
  d.rec = (size_t)(8 + d.depth) * 32;
  d.pool_recs = (uint8_t*)malloc(POOL * d.rec);
  d.pool_rs = (uint8_t*)malloc(POOL * 64);
  for (size_t i = 0; i < POOL * d.rec; i++) d.pool_recs[i] = (uint8_t)(rng_next() >> 56);
  for (size_t g = 0; g < POOL; g++) {
    uint8_t* r = d.pool_recs + g * d.rec;
    for (int f = 0; f < 8 + d.depth; f++) r[f * 32 + 31] &= 0x1F;   /* < 2^253 < r */
    memset(r + 3 * 32 + 20, 0, 12);                                  /* recipient, token: 160-bit addresses */
    memset(r + 6 * 32 + 20, 0, 12);
    memset(r + 7 * 32 + 8, 0, 24);                                   /* chain id, index: u64 */
    memset(r + 5 * 32 + 8, 0, 24);
    if (d.depth < 64) {
      uint64_t idx;
      memcpy(&idx, r + 5 * 32, 8);
      idx &= ((uint64_t)1 << d.depth) - 1;
      memcpy(r + 5 * 32, &idx, 8);
    }
  }
  for (size_t i = 0; i < POOL * 64; i++) d.pool_rs[i] = (uint8_t)(rng_next() >> 56);
  for (size_t g = 0; g < POOL; g++) { d.pool_rs[g * 64 + 31] &= 0x1F; d.pool_rs[g * 64 + 63] &= 0x1F; }
  for (int s = 0; s < 2; s++) {
    d.stage[s] = (uint8_t*)malloc(max_batch * d.rec);
    d.rs[s] = (uint8_t*)malloc(max_batch * 64);
    d.proofs[s] = (uint8_t*)malloc(max_batch * 256);
    d.pub[s] = (uint8_t*)malloc(max_batch * 192);
    CHECK(og_malloc(d.ctx, max_batch * d.rec, &d.inputs_d[s]));
  }
  /* warm-up: every schedule the sweep will meet grows its scratch now, not inside a measured request */
  {
    size_t sizes[8], ns = 0;
    for (size_t b = max_batch; b >= 1 && ns < 8; b = b > 4 ? b / 4 : b - 1) { sizes[ns++] = b; if (b == 1) break; }
    for (size_t k = 0; k < ns; k++)
      for (int rep = 0; rep < 2; rep++) {
        gather(&d, 0, 0, sizes[k]);
        CHECK(og_withdraw_prove_batch_d(d.ctx, d.pk, d.depth, d.n_pad3, d.n_pad2, (const uint8_t*)d.inputs_d[0], sizes[k], d.rs[0], d.proofs[0], d.pub[0]));
      }
    for (size_t g = 0; g < (max_batch < 4 ? max_batch : 4); g++) {  /* and the proofs are proofs */
      int ok = 0;
      gather(&d, 0, 0, max_batch < 4 ? max_batch : 4);
      CHECK(og_withdraw_prove_batch_d(d.ctx, d.pk, d.depth, d.n_pad3, d.n_pad2, (const uint8_t*)d.inputs_d[0], d.count[0], d.rs[0], d.proofs[0], d.pub[0]));
      CHECK(og_verify(vk_blob, vk_len, d.pub[0] + g * 192, 6, d.proofs[0] + g * 256, &ok));
      if (!ok) { fprintf(stderr, "warm-up proof %zu does not verify\n", g); return 4; }
    }
  }
  for (int a = 10; a < argc; a++) {
    const double rate = atof(argv[a]);
    /* arrivals */
    size_t cap = (size_t)(rate * duration_s * 1.5) + 64, total = 0;
    long long* arrival = (long long*)malloc(cap * sizeof(long long));
    double t = 0;
    for (;;) {
      t += -log(rng_unit()) / rate;
      if (t >= duration_s || total == cap) break;
      arrival[total++] = (long long)(t * 1e9);
    }
    if (total == 0) { free(arrival); continue; }
    long long* lat = (long long*)malloc(total * sizeof(long long));
    size_t n_batches = 0, largest = 0, next = 0, oldest = 0;
    int inflight = 0, head = 0;   /* mode 2: slots in flight, `head` = the older one */
    const long long t0 = now_ns() + 2000000;  /* the replay starts 2 ms from now */
    for (size_t i = 0; i < total; i++) arrival[i] += t0;
    long long t_last = t0;
    while (next < total || inflight > 0) {
      long long now = now_ns();
      if (mode == 2 && inflight > 0) {  /* reap the older call if it is done */
        int done = 0;
        CHECK(og_job_poll(d.ctx, d.job[head], &done));
        if (done || inflight == 2 || next >= total) {  /* (two in flight, or nothing left to submit: block) */
          CHECK(og_job_wait(d.ctx, d.job[head]));
          t_last = now_ns();
          for (size_t k = 0; k < d.count[head]; k++) lat[d.first[head] + k] = t_last - arrival[d.first[head] + k];
          oldest += d.count[head];
          d.job[head] = NULL;
          head ^= 1;
          inflight--;
          continue;
        }
      }
      if (next >= total) continue;
      if (arrival[next] > now) {  /* nobody is waiting */
        if (mode == 2 && inflight > 0) sleep_until(now + 50000 < arrival[next] ? now + 50000 : arrival[next]);  /* keep polling the call in flight */
        else sleep_until(arrival[next]);
        continue;
      }
      /* the drain rule: up to max_batch requests, or whoever has arrived `window` after the first */
      size_t k = 1;
      while (k < max_batch && next + k < total && arrival[next + k] <= now) k++;
      const long long deadline = arrival[next] + window_ns;
      if (k < max_batch && now < deadline && next + k < total) {
        const long long wake = arrival[next + k] < deadline ? arrival[next + k] : deadline;
        if (mode == 2 && inflight > 0 && wake > now + 50000) sleep_until(now + 50000); else sleep_until(wake);
        continue;
      }
      if (mode == 2) {
        const int slot = inflight ? head ^ 1 : head;
        gather(&d, slot, next, k);
        CHECK(og_withdraw_prove_batch_submit_d(d.ctx, d.pk, d.depth, d.n_pad3, d.n_pad2, (const uint8_t*)d.inputs_d[slot], k, d.rs[slot], d.proofs[slot],
                                               d.pub[slot], &d.job[slot]));
        inflight++;
      } else {
        gather(&d, 0, next, k);
        CHECK(og_withdraw_prove_batch_d(d.ctx, d.pk, d.depth, d.n_pad3, d.n_pad2, (const uint8_t*)d.inputs_d[0], k, d.rs[0], d.proofs[0], d.pub[0]));
        t_last = now_ns();
        for (size_t j = 0; j < k; j++) lat[next + j] = t_last - arrival[next + j];
      }
      next += k;
      n_batches++;
      if (k > largest) largest = k;
    }
    (void)oldest;
    double mean = 0;
    for (size_t i = 0; i < total; i++) mean += (double)lat[i];
    mean /= (double)total;
    qsort(lat, total, sizeof(long long), cmp_ll);
    const double span_s = (double)(t_last - t0) / 1e9;
    printf("{\"mode\": %d, \"offered_per_s\": %.1f, \"requests\": %zu, \"achieved_per_s\": %.2f, \"batches\": %zu, \"mean_batch\": %.2f, \"largest_batch\": %zu, "
           "\"max_batch\": %zu, \"window_us\": %lld, \"duration_s\": %.2f, \"drain_s\": %.3f, \"latency_ms\": {\"mean\": %.3f, \"p50\": %.3f, \"p90\": %.3f, \"p99\": %.3f, "
           "\"max\": %.3f}}\n",
           mode, rate, total, (double)total / span_s, n_batches, (double)total / (double)n_batches, largest, max_batch, window_ns / 1000, duration_s,
           span_s - duration_s > 0 ? span_s - duration_s : 0.0, mean / 1e6, (double)lat[total / 2] / 1e6, (double)lat[(size_t)((double)(total - 1) * 0.9)] / 1e6,
           (double)lat[(size_t)((double)(total - 1) * 0.99)] / 1e6, (double)lat[total - 1] / 1e6);
    fflush(stdout);
    free(arrival);
    free(lat);
  }
  for (int s = 0; s < 2; s++) {
    CHECK(og_free(d.ctx, d.inputs_d[s]));
    free(d.stage[s]); free(d.rs[s]); free(d.proofs[s]); free(d.pub[s]);
  }
  og_pk_free(d.pk);
  og_blob_free(vk_blob);
  og_shutdown(d.ctx);
  free(d.pool_recs); free(d.pool_rs);
  return 0;
}
