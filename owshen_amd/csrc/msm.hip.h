// Internal interface of the Pippenger MSM (msm.hip), shared with the Groth16 prover (groth16.hip).
#pragma once
#include "ctx.h"

struct og_bases {
  int is_g2 = 0;
  size_t n = 0;        // points per window table
  int c = 16;          // window bits (signed digits -> 2^(c-1) buckets)
  int nwin = 16;       // ceil(255 / c)
  int precomp = 0;     // 1: tab holds nwin tables, tab[k][i] = 2^(c k) P_i  (single bucket set)
  uint8_t* tab_d = nullptr;  // affine, Montgomery form; (precomp ? nwin : 1) * n points
  int device = 0;
};

namespace og {

// Sorted signed-digit decomposition of `batch` scalar vectors (shared by every MSM over the
// same scalars, e.g. Groth16's A / B1 / B2 / L queries all consume the witness z).
struct DigitSort {
  size_t n = 0;
  int batch = 0;
  int c = 16, nwin = 16, precomp = 0;
  uint32_t own_mask = 0xffffffffu;  // bit k: window k is accumulated by this rank (window-sharded MSM; all ones = every window)
  int n_own = 16;          // popcount(own_mask & windows)
  size_t nkeys = 0;        // bucket slots per batch item = (precomp ? 1 : n_own) * 2^(c-1)
  size_t ecap = 0;         // entry capacity per batch item = n * nwin
  uint32_t* offsets = nullptr;  // [batch][nkeys + 1] exclusive prefix of bucket sizes
  uint32_t* cursor = nullptr;   // [batch][nkeys] scatter cursors
  uint32_t* entries = nullptr;  // [batch][ecap]: (table_index << 1) | negate
  uint32_t* order = nullptr;    // [batch][nkeys]: bucket ids by descending size (lane -> bucket map of the
                                // accumulation kernel, so the 64 lanes of a wave carry equal work)
};

size_t msm_pick_c(size_t n);
size_t msm_pick_query_c(size_t n);
int msm_nwin(int c);

// workspace-managed (ctx arena); valid until the next digit_sort on the same slot
// map_d (optional, device, n x u32): entry i takes the scalar at index map_d[i] of each vector -- density
// compaction: a query whose bases are mostly the point at infinity is stored compacted and visits only the
// wires it has a base for (what bellman calls the query density).
int msm_digit_sort(og_ctx* ctx, int slot, const uint8_t* scalars_d, size_t scalar_stride_bytes, size_t n,
                   const uint32_t* map_d, int batch, int c, int precomp, DigitSort* out);

// same, restricted to the windows k with k % win_world == win_rank (window-sharded MSM over several GPUs)
int msm_digit_sort_windows(og_ctx* ctx, int slot, const uint8_t* scalars_d, size_t scalar_stride_bytes, size_t n,
                           const uint32_t* map_d, int batch, int c, int precomp, int win_rank, int win_world, DigitSort* out);

// bucket accumulation + reduction; result[g] (XYZZ, Montgomery) for g < batch written to out_d
// (G1: 128 B each, G2: 256 B each).
int msm_run(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_xyzz_d);
// Two queries whose results are only ever ADDED (Groth16's C = sum z_i L_i + sum h_j H_j) are ONE MSM over the concatenated
// points: they share a bucket set -- same window bits, same batch, G1 -- and with it one bucket reduction, one heavy-bucket tail
// and one window combine per proof instead of two.  The two halves keep their own tables and digit sorts (the scalars come
// from different buffers at different times):
//   MSM_FIRST   accumulate into a fresh bucket set named by ctx->msm_tag, fold the heavy buckets in, stop (out is not written)
//   MSM_SECOND  same ctx->lane / msm_tag: add to those buckets, then heavy buckets (added in), reduction, combine -> out
enum MsmPhase { MSM_FULL = 0, MSM_FIRST = 1, MSM_SECOND = 2 };
int msm_run_phase(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* out_xyzz_d, int phase);
// Window-sharded form: this rank's partial result as an array of msm_partial_slots(bases) XYZZ points per batch item --
// one point per window (the points of windows this rank does not own are the point at infinity) for plain bases, ONE
// partial sum for precomputed tables.  After an all-gather of every rank's array, msm_combine adds the ranks' arrays
// slot by slot and runs Horner over the windows: out[g] = sum_k 2^(c k) sum_r partial[r][g][k].
int msm_partial_slots(const og_bases* bases);
int msm_run_partial(og_ctx* ctx, const og_bases* bases, const DigitSort& ds, uint8_t* partial_xyzz_d);
int msm_combine(og_ctx* ctx, const og_bases* bases, const uint8_t* gathered_xyzz_d, int world, int batch, uint8_t* out_xyzz_d);

// out[g] = sum_r gathered[r * rank_stride + g * sizeof(XYZZ)] for g < batch: the ranks' partial sums of one query, added
int msm_sum_ranks(og_ctx* ctx, int is_g2, const uint8_t* gathered_xyzz_d, size_t rank_stride, int world, int batch, uint8_t* out_xyzz_d);

int bases_create(og_ctx* ctx, int is_g2, const uint8_t* points_d, size_t n, int c, int precomp, og_bases** out);
void bases_destroy(og_bases* b);

// out (canonical affine bytes, device) = to_affine(xyzz[i]) for i < count
int xyzz_to_affine_bytes(og_ctx* ctx, int is_g2, const uint8_t* xyzz_d, uint8_t* out_d, size_t count);

// arena: named scratch buffers that grow on demand and live until og_shutdown
int arena_get(og_ctx* ctx, const char* name, size_t bytes, void** out);
bool arena_has(og_ctx* ctx, const char* name);

}  // namespace og

// ---- other internal entry points used by the Groth16 prover (groth16.hip) ----------------
namespace og {
// H-polynomial on device buffers (ntt.hip): a, b, c Montgomery evaluations (destroyed), tmp scratch,
// h_out canonical coefficients; all batch x d x 32 B
int h_poly_device(og_ctx* ctx, uint8_t* a, uint8_t* b, uint8_t* c, uint8_t* tmp, uint8_t* h_out, int log_d, int batch,
                  const hipEvent_t* gates = nullptr);
}  // namespace og
