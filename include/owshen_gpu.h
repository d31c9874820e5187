/* owshen_gpu.h -- C ABI of libowshen_gpu.so, the MI355X (gfx950) Groth16 prover path.
 *
 * Drop-in boundary.  The reference snapshot has NO FFI seam and NO
 * prover, so none of these entry points *replaces* an existing
 * extern; each one cites the reference call site it is shaped to serve:
 *
 *   - proof generation would be called from `withdraw_handler`
 *     (/root/reference/src/services/api_services/withdraw.rs:27-71, between the
 *     ECDSA recover at :34 and the sequencer re-sign at :56);
 *   - proof verification would gate `burn_tx`
 *     (/root/reference/src/blockchain/tx/burn_tx.rs:11-32, before the balance debit);
 *   - every field element crossing this boundary uses the byte format of the
 *     reference's `Fp::to_repr()`: 32 bytes, little-endian, canonical (< modulus)
 *     (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11);
 *   - errors follow the callers' `anyhow::Result` convention
 *     (/root/reference/src/utils.rs:5-20): 0 = ok, negative = error, message via
 *     og_last_error(); nothing unwinds across the boundary.
 *
 * Conventions
 *   Fr / Fq element : 32 B little-endian canonical.
 *   G1 affine       : x || y (64 B); the point at infinity is 64 zero bytes.
 *   G2 affine       : x.c0 || x.c1 || y.c0 || y.c1 (128 B); infinity = 128 zero bytes.
 *   proof           : A(G1) || B(G2) || C(G1) = 256 B.
 *   "_d" pointers are DEVICE pointers (HBM, e.g. from og_malloc or a torch tensor's
 *   data_ptr()); all other pointers are host memory owned by the caller; the library
 *   never retains a caller pointer past return.  One og_ctx per GPU (one process per
 *   GPU); calls on one ctx are blocking and internally serialised.
 *   Randomness is never drawn inside the library: (r, s) are explicit inputs.
 */
#ifndef OWSHEN_GPU_H
#define OWSHEN_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct og_ctx og_ctx;
typedef struct og_bases og_bases; /* device-resident, Montgomery-form MSM bases */
typedef struct og_pk og_pk;       /* device-resident proving key + R1CS matrices */

#define OG_OK 0
#define OG_ERR_INVALID (-1)  /* bad argument / malformed input */
#define OG_ERR_HIP (-2)      /* HIP runtime failure (message has the hipError string) */
#define OG_ERR_NO_DEVICE (-3)
#define OG_ERR_UNSATISFIED (-4) /* witness does not satisfy the circuit: some row has a*b != c, or wire 0 != 1 (exact) */

/* ---- lifecycle ---------------------------------------------------------- */
int og_init(int device, og_ctx** out);
void og_shutdown(og_ctx* ctx);
const char* og_last_error(void); /* thread-local, valid until the next failing call */
int og_device_count(void);
int og_sync(og_ctx* ctx);
/* the HIP stream every kernel of this ctx is launched on (hipStream_t as void*) */
void* og_stream(og_ctx* ctx);

/* ---- device memory plumbing --------------------------------------------- */
int og_malloc(og_ctx* ctx, size_t bytes, void** out_d);
int og_free(og_ctx* ctx, void* ptr_d);
int og_memcpy_h2d(og_ctx* ctx, void* dst_d, const void* src, size_t bytes);
int og_memcpy_d2h(og_ctx* ctx, void* dst, const void* src_d, size_t bytes);

/* ---- N1: field arithmetic (test / micro-bench surface) --------------------
 * op: 0 add, 1 sub, 2 mul, 3 inv(a) (b ignored; inv(0) = 0).  field: 0 = Fr, 1 = Fq. */
int og_field_op_d(og_ctx* ctx, int field, int op, const uint8_t* a_d, const uint8_t* b_d,
                  uint8_t* out_d, size_t n);
/* chained Montgomery multiplications: x <- x*y repeated `iters` times per element;
 * returns kernel milliseconds (HIP events on the ctx stream) in *ms_out. */
int og_field_mulchain_d(og_ctx* ctx, int field, uint8_t* x_d, const uint8_t* y_d, size_t n,
                        int iters, float* ms_out);

/* the same chain as a LATENCY probe: form 0 = the lane-local product, one lane per element (64-lane workgroups); form 1 = the
 * wave-wide "w9" product (csrc/field_w9.hip.h: one element over nine lanes of a wave, ~80 instead of 205 instructions per
 * product), one wave per element, n <= 65535.  Same bytes out as og_field_mulchain_d.  *wave_cycles_out (may be NULL) = the
 * longest wave's loop in shader cycles. */
int og_field_mulchain_lat_d(og_ctx* ctx, int field, int form, uint8_t* x_d, const uint8_t* y_d, size_t n,
                            int iters, float* ms_out, uint64_t* wave_cycles_out);

/* VALU instruction-rate probe: every lane runs iters x 16 independent instructions.
 * kind: 0 v_mad_u64_u32, 1 v_mul_lo_u32, 2 v_mul_hi_u32, 3 v_add_co_u32, 4 v_addc_co_u32,
 * 5 v_lshl_add_u64, 6 v_add_u32, 7 v_mad_u32_u24, 8 v_mul_hi_u32_u24, 9 v_mov_b32.
 * blocks x 256 lanes.  Kernel milliseconds in *ms_out. */
int og_ubench(og_ctx* ctx, int kind, int iters, int blocks, float* ms_out);
/* same, plus kinds 10 v_fma_f64 and 11 v_add_f64; *wave_cycles_out = the longest wave's loop time in SHADER CYCLES
 * (s_memtime): with all waves resident, cycles / (iters x 16 x waves per SIMD) is the issue cost per wave-instruction
 * with no clock assumption, and cycles / ms the effective clock of the run.
 * Further kinds probe what the column-serial field products rely on: 12 one dependent v_mad_u64_u32 chain,
 * 13 v_lshrrev_b64, 14 the chain with an s_nop after every instruction, 15 / 16 two / four interleaved chains, 17 the chain
 * with vcc as the carry-out destination, 20 + k: k interleaved chains, 40 + k: the same ping-ponging between two register
 * pairs, 100 + 5 a + b: one chain on v[40:41] with its factors in VGPR banks a and b (b = 4: second factor in an SGPR);
 * 202: `iters` wave-wide MiMC7 rounds on a lone wave (blocks x 64 lanes; mimc7.hip.h w9_mimc7_round: two rows of the wave, the
 * 32-bit Montgomery digit -- what the launches use; 200 / 201, the one-row form and the 29-bit digit, in the hooks build only):
 * *wave_cycles_out / iters = cycles per round. */
int og_ubench_cycles(og_ctx* ctx, int kind, int iters, int blocks, float* ms_out, uint64_t* wave_cycles_out);
/* co-residency probe (what a short kernel on a second stream gets done beside a persistent kernel that holds `wgs_per_cu`
 * one-wave, 128-register workgroups on every CU): kind 0 = the resident waves run a dependent v_mad_u64_u32 chain, 1 = they
 * sleep.  out[0] = resident kernel ms, out[1] = filler ms beside it (queued delay_us later), out[2] = the filler alone. */
int og_ubench_coresidency(og_ctx* ctx, int wgs_per_cu, int kind, int iters, int filler_blocks, int filler_threads,
                          int filler_lds, int filler_prio, int filler_work, int delay_us, float out[3]);

/* ---- N5: MiMC7 (circomlib convention, 91 rounds) -------------------------- */
/* the 91 round constants, canonical LE, for cross-checking against the oracle */
int og_mimc7_constants(og_ctx* ctx, uint8_t out[91 * 32]);
/* out[i] = MultiMiMC7([left[i], right[i]], key = 0) */
int og_mimc7_hash2_d(og_ctx* ctx, const uint8_t* left_d, const uint8_t* right_d, uint8_t* out_d,
                     size_t n);
/* n paths of `depth` levels.  index bit l = 1 => node at level l is a right child.
 * siblings: n x depth x 32 B.  nodes_out: n x (depth+1) x 32 B (leaf first, root last). */
int og_mimc7_merkle_paths_d(og_ctx* ctx, const uint8_t* leaves_d, const uint64_t* indices_d,
                            const uint8_t* siblings_d, int depth, uint8_t* nodes_out_d, size_t n);
/* full tree over n = 2^k leaves.  nodes_out: (2n-1) x 32 B in level order:
 * [leaves (n) | level 1 (n/2) | ... | root (1)]. */
int og_mimc7_tree_build_d(og_ctx* ctx, const uint8_t* leaves_d, size_t n, uint8_t* nodes_out_d);
/* Batched append to an append-only depth-`depth` tree kept as a frontier (the "filled subtrees" of deposit contracts;
 * empty leaves are 0) -- the commitment tree `mint_tx` (/root/reference/src/blockchain/tx/mint_tx.rs:11-49) would feed,
 * SURVEY.md 8f-3.  frontier_*_d: depth x 32 B (must not alias); next_index = leaves already in the tree; k >= 1 new
 * leaves; root_out_d: 32 B.  Level by level on the GPU: ~k hashes in total, `depth` launches.  Frontier entries at
 * levels where bit l of (next_index + k) is 0 are unspecified: an append never reads them before rewriting them. */
int og_mimc7_append_d(og_ctx* ctx, int depth, const uint8_t* frontier_in_d, uint64_t next_index, const uint8_t* leaves_d,
                      size_t k, uint8_t* frontier_out_d, uint8_t* root_out_d);

/* ---- airdrop signatures: batched EdDSA verification on BabyJubJub with the MiMC7 sponge ----------
 * The reference's `PointCompressed::verify` (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:99-115)
 * with its placeholder product hash (:202-204) replaced by MultiMiMC7: accepts iff pk and R lie on the curve and
 *   s * BASE == R + h * pk,   h = MultiMiMC7([R.x, R.y, pk.x, pk.y, message], key 0).
 * records_d: n records of 6 x 32 B canonical LE: pk.x | pk.y | R.x | R.y | s | message (public keys affine: decompression,
 * mod.rs:88-98, stays on the host).  ok_out: HOST, n x u32 (1 accept / 0 reject; a non-canonical field element rejects). */
int og_eddsa_verify_batch_d(og_ctx* ctx, const uint8_t* records_d, size_t n, uint32_t* ok_out);

/* ---- N4: radix-2 Fr NTT (domain generator 7^((r-1)/n), coset generator 7) ------
 * batch transforms of size n = 2^log_n, natural order in and out, canonical bytes.
 *   inverse = 0, coset = 0 : out[i] = sum_j in[j] w^(ij)
 *   inverse = 0, coset = 1 : evaluates at 7 w^i
 *   inverse = 1            : the inverse maps (including the 1/n and 7^-j factors)
 * in_d and out_d must not overlap. */
int og_ntt_fr_d(og_ctx* ctx, const uint8_t* in_d, uint8_t* out_d, int log_n, int batch, int inverse,
                int coset);
/* Groth16 quotient: h = (A*B - C)/Z from the evaluations of A, B, C over the size-2^log_d
 * domain (3 iNTT, 3 coset NTT, pointwise, 1 coset iNTT).  batch x d x 32 B each; h_out_d gets
 * the d coefficients of h (coefficient d-1 is 0 for a satisfied R1CS). */
int og_h_poly_d(og_ctx* ctx, const uint8_t* a_d, const uint8_t* b_d, const uint8_t* c_d, int log_d,
                int batch, uint8_t* h_out_d);

/* ---- N2/N3: Pippenger MSM over G1 / G2 -------------------------------------
 * Bases are imported once (canonical affine -> device-resident Montgomery tables) and reused
 * by every MSM over them -- the shape of a Groth16 proving key, which is fixed across proofs.
 *   group        : 1 = G1 (64 B points), 2 = G2 (128 B points)
 *   window_bits  : 8, 12 or 16 (15 or 17 with precompute); 0 = choose from n
 *   precompute   : 1 = also store 2^(c k) P_i for every window k (nwin x the memory, one bucket
 *                  set instead of nwin: faster for repeated MSMs); 0 = plain
 * Points are NOT checked for curve membership (the key comes from a trusted setup). */
int og_bases_create_d(og_ctx* ctx, int group, const uint8_t* points_d, size_t n, int window_bits,
                      int precompute, og_bases** out);
void og_bases_free(og_bases* bases);
/* `batch` independent MSMs over the same bases: result[g] = sum_i s[g][i] * P_i, i < n.
 * scalars_d: batch vectors of n x 32 B, vector g starting at scalars_d + g * stride_bytes.
 * n may be smaller than the number of bases unless the bases were created with precompute.
 * out: HOST buffer, batch x (64 | 128) B canonical affine (zeros = point at infinity). */
int og_msm_d(og_ctx* ctx, const og_bases* bases, const uint8_t* scalars_d, size_t n, int batch,
             size_t stride_bytes, uint8_t* out);

/* Window-sharded MSM across GPUs (BASELINE.json configs[3]): bases replicated on every rank, scalars
 * broadcast, rank `win_rank` of `win_world` accumulates only the windows k with k % win_world == win_rank.
 *   og_msm_windows_d  this rank's share -> partial_out_d: og_msm_partial_slots(bases) points in the library's internal
 *                     extended form (128 B each in G1, 256 B in G2; opaque, only og_msm_combine_d reads them): one per
 *                     window for plain bases (infinity for windows the rank does not own), ONE partial sum for
 *                     precomputed tables.  n must equal the number of bases for precomputed tables.
 *   (all-gather of the ranks' partial arrays: <= 16 x 256 B per rank -- an RCCL all-gather, never an all-reduce: curve
 *    points do not add limb-wise.  owshen_amd/shard.py does it with torch.distributed, og_multi_* inside the library.)
 *   og_msm_combine_d  gathered_d: world x slots points, rank-major -> sum over ranks per slot, Horner over the windows;
 *                     out: HOST, 64 | 128 B canonical affine. */
int og_msm_partial_slots(const og_bases* bases);
int og_msm_windows_d(og_ctx* ctx, const og_bases* bases, const uint8_t* scalars_d, size_t n, int win_rank, int win_world,
                     uint8_t* partial_out_d);
int og_msm_combine_d(og_ctx* ctx, const og_bases* bases, const uint8_t* gathered_d, int world, uint8_t* out);

/* ---- N6: Groth16 proving -----------------------------------------------------
 * The proving key is parsed once and stays resident in HBM (R1CS matrices in CSR, the five query
 * vectors as precomputed window tables).  Serialized key, little-endian, every section padded to
 * a multiple of 32 B:
 *   u64 x 10 : "OWPK0001", n_wires, n_pub, log_d, n_rows, nnz_a, nnz_b, nnz_c, flags, 0
 *              (flags bit 0: no C matrix -- nnz_c = 0 -- and C z := (A z) o (B z): a key that og_zkey_import made)
 *   alpha_g1 | beta_g1 | delta_g1 | 64 B pad | beta_g2 | delta_g2
 *   for M in A, B, C: row_ptr (n_rows+1 u32) | col (nnz u32) | val (nnz x 32 B)
 *   a_query (m) | b_g1_query (m) | b_g2_query (m, G2) | l_query (m - n_pub - 1) | h_query (d - 1)
 * Rows = the constraints followed by the n_pub + 1 input-consistency rows (A = wire i, B = C = 0);
 * d = 2^log_d >= n_rows.  Wire 0 is the constant 1, wires 1..n_pub are public.
 * A witness is n_wires x 32 B; (r, s) are the caller's blinding scalars, r || s (64 B), explicit so that
 * a proof is a pure function of (key, witness, r, s).  A proof is A (G1) || B (G2) || C (G1) = 256 B.
 * OG_ERR_UNSATISFIED: a witness does not satisfy the R1CS (og_last_error names the first one).  The check is exact: every
 * QAP row product a_i b_i = c_i is tested and wire 0 must be the constant 1, so OG_OK means every returned proof verifies.
 * OG_ERR_INVALID: a witness holds a value that is not the canonical encoding of an Fr element (>= r); og_last_error names
 * the witness and its first such wire.  Checked on the GPU for every caller-supplied witness (one more read of it), before
 * "does not satisfy": the reference's `Fp::from_repr` rejects such bytes
 * (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11), and reducing them silently would prove a
 * statement about a different byte string than the caller holds. */
int og_pk_load(og_ctx* ctx, const uint8_t* blob, size_t len, og_pk** out);
/* Frees the key's device memory.  Waits for the device first, so a submitted call that still reads the key finishes its
 * kernels -- but its results are only delivered by og_job_wait, which the caller still owes (or og_job_abandon). */
void og_pk_free(og_pk* pk);
/* info[0..3] = n_wires, n_pub, log_d, n_rows */
int og_pk_info(const og_pk* pk, uint64_t info[4]);
/* Query density: a wire whose base is the point at infinity in a query (the wire never occurs in that matrix) is dropped
 * from that query's table and digit sort -- except that queries whose wire lists coincide up to a few wires share one list
 * (and one digit sort per sub-batch); the few bases a query lacks stay in its table as points at infinity and are skipped.
 * out[0..3] = points actually accumulated per proof by the A query (G1), the B query (once in G1 and once in G2), the L
 * query (G1) and the H query (G1, d - 1). */
int og_pk_density(const og_pk* pk, uint64_t out[4]);
/* Window bits of the A, B, L and H queries' precomputed tables: a query of n points costs n x ceil(255 / bits) bucket
 * additions per proof (16 bits from 8 k points on, 17 bits from 160 k points on: owshen_amd/csrc/msm.hip). */
int og_pk_windows(const og_pk* pk, uint64_t out[4]);
/* Witnesses in HOST memory (n x n_wires x 32 B; pageable is fine): each sub-batch is copied to the device inside the
 * prover's stage pipeline, under the accumulations of the sub-batch before it -- measured 510 proofs/s against 519 with
 * the same 2^18-wire witnesses already resident (og_prove_batch_d), 8.4 MB per proof over PCIe. */
int og_prove(og_ctx* ctx, const og_pk* pk, const uint8_t* witness, const uint8_t rs[64], uint8_t proof_out[256]);
int og_prove_batch(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses, size_t n, const uint8_t* rs,
                   uint8_t* proofs_out);
/* same, witnesses already resident in HBM (n x n_wires x 32 B); rs and proofs_out are host buffers */
int og_prove_batch_d(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses_d, size_t n, const uint8_t* rs,
                     uint8_t* proofs_out);

/* ---- multi-GPU from ONE process -----------------------------------------------------
 * The reference node is a single process with one Context (/root/reference/src/cli/node.rs:71-76), so the library itself
 * drives every GPU of the node: one og_ctx per device, one host thread per device per call, RCCL (ncclCommInitAll) for the
 * one exchange step that exists.  Handle arrays (og_pk**, og_bases**) have og_multi_size() entries, one per device.
 *   og_multi_init          n_devices = 0: every visible device.  Fails if fewer are visible than asked for.
 *   og_multi_prove_batch   proofs sharded across the devices in contiguous slices (key replicated, NO data-path collective:
 *   og_multi_withdraw_...  proofs are independent units); same bytes as the single-device calls.  All buffers are HOST.
 *   og_multi_msm           one MSM window-sharded over the devices (BASELINE.json configs[3] names this sharding): scalars
 *                          (host) -> device 0 -> ncclBroadcast over xGMI; rank g accumulates the windows k = g (mod G);
 *                          ncclAllGather of the per-window points (<= 16 x 256 B per rank; an all-gather, not an all-reduce:
 *                          curve points do not add limb-wise); Horner combine.  out: host, 64 | 128 B canonical affine. */
typedef struct og_multi og_multi;
int og_multi_init(int n_devices, og_multi** out);
void og_multi_shutdown(og_multi* m);
int og_multi_size(const og_multi* m);
/* out[0], out[1] = the contiguous slice [lo, hi) of a batch of n proofs that device `rank` proves (og_multi_prove_batch /
 * og_multi_withdraw_prove_batch): sizes differ by at most one, the first n mod G devices take the larger ones, a device whose
 * slice is empty sits the call out.  BASELINE.json configs[3] (4096 proofs over 8 GPUs) = 512 each. */
int og_multi_slice(const og_multi* m, size_t n, int rank, size_t out[2]);
og_ctx* og_multi_ctx(og_multi* m, int rank);
/* out[0] = the HIP device ordinal rank `rank` is bound to; pci_out = that device's PCI address ("0000:5d:00.0"); when the
 * devices share an RCCL communicator (more than one device): out[1] = ncclCommCount, out[2] = ncclCommUserRank, out[3] =
 * ncclCommCuDevice of this rank's communicator, else zeros.  A bench line quotes these so that "N ranks on N GPUs" is a
 * reading of the runtime, not a claim. */
int og_multi_device_info(const og_multi* m, int rank, uint64_t out[4], char pci_out[32]);
int og_multi_pk_load(og_multi* m, const uint8_t* blob, size_t len, og_pk** pks_out);
void og_multi_pk_free(og_multi* m, og_pk** pks);
int og_multi_prove_batch(og_multi* m, og_pk* const* pks, const uint8_t* witnesses, size_t n, const uint8_t* rs,
                         uint8_t* proofs_out);
/* public_out (host, may be NULL): n x 6 x 32 B, as og_withdraw_prove_batch_d */
int og_multi_withdraw_prove_batch(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2,
                                  const uint8_t* inputs, size_t n, const uint8_t* rs, uint8_t* proofs_out,
                                  uint8_t* public_out);
/* Window-sharded PROVING (BASELINE.json north_star: "MSM windows shard naturally across GPUs"; configs[3] as written): ONE
 * batch of n proofs on all N devices together -- every device generates the n witnesses and quotients itself (replicated:
 * they are latency chains / a few per cent of a proof), sorts and accumulates only the windows k = rank (mod N) of all five
 * queries over the replicated key, the per-proof partial points (768 B per proof and rank) are exchanged with ONE
 * ncclAllGather over xGMI -- never an all-reduce: curve points do not add limb-wise -- and device 0 adds the shares and
 * assembles the proofs.  Same arguments, same bytes out, same errors as og_multi_withdraw_prove_batch / og_multi_prove_batch
 * (which shard by PROOFS and stay the throughput form: a proof-sharded batch does no work twice); this form cuts the latency
 * of one request or a handful, the case of the one-request-per-call site
 * /root/reference/src/services/api_services/withdraw.rs:27-71.  N <= 15 (one rank per 17-bit window at most). */
int og_multi_withdraw_prove_sharded(og_multi* m, og_pk* const* pks, int depth, uint64_t n_pad3, uint64_t n_pad2,
                                    const uint8_t* inputs, size_t n, const uint8_t* rs, uint8_t* proofs_out,
                                    uint8_t* public_out);
int og_multi_prove_sharded(og_multi* m, og_pk* const* pks, const uint8_t* witnesses, size_t n, const uint8_t* rs,
                           uint8_t* proofs_out);
int og_multi_bases_create(og_multi* m, int group, const uint8_t* points, size_t n, int window_bits, int precompute,
                          og_bases** bases_out);
void og_multi_bases_free(og_multi* m, og_bases** bases);
int og_multi_msm(og_multi* m, og_bases* const* bases, const uint8_t* scalars, size_t n, uint8_t* out);

/* ---- N6: Groth16 verification (CPU only: no og_ctx, no GPU -- the `burn_tx` seam,
 * /root/reference/src/blockchain/tx/burn_tx.rs:11-32, must work on a sequencer without one) ----------
 * Evaluates the EIP-197 predicate e(-A,B) e(alpha,beta) e(IC_0 + sum x_i IC_i, gamma) e(C,delta) == 1.
 * vk: "OWVK0001" | u64 n_pub | alpha_g1 (64) | beta_g2 (128) | gamma_g2 (128) | delta_g2 (128) | IC ((n_pub+1) x 64).
 * public_inputs: n_pub x 32 B.  *ok_out = 1 accept / 0 reject; a proof with a non-canonical coordinate, a point
 * off the curve or outside the r-torsion, or a public input >= r is a REJECT (ok = 0, OG_OK), a malformed key
 * is OG_ERR_INVALID. */
int og_verify(const uint8_t* vk, size_t vk_len, const uint8_t* public_inputs, size_t n_pub,
              const uint8_t proof[256], int* ok_out);

/* ---- withdraw circuit: batched witness generation (N5 feeding N6) ----------------------------
 * The statement (public: root, nullifier_hash, recipient, amount, token, chain_id; private: nullifier, secret and a
 * depth-`depth` MiMC7 Merkle path; leaf = H(H(nullifier, secret), H(amount, token))) and its wire order are specified in
 * oracle/py/withdraw.py and built as an R1CS by og_withdraw_r1cs / owshen_amd/circuit.py.  The six public inputs cover
 * what the reference's gate signs (/root/reference/contracts/src/Owshen.sol:69: msg.sender, token, amount, id, chainid;
 * nullifier_hash stands in for the replay id).  n_pad3 / n_pad2 append synthetic multiplication gates
 * that size the statement (BASELINE.json: "MSM ~2^20 G1 points, Fr NTT 2^17"); 0 / 0 is the natural circuit.
 * inputs_d: n records of (8 + depth) x 32 B:
 *   nullifier | secret | amount | recipient | pad_seed | index (u64, low bytes) | token | chain_id | siblings[depth]
 * witness_out_d: n x n_wires x 32 B.  shape[0..2] = n_wires, n_constraints, n_pub. */
int og_withdraw_shape(int depth, uint64_t n_pad3, uint64_t n_pad2, uint64_t shape[3]);
int og_withdraw_witness_d(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, const uint8_t* inputs_d,
                          size_t n, uint8_t* witness_out_d);
/* input records -> proofs in one call (what `withdraw_handler` wants): every sub-batch's witnesses are generated by
 * the lane that proves them, in lane-private scratch, so the full n x n_wires witness array never exists and the
 * latency-bound MiMC7 walk overlaps the other lane's MSMs.  Same bytes as og_withdraw_witness_d + og_prove_batch_d.
 * pk must be a key for this (depth, n_pad3, n_pad2) shape.  rs: n x 64 B host, proofs_out: n x 256 B host.
 * public_out (host, may be NULL): n x 6 x 32 B, every proof's public inputs in verifier order (root, nullifier_hash,
 * recipient, amount, token, chain_id) -- root and nullifier_hash are COMPUTED by the witness generator, so the caller
 * needs them back to submit the proof.
 * OG_ERR_INVALID: an input record is malformed -- a field >= r (two encodings would map to one nullifier), or an index that
 * does not fit the tree (>= 2^depth, or bytes above the u64 set); og_last_error names the record and its first such field
 * (0 nullifier, 1 secret, 2 amount, 3 recipient, 4 pad_seed, 5 index, 6 token, 7 chain_id, 8 + l sibling l).  The records
 * arrive from HTTP (/root/reference/src/services/api_services/withdraw.rs:15-19); og_withdraw_witness_d checks the same. */
int og_withdraw_prove_batch_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2,
                              const uint8_t* inputs_d, size_t n, const uint8_t* rs, uint8_t* proofs_out,
                              uint8_t* public_out);

/* ---- deposit statement (BASELINE.json north_star: "deposit/withdraw circuits") ---------------------------------------------
 * public: commitment, depositor; private: nullifier, secret; commitment = H(nullifier, secret) (MultiMiMC7), depositor bound by
 * its square (oracle/py/deposit.py is the spec; 735 wires, 731 constraints, domain 2^10).  The reference's deposit credits an
 * account on the word of an L1 transaction hash (/root/reference/src/services/api_services/deposit.rs:32-154 ->
 * /root/reference/src/blockchain/tx/mint_tx.rs:11-49: no commitment, no proof); with notes, the ledger forms the leaf
 * H(c, H(amount, token)) from the depositor's c and the asset it credits -- and this proof lets it refuse a c nobody can open
 * (a typo, a c copied from somebody else's request), i.e. a note that could never be withdrawn.  The ledger's side is og_verify
 * with public inputs (c, depositor = DepositRequest.address, deposit.rs:19-24).
 * inputs_d: n records of 3 x 32 B: nullifier | secret | depositor.  shape[0..2] = n_wires, n_constraints, n_pub.
 * og_deposit_prove_batch_d: records -> proofs (rs: n x 64 B host, proofs_out: n x 256 B host, public_out (host, may be NULL):
 * n x 2 x 32 B = commitment | depositor per proof).  Errors as og_withdraw_prove_batch_d (a field >= r: OG_ERR_INVALID naming
 * the record and its field 0 nullifier, 1 secret, 2 depositor). */
int og_deposit_shape(uint64_t shape[3]);
int og_deposit_witness_d(og_ctx* ctx, const uint8_t* inputs_d, size_t n, uint8_t* witness_out_d);
int og_deposit_prove_batch_d(og_ctx* ctx, const og_pk* pk, const uint8_t* inputs_d, size_t n, const uint8_t* rs,
                             uint8_t* proofs_out, uint8_t* public_out);

/* The same call in two halves, for a host that keeps requests flowing (a sequencer proving batch after batch): submit
 * enqueues ALL the work of the batch on the ctx's streams and returns; og_job_wait blocks until it is done, fills
 * proofs_out / public_out (which, like rs, must stay valid until then) and frees the job.  At most two calls may be in
 * flight on a ctx, and they complete in submission order.  Submitting batch k + 1 before waiting for batch k lets its cold
 * start (first witnesses, sparse products, sorts) run beside batch k's last bucket accumulations instead of after them
 * (measured: +0.7 % proofs/s at batch 1024 -- the chip is already busy, what is gained is the latency of the first
 * witnesses and of the copy-out).  og_job_wait returns what og_withdraw_prove_batch_d would have (OG_ERR_UNSATISFIED ...).
 * The overlap exists for batches the library runs through its stage pipeline (sub-batches of >= 64 proofs: og_prove_plan mode
 * 3).  A smaller call (modes 1 / 2: what a request coalescer produces under light load) shares scratch with its predecessor
 * without per-slot guards, so its submit first waits -- holding the ctx -- until the call before it has finished: submit k + 1
 * then returns when batch k is done, and og_job_poll / other ctx calls from other threads queue behind it meanwhile.  Nothing
 * is lost against blocking calls (such batches take milliseconds); the call-ahead gain is a property of large batches.
 * Submit itself never proves: also a call that fits one sub-batch enqueues its witnesses with the rest and returns (until the
 * end of round 6 a small statement's witnesses were generated before the enqueue, with the host waiting for them).
 * While a thread is inside og_job_wait for a job, og_job_abandon and a second og_job_wait on that job are refused, and
 * og_shutdown waits for the waiter; og_msm_d / og_msm_windows_d / og_msm_combine_d are refused while a submitted call is
 * pending (they use the same scratch). */
typedef struct og_job og_job;
int og_withdraw_prove_batch_submit_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2,
                                     const uint8_t* inputs_d, size_t n, const uint8_t* rs, uint8_t* proofs_out,
                                     uint8_t* public_out, og_job** job_out);
int og_job_wait(og_ctx* ctx, og_job* job);
/* Non-blocking: *done_out = 1 once every kernel of the job has finished (og_job_wait would then return at once), else 0.  The
 * job stays pending either way.  Threading: og_job_wait blocks WITHOUT holding the context's lock, so one host thread may sit
 * in og_job_wait for batch k while another submits batch k + 1 -- the shape of a request coalescer that keeps one call ahead
 * from a blocking-task pool; a single-threaded host polls instead. */
int og_job_poll(og_ctx* ctx, og_job* job, int* done_out);
/* Every job must be consumed exactly once: by og_job_wait, or -- when the caller no longer wants the results, e.g. its
 * output buffers are going away -- by og_job_abandon, which waits for the job's kernels (they write device scratch the next
 * call reuses), copies nothing out and frees the call slot.  A handle that is not a pending job of this ctx (already
 * consumed, another ctx's) is refused with OG_ERR_INVALID by both. */
int og_job_abandon(og_ctx* ctx, og_job* job);

/* Window-sharded proving for a host that runs ONE PROCESS PER GPU and owns the collective (torch.distributed over RCCL:
 * owshen_amd/shard.py; og_multi_*_prove_sharded above is the same composition inside one process):
 *   og_withdraw_prove_partials_d / og_prove_partials_d   the front half on this rank: witnesses (or the caller's, device),
 *       sparse products, quotient, and the windows k = win_rank (mod win_world) of the five queries.  partials_d (device,
 *       n x OG_PARTIAL_BYTES): five arrays, A | B1 | L | H (n x 128 B: extended-Jacobian XYZZ over Fq, Montgomery form, the
 *       library's internal point format -- opaque to the host, only moved) | B2 (n x 256 B).  Blocking; errors and public_out
 *       as og_withdraw_prove_batch_d (every rank sees the same inputs, so every rank reports the same error).
 *   -- the host all-gathers the ranks' blocks: gathered_d = win_world x n x OG_PARTIAL_BYTES, rank-major --
 *   og_prove_from_partials_d   adds the ranks' shares query by query, applies the blinding rs (n x 64 B host) and writes the
 *       n x 256 B proofs (host): byte-identical to og_withdraw_prove_batch_d / og_prove_batch_d on the same inputs.  Any
 *       rank may run it (all hold the gathered block); the others skip it. */
#define OG_PARTIAL_BYTES 768
int og_withdraw_prove_partials_d(og_ctx* ctx, const og_pk* pk, int depth, uint64_t n_pad3, uint64_t n_pad2,
                                 const uint8_t* inputs_d, size_t n, int win_rank, int win_world, uint8_t* partials_d,
                                 uint8_t* public_out);
int og_prove_partials_d(og_ctx* ctx, const og_pk* pk, const uint8_t* witnesses_d, size_t n, int win_rank, int win_world,
                        uint8_t* partials_d);
int og_prove_from_partials_d(og_ctx* ctx, const og_pk* pk, const uint8_t* gathered_d, int win_world, size_t n,
                             const uint8_t* rs, uint8_t* proofs_out);

/* ---- key material: the withdraw circuit and Groth16 key generation (what a Rust host needs to obtain an OWPK0001 /
 * OWVK0001 blob without any Python) ----------------------------------------------------------------------------
 * og_r1cs: host-side R1CS, constraint rows only (the library appends the n_pub + 1 input-consistency rows), CSR per matrix
 * (A = 0, B = 1, C = 2) with canonical 32-byte coefficients.
 *   og_withdraw_r1cs  the statement of og_withdraw_witness_d: depth-`depth` MiMC7 Merkle withdraw circuit + n_pad3 / n_pad2
 *                     synthetic gates; dense != 0 appends the two density rows (every wire gets an A and a B base).
 *                     Wire and row order are specified by oracle/py/withdraw.py.
 *   og_r1cs_from_csr  any other circuit: ptr[k] has n_constraints + 1 entries.
 *   og_r1cs_info      info[0..5] = n_wires, n_pub, n_constraints, nnz_a, nnz_b, nnz_c
 *   og_r1cs_export    copies matrix k out (buffers sized from og_r1cs_info)
 *   og_setup          Groth16 key generation from explicit toxic waste tau | alpha | beta | gamma | delta (5 x 32 B, canonical,
 *                     non-zero; tests and benchmarks -- a production key comes from a ceremony), computed on the GPU.
 *                     *pk_out / *vk_out are malloc'd blobs in the OWPK0001 / OWVK0001 formats; release with og_blob_free. */
typedef struct og_r1cs og_r1cs;
int og_withdraw_r1cs(og_ctx* ctx, int depth, uint64_t n_pad3, uint64_t n_pad2, int dense, og_r1cs** out);
/* the statement of og_deposit_witness_d (wire and row order: oracle/py/deposit.py) */
int og_deposit_r1cs(og_ctx* ctx, og_r1cs** out);
int og_r1cs_from_csr(uint64_t n_wires, uint64_t n_pub, uint64_t n_constraints, const uint32_t* const ptr[3],
                     const uint32_t* const col[3], const uint8_t* const val[3], og_r1cs** out);
void og_r1cs_free(og_r1cs* r1cs);
int og_r1cs_info(const og_r1cs* r1cs, uint64_t info[6]);
int og_r1cs_export(const og_r1cs* r1cs, int matrix, uint32_t* ptr_out, uint32_t* col_out, uint8_t* val_out);
int og_setup(og_ctx* ctx, const og_r1cs* r1cs, const uint8_t toxic[160], uint8_t** pk_out, size_t* pk_len, uint8_t** vk_out,
             size_t* vk_len);
void og_blob_free(uint8_t* blob);

/* ---- snarkjs' files at the boundary (owshen_amd/csrc/zkey.hip; oracle/py/zkey.py restates the formats and is the oracle) ----
 * The lineage BASELINE.json's north_star names proved with circom / snarkjs; its keys are `.zkey` files, its witnesses `.wtns`
 * files (iden3 binfile containers).  No reference interface to replace: the snapshot holds no key; the
 * caller would be the node's start-up beside /root/reference/src/cli/node.rs:26-53.
 *   og_zkey_import  a Groth16 BN254 .zkey -> malloc'd "OWPK0001" / "OWVK0001" blobs (og_blob_free) for og_pk_load / og_verify.
 *                   Points leave the file's Montgomery form, the rows move from ffjavascript's root order to this library's, the
 *                   odd-coset Lagrange H section becomes the coefficient-basis H query through a DFT over G1 points on the
 *                   GPU, and the key carries header flag 1: a .zkey has no C matrix, the prover takes C z = (A z) o (B z) as
 *                   snarkjs does -- so a witness that violates a constraint is NOT refused (OG_ERR_UNSATISFIED needs a C
 *                   matrix), its proof simply does not verify.  Refused with OG_ERR_INVALID: another curve or protocol, a
 *                   section whose length disagrees with the header, a coordinate >= q, a point off its curve, a coefficient
 *                   >= r or outside the matrix.  Proofs made with the imported key are the ones snarkjs' prover makes for
 *                   the same (r, s).
 *                   r1cs (optional, may be NULL): the circuit the key was made for, as og_r1cs_read returns it.  Its A and B must
 *                   be the key's coefficient section row for row (else OG_ERR_INVALID: not this key's circuit); its C matrix
 *                   then rides in the imported key (flag 0) and OG_ERR_UNSATISFIED works as for a key of og_setup.
 *   og_r1cs_read    circom's .r1cs (iden3 r1csfile, version 1) -> og_r1cs (og_r1cs_free); n_pub = nPubOut + nPubIn; host only.
 *   og_r1cs_write   the inverse (malloc'd, og_blob_free): what `snarkjs r1cs info` / `snarkjs groth16 setup` read.
 *   og_zkey_export  the way back: an OWPK0001 + OWVK0001 pair as a .zkey that `snarkjs groth16 prove` accepts (C matrix left
 *                   behind, H section by the inverse transform, "no contributions": `snarkjs zkey verify` against a .ptau
 *                   will not pass, proving and verifying do).
 *   og_wtns_read    a .wtns -> n x 32 B canonical values (values_out == NULL: only *n_out); host only, no og_ctx.
 *   og_wtns_write   the inverse (malloc'd, og_blob_free).
 * The formats are written down from the published sources of snarkjs 0.7 / ffjavascript; no file made by snarkjs itself was
 * available to test against. */
int og_zkey_import(og_ctx* ctx, const uint8_t* zkey, size_t zkey_len, const og_r1cs* r1cs, uint8_t** pk_out, size_t* pk_len,
                   uint8_t** vk_out, size_t* vk_len);
int og_zkey_export(og_ctx* ctx, const uint8_t* pk, size_t pk_len, const uint8_t* vk, size_t vk_len, uint8_t** zkey_out,
                   size_t* zkey_len);
int og_r1cs_read(const uint8_t* r1cs_file, size_t len, og_r1cs** out);
int og_r1cs_write(const og_r1cs* r1cs, uint8_t** file_out, size_t* file_len);
int og_wtns_read(const uint8_t* wtns, size_t len, uint8_t* values_out, size_t capacity, uint64_t* n_out);
int og_wtns_write(const uint8_t* values, uint64_t n, uint8_t** wtns_out, size_t* wtns_len);

/* ---- key-generation helpers (trusted setup from explicit toxic waste; tests and bench) --------
 * out[i] = k_i * base.  base: host, canonical affine; scalars_d / out_d: device, canonical. */
int og_scalar_mul_d(og_ctx* ctx, int group, const uint8_t* base, const uint8_t* scalars_d, size_t n,
                    uint8_t* out_d);
/* Lagrange basis of the size-2^log_d NTT domain evaluated at tau: out[k] = L_k(tau), d x 32 B */
int og_lagrange_evals_d(og_ctx* ctx, int log_d, const uint8_t tau[32], uint8_t* out_d);
/* out[row] = sum_k val[k] * x[col[k]] over Fr, CSR with canonical values */
int og_spmv_fr_d(og_ctx* ctx, const uint32_t* row_ptr_d, const uint32_t* col_d, const uint8_t* val_d,
                 size_t n_rows, const uint8_t* x_d, uint8_t* out_d);

/* ---- region timing (HIP events on the ctx stream; for bench.py's roofline figures) ------------
 * og_profile(ctx, 1) clears the log and starts recording; og_profile(ctx, 0) stops.
 * og_profile_read: out[0] = total milliseconds, out[1] = number of regions, out[2] = units processed.
 * kind: 0 bucket accumulation G1 (one kernel launch per region; units = points x proofs), 1 same for G2,
 *       2 H-polynomial pipeline (units = domain elements), 3 digit sort (scalars), 4 / 5 bucket reduction
 *       G1 / G2 (buckets), 6 witness generation (witnesses), 7 R1CS sparse products (non-zeros),
 *       8 proof assembly (proofs), 9 / 10 heavy-bucket accumulation G1 / G2 (buckets above 2048 entries: a workgroup per
 *       bucket segment; a lone 2^26-point MSM is accumulated entirely here; units = points x proofs). */
int og_profile(og_ctx* ctx, int enable);
/* The batched prover pipelines its sub-batches over two HIP streams (memory-bound preparation of sub-batch k + 1 under
 * the VALU-bound arithmetic of k; scratch per sub-batch parity; default 2).  1 = strictly serial on one stream: kernel
 * timings free of co-scheduling. */
int og_set_lanes(og_ctx* ctx, int n_lanes);
int og_profile_read(og_ctx* ctx, int kind, double out[3]);
/* Frees the ctx's scratch arena (it regrows on demand): a long-lived host that proved a large batch hands the tens of GB of
 * sub-batch scratch back before, say, building 2^26-point window tables.  Waits for the ctx's streams first. */
int og_release_scratch(og_ctx* ctx);
/* Bound the HBM the prover reserves for sub-batch scratch on this ctx: the stage pipeline rotates over two scratch slots, each
 * sized for one sub-batch (~230 MB per proof of the 2^18-wire circuit, at most 256 proofs: ~70 GB at batch 1024 by default, or
 * 0.85 x the free memory if that is less); with a budget the sub-batches shrink until two slots fit into `bytes` (never below
 * one proof per sub-batch).  0 restores the default.  Takes effect from the next call; scratch already reserved stays until
 * og_release_scratch.  The environment variable OG_SUB_BATCH (a cap in proofs) remains as the operator's coarse knob. */
int og_set_scratch_budget(og_ctx* ctx, uint64_t bytes);
/* Latency knob for a host that proves a HANDFUL of requests per call (the reference's handler proves one per HTTP call,
 * /root/reference/src/services/api_services/withdraw.rs:27-71): withdraw calls of at most `max_requests` requests (0 .. 64;
 * default 0 = never) walk their MiMC7 chains on the HOST CPU -- one thread per request, the library's own field layer compiled
 * for the host, the wires copied up -- instead of on a lone GPU wave.  A request's walk is ~19 000 dependent modular products; a
 * lone wave takes ~0.42 us for each, a server core 20-50 ns: one request 10.7 -> ~4.5 ms.  Everything else of the call (padding
 * gates, sparse products, quotient, MSMs, assembly) stays on the GPU, larger calls are untouched, the bytes are the same.
 * og_mimc7_append_d follows the same bound for its leaves: an append of at most `max_requests` leaves hashes on the host (one
 * leaf into a depth-32 tree: 8.6 -> 0.32 ms).
 * Not a fallback: the call still fails without a GPU. */
int og_set_host_chains(og_ctx* ctx, int max_requests);
/* HBM accounting: out[0] = bytes of scratch this ctx's arena currently holds (sub-batch slots, call-level buffers, NTT
 * tables), out[1] = number of arena buffers, out[2] / out[3] = free / total bytes of the device (hipMemGetInfo). */
int og_mem_info(og_ctx* ctx, uint64_t out[4]);
/* How a call of n proofs with this key is scheduled on this ctx right now (it depends on the free HBM): sizes_out[0 .. *count_out)
 * = the sub-batches (at most `cap` are written), *mode_out = 0 one stream, serial; 1 a call of <= 16 requests, its five queries fanned out over the streams;
 * 2 whole sub-batches side by side on two streams; 3 the stage pipeline. */
int og_prove_plan(og_ctx* ctx, const og_pk* pk, size_t n, uint32_t* sizes_out, size_t cap, size_t* count_out, int* mode_out);
/* The GLV decomposition the library uses for the proof assembly of latency-bound calls, as a host function (no ctx, no GPU):
 * k (32 B LE, canonical) = k1 + lambda k2 (mod r) with |k1|, |k2| < 2^127, lambda the eigenvalue of the BN254 G1
 * endomorphism (x, y) -> (beta x, y).  out = |k1| (16 B LE, bit 127 = sign) || |k2| (the same).  Exposed so that the
 * constants can be checked from outside (tests/test_glv.py re-derives lambda from the modulus). */
int og_glv_decompose(const uint8_t k[32], uint8_t out[32]);
/* bytes of HBM a loaded proving key occupies (CSR matrices + the five per-window query tables + wire maps) */
int og_pk_bytes(const og_pk* pk, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* OWSHEN_GPU_H */
