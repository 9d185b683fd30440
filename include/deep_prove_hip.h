/* deep_prove_hip.h — C ABI of the MI355X-native deep-prove hot path (libdeepprove_hip.so).
 *
 * This is the boundary a Rust shim (or any FFI) binds to replace the reference's CPU rayon path. Every entry point
 * cites the reference interface it stands in for (paths relative to the deep-prove repository). Conventions:
 *   - every function returns int32 status: 0 = ok, negative = DP_ERR_*; nothing throws or aborts across the ABI;
 *     dp_last_error() returns the message of the last failure on the calling thread;
 *   - field elements cross the ABI as canonical little-endian uint64 (< p = 2^64-2^32+1); an extension element is
 *     two consecutive uint64 (c0, c1) of c0 + c1*X, X^2 = 7;
 *   - a dp_ctx owns one HIP device, one stream and a device arena: one ctx per GPU, one host thread per ctx — except the
 *     entry points the reference itself calls from worker threads (table uploads / frees, dp_pcs_commit, dp_pcs_open,
 *     dp_pcs_batch_open: see dp_pcs_commit), which any thread may call at any time and which are queued internally;
 *   - tables live in HBM behind opaque dp_buf handles; host<->device copies are explicit;
 *   - buffers returned through `uint64_t**` are malloc'ed by the library and released with dp_free();
 *   - there is NO CPU fallback: dp_ctx_create fails with DP_ERR_NODEVICE when no MI355X/HIP device is present.
 *     (dp_verify / dp_pcs_batch_verify / dp_transcript_* are host-only by design, as in the reference.)
 */
#ifndef DEEP_PROVE_HIP_H
#define DEEP_PROVE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DP_OK 0
#define DP_ERR_ARG (-1)
#define DP_ERR_OOM (-2)
#define DP_ERR_HIP (-3)
#define DP_ERR_SHAPE (-4)
#define DP_ERR_VERIFY (-5)
#define DP_ERR_NODEVICE (-6)

typedef struct dp_ctx dp_ctx;
typedef struct dp_buf dp_buf;
typedef struct dp_transcript dp_transcript;
typedef struct dp_commit dp_commit;
typedef struct dp_model dp_model;

const char* dp_last_error(void);
void dp_free(void* p); /* ONLY for buffers this library returned (they are recycled, not handed back to malloc: DP_OUT_POOL_BYTES) */

/* ---- device context
 * dp_ctx_create NARROWS the CPU affinity of the calling thread (threads it creates later inherit it) to the CPUs of the GPU's NUMA node
 * (/sys/bus/pci/devices/<bdf>/local_cpulist, intersected with what the thread may already use): every Fiat-Shamir round of a single proof
 * crosses PCIe twice, 34.7 ms per Dense-4M proof from the far socket of a two-socket host against 30.7 ms from the near one. DP_NUMA_PIN=0
 * leaves the affinity alone; so does a thread that is already confined to the node, or an empty intersection. */
int32_t dp_ctx_create(int32_t device_id, dp_ctx** out);
int32_t dp_ctx_destroy(dp_ctx* ctx);
const char* dp_ctx_name(const dp_ctx* ctx);
/* A context starts in LATENCY mode (one proof alone on the GPU: host-side Fiat-Shamir — the host's 1 us permutation beats the device's —,
 * one-workgroup kernels on a whole CU, wide sumcheck rounds spread over several workgroups). on != 0 switches it to THROUGHPUT mode, what
 * dp_model_prove_batch gives its workers: device-side Fiat-Shamir and the fused protocol kernels (a whole logup-GKR proof, the tail of a
 * sumcheck, the last commit rounds ... one launch and one wait each), one-workgroup kernels as 256-thread groups that reserve nothing.
 * For hosts that keep MANY seam-level calls in flight on many contexts (tests/support/seam_bench.c). Results are bit-identical in both
 * modes. Call it while no operation of the context is in flight. */
int32_t dp_ctx_set_throughput_mode(dp_ctx* ctx, int32_t on);

/* ---- asynchronous seam calls: submit / poll forms of dp_pcs_commit, dp_sumcheck_prove, dp_logup_prove and dp_pcs_batch_open.
 * The reference calls these seams from rayon workers — PCS::commit per witness column (zkml/src/commit/context.rs:79-103,
 * zkml/src/layers/activation.rs:294-304), one IOPProverState::prove_parallel per layer (sumcheck/src/prover.rs:498-501), batch_prove per lookup
 * (zkml/src/lookup/logup_gkr/prover.rs:24), one PCS::batch_open per proof (mpcs/src/lib.rs:111-226) — so a host keeps as many calls in flight as it
 * has threads. An engine lets ONE host thread keep hundreds in flight: a submit returns at once with a ticket; engine threads (DP_HOST_THREADS, default
 * the usable CPUs - 2) run the calls as fibers on `max_in_flight` worker contexts of the device of `ctx` (each with an arena of `worker_arena_bytes`,
 * 0 = 512 MB, sharing the PCS tables of `ctx`: call dp_pcs_setup first) in throughput mode, and calls of IDENTICAL SHAPE that are queued together are
 * proved in lock step with their kernel launches merged (what dp_model_prove_batch does for whole proofs). Outputs are bit-identical to the blocking forms.
 *  - arguments are read at submit time, except: the tables / columns / commitments must stay alive, and the dp_transcript must not be touched, until the
 *    ticket has completed (the transcript is advanced by the call, exactly as by the blocking form);
 *  - dp_poll: 0 = still running, 1 = completed, < 0 = failed with that DP_ERR_* code (dp_last_error() has the message); dp_wait blocks;
 *  - results of a completed ticket: dp_ticket_words(which = 0) = the seam's proof stream (malloc'ed, dp_free), dp_ticket_values = the fixed-size
 *    values (dp_sumcheck_prove_submit: the final evaluations, 2 words per table), dp_ticket_commit = the dp_commit and root of a dp_pcs_commit_submit;
 *  - dp_ticket_free after the results have been taken; dp_async_destroy after every ticket has completed (it waits for queued calls). */
typedef struct dp_async dp_async;
typedef struct dp_ticket dp_ticket;
int32_t dp_async_create(dp_ctx* ctx, int32_t max_in_flight, size_t worker_arena_bytes, dp_async** out);
/* Blocking calls that merge (round 6): after this the BLOCKING forms of the seams on `ctx` — dp_pcs_commit, dp_pcs_batch_open (mpcs/src/lib.rs:111-226),
 * dp_sumcheck_prove (sumcheck/src/prover.rs:498-501), dp_logup_prove (zkml/src/lookup/logup_gkr/prover.rs:24), dp_mle_fix_high, dp_mle_eval — are a submit to
 * `engine` plus a wait: calls of identical shape made by OTHER threads (on this or any other context routed to the same engine) within its linger window
 * are proved in lock step with merged launches, and the calling thread sleeps meanwhile. What a host written against the reference's synchronous traits
 * gets without restructuring: the rayon workers of commit/context.rs:79-103 blocked inside PCS::commit are exactly such threads. engine = NULL detaches.
 * The engine must drive the same device as `ctx`; results, errors and transcript effects are those of the un-routed calls. */
int32_t dp_ctx_route_to_engine(dp_ctx* ctx, dp_async* engine);
int32_t dp_async_destroy(dp_async* a);
/* counters since creation: calls executed, groups they ran in, calls that ran merged with at least one other, worker contexts */
int32_t dp_async_stats(dp_async* a, size_t* calls, size_t* groups, size_t* merged_calls, size_t* workers);
int32_t dp_pcs_commit_submit(dp_async* a, const dp_buf* poly, dp_ticket** ticket);
/* PCS::commit(pp, &poly) with the polynomial on the HOST, as the trait passes it (mpcs/src/lib.rs:126-129): upload + commit in one ticket; the device table comes
 * back with dp_ticket_buf (the commitment refers to it: free the commitment first), commitment and root with dp_ticket_commit */
int32_t dp_pcs_commit_host_submit(dp_async* a, const uint64_t* words, size_t n, int32_t is_ext, dp_ticket** ticket);
/* dp_mle_fix_high / dp_mle_eval as tickets: the fixed table comes back with dp_ticket_buf (free it with dp_buf_free on the engine's context), the evaluation
 * with dp_ticket_values (2 words) */
int32_t dp_mle_fix_high_submit(dp_async* a, const dp_buf* matrix, size_t rows, size_t cols, const uint64_t* point, dp_ticket** ticket);
int32_t dp_mle_eval_submit(dp_async* a, const dp_buf* f, const uint64_t* point, uint32_t k, dp_ticket** ticket);
int32_t dp_sumcheck_prove_submit(dp_async* a, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree,
                                 const int32_t* term_tables, const uint64_t* term_coeffs, int32_t nterms, dp_transcript* t, dp_ticket** ticket);
int32_t dp_logup_prove_submit(dp_async* a, const dp_buf* const* columns, int32_t ncols, int32_t cols_per_instance, const dp_buf* multiplicities,
                              const uint64_t constant_challenge[2], const uint64_t column_separation_challenge[2], dp_transcript* t, dp_ticket** ticket);
int32_t dp_pcs_batch_open_submit(dp_async* a, const dp_commit* const* comms, int32_t n, const uint64_t* points_flat, const uint64_t* evals,
                                 dp_transcript* t, dp_ticket** ticket);
int32_t dp_poll(dp_ticket* t);
int32_t dp_wait(dp_ticket* t);
int32_t dp_ticket_words(dp_ticket* t, int32_t which, uint64_t** words, size_t* nwords);
int32_t dp_ticket_values(dp_ticket* t, uint64_t* values, size_t nvalues);
int32_t dp_ticket_commit(dp_ticket* t, dp_commit** out, uint64_t root[4]);
int32_t dp_ticket_buf(dp_ticket* t, dp_buf** out);
int32_t dp_ticket_free(dp_ticket* t);

/* ---- measurement: per-kernel HIP-event timing on the ctx's launch stream (used by bench.py for the roofline object).
 * dp_profile_report returns a malloc'ed JSON array [{"kernel","launches","total_ms","alg_bytes"}...]; free with dp_free. */
int32_t dp_profile_enable(dp_ctx* ctx, int32_t on);
int32_t dp_profile_report(dp_ctx* ctx, char** json);
/* Poseidon2 compress() (poseidon/src/poseidon_hash.rs:65-70, two permutations) per second of the Merkle-layer kernel on a
 * layer of `nodes` nodes, HIP-event timed on the ctx's stream: the VALU-integer peak of the chip for the hash the whole
 * protocol is made of (bench.py prices the job's hashing against it). */
int32_t dp_probe_compress_rate(dp_ctx* ctx, size_t nodes, int32_t reps, double* per_second);

/* ---- tables: DenseMultilinearExtension{evaluations: FieldType::{Base,Ext}} (multilinear_extensions/src/mle.rs:137-181) */
/* Fieldizer::to_field on i64 (zkml/src/quantization/mod.rs:210-220), done on device */
int32_t dp_buf_from_i64(dp_ctx* ctx, const int64_t* v, size_t n, dp_buf** out);
/* n elements; words holds n (base) or 2n (ext) canonical uint64 */
int32_t dp_buf_upload(dp_ctx* ctx, const uint64_t* words, size_t n, int32_t is_ext, dp_buf** out);
int32_t dp_buf_download(dp_ctx* ctx, const dp_buf* buf, uint64_t* out_words);
size_t dp_buf_len(const dp_buf* buf);
int32_t dp_buf_is_ext(const dp_buf* buf);
int32_t dp_buf_free(dp_ctx* ctx, dp_buf* buf);

/* ---- Fiat-Shamir transcript: transcript::BasicTranscript (transcript/src/basic.rs:8-54, lib.rs:42-78). Host side. */
dp_transcript* dp_transcript_new(const char* label); /* BasicTranscript::new(label); NULL -> no label absorbed */
void dp_transcript_free(dp_transcript* t);
int32_t dp_transcript_append_elements(dp_transcript* t, const uint64_t* base_elems, size_t n); /* append_field_elements */
int32_t dp_transcript_append_message(dp_transcript* t, const uint8_t* bytes, size_t n);       /* append_message */
/* label != NULL: get_and_append_challenge(label); label == NULL: read_challenge() */
int32_t dp_transcript_challenge(dp_transcript* t, const char* label, uint64_t out[2]);

/* ---- MLE primitives */
/* build_eq_x_r_vec (multilinear_extensions/src/virtual_poly.rs:414-453) / compute_betas_eval (zkml/src/commit/mod.rs:10-28) */
int32_t dp_eq_table(dp_ctx* ctx, const uint64_t* point, uint32_t k, dp_buf** out);
/* MultilinearExtension::evaluate (mle.rs:607-623) */
int32_t dp_mle_eval(dp_ctx* ctx, const dp_buf* f, const uint64_t* point, uint32_t k, uint64_t out[2]);
/* fix_high_variables_in_place of a rows x cols base matrix at a log2(rows)-coordinate point (mle.rs:562-603 as used by
 * zkml/src/layers/dense.rs:470-475), in one pass from the base-field weights */
int32_t dp_mle_fix_high(dp_ctx* ctx, const dp_buf* matrix, size_t rows, size_t cols, const uint64_t* point, dp_buf** out);

/* ---- sumcheck: IOPProverState::prove_parallel(VirtualPolynomial, transcript) (sumcheck/src/prover.rs:498-585).
 * The virtual polynomial is sum_i coeff_i * prod_{j<degree_i} tables[term_tables[o_i + j]], o_i = degree_0 + .. + degree_(i-1)
 * (term_tables is the concatenation of the terms' table lists), degree_i in 1..5 as the reference dispatches
 * (sumcheck/src/prover.rs:706-713). A table has 2^k entries, 1 <= k <= num_vars (VirtualPolynomial::add_mle_list only asserts
 * num_vars <= max_num_variables, virtual_poly.rs:147-180): a table with fewer variables is constant in the missing (high)
 * ones and gets the 2^(missing) factor of sumcheck_macro/src/lib.rs:236-247; the tables of ONE product share their length, as
 * the generated round function assumes. Each table is passed once (they are de-duplicated by identity, like the Arc
 * pointers of the reference). proof_words receives the IOPProof stream {point: len, ext...; rounds: count, (len, ext...)...};
 * finals receives get_mle_final_evaluations() (2 words per table, table order). */
int32_t dp_sumcheck_prove(dp_ctx* ctx, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables,
                          const int32_t* term_degree, const int32_t* term_tables, const uint64_t* term_coeffs,
                          int32_t nterms, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords,
                          uint64_t* finals);

/* IOPVerifierState::verify(claimed_sum, proof, aux_info{max_degree, num_variables}, transcript) -> SubClaim
 * (sumcheck/src/verifier.rs:12-168). Host only (no dp_ctx: a verifier needs no device). proof_words: the IOPProof stream of
 * dp_sumcheck_prove. On acceptance point receives the num_vars challenges (2 words each) and expected_evaluation the value
 * the virtual polynomial must take there; DP_ERR_VERIFY on rejection (the reference panics / returns Err). */
int32_t dp_sumcheck_verify(uint32_t num_vars, uint32_t max_degree, const uint64_t claimed_sum[2], const uint64_t* proof_words,
                           size_t proof_nwords, dp_transcript* t, uint64_t* point, uint64_t expected_evaluation[2]);

/* ---- round-level sumcheck: what a sharded prover exchanges between devices. The reference's thread-sharded
 * IOPProverState::prove_batch_polys (sumcheck/src/prover.rs:37-321) gives every worker a contiguous 1/2^k chunk of each
 * table, sums the workers' round evaluations and broadcasts one challenge; a session is one worker's side of it: the raw
 * per-term sums of its chunk per round (the caller adds the shares, applies coefficients / extrapolation and runs the
 * transcript), then one folded value per table. Tables and term layout as in dp_sumcheck_prove (degrees 1..5, ragged term
 * lists, tables of 2^k <= 2^num_vars entries). One session at a time
 * per dp_ctx; the session borrows the ctx's arena until dp_sc_session_free. */
typedef struct dp_sc_session dp_sc_session;
int32_t dp_sc_session_new(dp_ctx* ctx, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables,
                          const int32_t* term_degree, const int32_t* term_tables, int32_t nterms, dp_sc_session** out);
/* r_prev == NULL in the first round. raw_out: (degree_i + 1) extension values per term, terms back to back */
int32_t dp_sc_session_round(dp_sc_session* s, const uint64_t* r_prev, uint64_t* raw_out, size_t* nraw_ext);
int32_t dp_sc_session_finish(dp_sc_session* s, const uint64_t r_last[2], uint64_t* finals);
int32_t dp_sc_session_free(dp_sc_session* s);

/* ---- the sharded prover with its round loop IN the library: IOPProverState::prove_batch_polys (sumcheck/src/prover.rs:37-321,
 * merge step sumcheck/src/util.rs:215-243) across GPUs, one rank per GPU. Rank g of W = 2^k holds the contiguous slice
 * [g N/W, (g+1) N/W) of every table (tables: THIS rank's slices, 2^(num_vars - k) entries each; num_vars: of the whole
 * polynomial). Per round: the local round sums (device), ncclAllGather of the shares as u64 device words over xGMI, mod-p sum and
 * Fiat-Shamir on the host, the same challenge on every rank; after num_vars - k rounds one value per table per rank is gathered
 * and the last k rounds run identically everywhere. The proof stream and final evaluations are bit-identical to dp_sumcheck_prove
 * on the whole tables, on every rank. Terms as in dp_sumcheck_prove (full-length local tables only).
 * Communicator: rank 0 calls dp_dist_unique_id and hands the 128 bytes to the other ranks through whatever control plane the
 * host has (MPI, torch.distributed, a socket); every rank then calls dp_dist_init (ncclCommInitRank; librccl is loaded with
 * dlopen on first use). dist == NULL: a world of one. */
typedef struct dp_dist dp_dist;
int32_t dp_dist_unique_id(uint8_t id[128]);
int32_t dp_dist_init(dp_ctx* ctx, const uint8_t id[128], int32_t rank, int32_t world, dp_dist** out);
int32_t dp_dist_free(dp_dist* d);
int32_t dp_sumcheck_prove_sharded(dp_ctx* ctx, dp_dist* dist, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables,
                                  const int32_t* term_degree, const int32_t* term_tables, const uint64_t* term_coeffs,
                                  int32_t nterms, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals);
/* the same loop with `world` contexts driven from ONE process (one thread per rank, exchange in host memory): several
 * contexts on one GPU. tables: world x ntables handles, rank-major; transcripts: one per rank, all in the same state. */
int32_t dp_sumcheck_prove_sharded_local(dp_ctx* const* ctxs, int32_t world, uint32_t num_vars, const dp_buf* const* tables,
                                        int32_t ntables, const int32_t* term_degree, const int32_t* term_tables,
                                        const uint64_t* term_coeffs, int32_t nterms, dp_transcript* const* transcripts,
                                        uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals);

/* ---- logup-GKR: logup_gkr::prover::batch_prove(LogUpInput, transcript) (zkml/src/lookup/logup_gkr/prover.rs:24-198).
 * multiplicities == NULL -> LogUpInput::Lookup with `cols_per_instance`; else LogUpInput::Table. */
int32_t dp_logup_prove(dp_ctx* ctx, const dp_buf* const* columns, int32_t ncols, int32_t cols_per_instance,
                       const dp_buf* multiplicities, const uint64_t constant_challenge[2],
                       const uint64_t column_separation_challenge[2], dp_transcript* t, uint64_t** proof_words,
                       size_t* proof_nwords);
/* logup_gkr::verifier::verify_logup_proof(proof, num_instances, constant_challenge, column_separation_challenge, transcript)
 * (zkml/src/lookup/logup_gkr/verifier.rs:16-211). Host only. On acceptance numerators / denominators receive the fractional
 * sum of every instance (2 words each; the caller checks that all lookups and their table cancel, lookup/context.rs) and
 * claims_words the output claims {count; per claim: point (len, ext...), eval} (malloc'ed, release with dp_free);
 * DP_ERR_VERIFY on rejection. */
int32_t dp_logup_verify(const uint64_t* proof_words, size_t proof_nwords, int32_t num_instances,
                        const uint64_t constant_challenge[2], const uint64_t column_separation_challenge[2], dp_transcript* t,
                        uint64_t* numerators, uint64_t* denominators, uint64_t** claims_words, size_t* claims_nwords);

/* ---- PCS: mpcs::PolynomialCommitmentScheme for Basefold<GoldilocksExt2, BasefoldRSParams<PoseidonHasher>> */
/* PCS::setup + PCS::trim (mpcs/src/basefold.rs:278-303): max_poly_size must be a power of two */
int32_t dp_pcs_setup(dp_ctx* ctx, size_t max_poly_size);
/* PCS::commit (mpcs/src/basefold.rs:304-354); the commitment keeps a reference to `poly` (do not free it first).
 * Thread-safe: the reference commits from rayon workers (zkml/src/commit/context.rs:79-103, layers/activation.rs:294-304), so
 * dp_buf_from_i64 / dp_buf_upload / dp_buf_download / dp_buf_free / dp_pcs_commit / dp_pcs_commit_free / dp_pcs_open /
 * dp_pcs_batch_open may be called on one dp_ctx from several host threads at once; the calls are queued on the context's
 * stream (a lock, not parallel execution). Everything else on a dp_ctx keeps the one-host-thread-per-context rule. */
int32_t dp_pcs_commit(dp_ctx* ctx, const dp_buf* poly, dp_commit** out, uint64_t root[4]);
int32_t dp_pcs_commit_free(dp_ctx* ctx, dp_commit* c);
/* PCS::get_pure_commitment (mpcs/src/basefold.rs:459-461): BasefoldCommitment{root, num_vars, is_base} (structure.rs:161-166) */
int32_t dp_pcs_commitment(const dp_commit* c, uint64_t root[4], uint32_t* num_vars, int32_t* is_base);
/* PCS::open / PCS::verify (mpcs/src/basefold.rs:466-544, 863-962): one committed polynomial at one point.
 *  - at most PCS::trivial_num_vars() = 7 variables (the case zkml opens one by one, zkml/src/commit/context.rs:295,395,477):
 *    the proof is the evaluation table itself (BasefoldProof::trivial, structure.rs:352-363), verification is the Merkle root of
 *    that table plus its evaluation; the transcript is not touched (t may be NULL);
 *  - more variables: commit_phase (basefold/commit_phase.rs:30-185: sumcheck on eq(point, x) f(x) interleaved with the FRI folds
 *    of the committed codeword, one Merkle tree per folded oracle) and prover_query_phase (basefold/query_phase.rs:31-66,
 *    373-417: 200 indices, the codeword pair and one pair per oracle with their Merkle paths) on the device; the proof stream
 *    is the Basefold proof layout of dp_pcs_batch_open with an empty batch sumcheck and one commitment pair per query
 *    (ProofQueriesResultWithMerklePath::Single). `eval` is not needed to open (basefold.rs:471).
 * dp_pcs_verify is host only; max_poly_size = the dp_pcs_setup size of the prover (the code's coset shift depends on it). */
int32_t dp_pcs_open(dp_ctx* ctx, const dp_commit* comm, const uint64_t* point, uint32_t num_vars, const uint64_t eval[2],
                    dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords);
int32_t dp_pcs_verify(size_t max_poly_size, const uint64_t root[4], uint32_t num_vars, int32_t is_base, const uint64_t* point,
                      const uint64_t eval[2], const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t);
/* PCS::batch_open with Evaluation::new(i, i, evals[i]) (mpcs/src/basefold.rs:546-770 as called from
 * zkml/src/commit/context.rs:355-418). points_flat = concatenation of the n points (2*num_vars_i words each). */
int32_t dp_pcs_batch_open(dp_ctx* ctx, const dp_commit* const* comms, int32_t n, const uint64_t* points_flat,
                          const uint64_t* evals, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords);
/* PCS::batch_verify (mpcs/src/basefold.rs:964-1098). Host only. roots: 4 words per commitment. */
int32_t dp_pcs_batch_verify(size_t max_poly_size, const uint64_t* roots, const uint32_t* num_vars,
                            const int32_t* is_base, int32_t n, const uint64_t* points_flat, const uint64_t* evals,
                            const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t);

/* PCS::batch_open / PCS::batch_verify with a general `evals: &[Evaluation<E>]` (mpcs/src/basefold.rs:546-770, 964-1098; Evaluation =
 * {poly, point, value}, mpcs/src/lib.rs:283-304): n_evals evaluations over n_polys commitments and n_points points — any polynomial at
 * any point of its size, several evaluations per polynomial or per point (the reference's own batch_commit_open_verify tests do that,
 * mpcs/src/lib.rs:508-700). points_flat = the points one after the other, point_num_vars[i] extension elements each; eval_poly /
 * eval_point index comms / points; eval_values 2 words each. Same proof layout as dp_pcs_batch_open (one commitment pair per query and
 * COMMITMENT). dp_pcs_batch_open(comms, n, ..) is the special case eval i = (polynomial i, point i). */
int32_t dp_pcs_batch_open_evals(dp_ctx* ctx, const dp_commit* const* comms, int32_t n_polys, const uint64_t* points_flat,
                                const uint32_t* point_num_vars, int32_t n_points, const uint32_t* eval_poly, const uint32_t* eval_point,
                                const uint64_t* eval_values, int32_t n_evals, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords);
int32_t dp_pcs_batch_verify_evals(size_t max_poly_size, const uint64_t* roots, const uint32_t* num_vars, const int32_t* is_base,
                                  int32_t n_polys, const uint64_t* points_flat, const uint32_t* point_num_vars, int32_t n_points,
                                  const uint32_t* eval_poly, const uint32_t* eval_point, const uint64_t* eval_values, int32_t n_evals,
                                  const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t);

/* PCS::batch_commit (mpcs/src/basefold.rs:356-446): n polynomials of ONE size and ONE field behind one Merkle root. Every polynomial is
 * encoded as in dp_pcs_commit; leaf j of the common tree is the row [codeword_0[j], .., codeword_{n-1}[j]] (MerkleTree::from_batch_leaves,
 * util/merkle_tree.rs:68-74, 261-329: pair hash = hash_two_digests(hash(row 2i), hash(row 2i+1)), util/hash.rs:32-41; raw tables when
 * trivial). One polynomial gives the ordinary commitment. 1 <= n <= 32. The handle keeps references to the dp_bufs. Not called by zkml. */
typedef struct dp_batch_commit dp_batch_commit;
int32_t dp_pcs_batch_commit(dp_ctx* ctx, const dp_buf* const* polys, int32_t n, dp_batch_commit** out, uint64_t root[4]);
int32_t dp_pcs_batch_commit_free(dp_ctx* ctx, dp_batch_commit* c);
/* PCS::simple_batch_open / PCS::simple_batch_verify (mpcs/src/basefold.rs:777-861, 1100-1203): all polynomials of a batch commitment at
 * ONE point. log2(next_pow2(n)) "batch coeffs" challenges t, then the commit phase of dp_pcs_open on sum_k eq(t)_k f_k
 * (basefold/commit_phase.rs:363-503) and 200 queries, each opening the ROW pair (one pair per polynomial, one Merkle path) and one pair
 * per folded oracle (basefold/query_phase.rs:104-139, 474-538). Proof stream: the Basefold layout with an empty batch sumcheck and n
 * commitment entries per query, all with the pair's index, the path on entry 0 (..::SimpleBatched). Trivial sizes (<= 7 variables): the
 * proof is the n tables, the transcript is not touched; the verifier checks their common root AND (stricter than the reference, which
 * stops at the root, basefold.rs:1114-1124) each table's shape and evaluation. evals: 2 words per polynomial, commitment order.
 * dp_pcs_simple_batch_verify is host only. */
int32_t dp_pcs_simple_batch_open(dp_ctx* ctx, const dp_batch_commit* comm, const uint64_t* point, uint32_t num_vars, dp_transcript* t,
                                 uint64_t** proof_words, size_t* proof_nwords);
int32_t dp_pcs_simple_batch_verify(size_t max_poly_size, const uint64_t root[4], uint32_t num_vars, int32_t is_base, const uint64_t* point,
                                   const uint64_t* evals, int32_t n, const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t);

/* ---- model level: Context::generate / Prover::prove / verify (zkml/src/iop/context.rs:109, prover.rs:401,
 * verifier.rs:306). model_blob (int64): input_len, nlayers, then per layer kind (0 Dense, 1 Requant, 2 Relu, 3 Conv,
 * 4 MaxPool, 5 Flatten, 6 MatMul, 7 Add, 8 Embeddings, 9 Positional) followed by
 *   Dense: nrows, ncols, weights[nrows*ncols] row-major, bias[nrows]   (padded to powers of two, already quantised; after a
 *          Flatten the columns follow the padded (c,h,w) layout with zeros at padding positions, tensor.rs:1627-1675)
 *   Requant: right_shift, fp_scale, fixed_point_multiplier, intermediate_bit_size   (zkml/src/layers/requant.rs:46-73)
 *   Relu, Flatten: (nothing)
 *   Conv: kw, kx, real_nw, nw, unpadded output shape (c, h, w), filter[kw*kx*real_nw*real_nw], bias[kw] — the layer as
 *          pad_conv / into_padded_and_ffted leave it (zkml/src/padding.rs:218-260, tensor.rs:409-431): every dimension a
 *          power of two, nw = padded input side, stride 1, no padding; proven with the zkCNN FFT protocol
 *          (zkml/src/layers/convolution.rs:697-1080)
 *   MaxPool: padded input shape (c, h, w); kernel 2, stride 2 (zkml/src/layers/pooling.rs:342-520)
 *   MatMul: k, n, flags (1 = bias, 2 = Config::TransposeB), weights[k*n] ([k][n] row major; [n][k] with TransposeB), bias[n] if
 *          flagged — MatMul::new_constant (zkml/src/layers/matrix_mul.rs:176, 701-873): the activation is a row-major
 *          [s][k] matrix (s = its length / k, a power of two >= 2), the output [s][n]; the Linear layer of a transformer block
 *          applied to every row of a sequence.
 *   Add: left multiplier, right multiplier, n, operand[n] — Add::new_with(operand) (zkml/src/layers/add.rs:72-145): out = left * x +
 *          right * operand for a constant operand as long as the activation (how learned positional embeddings enter,
 *          transformer/positional.rs); no sumcheck: the proof is the two evaluations, the operand's goes to its commitment.
 *   Embeddings: vocabulary, embedding size (both padded to powers of two), table[vocabulary * size] row major — only as the FIRST layer
 *          (zkml/src/layers/transformer/embeddings.rs:359-571): the model input is then a vector of token ids, the output the
 *          [tokens][size] matrix of their rows; proved as one-hot(tokens) x table without building the one-hot matrix, the verifier
 *          checks the resulting one-hot claim against the tokens.
 *   Positional: left multiplier, right multiplier, positions, embedding size (both padded), table[positions * size] — Positional::Learned
 *          (zkml/src/layers/transformer/positional.rs:327-583): the first `tokens` rows of the committed table are added to the
 *          [tokens][size] activation; the proof lifts the claim on that slice to the whole table with one transcript coordinate and
 *          one sub-matrix evaluation per doubling.
 * GRAPH form (zkml/src/layers/provable/mod.rs:195-565: nodes with several inputs / outputs, models with several input / output tensors):
 *   input_len (the sum of the input tensors), -(number of nodes) — the NEGATIVE count marks the form —, number of input tensors and their
 *   lengths, number of output tensors and one (node, slot) pair each; then per node: kind, number of inputs (1 .. 3), one (node, slot) pair
 *   per input (node = -1: input tensor `slot` of the model), then the parameters of the kind as above. A node reads only nodes with
 *   smaller ids; every tensor has exactly one reader (what the reference proves, provable/mod.rs:235-270). An input vector is the
 *   concatenation of the input tensors, an output vector that of the output tensors. Kinds that only exist in a graph:
 *   10 MatMul of two inputs: k, n, flags (2 = TransposeB) — [s][k] x [k][n] ([n][k] transposed), no bias (layers/matrix_mul.rs:633-873)
 *   11 Add of two inputs: left multiplier, right multiplier (layers/add.rs:81-145)
 *   12 ConcatMatMul: shape of A (3), shape of B (3), (concat, mat_mul, output) axis of A (3) and of B (3), 0 | 1 followed by the permutation
 *      (3) of the [concat][rows][cols] result — chunk c of the result = chunk c of A times chunk c of B (layers/concat_matmul.rs:467-616)
 *   13 QKV: k, n, W_q | W_k | W_v ([k][n] each), b_q | b_k | b_v ([n] each): one [s][k] input, the three outputs X W + b (slots 0, 1, 2)
 *   14 LayerNorm (zkml/src/layers/transformer/layernorm.rs:74-101,140-257): dim (the padded normalisation dimension, a power of two), N (its
 *      unpadded size, next_pow2(N) == dim), multiplier, the f32 bits of the rescaled epsilon, range_check_bits, log2 of the scalar of the top
 *      range-checked chunk (QuantisedLayerNormData), gamma[dim], beta[dim] (zero on the padding). Input [rows >= 4][dim];
 *      out = gamma (N x - sum x) lut(multiplier (N sum x^2 - (sum x)^2) >> range_check_bits) + beta with the inverse-square-root table of
 *      lookup/context.rs:124-157 (2^15 entries; its output column is committed once per context). Usually followed by a Requant (1) whose
 *      multiplier is a power of two (Requant::new_shift, layernorm.rs:473-513).
 *   15 Softmax (zkml/src/layers/transformer/softmax.rs:66-99,153-233) over the last dimension of a padded [c][n][n] tensor under the causal mask
 *      (entries j > i of row i count as minus infinity): c, n, n, multiplier (brings the input to the scale 2^24), f32 bits of 1 / temperature,
 *      f32 bits of the input scale (both only steer the row shifts the prover computes, softmax.rs:250-320), table_size (the exponential table has
 *      2^table_size entries), bkm (inputs from here on are mapped to zero; table_size == ceil_log2(bkm >> 16)), number of zero chunks (<= 3),
 *      bits per zero chunk (0 or 2..22), allowable error of a row sum (2..2048; QuantisedSoftmaxData / SoftmaxCtx). The output has the scale 2^-12.
 *      The exponential table and the error table commit their output column once per context. Lookup tables of fewer than four entries (a one-bit
 *      zero table) are refused: widen the zero table by a bit.
 *   16 Mha (zkml/src/layers/transformer/mha.rs:133-186,633-724) as ONE node with THREE inputs Q, K, V, each a padded [seq][heads * head_dim] matrix
 *      read as [seq][heads][head_dim]: seq, heads, head_dim (powers of two, seq >= 2, head_dim >= 2, heads * seq >= 4), then the parameters of its
 *      softmax exactly as kind 15 lists them after the shape (multiplier .. allowable error; the softmax works straight on the products Q_h K_h^T,
 *      so its input scale is scale(Q) scale(K), its domain qk.output_domain(), 1 / temperature = sqrt(head_dim)). Output [seq][heads * head_dim]
 *      = softmax(Q_h K_h^T) V_h per head at the scale 2^-12 scale(V). One proof step per node: MhaProof {final_mul, softmax, qk}; the claims it
 *      hands on are those on Q, K, V in this order.
 *   17 Gelu (zkml/src/layers/activation.rs:559-671, Activation::Gelu; also a layer of a chain): multiplier = round(2^12 * input scale), 1 .. 4096
 *      (GELU::quantize). The input (>= 4 entries) times the multiplier must lie in [-2^(7 + ceil_log2(multiplier)), 2^(7 + ceil_log2(multiplier))), the
 *      rows of its table (its output column, round(127 gelu(i / 2^12)) in f32 with the C library's tanhf, is a commitment of the context). Proved as the
 *      reference's ActivationProof; the committed scaled-input column is opened at the lookup's own claim — what verify_activation checks (:495-505); the
 *      reference's prover files that claim divided by the multiplier (:405-430) and only verifies where the column is opened by showing it (<= 2^7 entries).
 *   A node has one, two or (kind 16) three inputs. Embeddings stay the first node of a chain. Not built: Logits (no consistent transcript in the
 *   reference outside cfg(test)). */
int32_t dp_model_setup(dp_ctx* ctx, const int64_t* model_blob, size_t nwords, dp_model** out);
int32_t dp_model_free(dp_model* m);
/* runs inference on the host (Model::run, not part of proving time) then Prover::prove on the device.
 * output receives the model output (capacity *noutput on entry, length on exit). prove_ms (nullable) = prove() wall ms */
int32_t dp_model_prove(dp_model* m, const int64_t* input, size_t ninput, uint64_t** proof_words, size_t* proof_nwords,
                       int64_t* output, size_t* noutput, double* prove_ms);
/* `nproofs` independent proofs (inputs concatenated, `ninput` words each) with up to `concurrency` (<= 1024) proofs in
 * flight on the model's GPU: every in-flight proof has its own HIP stream, arena and host<->device mailbox; the model
 * commitments are shared read-only. A single proof is a chain of ~10^3 sequential Fiat-Shamir round trips that cannot
 * fill an MI355X, so this is how one GPU is saturated (and how BASELINE config 4, a batch of independent proofs, is
 * served). The proofs in flight are grouped into cohorts of DP_COHORT (default: in flight / 22, rounded up) proofs that run in lock step: launch
 * number i of all members of a cohort is ONE kernel launch (blockIdx.z = proof) on the cohort's stream. The proofs are
 * driven by min(#cohorts, DP_HOST_THREADS or dp_host_cpu_budget() - 2) host threads; a thread runs its proofs as
 * cooperative fibers and switches proof at every device wait. `concurrency` is a cap: worker arenas are sized from the
 * footprint of the model's earlier proofs (DP_WORKER_ARENA_BYTES overrides) and the number in flight is cut to what fits
 * in the free HBM; dp_model_in_flight reports the number the last batch ran with.
 * proof_words / proof_nwords: arrays of nproofs entries (each buffer malloc'ed, release with dp_free);
 * outputs: nproofs * noutput_cap words (nullable). */
int32_t dp_model_prove_batch(dp_model* m, const int64_t* inputs, size_t nproofs, size_t ninput, int32_t concurrency,
                             uint64_t** proof_words, size_t* proof_nwords, int64_t* outputs, size_t noutput_cap,
                             size_t* noutput, double* wall_ms);
/* proofs the last dp_model_prove_batch of this model kept in flight (0 before the first batch) */
/* Model::run (zkml/src/model.rs: quantised inference of the padded model) on the HOST, no device involved: the inference dp_model_prove /
 * dp_model_prove_batch run before every proof (int16 weights, 32-bit accumulators where exact). *noutput: capacity in, length out. */
int32_t dp_model_infer_host(const int64_t* model_blob, size_t nwords, const int64_t* input, size_t ninput, int64_t* output, size_t* noutput);
/* The Poseidon2-w8 permutation of the HOST transcript (transcript/src/basic.rs over ff_ext/src/lib.rs:167-236), in place on eight canonical
 * words: the AVX-512 code (csrc/p2_avx512.cpp) when the CPU has AVX-512F/DQ and DP_NO_AVX512 is not set, else scalar; force_scalar != 0
 * always takes the scalar code. *vectorised (may be NULL) tells which one the library's transcripts use. Host only. */
int32_t dp_host_poseidon2(uint64_t state[8], int32_t force_scalar, int32_t* vectorised);
int32_t dp_model_in_flight(const dp_model* m, size_t* in_flight);
/* length (in int64 words) of the model's output tensor: what `output` / `outputs` of the prove calls must hold */
int32_t dp_model_output_len(const dp_model* m, size_t* noutput);
/* CPUs the process may use: the cgroup CPU quota when there is one, else the number of hardware threads */
double dp_host_cpu_budget(void);
/* serialisable verifier-side context (model commitments, shapes, tables) */
int32_t dp_model_verifier_blob(const dp_model* m, uint64_t** words, size_t* nwords);
/* zkml::verify(ctx, proof, io, transcript) — host only, default transcript "m2vec" */
/* zkml::verify for a batch of proofs of one model at prover speed: a verifier's time is Merkle-path hashing (~125 000 Poseidon2
 * compress() for one Dense-4M proof: 0.4 s of a host core, 0.1 ms of the GPU), so the protocol checks (transcript, sumchecks,
 * logup, Basefold fold checks: zkml/src/iop/verifier.rs:72-318, mpcs/src/basefold.rs:964-1098) run on `threads` host threads
 * (0 = CPU budget - 2) and every Merkle path of a proof is authenticated in ONE device launch (mpcs/src/util/merkle_tree.rs:
 * 331-420; ctx == NULL: on the host threads too). inputs / outputs: nproofs x ninput / noutput words. results[i] = DP_OK,
 * DP_ERR_VERIFY (rejected) or DP_ERR_ARG (malformed stream); the return value only reports whether the batch was processed. */
int32_t dp_verify_batch(dp_ctx* ctx, const uint64_t* verifier_blob, size_t blob_nwords, const uint64_t* const* proof_words,
                        const size_t* proof_nwords, const int64_t* inputs, size_t ninput, const int64_t* outputs, size_t noutput,
                        size_t nproofs, int32_t threads, int32_t* results, double* wall_ms);
int32_t dp_verify(const uint64_t* verifier_blob, size_t blob_nwords, const uint64_t* proof_words, size_t proof_nwords,
                  const int64_t* input, size_t ninput, const int64_t* output, size_t noutput);

#ifdef __cplusplus
}
#endif
#endif
