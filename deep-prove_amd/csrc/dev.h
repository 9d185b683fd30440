// Device-operations interface between the host orchestrator (transcript, claim routing, proof assembly — the
// reference's O(log n) host work) and the O(n) table work that lives on the MI355X.
// One implementation ships in the product: HipDev (hip_dev.hip, hand-written gfx950 kernels). The interface is
// abstract so that tests/ can plug a CPU test double under the same orchestrator to check host logic without a GPU;
// there is NO CPU implementation inside the product library (dp_ctx_create fails without a HIP device).
//
// K-numbers refer to SURVEY.md §2.3 (the reference's rayon hot loops these ops replace).
#pragma once
#include "gl64.h"
#include "poseidon2.h"
#include <vector>
#include <stdexcept>
#include <string>

namespace dp {

struct DpError : std::runtime_error {
  int code;
  DpError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#ifndef DEEP_PROVE_HIP_H  // same values as the macros of include/deep_prove_hip.h
enum { DP_OK = 0, DP_ERR_ARG = -1, DP_ERR_OOM = -2, DP_ERR_HIP = -3, DP_ERR_SHAPE = -4, DP_ERR_VERIFY = -5, DP_ERR_NODEVICE = -6 };
#endif
#define DP_REQUIRE(cond, code, msg) do { if (!(cond)) throw ::dp::DpError((code), std::string(msg)); } while (0)

// A table of field elements resident in HBM. Base elements are canonical u64, extension elements two u64 (c0,c1).
struct DBuf {
  void* p = nullptr;
  size_t n = 0;
  bool ext = false;
  size_t elem_bytes() const { return ext ? 16 : 8; }
  size_t bytes() const { return n * elem_bytes(); }
  DBuf slice(size_t off, size_t cnt) const {
    DBuf r; r.p = (char*)p + off * elem_bytes(); r.n = cnt; r.ext = ext; return r;
  }
  bool null() const { return p == nullptr; }
};

constexpr int SC_MAXK = 5;  // largest product in the reference's graphs: the maxpool zero-check (4 differences x eq, pooling.rs:405-406)
struct ScTerm { int k; int t[SC_MAXK]; };  // product of k (<= SC_MAXK) tables, indices into the table list

// Merkle tree over `nleaves` field elements (merkle_tree.rs:261-329): layer 0 packs leaf pairs (no hashing),
// upper layers = Poseidon2 compress. All layers live in one buffer of (nleaves-1) digests; layer l starts at digest
// offset nleaves - (nleaves >> l).
struct DevTree {
  DBuf leaves;
  DBuf nodes;  // base buffer, 4*(nleaves-1) words
  size_t nleaves = 0;
  Digest root;
  unsigned height() const { return dp_ceil_log2(nleaves); }
};
// BasefoldCommitmentWithWitness (structure.rs:63-72)
struct DevCommit {
  unsigned nv = 0;
  bool is_base = true;
  DBuf evals;     // natural order (the polynomial itself)
  DBuf bh_evals;  // bit-reversed evaluations (== evals when trivial)
  DevTree tree;   // leaves = bit-reversed RS codeword (raw evaluations when trivial)
  bool trivial() const { return nv <= 7; }
  size_t codeword_size() const { return tree.nleaves; }
};
// fractional-sum tree of one logup instance (zkml/src/lookup/logup_gkr/circuit.rs): layer j has length n >> j;
// num[0] is null (lookup: all numerators are -1) or the multiplicity column (table)
struct LogupCircuitDev { std::vector<DBuf> num, den; };
struct QueryDesc {  // one (query, tree) pair of the Basefold query phase (K14)
  const DevTree* tree;
  size_t p0;  // even index of the opened leaf pair
};

class Dev {
 public:
  virtual ~Dev() {}
  virtual const char* name() const = 0;
  virtual void bind_thread() {}  // make this context current for the calling host thread
  virtual void pin_thread() {}   // a thread the library spawned for this context keeps to the CPUs of the device's NUMA node (HipDev::pin_thread)
  // ---- memory. alloc() is an arena (stack discipline via mark/release); persistent allocations outlive proofs.
  virtual DBuf alloc(size_t n, bool ext) = 0;
  virtual size_t mark() = 0;
  virtual void release(size_t m) = 0;
  virtual DBuf alloc_persistent(size_t n, bool ext) = 0;
  virtual void free_persistent(DBuf& b) = 0;
  virtual void upload(const DBuf& dst, const u64* src) = 0;
  virtual void upload_i64(const DBuf& dst, const int64_t* src) = 0;  // Fieldizer on device
  virtual void download(const DBuf& src, u64* dst) = 0;
  virtual void copy(const DBuf& dst, const DBuf& src) = 0;
  virtual void zero(const DBuf& dst) = 0;
  virtual void sync() = 0;
  // uploads this context has queued but not yet run (throughput mode: copies of up to 2 MB return before they ran) are complete when this returns: called before
  // ANOTHER context's stream reads the tables (the asynchronous seam engine, capi.cpp)
  virtual void flush_uploads() { sync(); }
  // a call on this context failed half-way (an exception crossed it): drop per-call state (an open sumcheck session) so that the next call starts clean
  virtual void abort_call() {}
  // ---- MLE primitives
  // K4: out[idx] (+)= scale * prod_t (idx_t ? pt[t] : 1 - pt[t])
  virtual void eq_table(const DBuf& out, const Ext* pt, unsigned k, Ext scale, bool accumulate) = 0;
  // many plain eq tables (scale 1, no accumulation) in one submission — batch_open builds one per opened polynomial
  struct EqJob { DBuf out; const Ext* pt; unsigned k; };
  virtual void eq_table_many(const EqJob* jobs, size_t n) {
    for (size_t i = 0; i < n; i++) eq_table(jobs[i].out, jobs[i].pt, jobs[i].k, ex_one(), false);
  }
  // out[i] = eq(i mod 2^k, pt): the eq table repeated out.n / 2^k times (the `beta_acc` of convolution.rs:859)
  virtual void eq_table_tiled(const DBuf& out, const Ext* pt, unsigned k) {
    size_t n = size_t(1) << k;
    DP_REQUIRE(out.ext && out.n % n == 0, DP_ERR_SHAPE, "eq_table_tiled: output shape");
    eq_table(out.slice(0, n), pt, k, ex_one(), false);
    for (size_t o = n; o < out.n; o += n) copy(out.slice(o, n), out.slice(0, n));
  }
  // eq table that is only ever read by the sumcheck started next (the per-layer eq of logup-GKR): a device may build it
  // inside that sumcheck's kernel instead of spending a launch on it
  virtual void eq_table_lazy(const DBuf& out, const Ext* pt, unsigned k) { eq_table(out, pt, k, ex_one(), false); }
  // K1 chain collapsed to one pass: out[i] = sum_x fs[i](x) * eq(x, pt)
  virtual void mle_eval_batch(const DBuf* fs, int nf, const Ext* pt, unsigned k, Ext* out) = 0;
  // K2 in one pass: out[c] = sum_r eq(r, pt) * W[r*C + c]      (W base field, R = 2^k rows)
  virtual void fix_high(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) = 0;
  // out[r] = sum_c eq(pt, c) * W[r][c] for a row-major base table W[R][C]: the LOW log2(C) variables of the MLE fixed at pt in one pass
  // (fix_variables_in_place by a multi-variable point, what MatMul does with its right matrix: matrix_mul.rs:823). Default: one MLE
  // evaluation per row; devices override it with one pass over the table.
  virtual void fix_low(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) {
    DP_REQUIRE(!W.ext && W.n == R * C && out.ext && out.n == R && C >= 2 && (C & (C - 1)) == 0, DP_ERR_SHAPE, "fix_low: shapes");
    std::vector<DBuf> rows(R);
    for (size_t r = 0; r < R; r++) rows[r] = W.slice(r * C, C);
    std::vector<Ext> v(R);
    const size_t step = 256;
    for (size_t r = 0; r < R; r += step) mle_eval_batch(rows.data() + r, (int)std::min(step, R - r), pt, dp_ceil_log2(C), v.data() + r);
    upload(out, (const u64*)v.data());
  }
  // ---- sumcheck (K1 + K3): fold every table with r (if given; tabs[i] is replaced), then per term the sums
  // sum_b prod_j (v_j[2b] + t (v_j[2b+1] - v_j[2b])) for t = 0..k, written consecutively to `out`.
  virtual void sc_round(DBuf* tabs, int ntabs, const Ext* r, const ScTerm* terms, int nterms, Ext* out) = 0;
  // The same round when the caller knows what the round must sum to (only meaningful for ONE term: out[0] + out[1] ==
  // *claim, the previous round polynomial at its challenge, in raw units): a device may then skip the t = 1 pass and
  // return out[1] = claim - out[0] — the identity is exact in the field, so the message is bit-identical.
  virtual void sc_round_claim(DBuf* tabs, int ntabs, const Ext* r, const ScTerm* terms, int nterms, const Ext* claim, Ext* out) {
    (void)claim;
    sc_round(tabs, ntabs, r, terms, nterms, out);
  }
  virtual void sc_finish(DBuf* tabs, int ntabs, Ext r, Ext* finals) = 0;
  // A sharded sumcheck (csrc/sharded.h) whose shares stay on the device: while an exchange is set, sc_round leaves THIS rank's raw
  // round sums in device memory, has them all-gathered there (ShareExchange: ncclAllGather over xGMI on the context's own stream),
  // adds the ranks' shares mod p in a one-workgroup kernel and publishes the TOTAL to the host — `out` then holds the sums over
  // all ranks and the round has cost one host wait. `false`: this device keeps the host exchange (the CPU test double).
  struct ShareExchange {
    virtual ~ShareExchange() {}
    virtual int world() const = 0;
    // every rank contributes `nwords` u64 words at dsend; drecv receives world * nwords words in rank order; ordered on `stream`
    virtual void all_gather_device(const u64* dsend, size_t nwords, u64* drecv, void* stream) = 0;
  };
  virtual bool sc_set_share_exchange(ShareExchange* x) { (void)x; return false; }
  // All remaining rounds of a sumcheck WITH its Fiat-Shamir transcript on the device, for devices that can keep the sponge
  // to themselves: called at the top of a round with the tables as sc_round would get them (r = the previous challenge,
  // not yet folded in, or null in the first round), the coefficient of every term and the transcript's sponge. On `true`
  // the round messages (max_degree + 1 values each) and challenges of every remaining round have been appended to
  // `msgs` / `point`, `finals` holds the final evaluation of every table and `ch` is the sponge after the last challenge —
  // exactly what the per-round path (sc_round / host transcript / sc_finish) produces. `false`: not taken, nothing changed.
  virtual bool sc_tail(DBuf* tabs, int ntabs, const Ext* r, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned max_degree, Challenger& ch,
                       std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& point, Ext* finals) {
    (void)tabs; (void)ntabs; (void)r; (void)terms; (void)coeffs; (void)nterms; (void)max_degree; (void)ch; (void)msgs; (void)point; (void)finals;
    return false;
  }
  // eq tables followed by a whole sumcheck over tables that include them (the accumulation sumchecks of Requant and of
  // same_poly, zkml.h) with the transcript on the device: job j does out[idx] (+)= scale * eq(idx, pt) exactly like eq_table,
  // in order; then the sumcheck of sum_i coeff_i * prod tables[terms[i]] over `nv` variables — header, rounds, challenges —
  // as sumcheck_prove runs it. On `true`: round messages (max_degree + 1 evaluations each), challenges, the final evaluation
  // of every table, `ch` the sponge after the last challenge. `false`: not taken, nothing changed.
  struct EqAccJob { DBuf out; const Ext* pt; unsigned k; Ext scale; bool accumulate; };
  struct EqSumOut { std::vector<std::vector<Ext>> msgs; std::vector<Ext> point, finals; };
  virtual bool eqsum_tail(const EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms,
                          unsigned nv, unsigned max_degree, Challenger& ch, EqSumOut& out) {
    (void)jobs; (void)njobs; (void)tabs; (void)ntabs; (void)terms; (void)coeffs; (void)nterms; (void)nv; (void)max_degree; (void)ch; (void)out;
    return false;
  }
  // Every delegation sumcheck of one batch FFT / iFFT of the convolution protocol (zkml.h delegate_matrix_evaluation, iop/prover.rs:164-211)
  // in one go, with the transcript on the device: f_middle[l] (2^(l+1) entries) are the intermediate tables of phi_g_init, r1 (n1 =
  // |f_middle| + 1 coordinates) the FFT point, r2 (n1 coordinates) the point the batch sumcheck ended at, omegas the 2^n1 powers of the
  // root of unity. On `true`: for l = |f_middle| - 1 down to 0 the round messages (4 evaluations), the point and the three final
  // evaluations (beta, phi, f_middle[l]) of that sumcheck; `ch` is the sponge after the last one. `false`: not taken, nothing changed.
  struct DelegTailArgs { const std::vector<std::vector<Ext>>* f_middle; const Ext* r1; unsigned n1; const Ext* r2; const u64* omegas; size_t nomegas; bool is_fft; };
  struct DelegTailOut { std::vector<std::vector<std::vector<Ext>>> msgs; std::vector<std::vector<Ext>> points, finals; };
  virtual bool deleg_tail(const DelegTailArgs& a, Challenger& ch, DelegTailOut& out) { (void)a; (void)ch; (void)out; return false; }
  // The device part of Dense::prove_step (zkml.h prove_dense) in one go, with the transcript on the device: the bias at the
  // output point, W(point, .) (fix_high), and the whole sumcheck of sum_c W(point, c) * in(c) — header, rounds, challenges.
  // On `true`: bias_eval, the round messages (3 evaluations each: degree 2) and challenges of the log2(C) rounds, the final
  // evaluations [W(point, point'), in(point')], and `ch` the sponge after the last challenge. `false`: not taken.
  struct DenseTailOut { Ext bias_eval; std::vector<std::vector<Ext>> msgs; std::vector<Ext> point; Ext finals[2]; };
  virtual bool dense_tail(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, Challenger& ch, DenseTailOut& out) {
    (void)bias; (void)W; (void)R; (void)C; (void)in; (void)pt; (void)ch; (void)out;
    return false;
  }
  // ---- logup-GKR (K13)
  virtual void logup_den(const DBuf& out, const DBuf* cols, int ncols, Ext c, Ext chi) = 0;
  virtual void logup_layer(const DBuf& num_in, const DBuf& den_in, const DBuf& num_out, const DBuf& den_out) = 0;
  // all layers of `ninst` instances (columns [i*cpi, (i+1)*cpi) each) and their 4 output values [n0,n1,d0,d1];
  // devices fuse this into one launch for small tables, the default composes logup_den / logup_layer
  virtual void logup_build(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi,
                           std::vector<LogupCircuitDev>& circuits, std::vector<Ext>& outputs) {
    size_t n = cols[0].n;
    circuits.clear(); outputs.clear();
    for (int s = 0; s < ninst; s++) {
      LogupCircuitDev cd;
      DBuf den0 = alloc(n, true);
      logup_den(den0, cols + (size_t)s * cpi, cpi, c, chi);
      cd.den.push_back(den0);
      cd.num.push_back(mult);
      for (size_t len = n; len > 2; len >>= 1) {
        DBuf nn = alloc(len / 2, true), dn = alloc(len / 2, true);
        logup_layer(cd.num.back(), cd.den.back(), nn, dn);
        cd.num.push_back(nn); cd.den.push_back(dn);
      }
      u64 w[8];
      download(cd.num.back(), w); download(cd.den.back(), w + 4);
      for (int k = 0; k < 4; k++) outputs.push_back(ex(w[2 * k], w[2 * k + 1]));
      circuits.push_back(cd);
    }
  }
  // All layers of a logup-GKR batch proof WITH the transcript on the device (the step after sc_tail: one device wait per
  // lookup argument instead of one per tree layer): from the sponge `ch`, the initial (batching, alpha, lambda) and claim —
  // the state of logup_batch_prove (logup.h) right before its layer loop — run every layer: absorb the claim, the batched
  // sumcheck over eq(point, .) and the layer's numerators / denominators, the three challenges, the next claim.
  // On `true`: layer_msgs[l] / layer_points[l] hold the round messages and challenges of layer l's sumcheck, round_evals[l]
  // its final evaluations without the eq table, `point` the final point (last sumcheck point + last batching challenge) and
  // `ch` the sponge after the last challenge. `false`: not taken, nothing changed. No shipped device implements it yet
  // (the contract is pinned by the CPU double of tests/, which runs logup_layers of logup.h on a private transcript).
  struct LogupTailArgs { const std::vector<LogupCircuitDev>* circuits; bool initial_lookup, is_table; unsigned total_layers; Ext batching, alpha, lambda, claim; };
  virtual bool logup_tail(const LogupTailArgs& a, Challenger& ch, std::vector<std::vector<std::vector<Ext>>>& layer_msgs,
                          std::vector<std::vector<Ext>>& layer_points, std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) {
    (void)a; (void)ch; (void)layer_msgs; (void)layer_points; (void)round_evals; (void)point;
    return false;
  }
  // A whole logup-GKR batch proof with the transcript on the device: the fractional-sum trees of `ninst` instances (columns
  // [i*cpi, (i+1)*cpi) each; `mult` non-null: ONE table instance with these multiplicities), the circuit outputs absorbed,
  // the initial challenges, every layer (as logup_tail) and the evaluations of [mult,] columns at the final point — what
  // logup_batch_prove (logup.h) does between its shape checks and the assembly of the proof, from the sponge `ch` and back.
  // `false`: not taken, nothing changed.
  struct LogupFullOut {
    std::vector<Ext> outputs;                                   // [n0, n1, d0, d1] per instance
    std::vector<std::vector<std::vector<Ext>>> layer_msgs;      // per layer: the round messages of its sumcheck
    std::vector<std::vector<Ext>> layer_points, round_evals;    // per layer: its challenges / final evaluations without eq
    std::vector<Ext> point, col_evals;                          // final point; [mult,] columns evaluated there
  };
  virtual bool logup_full(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, Challenger& ch, LogupFullOut& out) {
    (void)cols; (void)cpi; (void)ninst; (void)mult; (void)c; (void)chi; (void)ch; (void)out;
    return false;
  }
  // ---- Basefold (K5-K12, K14)
  virtual void pcs_init(unsigned full_message_size_log) = 0;
  virtual DevCommit commit(const DBuf& evals, bool persistent) = 0;
  // many commitments at once (same results as commit() on each); devices batch equal-size polynomials
  virtual std::vector<DevCommit> commit_many(const std::vector<DBuf>& evals, bool persistent) {
    std::vector<DevCommit> out;
    for (const DBuf& e : evals) out.push_back(commit(e, persistent));
    return out;
  }
  virtual void free_commit(DevCommit& c) = 0;
  virtual DevTree merkle_ext(const DBuf& leaves) = 0;
  // MerkleTree::from_batch_leaves over k >= 2 codewords of one size and one field (merkle_tree.rs:68-74): the returned tree's `leaves`
  // are the row hashes (two extension entries per row: hash_or_noop of [cws[0][j], .., cws[k-1][j]]), its nodes from the first hashed
  // layer up are the batch tree's (pair i = hash_two_digests(hash(row 2i), hash(row 2i+1))), its root the commitment.
  virtual DevTree batch_tree(const DBuf* cws, int k, bool persistent) { (void)cws; (void)k; (void)persistent; throw DpError(DP_ERR_SHAPE, "batch_tree: not provided by this device"); }
  // classic sumcheck round (K12): fold every (f_i, eq_i) of length > 1 with r (if given), then
  // out[2i] = sum_j f[2j]*eq[2j], out[2i+1] = sum_j (f[2j+1]-f[2j])*(eq[2j+1]-eq[2j]); length-1 pairs give (f*eq, 0)
  // `los` (null: none) holds the FACTORED eq tables: los[i].n == 0 means eqs[i] is the table itself (as long as fs[i]); otherwise
  // table_i[j] = los[i][j mod L] * eqs[i][j / L] with L = los[i].n and fs[i].n = L * eqs[i].n — eq(x, z) is the outer product of the eq
  // tables of the low and the high coordinates of z (the products are exact: the same field elements as the materialised table),
  // so a 2^nv-entry table is never written or read. A fold halves los[i] while it is longer than one entry, then eqs[i].
  virtual void classic_round(DBuf* fs, DBuf* eqs, DBuf* los, int np, const Ext* r, Ext* out) = 0;
  // How pcs_batch_open represents the eq table of an opened nv-variable polynomial: the number of LOW variables of the factored form
  // (0: materialise the table), and the length from which on every table is materialised again (the one-workgroup tails and the
  // last rounds work on plain tables).
  virtual unsigned classic_eq_split(unsigned nv) { (void)nv; return 0; }
  virtual size_t classic_eq_materialise_n() { return 8192; }
  struct EqOuterJob { DBuf out, lo, hi; };  // out[j] = lo[j mod lo.n] * hi[j / lo.n]
  virtual void eq_outer_many(const EqOuterJob* jobs, size_t n) { (void)jobs; (void)n; throw DpError(DP_ERR_SHAPE, "eq_outer_many: not provided by this device"); }
  // The remaining rounds of the batch-opening ("classic") sumcheck of pcs_batch_open (pcs.h) with the transcript on the device.
  // Called at the top of round `round`, exactly where classic_round would be called (r = the previous challenge, not yet
  // folded in, or null in round 0), with eq_xt[i] the batching coefficient of pair i and `sum` the running claim. On `true`
  // the 3-coefficient message and the challenge of every remaining round have been appended to `msgs` / `challenges` and `ch`
  // is the sponge after the last challenge; fs / eqs are left in an unspecified state (the caller only needs the
  // challenges from here on). `false`: not taken, nothing changed.
  struct ClassicTailArgs { DBuf* fs; DBuf* eqs; DBuf* los; int np; const Ext* r; const Ext* eq_xt; unsigned num_vars, round; Ext sum; };
  virtual bool classic_tail(const ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) {
    (void)a; (void)ch; (void)msgs; (void)challenges;
    return false;
  }
  // K11: acc[j*rep + q] += x[j] * coeff  for q < rep
  virtual void axpy_rep(const DBuf& acc, const DBuf& x, Ext coeff, size_t rep) = 0;
  // K11 batched: acc = (init ? *init : 0) + sum_d rep_d(x_d) * coeff_d in one pass over acc (field addition is exact,
  // so the order of accumulation does not matter)
  struct AxpyJob { DBuf x; Ext coeff; size_t rep; };
  virtual void axpy_many(const DBuf& acc, const DBuf* init, const AxpyJob* jobs, size_t n) {
    if (init) copy(acc, *init); else zero(acc);
    for (size_t i = 0; i < n; i++) axpy_rep(acc, jobs[i].x, jobs[i].coeff, jobs[i].rep);
  }
  // The remaining rounds of the Basefold commit phase (pcs.h commit_rounds: batch_commit_phase, commit_phase.rs:187-359) with
  // the transcript on the device, from the top of a round i >= 1: absorb `last`, draw the folding challenge, merge the committed
  // codewords of the running oracle's size into `folded` (merges[j] for tail round j; never in place: `folded` is the leaf
  // array of a tree), FRI-fold, fold the sumcheck pairs (eq, sum_evals), and — except in the final round — the next message,
  // the Merkle tree of the folded oracle and its root absorbed; in the final round the final message (sum_evals, bit-reversed
  // index order) absorbed. On `true`: one message and one tree (leaves, nodes, root; allocated so that they outlive the call)
  // per non-final round, the final message, `ch` the sponge at the end. `false`: not taken, nothing changed.
  struct CommitTailArgs { const Ext* last; DBuf folded, eq, sum_evals; unsigned rounds_left; const std::vector<std::vector<AxpyJob>>* merges; };
  struct CommitTailOut { std::vector<std::vector<Ext>> msgs; std::vector<DevTree> trees; std::vector<Ext> final_message; };
  virtual bool commit_tail(const CommitTailArgs& a, Challenger& ch, CommitTailOut& out) {
    (void)a; (void)ch; (void)out;
    return false;
  }
  // K10: (optionally) fold the evaluation-form pair arrays with ch, then (optionally) the coefficient-form message
  virtual void bf_round(DBuf& eq, DBuf& f, const Ext* ch, Ext* msg3) = 0;
  // K9: FRI fold of a bit-reversed codeword of length 2^(level+1)
  virtual DBuf fri_fold(const DBuf& oracle, unsigned level, Ext ch) = 0;
  virtual void bitrev_copy(const DBuf& dst, const DBuf& src) = 0;
  // verifier side (K8 in reverse): authenticate `n` Merkle paths, each from its leaf-pair digest up to its root
  // (mpcs/src/util/merkle_tree.rs:331-420): true iff all authenticate; *first_bad = index of one that does not.
  // leaf / root: 4 words per job; x: index of the leaf pair; path_off / depth: digests [path_off, path_off + depth) of `pool`
  virtual bool merkle_paths_check(const u64* leaf, const u64* root, const u64* x, const u64* path_off, const u64* depth, size_t n, const u64* pool, size_t pool_digests, size_t* first_bad) {
    for (size_t j = 0; j < n; j++) {
      Digest h; for (int k = 0; k < 4; k++) h.v[k] = leaf[4 * j + k];
      size_t xi = (size_t)x[j];
      DP_REQUIRE(path_off[j] + depth[j] <= pool_digests, DP_ERR_ARG, "merkle_paths_check: path outside the pool");
      for (size_t l = 0; l < depth[j]; l++) { Digest sib; for (int k = 0; k < 4; k++) sib.v[k] = pool[4 * (path_off[j] + l) + k]; h = (xi & 1) ? host_compress(sib, h) : host_compress(h, sib); xi >>= 1; }
      for (int k = 0; k < 4; k++) if (h.v[k] != root[4 * j + k]) { if (first_bad) *first_bad = j; return false; }
    }
    return true;
  }
  // K14: for each descriptor: the leaf pair (as stored) followed by the Merkle path (height-1 digests)
  virtual void query_gather(const QueryDesc* d, size_t nd, std::vector<std::vector<u64>>& out) = 0;
  // the same words scattered into a buffer the caller lays out: descriptor i's pair words (4 for an extension tree, 2 for a base one) at dst[pair_off[i]], its
  // path words at dst[path_off[i]]; `total` = the length of dst. Words of dst outside those ranges are UNSPECIFIED afterwards (the caller writes its headers
  // there after the call): the device gathers into an image of dst and the image comes back in one copy
  virtual void query_gather_into(const QueryDesc* d, size_t nd, const size_t* pair_off, const size_t* path_off, u64* dst, size_t total) {
    std::vector<std::vector<u64>> out;
    query_gather(d, nd, out);
    for (size_t i = 0; i < nd; i++) {
      const size_t np = d[i].tree->leaves.ext ? 4 : 2;
      DP_REQUIRE(out[i].size() >= np && pair_off[i] + np <= total && path_off[i] + (out[i].size() - np) <= total, DP_ERR_SHAPE, "query_gather_into: layout outside the buffer");
      std::copy(out[i].begin(), out[i].begin() + np, dst + pair_off[i]);
      std::copy(out[i].begin() + np, out[i].end(), dst + path_off[i]);
    }
  }
  // The whole query section of a batch opening as its stream words (proof.h Writer::basefold; batch_prover_query_phase, query_phase.rs:67-102, 419-472): the count,
  // then per query [index x] [#oracle] entries [#commitments] entries, an entry = [is_ext] [pair: 4 or 2 words] [index of the pair's left element] [path length]
  // [path digests]. Every query opens the same trees — the oracles of the commit phase first, the committed codewords after them — at the pair
  // p0 = (x >> shift) & ~1 (oracle k: shift = 1 + k; a codeword of height h below a running oracle of cw_log: shift = cw_log - h), so the section is a regular
  // image: a query's block has a fixed length and tree k sits at a fixed offset in it. A device that can (HipDev) writes the image itself from the 200 query
  // indices and the list of trees — headers included — instead of reading 11 200 expanded descriptors (56 B each) from the host; this default builds the
  // descriptors and the headers on the host (the CPU double, devices without the kernel).
  struct QueryTree { const DevTree* tree; unsigned shift; };
  // offsets of the entries in a query's block (words from the block's first word), the position of the [#commitments] word, the block's length
  static size_t query_section_layout(const QueryTree* trees, size_t noracle, size_t ncomm, std::vector<size_t>& rel, size_t& ncomm_pos) {
    rel.assign(noracle + ncomm, 0);
    size_t pos = 2;
    ncomm_pos = 0;
    for (size_t k = 0; k < noracle + ncomm; k++) {
      if (k == noracle) ncomm_pos = pos++;
      rel[k] = pos;
      pos += 1 + (trees[k].tree->leaves.ext ? 4 : 2) + 2 + 4 * (size_t)(trees[k].tree->height() - 1);
    }
    if (ncomm == 0) ncomm_pos = pos++;
    return pos;
  }
  virtual void query_section(const size_t* qidx, size_t nq, const QueryTree* trees, size_t noracle, size_t ncomm, u64* dst, size_t total) {
    const size_t nt = noracle + ncomm;
    std::vector<size_t> rel; size_t cpos;
    const size_t stride = query_section_layout(trees, noracle, ncomm, rel, cpos);
    DP_REQUIRE(total == 1 + nq * stride, DP_ERR_SHAPE, "query_section: layout");
    std::vector<QueryDesc> descs(nq * nt);
    std::vector<size_t> pair_off(nq * nt), path_off(nq * nt);
    for (size_t qi = 0, di = 0; qi < nq; qi++)
      for (size_t k = 0; k < nt; k++, di++) {
        const size_t e = 1 + qi * stride + rel[k];
        descs[di] = {trees[k].tree, (qidx[qi] >> trees[k].shift) & ~size_t(1)};
        pair_off[di] = e + 1; path_off[di] = e + 1 + (trees[k].tree->leaves.ext ? 4 : 2) + 2;
      }
    query_gather_into(descs.data(), descs.size(), pair_off.data(), path_off.data(), dst, total);
    dst[0] = nq;
    for (size_t qi = 0, di = 0; qi < nq; qi++) {
      u64* o = dst + 1 + qi * stride;
      o[0] = qidx[qi]; o[1] = noracle; o[cpos] = ncomm;
      for (size_t k = 0; k < nt; k++, di++) {
        const size_t nw = trees[k].tree->leaves.ext ? 4 : 2;
        u64* e = o + rel[k];
        e[0] = nw == 4 ? 1 : 0; e[1 + nw] = descs[di].p0; e[2 + nw] = (u64)(trees[k].tree->height() - 1);
      }
    }
  }
  // the same words in ONE buffer: descriptor i at flat[off[i] .. off[i + 1]). A Dense-4M batch opening gathers 10 000 (pair, path) records, 5.8 MB: the
  // vector-per-record form cost the proving thread ~20 000 heap allocations and two extra copies per proof (the members of a cohort run it one after the other)
  virtual void query_gather_flat(const QueryDesc* d, size_t nd, std::vector<u64>& flat, std::vector<size_t>& off) {
    std::vector<std::vector<u64>> out;
    query_gather(d, nd, out);
    off.assign(nd + 1, 0);
    for (size_t i = 0; i < nd; i++) off[i + 1] = off[i] + out[i].size();
    flat.resize(off[nd]);
    for (size_t i = 0; i < nd; i++) std::copy(out[i].begin(), out[i].end(), flat.begin() + off[i]);
  }
};

}  // namespace dp
