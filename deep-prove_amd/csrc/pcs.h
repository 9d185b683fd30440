// Basefold<GoldilocksExt2, BasefoldRSParams<PoseidonHasher>> over device-resident polynomials:
// commit / trivial open / batch_open (mpcs/src/basefold.rs:304-354,466-483,546-770; basefold/commit_phase.rs:187-359;
// basefold/query_phase.rs:67-102,419-472) and the host-side verifier (basefold.rs:863-1098; query_phase.rs:220-288,
// 1116-1236). RS code rate 1/2, 200 queries, base-case message 2^7 (encoding/rs.rs:194-215).
#pragma once
#include "sumcheck.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <system_error>
#include <thread>

namespace dp {

constexpr unsigned PCS_RATE_LOG = 1;
constexpr unsigned PCS_BASECODE_LOG = 7;
constexpr unsigned PCS_NUM_QUERIES = 200;

inline Commitment pure_commitment(const DevCommit& c) { Commitment r; r.root = c.tree.root; r.num_vars = c.nv; r.is_base = c.is_base; return r; }

// PCS::open for a trivial (<= 7 variables) commitment: the proof is the raw evaluation table (basefold.rs:481-483)
inline BasefoldProof pcs_open_trivial(Dev& dev, const DevCommit& c) {
  DP_REQUIRE(c.trivial(), DP_ERR_SHAPE, "pcs_open: only trivial commitments are opened individually (zkml/src/commit/context.rs:295)");
  BasefoldProof p;
  FieldVec fv; fv.is_ext = c.evals.ext; fv.w.resize(c.evals.n * (fv.is_ext ? 2 : 1));
  dev.download(c.evals, fv.w.data());
  p.trivial_proof.push_back(fv);
  return p;
}

struct OpenClaim { const DevCommit* comm; std::vector<Ext> point; Ext eval; };

// words of Dev::query_gather for one (tree, pair) -> the opened pair with its Merkle path (query_phase.rs:669-700)
inline CodewordQuery query_from_words(const QueryDesc& d, const u64* w, size_t n) {
  CodewordQuery q; q.is_ext = d.tree->leaves.ext; q.index = d.p0;
  size_t o = 0;
  if (q.is_ext) { q.left = ex(w[0], w[1]); q.right = ex(w[2], w[3]); o = 4; } else { q.left = ex(w[0], 0); q.right = ex(w[1], 0); o = 2; }
  const size_t npath = (n - o) / 4;
  q.path.resize(npath);
  if (npath) memcpy((void*)q.path.data(), w + o, npath * sizeof(Digest));  // (a digest is its four words)
  return q;
}
inline CodewordQuery query_from_words(const QueryDesc& d, const std::vector<u64>& flat, const std::vector<size_t>& off, size_t i) { return query_from_words(d, flat.data() + off[i], off[i + 1] - off[i]); }

// Rounds [first, num_rounds) of batch_commit_phase (commit_phase.rs:187-359): per round absorb the pending sumcheck message,
// draw the folding challenge, merge the committed codewords of the running oracle's size, FRI-fold, fold the sumcheck pairs;
// then the next message, the Merkle tree of the folded oracle and its root — or, in the last round, the final message.
// With `allow_tail` a device may take the remaining rounds over at the top of any round >= 1 (Dev::commit_tail).
struct CommitLoopState { std::vector<Ext> last; DBuf running, folded, eq, sum_evals; DevTree pending; };
inline void commit_rounds(Dev& dev, const std::vector<std::vector<Dev::AxpyJob>>& merges, unsigned num_rounds, unsigned first, bool allow_tail, CommitLoopState& st,
                          Transcript& t, std::vector<std::vector<Ext>>& msgs, std::vector<Digest>& roots, std::vector<DevTree>& trees, std::vector<Ext>& final_message) {
  for (unsigned i = first; i < num_rounds; i++) {
    if (allow_tail && i > 0) {
      std::vector<std::vector<Dev::AxpyJob>> rest(merges.begin() + i, merges.end());
      Dev::CommitTailArgs ta{st.last.data(), st.folded, st.eq, st.sum_evals, num_rounds - i, &rest};
      Dev::CommitTailOut to;
      if (dev.commit_tail(ta, t.challenger(), to)) {
        DP_REQUIRE(to.msgs.size() == num_rounds - i - 1 && to.trees.size() == num_rounds - i - 1, DP_ERR_SHAPE, "commit_tail: one message and one tree per non-final round expected");
        trees.push_back(st.pending);
        for (auto& m : to.msgs) msgs.push_back(m);
        for (auto& tr : to.trees) { roots.push_back(tr.root); trees.push_back(tr); }
        final_message = to.final_message;
        return;
      }
    }
    for (const Ext& e : st.last) t.append_ext(e);
    Ext c = t.get_and_append_challenge("commit round");
    if (i > 0) {
      trees.push_back(st.pending);
      if (!merges[i].empty()) {  // a fresh buffer: the committed oracle (tree leaves) must stay untouched
        DBuf b = dev.alloc(st.folded.n, true);
        dev.axpy_many(b, &st.folded, merges[i].data(), merges[i].size());
        st.running = b;
      } else st.running = st.folded;
    }
    st.folded = dev.fri_fold(st.running, dp_ceil_log2(st.running.n) - 1, c);
    if (i + 1 < num_rounds) {
      dev.bf_round(st.eq, st.sum_evals, &c, st.last.data());
      msgs.push_back(st.last);
      st.pending = dev.merkle_ext(st.folded);
      t.append_digest(st.pending.root);
      roots.push_back(st.pending.root);
    } else {
      dev.bf_round(st.eq, st.sum_evals, &c, nullptr);
      std::vector<u64> w(2 * st.sum_evals.n);
      dev.download(st.sum_evals, w.data());
      size_t m = st.sum_evals.n; unsigned lg = dp_ceil_log2(m);
      final_message.resize(m);
      for (size_t j = 0; j < m; j++) { size_t r = dp_reverse_bits(j, lg); final_message[r] = ex(w[2 * j], w[2 * j + 1]); }
      t.append_exts(final_message);
    }
  }
}

// One round message [h0, h1, h2] of the batch-opening sumcheck from the per-pair sums of Dev::classic_round
// (CoefficientsProver::prove_round, sum_check/classic/coeff.rs:198-345): pair i contributes eq_xt[i] * (c0, c2), scaled by the
// number of times its (shorter) hypercube repeats in the round's; h1 follows from the running claim.
inline std::vector<Ext> classic_round_message(const Ext* raw, const DBuf* fs, const Ext* eq_xt, size_t np, unsigned num_vars, unsigned round, Ext sum) {
  size_t size = size_t(1) << (num_vars - round - 1);
  Ext h0 = ex_zero(), h2 = ex_zero();
  for (size_t i = 0; i < np; i++) {
    size_t poly_len = fs[i].n;  // current length after the folds so far
    Ext c0 = raw[2 * i], c2 = raw[2 * i + 1];
    size_t multiple;
    if (poly_len == 1) multiple = size;
    else if (size < poly_len || size == 1) multiple = 1;
    else multiple = size / (poly_len >> 1);
    if (multiple != 1) { Ext m = ex_from_u64(multiple); c0 = ex_mul(c0, m); c2 = ex_mul(c2, m); }
    h0 = ex_add(h0, ex_mul(eq_xt[i], c0));
    h2 = ex_add(h2, ex_mul(eq_xt[i], c2));
  }
  Ext h1 = ex_sub(ex_sub(sum, ex_dbl(h0)), h2);
  return {h0, h1, h2};
}

// PCS::batch_open (basefold.rs:546-770): `evals` = Evaluation{poly, point, value} over the commitments `comms` and the points `points` —
// any polynomial at any point of its size, several evaluations per polynomial or per point. The classic sumcheck runs on one
// (f, eq) pair per EVALUATION (the reference merges the polynomials that share a point first, basefold.rs:607-643: the same round
// sums, field arithmetic being exact); the commit phase and the queries run per COMMITMENT, with the coefficient of a commitment the
// sum over its evaluations (basefold.rs:690-701).
struct EvalClaim { size_t poly, point; Ext eval; };
inline BasefoldProof pcs_batch_open_evals(Dev& dev, unsigned full_log, const std::vector<const DevCommit*>& comms, const std::vector<std::vector<Ext>>& points,
                                          const std::vector<EvalClaim>& evals, Transcript& t) {
  BasefoldProof proof;
  if (comms.empty() && points.empty() && evals.empty()) return proof;  // Proof::trivial(vec![])
  DP_REQUIRE(!comms.empty() && !evals.empty(), DP_ERR_SHAPE, "batch_open: commitments and evaluations expected");
  const bool timing = getenv("DP_TIMING") && atoi(getenv("DP_TIMING"));
  auto tl0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (!timing) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[dp timing]   batch_open: %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tl0).count()); tl0 = t1; };
  size_t mk = dev.mark();
  unsigned num_vars = 0;
  for (const DevCommit* c : comms) {
    DP_REQUIRE(c && !c->trivial(), DP_ERR_SHAPE, "batch_open: trivial commitment");
    if (c->nv > num_vars) num_vars = c->nv;
  }
  for (const EvalClaim& e : evals) {
    DP_REQUIRE(e.poly < comms.size() && e.point < points.size(), DP_ERR_ARG, "batch_open: evaluation refers to a missing polynomial or point");
    DP_REQUIRE(points[e.point].size() == comms[e.poly]->nv, DP_ERR_SHAPE, "batch_open: point length != num_vars");
  }
  DP_REQUIRE(num_vars <= full_log, DP_ERR_SHAPE, "batch_open: polynomial larger than the PCS parameters");
  const size_t np = evals.size(), nc = comms.size();  // np: (f, eq) pairs of the sumcheck; nc: committed codewords
  unsigned bsl = dp_ceil_log2(np);
  std::vector<Ext> tt;
  for (unsigned i = 0; i < bsl; i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
  std::vector<Ext> eq_xt = host_eq_table(tt);
  Ext target = ex_zero();
  for (size_t i = 0; i < np; i++)
    target = ex_add(target, ex_mul(ex_mul(evals[i].eval, ex_from_u64(u64(1) << (num_vars - comms[evals[i].poly]->nv))), eq_xt[i]));
  // ---- classic sumcheck on sum_i eq_xt[i] * eq(x, z_i) * f_i(x)  (sum_check/classic.rs:232-285, coeff.rs:198-345)
  // eq(x, z_i) of a long polynomial is kept as the outer product of two short tables (Dev::classic_round, `los`): it is never
  // materialised at 2^nv entries, only once every table has become short
  std::vector<DBuf> fs(np), eqs(np), los(np);
  size_t n_factored = 0;
  {
    std::vector<Dev::EqJob> jobs;
    for (size_t i = 0; i < np; i++) {
      fs[i] = comms[evals[i].poly]->evals;
      const unsigned nv = comms[evals[i].poly]->nv, m = fs[i].n > dev.classic_eq_materialise_n() ? dev.classic_eq_split(nv) : 0;
      const Ext* pt = points[evals[i].point].data();
      if (m > 0 && m < nv) {
        los[i] = dev.alloc(size_t(1) << m, true); eqs[i] = dev.alloc(size_t(1) << (nv - m), true);
        jobs.push_back({los[i], pt, m}); jobs.push_back({eqs[i], pt + m, nv - m});
        n_factored++;
      } else {
        eqs[i] = dev.alloc(fs[i].n, true);
        jobs.push_back({eqs[i], pt, nv});
      }
    }
    dev.eq_table_many(jobs.data(), jobs.size());
  }
  std::vector<Ext> challenges, raw(2 * np);
  Ext sum = target, ch = ex_zero();
  for (unsigned round = 0; round < num_vars; round++) {
    if (n_factored) {
      size_t maxn = 0; for (size_t i = 0; i < np; i++) maxn = std::max(maxn, fs[i].n);
      if (maxn <= dev.classic_eq_materialise_n()) {
        std::vector<Dev::EqOuterJob> oj;
        for (size_t i = 0; i < np; i++) if (los[i].n) { DBuf full = dev.alloc(fs[i].n, true); oj.push_back({full, los[i], eqs[i]}); eqs[i] = full; los[i] = DBuf(); }
        dev.eq_outer_many(oj.data(), oj.size());
        n_factored = 0;
      }
    }
    // a device that keeps the sponge to itself runs every remaining round in one go (Dev::classic_tail)
    Dev::ClassicTailArgs ta{fs.data(), eqs.data(), los.data(), (int)np, round ? &ch : nullptr, eq_xt.data(), num_vars, round, sum};
    if (dev.classic_tail(ta, t.challenger(), proof.sumcheck_proof, challenges)) {
      DP_REQUIRE(challenges.size() == num_vars && proof.sumcheck_proof.size() == num_vars, DP_ERR_SHAPE, "classic_tail: one message and one challenge per round expected");
      break;
    }
    dev.classic_round(fs.data(), eqs.data(), los.data(), (int)np, round ? &ch : nullptr, raw.data());
    std::vector<Ext> msg = classic_round_message(raw.data(), fs.data(), eq_xt.data(), np, num_vars, round, sum);
    for (const Ext& e : msg) t.append_ext(e);
    ch = t.get_and_append_challenge("sumcheck round");
    challenges.push_back(ch);
    sum = ex_add(msg[0], ex_mul(ch, ex_add(msg[1], ex_mul(ch, msg[2]))));
    proof.sumcheck_proof.push_back(msg);
  }
  lap("classic sumcheck");
  // (the last challenge never needs to be folded in: only the challenges are used below)
  std::vector<Ext> coeffs(nc, ex_zero());
  for (size_t i = 0; i < np; i++) {
    const std::vector<Ext>& pt = points[evals[i].point];
    coeffs[evals[i].poly] = ex_add(coeffs[evals[i].poly], ex_mul(eq_eval(challenges.data(), pt.data(), pt.size()), eq_xt[i]));
  }

  // ---- batch_commit_phase (commit_phase.rs:187-359)
  unsigned num_rounds = num_vars - PCS_BASECODE_LOG;
  size_t cw_size = size_t(1) << (num_vars + PCS_RATE_LOG);
  std::vector<Dev::AxpyJob> jobs;
  // running oracle of size `n` = init + sum of the committed codewords of that size, each times its coefficient
  auto merge_codewords = [&](size_t n, const DBuf* init) {
    jobs.clear();
    for (size_t i = 0; i < nc; i++) if (comms[i]->codeword_size() == n) jobs.push_back({comms[i]->tree.leaves, coeffs[i], 1});
    DBuf b = dev.alloc(n, true);
    dev.axpy_many(b, init, jobs.data(), jobs.size());
    return b;
  };
  DBuf running = merge_codewords(cw_size, nullptr);
  DBuf sum_evals = dev.alloc(size_t(1) << num_vars, true);
  jobs.clear();
  for (size_t i = 0; i < nc; i++) jobs.push_back({comms[i]->bh_evals, coeffs[i], size_t(1) << (num_vars - comms[i]->nv)});
  dev.axpy_many(sum_evals, nullptr, jobs.data(), jobs.size());
  std::vector<Ext> rev_point(challenges.rbegin(), challenges.rend());
  DBuf eq = dev.alloc(size_t(1) << num_vars, true);
  dev.eq_table(eq, rev_point.data(), num_vars, ex_one(), false);  // == bit-reversed eq(point)
  std::vector<Ext> last(3);
  dev.bf_round(eq, sum_evals, nullptr, last.data());
  proof.sumcheck_messages.push_back(last);
  std::vector<DevTree> trees;
  // merges[i]: the committed codewords that join the running oracle at the top of round i (those as long as it is then)
  std::vector<std::vector<Dev::AxpyJob>> merges(num_rounds);
  for (unsigned i = 1; i < num_rounds; i++)
    for (size_t k = 0; k < nc; k++) if (comms[k]->codeword_size() == (cw_size >> i)) merges[i].push_back({comms[k]->tree.leaves, coeffs[k], 1});
  CommitLoopState st; st.last = last; st.running = running; st.eq = eq; st.sum_evals = sum_evals;
  commit_rounds(dev, merges, num_rounds, 0, true, st, t, proof.sumcheck_messages, proof.roots, trees, proof.final_message);
  lap("commit phase");
  // ---- batch_prover_query_phase (query_phase.rs:67-102, 419-472) and Merkle paths (:1062-1087)
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % cw_size));
  // The query section is laid out as its stream words (proof.h Writer::basefold / cq) and the device writes it in that form (Dev::query_section): every query opens
  // the pair (index | 1) - 1 of each oracle, index = x >> 1 halving from oracle to oracle, and the pair (xi | 1) - 1, xi = x >> (cw_log - height), of each codeword
  {
    const unsigned cw_log = num_vars + PCS_RATE_LOG;
    std::vector<Dev::QueryTree> qt;
    for (size_t k = 0; k < trees.size(); k++) qt.push_back({&trees[k], (unsigned)(1 + k)});
    for (const DevCommit* c : comms) qt.push_back({&c->tree, cw_log - c->tree.height()});
    std::vector<size_t> rel; size_t cpos;
    const size_t total = 1 + qidx.size() * Dev::query_section_layout(qt.data(), trees.size(), nc, rel, cpos);
    proof.queries_ser.acquire(total);
    dev.query_section(qidx.data(), qidx.size(), qt.data(), trees.size(), nc, proof.queries_ser.data(), total);
  }
  lap("query phase");
  dev.release(mk);
  return proof;
}

// ... in the shape zkml uses it: claim i = (polynomial i, point i) (commit/context.rs:370-383)
inline BasefoldProof pcs_batch_open(Dev& dev, unsigned full_log, const std::vector<OpenClaim>& claims, Transcript& t) {
  std::vector<const DevCommit*> comms; std::vector<std::vector<Ext>> points; std::vector<EvalClaim> evals;
  for (size_t i = 0; i < claims.size(); i++) { comms.push_back(claims[i].comm); points.push_back(claims[i].point); evals.push_back({i, i, claims[i].eval}); }
  return pcs_batch_open_evals(dev, full_log, comms, points, evals, t);
}

// PCS::open of one committed polynomial at one point (basefold.rs:466-544). commit_phase (commit_phase.rs:30-185) is the batch
// commit phase with nothing to merge: the first oracle is the committed codeword itself (read as extension elements), the
// sumcheck runs on eq(point, x) * f(x) over the bit-reversed hypercube evaluations; prover_query_phase (query_phase.rs:31-66,
// 373-417) opens the pair of the codeword at the query index and one pair per folded oracle. No classic sumcheck, no batch
// challenge: `sumcheck_proof` stays empty, every query carries exactly one commitment pair (ProofQueriesResultWithMerklePath::Single).
inline BasefoldProof pcs_open(Dev& dev, unsigned full_log, const DevCommit& c, const std::vector<Ext>& point, Transcript& t) {
  if (c.trivial()) return pcs_open_trivial(dev, c);  // Proof::trivial(evaluations): the transcript is not touched (basefold.rs:481-483)
  DP_REQUIRE(point.size() == c.nv, DP_ERR_SHAPE, "open: point length != num_vars");
  DP_REQUIRE(c.nv <= full_log, DP_ERR_SHAPE, "open: polynomial larger than the PCS parameters");
  BasefoldProof proof;
  size_t mk = dev.mark();
  const unsigned num_vars = c.nv, num_rounds = num_vars - PCS_BASECODE_LOG;
  const size_t cw_size = size_t(1) << (num_vars + PCS_RATE_LOG);
  CommitLoopState st;
  if (c.tree.leaves.ext) st.running = c.tree.leaves;  // fri_fold reads it, the committed leaves stay untouched
  else { Dev::AxpyJob j{c.tree.leaves, ex_one(), 1}; st.running = dev.alloc(cw_size, true); dev.axpy_many(st.running, nullptr, &j, 1); }
  st.sum_evals = dev.alloc(size_t(1) << num_vars, true);  // bf_round folds in place: always a copy
  { Dev::AxpyJob j{c.bh_evals, ex_one(), 1}; dev.axpy_many(st.sum_evals, nullptr, &j, 1); }
  std::vector<Ext> rev_point(point.rbegin(), point.rend());
  st.eq = dev.alloc(size_t(1) << num_vars, true);
  dev.eq_table(st.eq, rev_point.data(), num_vars, ex_one(), false);  // == bit-reversed eq(point)
  st.last.resize(3);
  dev.bf_round(st.eq, st.sum_evals, nullptr, st.last.data());
  proof.sumcheck_messages.push_back(st.last);
  std::vector<DevTree> trees;
  std::vector<std::vector<Dev::AxpyJob>> merges(num_rounds);
  commit_rounds(dev, merges, num_rounds, 0, true, st, t, proof.sumcheck_messages, proof.roots, trees, proof.final_message);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % cw_size));
  std::vector<QueryDesc> descs;
  for (size_t x : qidx) {
    size_t index = x >> 1;
    for (auto& tr : trees) { descs.push_back({&tr, (index | 1) - 1}); index >>= 1; }
    descs.push_back({&c.tree, (x | 1) - 1});
  }
  std::vector<u64> got; std::vector<size_t> goff;
  dev.query_gather_flat(descs.data(), descs.size(), got, goff);
  size_t di = 0;
  for (size_t x : qidx) {
    BatchedQuery bq; bq.index = x;
    for (size_t k = 0; k < trees.size(); k++, di++) bq.oracle_query.push_back(query_from_words(descs[di], got, goff, di));
    bq.commitments_query.push_back(query_from_words(descs[di], got, goff, di)); di++;
    proof.queries.push_back(std::move(bq));
  }
  dev.release(mk);
  return proof;
}

// PCS::batch_commit (basefold.rs:356-446): k polynomials of one size and one field behind ONE Merkle root. Every polynomial is encoded as in
// commit() (its own tree is kept: the query phase gathers the opened pairs through it); the common tree is Dev::batch_tree over the
// codewords (over the raw tables when trivial). One polynomial: the ordinary commitment (merkle_tree.rs:273-285).
struct DevBatchCommit {
  std::vector<DevCommit> polys;
  DevTree tree;  // leaves = row hashes (k >= 2)
  unsigned nv = 0;
  bool is_base = true;
  Digest root;
  bool trivial() const { return nv <= PCS_BASECODE_LOG; }
};
inline DevBatchCommit pcs_batch_commit(Dev& dev, const std::vector<DBuf>& evals, bool persistent) {
  DP_REQUIRE(!evals.empty(), DP_ERR_ARG, "cannot batch commit to zero polynomials");
  for (const DBuf& e : evals) DP_REQUIRE(e.n == evals[0].n && e.ext == evals[0].ext, DP_ERR_SHAPE, "cannot batch commit to polynomials of different sizes or fields");
  DevBatchCommit c;
  c.polys = dev.commit_many(evals, persistent);
  c.nv = c.polys[0].nv; c.is_base = c.polys[0].is_base;
  if (evals.size() == 1) { c.root = c.polys[0].tree.root; return c; }
  std::vector<DBuf> cws;
  for (const DevCommit& p : c.polys) cws.push_back(p.tree.leaves);
  c.tree = dev.batch_tree(cws.data(), (int)cws.size(), persistent);
  c.root = c.tree.root;
  return c;
}
// PCS::simple_batch_open (basefold.rs:777-861): all polynomials of a batch commitment at ONE point. "batch coeffs" challenges t, eq(t) as
// the random linear combination; simple_batch_commit_phase (commit_phase.rs:363-503) is the single-polynomial commit phase on
// sum_k eq(t)_k codeword_k / sum_k eq(t)_k bh_evals_k; simple_batch_prover_query_phase (query_phase.rs:104-139, 474-538) opens the row pair
// (every polynomial's pair at p0) with ONE Merkle path plus one pair per folded oracle.
// Stream form of ..::SimpleBatched: `commitments_query` has one entry per polynomial, all with index p0; the path rides on entry 0.
inline BasefoldProof pcs_simple_batch_open(Dev& dev, const DevBatchCommit& c, const std::vector<Ext>& point, Transcript& t) {
  BasefoldProof proof;
  const size_t k = c.polys.size();
  if (c.trivial()) {  // Proof::trivial(polynomials_bh_evals): the transcript is not touched (basefold.rs:788-790)
    for (const DevCommit& p : c.polys) { FieldVec fv; fv.is_ext = p.evals.ext; fv.w.resize(p.evals.n * (fv.is_ext ? 2 : 1)); dev.download(p.evals, fv.w.data()); proof.trivial_proof.push_back(std::move(fv)); }
    return proof;
  }
  DP_REQUIRE(point.size() == c.nv, DP_ERR_SHAPE, "simple_batch_open: point length != num_vars");
  unsigned batch_size_log = 0; while ((size_t(1) << batch_size_log) < k) batch_size_log++;
  std::vector<Ext> tt;
  for (unsigned i = 0; i < batch_size_log; i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
  std::vector<Ext> eq_xt = host_eq_table(tt);
  size_t mk = dev.mark();
  const unsigned num_vars = c.nv, num_rounds = num_vars - PCS_BASECODE_LOG;
  const size_t cw_size = size_t(1) << (num_vars + PCS_RATE_LOG);
  CommitLoopState st;
  std::vector<Dev::AxpyJob> jc, je;
  for (size_t q = 0; q < k; q++) { jc.push_back({c.polys[q].tree.leaves, eq_xt[q], 1}); je.push_back({c.polys[q].bh_evals, eq_xt[q], 1}); }
  st.running = dev.alloc(cw_size, true); dev.axpy_many(st.running, nullptr, jc.data(), jc.size());
  st.sum_evals = dev.alloc(size_t(1) << num_vars, true); dev.axpy_many(st.sum_evals, nullptr, je.data(), je.size());
  std::vector<Ext> rev_point(point.rbegin(), point.rend());
  st.eq = dev.alloc(size_t(1) << num_vars, true);
  dev.eq_table(st.eq, rev_point.data(), num_vars, ex_one(), false);
  st.last.resize(3);
  dev.bf_round(st.eq, st.sum_evals, nullptr, st.last.data());
  proof.sumcheck_messages.push_back(st.last);
  std::vector<DevTree> trees;
  std::vector<std::vector<Dev::AxpyJob>> merges(num_rounds);
  commit_rounds(dev, merges, num_rounds, 0, true, st, t, proof.sumcheck_messages, proof.roots, trees, proof.final_message);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % cw_size));
  std::vector<QueryDesc> descs;
  for (size_t x : qidx) {
    size_t index = x >> 1;
    for (auto& tr : trees) { descs.push_back({&tr, (index | 1) - 1}); index >>= 1; }
    const size_t p0 = (x | 1) - 1;
    for (size_t q = 0; q < k; q++) descs.push_back({&c.polys[q].tree, p0});
    if (k > 1) descs.push_back({&c.tree, 2 * p0});  // row hashes: two entries per row; the first digest of that path is hash(row p0 + 1)
  }
  std::vector<u64> got; std::vector<size_t> goff;
  dev.query_gather_flat(descs.data(), descs.size(), got, goff);
  size_t di = 0;
  for (size_t x : qidx) {
    BatchedQuery bq; bq.index = x;
    for (size_t j = 0; j < trees.size(); j++, di++) bq.oracle_query.push_back(query_from_words(descs[di], got, goff, di));
    for (size_t q = 0; q < k; q++, di++) { CodewordQuery cq = query_from_words(descs[di], got, goff, di); if (k > 1) cq.path.clear(); bq.commitments_query.push_back(std::move(cq)); }
    if (k > 1) { CodewordQuery rows = query_from_words(descs[di], got, goff, di); di++; bq.commitments_query[0].path.assign(rows.path.begin() + 1, rows.path.end()); }
    proof.queries.push_back(std::move(bq));
  }
  dev.release(mk);
  return proof;
}

// =================================================================== host verifier
struct VerifierParams { unsigned full_log = 0; };

inline Digest host_leaf_pair_digest(bool is_ext, Ext l, Ext r) {  // hash_two_leaves* -> hash_or_noop (<= 4 elements: no hash)
  Digest d;
  if (is_ext) { d.v[0] = l.c0; d.v[1] = l.c1; d.v[2] = r.c0; d.v[3] = r.c1; }
  else { d.v[0] = l.c0; d.v[1] = r.c0; d.v[2] = 0; d.v[3] = 0; }
  return d;
}
// A verifier spends its time here: ~125 000 compress() for one Dense-4M proof (200 queries x (12 oracle trees + ~35
// commitments) x path depth), 0.4 s of one host core against 0.1 ms of the GPU. A batch verifier therefore only RECORDS the
// paths while it runs the protocol checks (g_merkle_sink set, per thread) and authenticates all of them at once afterwards
// (Dev::merkle_paths_check: one path per lane); the paths point into the Proof object, which outlives the check.
struct MerkleJob { Digest leaf; size_t x; const Digest* path; size_t depth; Digest root; };
inline std::vector<MerkleJob>*& merkle_sink() { static thread_local std::vector<MerkleJob>* s = nullptr; return s; }
inline bool merkle_job_ok(const MerkleJob& j) {
  Digest h = j.leaf; size_t x = j.x;
  for (size_t l = 0; l < j.depth; l++) { h = (x & 1) ? host_compress(j.path[l], h) : host_compress(h, j.path[l]); x >>= 1; }
  return h == j.root;
}
// all recorded paths of a proof. With the vectorised compression (p2_avx512.cpp, installed by the library on AVX-512 CPUs) eight paths
// climb side by side, one per lane, sorted by depth so that the lanes of a group finish together; a finished lane idles masked.
// groups [first, .., step `stride`) of eight paths each, `order` = the jobs by decreasing depth
using P2Compress8 = void (*)(const u64 (*)[4], const u64 (*)[4], u64 (*)[4]);
inline bool merkle_job_groups_ok(const std::vector<MerkleJob>& jobs, const std::vector<size_t>& order, size_t first, size_t stride, P2Compress8 c8) {
  for (size_t g = first * 8; g < order.size(); g += stride * 8) {
    const size_t m = std::min<size_t>(8, order.size() - g);
    u64 h[8][4], l[8][4], r[8][4], o[8][4]; size_t x[8], depth[8]; size_t maxd = 0;
    for (size_t k = 0; k < 8; k++) {
      const MerkleJob& j = jobs[order[g + (k < m ? k : 0)]];  // (a short last group repeats its first path)
      for (int q = 0; q < 4; q++) h[k][q] = j.leaf.v[q];
      x[k] = j.x; depth[k] = j.depth; maxd = std::max(maxd, j.depth);
    }
    for (size_t lv = 0; lv < maxd; lv++) {
      for (size_t k = 0; k < 8; k++) {
        const MerkleJob& j = jobs[order[g + (k < m ? k : 0)]];
        static const Digest idle{};
        const Digest& sib = lv < depth[k] ? j.path[lv] : idle;  // (a lane whose path has ended hashes on, its result is dropped)
        const bool right_child = (x[k] >> lv) & 1;
        for (int q = 0; q < 4; q++) { l[k][q] = right_child ? sib.v[q] : h[k][q]; r[k][q] = right_child ? h[k][q] : sib.v[q]; }
      }
      c8(l, r, o);
      for (size_t k = 0; k < 8; k++) if (lv < depth[k]) for (int q = 0; q < 4; q++) h[k][q] = o[k][q];
    }
    for (size_t k = 0; k < m; k++) { const MerkleJob& j = jobs[order[g + k]]; for (int q = 0; q < 4; q++) if (h[k][q] != j.root.v[q]) return false; }
  }
  return true;
}
// `threads` > 1 (the single-proof entry points, whose caller has nothing else to do meanwhile): the groups are independent, thread t takes
// every threads-th one. DP_VERIFY_THREADS (default: up to 8 of the machine's cores).
inline unsigned verify_threads() {
  static const unsigned n = [] { const char* e = getenv("DP_VERIFY_THREADS"); unsigned hw = std::thread::hardware_concurrency(); unsigned d = std::max(1u, std::min(8u, hw));
                                 return e ? (unsigned)std::max(1, atoi(e)) : d; }();
  return n;
}
inline bool merkle_jobs_ok(const std::vector<MerkleJob>& jobs, unsigned threads = 1) {
  auto c8 = p2_fast_compress8();
  if (threads > 1 && jobs.size() >= 4096) {
    std::vector<char> ok(threads, 1);
    std::vector<std::thread> th;
    if (!c8) {
      auto part = [&](unsigned t) { for (size_t i = t; i < jobs.size(); i += threads) if (!merkle_job_ok(jobs[i])) { ok[t] = 0; return; } };
      for (unsigned t = 1; t < threads; t++) { try { th.emplace_back(part, t); } catch (const std::system_error&) { part(t); } }  // (no thread to be had: the share is done here)
      part(0);
    } else {
      std::vector<size_t> order(jobs.size());
      for (size_t i = 0; i < order.size(); i++) order[i] = i;
      std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].depth > jobs[b].depth; });
      auto part = [&](unsigned t) { ok[t] = merkle_job_groups_ok(jobs, order, t, threads, c8) ? 1 : 0; };
      for (unsigned t = 1; t < threads; t++) { try { th.emplace_back(part, t); } catch (const std::system_error&) { part(t); } }
      part(0);
    }
    for (auto& t : th) t.join();
    for (char v : ok) if (!v) return false;
    return true;
  }
  if (!c8 || jobs.size() < 8) { for (const MerkleJob& j : jobs) if (!merkle_job_ok(j)) return false; return true; }
  std::vector<size_t> order(jobs.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].depth > jobs[b].depth; });
  return merkle_job_groups_ok(jobs, order, 0, 1, c8);
}
// authenticate_merkle_path_root (merkle_tree.rs:331-420)
// `depth`: the height the tree of this codeword must have — 2^depth leaf pairs (a proof that carries a shorter or longer path would
// otherwise pick its own tree size; the reference takes the path as it comes)
inline void check_merkle_path(const CodewordQuery& q, const Digest& root, size_t depth) {
  DP_REQUIRE(q.path.size() == depth, DP_ERR_VERIFY, "merkle path: wrong length for the size of this codeword");
  if (std::vector<MerkleJob>* sink = merkle_sink()) { sink->push_back({host_leaf_pair_digest(q.is_ext, q.left, q.right), q.index >> 1, q.path.data(), q.path.size(), root}); return; }
  Digest h = host_leaf_pair_digest(q.is_ext, q.left, q.right);
  size_t x = q.index >> 1;
  for (const Digest& sib : q.path) {
    h = (x & 1) ? host_compress(sib, h) : host_compress(h, sib);
    x >>= 1;
  }
  DP_REQUIRE(h == root, DP_ERR_VERIFY, "merkle path does not authenticate against the root");
}
inline Digest host_merkle_root(const FieldVec& leaves) {  // MerkleTree::from_leaves on a small vector
  size_t n = leaves.len();
  DP_REQUIRE(n >= 2 && (n & (n - 1)) == 0, DP_ERR_VERIFY, "trivial proof: bad leaf count");
  std::vector<Digest> cur(n / 2);
  for (size_t i = 0; i < n / 2; i++) {
    if (leaves.is_ext) cur[i] = host_leaf_pair_digest(true, ex(leaves.w[4 * i], leaves.w[4 * i + 1]), ex(leaves.w[4 * i + 2], leaves.w[4 * i + 3]));
    else cur[i] = host_leaf_pair_digest(false, ex(leaves.w[2 * i], 0), ex(leaves.w[2 * i + 1], 0));
  }
  while (cur.size() > 1) {
    std::vector<Digest> nx(cur.size() / 2);
    for (size_t i = 0; i < nx.size(); i++) nx[i] = host_compress(cur[2 * i], cur[2 * i + 1]);
    cur = nx;
  }
  return cur[0];
}
// PCS::verify for trivial proofs (basefold.rs:873-894); does not touch the transcript
inline void pcs_verify_trivial(const Commitment& comm, const std::vector<Ext>& point, Ext eval, const BasefoldProof& proof) {
  DP_REQUIRE(proof.is_trivial() && proof.trivial_proof.size() == 1, DP_ERR_VERIFY, "expected a trivial opening proof");
  const FieldVec& fv = proof.trivial_proof[0];
  DP_REQUIRE(host_merkle_root(fv) == comm.root, DP_ERR_VERIFY, "trivial proof: Merkle root mismatch");
  // (the commitment's own description of the polynomial must fit the opened table: the reference never looks at it for a trivial opening)
  DP_REQUIRE((size_t(1) << comm.num_vars) == fv.len() && comm.is_base == !fv.is_ext, DP_ERR_VERIFY, "trivial proof: table does not match the commitment's shape");
  std::vector<Ext> v(fv.len());
  for (size_t i = 0; i < v.size(); i++) v[i] = fv.is_ext ? ex(fv.w[2 * i], fv.w[2 * i + 1]) : ex(fv.w[i], 0);
  DP_REQUIRE(ex_eq(host_mle_eval(v, point), eval), DP_ERR_VERIFY, "trivial proof: wrong evaluation");
}
// x0 (and w = -1/(2 x0)) of verifier_folding_coeffs (rs.rs:412-456), computed from first principles:
// x0 = gamma^(2^(full_log+1-level-1)) * omega_{2^(level+1)}^{bitrev(index, level)}
inline void folding_coeffs(unsigned full_log, unsigned level, size_t index, u64& x0, u64& w) {
  size_t bi = dp_reverse_bits(index, level);
  u64 g = GL_G32;  // generator of the 2^(level+1) subgroup
  for (unsigned i = level + 1; i < 32; i++) g = gl_sqr(g);
  u64 root = gl_pow(g, bi);
  u64 gam = GL_GENERATOR;
  for (unsigned i = 0; i < full_log + PCS_RATE_LOG - level - 1; i++) gam = gl_sqr(gam);
  x0 = gl_mul(root, gam);
  w = gl_neg(gl_inv(gl_dbl(x0)));
}
inline Ext interpolate2_weights(Ext a0, Ext a1, Ext b1, Ext w, Ext x) { return ex_add(a1, ex_mul(ex_mul(ex_sub(x, a0), ex_sub(b1, a1)), w)); }

struct VerifyClaim { Commitment comm; std::vector<Ext> point; Ext eval; };

// final codeword: encode_small(interpolate(bitrev(final_message))) then bit-reverse (query_phase.rs:158-172, 239-251).
// Evaluated directly: message coefficients c (multilinear, bit-reversed order), codeword[k] = sum_i c_i (shift w^k)^i.
inline std::vector<Ext> final_codeword_of(const VerifierParams& vp, const std::vector<Ext>& final_message) {
  size_t mlen = final_message.size();
  std::vector<Ext> msg(mlen);
  for (size_t j = 0; j < mlen; j++) msg[dp_reverse_bits(j, PCS_BASECODE_LOG)] = final_message[j];
  for (unsigned i = 1; i <= PCS_BASECODE_LOG; i++) {
    size_t chunk = size_t(1) << i, half = chunk >> 1;
    for (size_t c = 0; c < mlen; c += chunk) for (size_t j = half; j < chunk; j++) msg[c + j] = ex_sub(msg[c + j], msg[c + j - half]);
  }
  u64 shift = GL_GENERATOR;
  for (unsigned i = 0; i < vp.full_log - PCS_BASECODE_LOG; i++) shift = gl_sqr(shift);
  u64 w256 = GL_G32;
  for (unsigned i = PCS_BASECODE_LOG + PCS_RATE_LOG; i < 32; i++) w256 = gl_sqr(w256);
  size_t flen = mlen << PCS_RATE_LOG;
  std::vector<Ext> final_codeword(flen);
  for (size_t k = 0; k < flen; k++) {
    u64 x = gl_mul(shift, gl_pow(w256, k));
    Ext acc = ex_zero();
    for (size_t i = mlen; i-- > 0;) acc = ex_add(ex_mul_base(acc, x), msg[i]);
    final_codeword[dp_reverse_bits(k, PCS_BASECODE_LOG + PCS_RATE_LOG)] = acc;
  }
  return final_codeword;
}

// PCS::verify (basefold.rs:863-962) + verifier_query_phase (query_phase.rs:141-211) + SingleQueryResultWithMerklePath::check
// (:915-974): replay the commit-phase transcript, authenticate every opened pair, fold the codeword pair down the oracles to
// the final codeword, then the sumcheck chain: eval = h_0(0) + h_0(1), h_i(r_i) = h_{i+1}(0) + h_{i+1}(1), and
// h_last(r_last) = <final_message, eq(point_head) * eq(point_tail, reversed challenges)>.
// pcs_verify_core, `batch` false: PCS::verify of one polynomial (evals = {eval}); true: PCS::simple_batch_verify (basefold.rs:1100-1203,
// query_phase.rs:292-371, 1441-1530) of the evals.size() polynomials behind `comm`: "batch coeffs" are drawn first, each query carries
// one pair per polynomial under ONE Merkle path (row pair digest = hash_two_leaves_batch), the fold chain starts from the eq(t)-weighted
// sums of the pairs, and the first commit-phase message must sum to <eq(t), evals>.
inline Digest host_row_hash(const std::vector<CodewordQuery>& qs, bool right) {  // hash_or_noop of [v_0, .., v_{k-1}] (util/hash.rs:17-24)
  std::vector<u64> w;
  for (const CodewordQuery& q : qs) { const Ext v = right ? q.right : q.left; w.push_back(v.c0); if (q.is_ext) w.push_back(v.c1); }
  Digest d{};
  if (w.size() <= 4) { for (size_t i = 0; i < w.size(); i++) d.v[i] = w[i]; return d; }
  Challenger ch;
  for (u64 x : w) ch.observe(x);
  for (int i = 0; i < 4; i++) d.v[i] = ch.sample();
  return d;
}
inline void pcs_verify_core(const VerifierParams& vp, const Commitment& comm, const std::vector<Ext>& point, const std::vector<Ext>& evals, bool batch, const BasefoldProof& proof, Transcript& t) {
  const size_t np = evals.size();
  DP_REQUIRE(np >= 1 && (batch || np == 1), DP_ERR_ARG, "verify: evaluations");
  std::vector<Ext> eq_xt(1, ex_one());
  if (batch) {
    std::vector<Ext> tt;
    for (unsigned i = 0; i < dp_ceil_log2(np); i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
    eq_xt = host_eq_table(tt);
  }
  Ext eval = ex_zero();
  for (size_t k = 0; k < np; k++) eval = ex_add(eval, ex_mul(evals[k], eq_xt[k]));
  const unsigned num_vars = (unsigned)point.size();
  DP_REQUIRE(num_vars == comm.num_vars && num_vars > PCS_BASECODE_LOG && num_vars <= vp.full_log, DP_ERR_VERIFY, "verify: bad shapes");
  DP_REQUIRE(proof.sumcheck_proof.empty() && proof.trivial_proof.empty(), DP_ERR_VERIFY, "verify: a single opening carries no batch sumcheck");
  const unsigned num_rounds = num_vars - PCS_BASECODE_LOG;
  DP_REQUIRE(proof.sumcheck_messages.size() == num_rounds && proof.roots.size() + 1 == num_rounds, DP_ERR_VERIFY, "verify: commit-phase shape");
  std::vector<Ext> fold_ch;
  for (unsigned i = 0; i < num_rounds; i++) {
    DP_REQUIRE(proof.sumcheck_messages[i].size() == 3, DP_ERR_VERIFY, "verify: commit message size");
    t.append_exts(proof.sumcheck_messages[i]);
    fold_ch.push_back(t.get_and_append_challenge("commit round"));
    if (i + 1 < num_rounds) t.append_digest(proof.roots[i]);
  }
  DP_REQUIRE(proof.final_message.size() == (size_t(1) << PCS_BASECODE_LOG), DP_ERR_VERIFY, "verify: final message size");
  t.append_exts(proof.final_message);
  const size_t cw_size = size_t(1) << (num_vars + PCS_RATE_LOG);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % cw_size));
  std::vector<Ext> rev(fold_ch.rbegin(), fold_ch.rend());
  Ext coeff = eq_eval(point.data() + (point.size() - fold_ch.size()), rev.data(), fold_ch.size());
  std::vector<Ext> head(point.begin(), point.end() - fold_ch.size());
  std::vector<Ext> peq = host_eq_table(head);
  for (auto& e : peq) e = ex_mul(e, coeff);
  std::vector<Ext> final_codeword = final_codeword_of(vp, proof.final_message);
  DP_REQUIRE(proof.queries.size() == PCS_NUM_QUERIES, DP_ERR_VERIFY, "verify: wrong number of queries");
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) {
    const BatchedQuery& bq = proof.queries[q];
    const size_t index = qidx[q];
    DP_REQUIRE(bq.index == index, DP_ERR_VERIFY, "verify: query index mismatch");
    DP_REQUIRE(bq.oracle_query.size() == proof.roots.size() && bq.commitments_query.size() == np, DP_ERR_VERIFY, "verify: query shape");
    for (size_t k = 0; k < bq.oracle_query.size(); k++) check_merkle_path(bq.oracle_query[k], proof.roots[k], num_vars + PCS_RATE_LOG - k - 2);
    size_t right_index = index | 1, left_index = right_index - 1;
    Ext cur_l = ex_zero(), cur_r = ex_zero();
    for (size_t k = 0; k < np; k++) {
      const CodewordQuery& cq = bq.commitments_query[k];
      DP_REQUIRE(cq.is_ext == !comm.is_base, DP_ERR_VERIFY, "verify: field type of the opened codeword");
      DP_REQUIRE(cq.index == left_index, DP_ERR_VERIFY, "verify: commitment query index");
      DP_REQUIRE(k == 0 || cq.path.empty(), DP_ERR_VERIFY, "verify: the row pair has one Merkle path");
      cur_l = ex_add(cur_l, ex_mul(cq.left, eq_xt[k])); cur_r = ex_add(cur_r, ex_mul(cq.right, eq_xt[k]));
    }
    if (np == 1) check_merkle_path(bq.commitments_query[0], comm.root, num_vars + PCS_RATE_LOG - 1);
    else {  // authenticate_merkle_path_root_batch (merkle_tree.rs:449-490)
      const CodewordQuery& c0 = bq.commitments_query[0];
      DP_REQUIRE(c0.path.size() == num_vars + PCS_RATE_LOG - 1, DP_ERR_VERIFY, "merkle path: wrong length for the size of this codeword");
      Digest h = host_compress(host_row_hash(bq.commitments_query, false), host_row_hash(bq.commitments_query, true));
      if (std::vector<MerkleJob>* sink = merkle_sink()) sink->push_back({h, c0.index >> 1, c0.path.data(), c0.path.size(), comm.root});
      else { MerkleJob j{h, c0.index >> 1, c0.path.data(), c0.path.size(), comm.root}; DP_REQUIRE(merkle_job_ok(j), DP_ERR_VERIFY, "merkle path does not authenticate against the root"); }
    }
    for (unsigned i = 0; i < num_rounds; i++) {
      u64 x0, w;
      folding_coeffs(vp.full_log, num_vars + PCS_RATE_LOG - i - 1, left_index >> 1, x0, w);
      Ext res = interpolate2_weights(ex_base(x0), cur_l, cur_r, ex_base(w), fold_ch[i]);
      size_t next_index = right_index >> 1;
      Ext next_val;
      if (i + 1 < num_rounds) {
        right_index = next_index | 1; left_index = right_index - 1;
        const CodewordQuery& oq = bq.oracle_query[i];
        DP_REQUIRE(oq.index == left_index && oq.is_ext, DP_ERR_VERIFY, "verify: oracle query index");
        cur_l = oq.left; cur_r = oq.right;
        next_val = (next_index & 1) ? cur_r : cur_l;
      } else next_val = final_codeword[next_index];
      DP_REQUIRE(ex_eq(res, next_val), DP_ERR_VERIFY, "verify: folding check failed");
    }
  }
  auto zero_plus_one = [](const std::vector<Ext>& p) { return ex_add(ex_add(ex_dbl(p[0]), p[1]), p[2]); };
  auto eval2 = [](const std::vector<Ext>& p, Ext x) { return ex_add(p[0], ex_add(ex_mul(x, p[1]), ex_mul(ex_mul(x, x), p[2]))); };
  DP_REQUIRE(ex_eq(eval, zero_plus_one(proof.sumcheck_messages[0])), DP_ERR_VERIFY, "verify: first commit-phase message does not match the evaluation");
  for (unsigned i = 0; i + 1 < num_rounds; i++)
    DP_REQUIRE(ex_eq(eval2(proof.sumcheck_messages[i], fold_ch[i]), zero_plus_one(proof.sumcheck_messages[i + 1])), DP_ERR_VERIFY, "verify: commit-phase sumcheck chain");
  Ext ip = ex_zero();
  for (size_t i = 0; i < peq.size(); i++) ip = ex_add(ip, ex_mul(proof.final_message[i], peq[i]));
  DP_REQUIRE(ex_eq(eval2(proof.sumcheck_messages[num_rounds - 1], fold_ch[num_rounds - 1]), ip), DP_ERR_VERIFY, "verify: final message inner product");
}
inline void pcs_verify(const VerifierParams& vp, const Commitment& comm, const std::vector<Ext>& point, Ext eval, const BasefoldProof& proof, Transcript& t) {
  if (proof.is_trivial()) { pcs_verify_trivial(comm, point, eval, proof); return; }
  pcs_verify_core(vp, comm, point, {eval}, false, proof, t);
}
// root of MerkleTree::from_batch_leaves over small tables (the trivial branch of simple_batch_verify, basefold.rs:1114-1124)
inline Digest host_batch_merkle_root(const std::vector<FieldVec>& tabs) {
  if (tabs.size() == 1) return host_merkle_root(tabs[0]);
  const size_t n = tabs[0].len();
  DP_REQUIRE(n >= 2 && (n & (n - 1)) == 0, DP_ERR_VERIFY, "trivial proof: bad leaf count");
  auto row = [&](size_t j) {
    std::vector<CodewordQuery> qs;
    for (const FieldVec& f : tabs) { CodewordQuery q; q.is_ext = f.is_ext; q.left = f.is_ext ? ex(f.w[2 * j], f.w[2 * j + 1]) : ex(f.w[j], 0); qs.push_back(q); }
    return host_row_hash(qs, false);
  };
  std::vector<Digest> cur(n / 2);
  for (size_t i = 0; i < n / 2; i++) cur[i] = host_compress(row(2 * i), row(2 * i + 1));
  while (cur.size() > 1) {
    std::vector<Digest> nx(cur.size() / 2);
    for (size_t i = 0; i < nx.size(); i++) nx[i] = host_compress(cur[2 * i], cur[2 * i + 1]);
    cur = nx;
  }
  return cur[0];
}
// PCS::simple_batch_verify. Trivial proofs: the reference only compares the root of the opened tables (basefold.rs:1114-1124); here every
// table must also have the commitment's shape and evaluate to its claimed value at the point (costs an honest prover nothing).
inline void pcs_simple_batch_verify(const VerifierParams& vp, const Commitment& comm, const std::vector<Ext>& point, const std::vector<Ext>& evals, const BasefoldProof& proof, Transcript& t) {
  DP_REQUIRE(!evals.empty(), DP_ERR_ARG, "simple_batch_verify: no evaluations");
  if (proof.is_trivial()) {
    DP_REQUIRE(comm.num_vars <= PCS_BASECODE_LOG && point.size() == comm.num_vars, DP_ERR_VERIFY, "trivial proof for a non-trivial commitment");
    DP_REQUIRE(proof.trivial_proof.size() == evals.size(), DP_ERR_VERIFY, "trivial proof: one table per polynomial expected");
    for (const FieldVec& fv : proof.trivial_proof) DP_REQUIRE((size_t(1) << comm.num_vars) == fv.len() && comm.is_base == !fv.is_ext, DP_ERR_VERIFY, "trivial proof: table does not match the commitment's shape");
    DP_REQUIRE(host_batch_merkle_root(proof.trivial_proof) == comm.root, DP_ERR_VERIFY, "trivial proof: Merkle root mismatch");
    for (size_t k = 0; k < evals.size(); k++) {
      const FieldVec& fv = proof.trivial_proof[k];
      std::vector<Ext> v(fv.len());
      for (size_t i = 0; i < v.size(); i++) v[i] = fv.is_ext ? ex(fv.w[2 * i], fv.w[2 * i + 1]) : ex(fv.w[i], 0);
      DP_REQUIRE(ex_eq(host_mle_eval(v, point), evals[k]), DP_ERR_VERIFY, "trivial proof: wrong evaluation");
    }
    return;
  }
  pcs_verify_core(vp, comm, point, evals, true, proof, t);
}

// PCS::batch_verify (basefold.rs:964-1098) + batch_verifier_query_phase (query_phase.rs:220-288) + check (:1116-1236)
struct VerifyEval { size_t poly, point; Ext eval; };
inline void pcs_batch_verify_evals(const VerifierParams& vp, const std::vector<Commitment>& comms, const std::vector<std::vector<Ext>>& points, const std::vector<VerifyEval>& evals,
                                   const BasefoldProof& proof, Transcript& t) {
  if (comms.empty() && points.empty() && evals.empty() && proof.trivial_proof.empty() && proof.is_trivial()) return;
  DP_REQUIRE(!comms.empty() && !evals.empty(), DP_ERR_VERIFY, "batch_verify: proof given but no claims");
  const size_t np = evals.size(), nc = comms.size();
  unsigned num_vars = 0, min_nv = ~0u;
  for (const Commitment& c : comms) { num_vars = std::max(num_vars, c.num_vars); min_nv = std::min(min_nv, c.num_vars); }
  for (const VerifyEval& e : evals) {
    DP_REQUIRE(e.poly < nc && e.point < points.size(), DP_ERR_ARG, "batch_verify: evaluation refers to a missing polynomial or point");
    DP_REQUIRE(points[e.point].size() == comms[e.poly].num_vars, DP_ERR_VERIFY, "batch_verify: point length != num_vars");
  }
  DP_REQUIRE(min_nv >= PCS_BASECODE_LOG && num_vars <= vp.full_log && !proof.is_trivial(), DP_ERR_VERIFY, "batch_verify: bad shapes");
  unsigned num_rounds = num_vars - PCS_BASECODE_LOG;
  unsigned bsl = dp_ceil_log2(np);
  std::vector<Ext> tt;
  for (unsigned i = 0; i < bsl; i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
  std::vector<Ext> eq_xt = host_eq_table(tt);
  Ext target = ex_zero();
  for (size_t i = 0; i < np; i++)
    target = ex_add(target, ex_mul(ex_mul(evals[i].eval, ex_from_u64(u64(1) << (num_vars - comms[evals[i].poly].num_vars))), eq_xt[i]));
  // SumCheck::verify (classic.rs:287-330): coefficients form, degree 2
  DP_REQUIRE(proof.sumcheck_proof.size() == num_vars, DP_ERR_VERIFY, "batch_verify: wrong number of sumcheck rounds");
  std::vector<Ext> vpoint;
  Ext sum = target;
  for (unsigned i = 0; i < num_vars; i++) {
    const auto& m = proof.sumcheck_proof[i];
    DP_REQUIRE(m.size() == 3, DP_ERR_VERIFY, "batch_verify: round message must hold 3 coefficients");
    for (const Ext& e : m) t.append_ext(e);
    Ext ch = t.get_and_append_challenge("sumcheck round");
    vpoint.push_back(ch);
    // msg.sum() = c0 + (c0+c1+c2) must equal the running sum
    DP_REQUIRE(ex_eq(ex_add(m[0], ex_add(ex_add(m[0], m[1]), m[2])), sum), DP_ERR_VERIFY, "batch_verify: classic sumcheck consistency failure");
    sum = ex_add(m[0], ex_mul(ch, ex_add(m[1], ex_mul(ch, m[2]))));
  }
  Ext new_target = sum;
  std::vector<Ext> coeffs(nc, ex_zero());  // per commitment: the sum over its evaluations (basefold.rs:1040-1051)
  for (size_t i = 0; i < np; i++) {
    const std::vector<Ext>& pt = points[evals[i].point];
    coeffs[evals[i].poly] = ex_add(coeffs[evals[i].poly], ex_mul(eq_eval(vpoint.data(), pt.data(), pt.size()), eq_xt[i]));
  }
  DP_REQUIRE(proof.sumcheck_messages.size() == num_rounds && proof.roots.size() + 1 == num_rounds, DP_ERR_VERIFY, "batch_verify: commit-phase shape");
  std::vector<Ext> fold_ch;
  for (unsigned i = 0; i < num_rounds; i++) {
    DP_REQUIRE(proof.sumcheck_messages[i].size() == 3, DP_ERR_VERIFY, "batch_verify: commit message size");
    t.append_exts(proof.sumcheck_messages[i]);
    fold_ch.push_back(t.get_and_append_challenge("commit round"));
    if (i + 1 < num_rounds) t.append_digest(proof.roots[i]);
  }
  DP_REQUIRE(proof.final_message.size() == (size_t(1) << PCS_BASECODE_LOG), DP_ERR_VERIFY, "batch_verify: final message size");
  t.append_exts(proof.final_message);
  size_t cw_size = size_t(1) << (num_vars + PCS_RATE_LOG);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % cw_size));
  // partial eq (basefold.rs:1066-1078)
  std::vector<Ext> rev(fold_ch.rbegin(), fold_ch.rend());
  Ext coeff = eq_eval(vpoint.data() + (vpoint.size() - fold_ch.size()), rev.data(), fold_ch.size());
  std::vector<Ext> head(vpoint.begin(), vpoint.end() - fold_ch.size());
  std::vector<Ext> peq = host_eq_table(head);
  for (auto& e : peq) e = ex_mul(e, coeff);
  size_t mlen = proof.final_message.size();
  std::vector<Ext> final_codeword = final_codeword_of(vp, proof.final_message);
  DP_REQUIRE(proof.queries.size() == PCS_NUM_QUERIES, DP_ERR_VERIFY, "batch_verify: wrong number of queries");
  for (unsigned q = 0; q < PCS_NUM_QUERIES; q++) {
    const BatchedQuery& bq = proof.queries[q];
    size_t index = qidx[q];
    DP_REQUIRE(bq.index == index, DP_ERR_VERIFY, "batch_verify: query index mismatch");
    DP_REQUIRE(bq.oracle_query.size() == proof.roots.size() && bq.commitments_query.size() == nc, DP_ERR_VERIFY, "batch_verify: query shape");
    for (size_t k = 0; k < bq.oracle_query.size(); k++) check_merkle_path(bq.oracle_query[k], proof.roots[k], num_vars + PCS_RATE_LOG - k - 2);
    for (size_t k = 0; k < nc; k++) {
      DP_REQUIRE(bq.commitments_query[k].is_ext == !comms[k].is_base, DP_ERR_VERIFY, "batch_verify: field type of opened codeword");
      check_merkle_path(bq.commitments_query[k], comms[k].root, comms[k].num_vars + PCS_RATE_LOG - 1);
    }
    Ext cur_l = ex_zero(), cur_r = ex_zero();
    size_t right_index = index | 1, left_index = right_index - 1;
    for (unsigned i = 0; i < num_rounds; i++) {
      for (size_t k = 0; k < nc; k++) if (comms[k].num_vars == num_vars - i) {
        const CodewordQuery& cq = bq.commitments_query[k];
        DP_REQUIRE(cq.index == left_index, DP_ERR_VERIFY, "batch_verify: commitment query index");  // provers emit the index of the LEFT element of the pair
        cur_l = ex_add(cur_l, ex_mul(cq.left, coeffs[k]));
        cur_r = ex_add(cur_r, ex_mul(cq.right, coeffs[k]));
      }
      u64 x0, w;
      folding_coeffs(vp.full_log, num_vars + PCS_RATE_LOG - i - 1, left_index >> 1, x0, w);
      Ext res = interpolate2_weights(ex_base(x0), cur_l, cur_r, ex_base(w), fold_ch[i]);
      size_t next_index = right_index >> 1;
      Ext next_val;
      if (i + 1 < num_rounds) {
        right_index = next_index | 1; left_index = right_index - 1;
        const CodewordQuery& oq = bq.oracle_query[i];
        DP_REQUIRE(oq.index == left_index && oq.is_ext, DP_ERR_VERIFY, "batch_verify: oracle query index");
        cur_l = oq.left; cur_r = oq.right;
        next_val = (next_index & 1) ? cur_r : cur_l;
      } else {
        for (size_t k = 0; k < nc; k++) if (comms[k].num_vars == num_vars - i - 1) {
          const CodewordQuery& cq = bq.commitments_query[k];
          DP_REQUIRE(cq.index == (next_index | 1) - 1, DP_ERR_VERIFY, "batch_verify: last-round commitment query index");
          res = ex_add(res, ex_mul((next_index & 1) ? cq.right : cq.left, coeffs[k]));
        }
        next_val = final_codeword[next_index];
      }
      DP_REQUIRE(ex_eq(res, next_val), DP_ERR_VERIFY, "batch_verify: folding check failed");
    }
  }
  // final checks (query_phase.rs:264-288)
  auto zero_plus_one = [](const std::vector<Ext>& p) { return ex_add(ex_add(ex_dbl(p[0]), p[1]), p[2]); };
  auto eval2 = [](const std::vector<Ext>& p, Ext x) { return ex_add(p[0], ex_add(ex_mul(x, p[1]), ex_mul(ex_mul(x, x), p[2]))); };
  DP_REQUIRE(ex_eq(new_target, zero_plus_one(proof.sumcheck_messages[0])), DP_ERR_VERIFY, "batch_verify: first commit-phase message does not match the sum");
  for (unsigned i = 0; i + 1 < num_rounds; i++)
    DP_REQUIRE(ex_eq(eval2(proof.sumcheck_messages[i], fold_ch[i]), zero_plus_one(proof.sumcheck_messages[i + 1])), DP_ERR_VERIFY, "batch_verify: commit-phase sumcheck chain");
  Ext ip = ex_zero();
  for (size_t i = 0; i < mlen; i++) ip = ex_add(ip, ex_mul(proof.final_message[i], peq[i]));
  DP_REQUIRE(ex_eq(eval2(proof.sumcheck_messages[num_rounds - 1], fold_ch[num_rounds - 1]), ip), DP_ERR_VERIFY, "batch_verify: final message inner product");
}

// ... in the shape zkml uses it: claim i = (commitment i, point i)
inline void pcs_batch_verify(const VerifierParams& vp, const std::vector<VerifyClaim>& claims, const BasefoldProof& proof, Transcript& t) {
  std::vector<Commitment> comms; std::vector<std::vector<Ext>> points; std::vector<VerifyEval> evals;
  for (size_t i = 0; i < claims.size(); i++) { comms.push_back(claims[i].comm); points.push_back(claims[i].point); evals.push_back({i, i, claims[i].eval}); }
  pcs_batch_verify_evals(vp, comms, points, evals, proof, t);
}

}  // namespace dp
