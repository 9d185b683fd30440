// TEST DOUBLE — lives under tests/, is never compiled into the product library.
// A plain-loop CPU implementation of dp::Dev so that the host orchestrator (transcript order, claim routing, proof
// assembly) can be byte-compared against the oracle in the `-m "not gpu"` suite, where no MI355X is present.
// Every method is the literal spec of the corresponding HIP kernel family in deep-prove_amd/csrc/hip_dev.hip.
#pragma once
#include "../../deep-prove_amd/csrc/dev.h"
#include "../../deep-prove_amd/csrc/sumcheck.h"
#include "../../deep-prove_amd/csrc/logup.h"
#include "../../deep-prove_amd/csrc/pcs.h"
#include <cstdlib>
#include <cstring>

namespace dp {

class CpuDev : public Dev {
  std::vector<void*> arena_;
 protected:
  unsigned full_log_ = 0;
 private:
  static u64* B(const DBuf& b) { return (u64*)b.p; }
  static Ext* X(const DBuf& b) { return (Ext*)b.p; }
  static Ext at(const DBuf& b, size_t i) { return b.ext ? X(b)[i] : ex_base(B(b)[i]); }
 public:
  ~CpuDev() { release(0); }
  const char* name() const override { return "cpu-test-double"; }
  DBuf alloc(size_t n, bool ext) override {
    DBuf b; b.n = n; b.ext = ext; b.p = aligned_alloc(16, std::max<size_t>(16, n * (ext ? 16 : 8))); arena_.push_back(b.p); return b;
  }
  size_t mark() override { return arena_.size(); }
  void release(size_t m) override { while (arena_.size() > m) { free(arena_.back()); arena_.pop_back(); } }
  DBuf alloc_persistent(size_t n, bool ext) override { DBuf b; b.n = n; b.ext = ext; b.p = aligned_alloc(16, std::max<size_t>(16, n * (ext ? 16 : 8))); return b; }
  void free_persistent(DBuf& b) override { if (b.p) free(b.p); b.p = nullptr; }
  void upload(const DBuf& d, const u64* s) override { memcpy(d.p, s, d.bytes()); }
  void upload_i64(const DBuf& d, const int64_t* s) override { for (size_t i = 0; i < d.n; i++) B(d)[i] = gl_from_i64(s[i]); }
  void download(const DBuf& s, u64* d) override { memcpy(d, s.p, s.bytes()); }
  void copy(const DBuf& d, const DBuf& s) override { memcpy(d.p, s.p, s.bytes()); }
  void zero(const DBuf& d) override { memset(d.p, 0, d.bytes()); }
  void sync() override {}
  void eq_table(const DBuf& out, const Ext* pt, unsigned k, Ext scale, bool acc) override {
    for (size_t i = 0; i < (size_t(1) << k); i++) {
      Ext v = scale;
      for (unsigned t = 0; t < k; t++) v = ex_mul(v, ((i >> t) & 1) ? pt[t] : ex_sub(ex_one(), pt[t]));
      X(out)[i] = acc ? ex_add(X(out)[i], v) : v;
    }
  }
  void mle_eval_batch(const DBuf* fs, int nf, const Ext* pt, unsigned k, Ext* out) override {
    DBuf eq = alloc(size_t(1) << k, true);
    eq_table(eq, pt, k, ex_one(), false);
    for (int f = 0; f < nf; f++) {
      Ext s = ex_zero();
      for (size_t i = 0; i < fs[f].n; i++) s = ex_add(s, ex_mul(X(eq)[i], at(fs[f], i)));
      out[f] = s;
    }
    release(mark() - 1);
  }
  void fix_high(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) override {
    unsigned k = dp_ceil_log2(R);
    DBuf eq = alloc(R, true);
    eq_table(eq, pt, k, ex_one(), false);
    for (size_t c = 0; c < C; c++) {
      Ext s = ex_zero();
      for (size_t r = 0; r < R; r++) s = ex_add(s, ex_mul_base(X(eq)[r], B(W)[r * C + c]));
      X(out)[c] = s;
    }
    release(mark() - 1);
  }
  // (Dev::fix_low: the double keeps the interface's default — one MLE evaluation per row — so that path is exercised too)
  DBuf fold(const DBuf& in, Ext r) {
    DBuf o = alloc(in.n / 2, true);
    for (size_t i = 0; i < in.n / 2; i++) X(o)[i] = in.ext ? ex_lerp(X(in)[2 * i], X(in)[2 * i + 1], r) : ex_lerp_base(B(in)[2 * i], B(in)[2 * i + 1], r);
    return o;
  }
  void sc_round(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, int nterms, Ext* out) override {
    if (r) for (int i = 0; i < nt; i++) tabs[i] = fold(tabs[i], *r);
    size_t o = 0;
    for (int ti = 0; ti < nterms; ti++) {
      int k = terms[ti].k;
      for (int t = 0; t <= k; t++) out[o + t] = ex_zero();
      size_t n = tabs[terms[ti].t[0]].n;
      for (size_t b = 0; b + 1 < n; b += 2)
        for (int t = 0; t <= k; t++) {
          Ext p = ex_one();
          for (int j = 0; j < k; j++) { Ext a = at(tabs[terms[ti].t[j]], b), bb = at(tabs[terms[ti].t[j]], b + 1); p = ex_mul(p, ex_add(a, ex_mul(ex_from_u64(t), ex_sub(bb, a)))); }
          out[o + t] = ex_add(out[o + t], p);
        }
      o += k + 1;
    }
  }
  void sc_finish(DBuf* tabs, int nt, Ext r, Ext* fin) override { for (int i = 0; i < nt; i++) { tabs[i] = fold(tabs[i], r); fin[i] = X(tabs[i])[0]; } }
  // The contract of Dev::sc_tail (device-side Fiat-Shamir, HipDev's throughput mode), spelled out on the CPU: from the sponge
  // the host hands over, run every remaining round — raw sums, coefficients and extrapolation to max_degree + 1 evaluations,
  // absorb, squeeze "Internal round", fold — and hand the sponge back. Off unless device_fs is set (DP_DOUBLE_DEVICE_FS=1 in
  // the harness); like the device it declines tables outside a size window, so both host paths get exercised in one proof.
  bool device_fs = false;
  size_t tails_taken = 0, tails_declined = 0;
  bool sc_tail(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned md, Challenger& ch,
               std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& point, Ext* finals) override {
    if (!device_fs) return false;
    size_t n_after = r ? tabs[0].n / 2 : tabs[0].n;
    if (n_after < 4 || n_after > 256) { tails_declined++; return false; }
    tails_taken++;
    Transcript t("");  // only a carrier for the sponge: its own state is replaced by the caller's
    t.challenger() = ch;
    Ext chal = r ? *r : ex_zero();
    bool have = r != nullptr;
    size_t nraw = 0; for (int i = 0; i < nterms; i++) nraw += terms[i].k + 1;
    std::vector<Ext> raw(nraw);
    for (size_t m = n_after; m > 1; m >>= 1) {
      sc_round(tabs, nt, have ? &chal : nullptr, terms, nterms, raw.data());
      std::vector<Ext> msg(md + 1, ex_zero());
      size_t off = 0;
      for (int ti = 0; ti < nterms; ti++) {
        unsigned k = terms[ti].k;
        std::vector<Ext> s(k + 1);
        for (unsigned j = 0; j <= k; j++) s[j] = ex_mul(raw[off + j], coeffs[ti]);
        off += k + 1;
        for (unsigned j = 0; j <= md; j++) msg[j] = ex_add(msg[j], j <= k ? s[j] : extrapolate_small(s.data(), k, j));
      }
      for (const Ext& e : msg) t.append_ext(e);
      chal = t.get_and_append_challenge("Internal round");
      have = true;
      msgs.push_back(msg); point.push_back(chal);
    }
    sc_finish(tabs, nt, chal, finals);
    ch = t.challenger();
    return true;
  }
  // The contract of Dev::logup_tail: the whole layer loop of a logup-GKR proof on a private transcript that starts from the
  // host's sponge and is handed back at the end (device_logup, DP_DOUBLE_DEVICE_LOGUP=1 in the harness).
  bool device_logup = false;
  size_t logup_tails = 0;
  bool logup_tail(const LogupTailArgs& a, Challenger& ch, std::vector<std::vector<std::vector<Ext>>>& layer_msgs,
                  std::vector<std::vector<Ext>>& layer_points, std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) override {
    if (!device_logup) return false;
    logup_tails++;
    Transcript t("");
    t.challenger() = ch;
    std::vector<IOPProof> proofs;
    point = logup_layers(*this, *a.circuits, a.initial_lookup, a.is_table, a.total_layers, a.batching, a.alpha, a.lambda, a.claim, t, proofs, round_evals);
    for (auto& p : proofs) { layer_msgs.push_back(p.proofs); layer_points.push_back(p.point); }
    ch = t.challenger();
    return true;
  }
  // The contract of Dev::logup_full: the whole of logup_batch_prove on a private transcript seeded with the host's sponge
  // (device_logup_full, DP_DOUBLE_DEVICE_LOGUP=2 in the harness).
  bool device_logup_full = false;
  bool in_full_ = false;
  size_t logup_fulls = 0;
  bool logup_full(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, Challenger& ch, LogupFullOut& out) override {
    if (!device_logup_full || in_full_) return false;
    in_full_ = true;
    LogUpInputDev in;
    in.is_table = !mult.null(); in.multiplicities = mult; in.columns_per_instance = (size_t)cpi;
    in.columns.assign(cols, cols + (size_t)cpi * ninst);
    in.constant_challenge = c; in.column_separation_challenge = chi;
    Transcript t("");
    t.challenger() = ch;
    LogUpProof p = logup_batch_prove(*this, in, t);
    in_full_ = false;
    for (auto& o : p.circuit_outputs) for (const Ext& e : o) out.outputs.push_back(e);
    for (auto& sp : p.sumcheck_proofs) { out.layer_msgs.push_back(sp.proofs); out.layer_points.push_back(sp.point); }
    out.round_evals = p.round_evaluations;
    out.point = p.output_claims[0].point;
    for (auto& oc : p.output_claims) out.col_evals.push_back(oc.eval);
    ch = t.challenger();
    logup_fulls++;
    return true;
  }
  void logup_den(const DBuf& out, const DBuf* cols, int nc, Ext c, Ext chi) override {
    for (size_t i = 0; i < out.n; i++) { Ext acc = c, pw = ex_one(); for (int j = 0; j < nc; j++) { acc = ex_add(acc, ex_mul_base(pw, B(cols[j])[i])); pw = ex_mul(pw, chi); } X(out)[i] = acc; }
  }
  void logup_layer(const DBuf& ni, const DBuf& di, const DBuf& no, const DBuf& dout) override {
    size_t h = di.n / 2;
    Ext m1 = ex_neg(ex_one());
    for (size_t i = 0; i < h; i++) {
      Ext n1 = ni.null() ? m1 : at(ni, i), n2 = ni.null() ? m1 : at(ni, i + h), d1 = X(di)[i], d2 = X(di)[i + h];
      X(no)[i] = ex_add(ex_mul(n1, d2), ex_mul(d1, n2));
      X(dout)[i] = ex_mul(d1, d2);
    }
  }
  void pcs_init(unsigned fl) override { full_log_ = fl; }
  void bitrev_copy(const DBuf& d, const DBuf& s) override {
    unsigned lg = dp_ceil_log2(s.n);
    for (size_t i = 0; i < s.n; i++) { size_t j = dp_reverse_bits(i, lg); if (s.ext) X(d)[j] = X(s)[i]; else B(d)[j] = B(s)[i]; }
  }
  DevTree build_tree(const DBuf& leaves, bool persistent) {
    DevTree t; t.leaves = leaves; t.nleaves = leaves.n;
    size_t n = leaves.n;
    t.nodes = persistent ? alloc_persistent(4 * (n - 1), false) : alloc(4 * (n - 1), false);
    u64* nd = B(t.nodes);
    for (size_t i = 0; i < n / 2; i++) {
      u64* d = nd + 4 * i;
      if (leaves.ext) { d[0] = X(leaves)[2 * i].c0; d[1] = X(leaves)[2 * i].c1; d[2] = X(leaves)[2 * i + 1].c0; d[3] = X(leaves)[2 * i + 1].c1; }
      else { d[0] = B(leaves)[2 * i]; d[1] = B(leaves)[2 * i + 1]; d[2] = 0; d[3] = 0; }
    }
    size_t off = 0, cnt = n / 2;
    while (cnt > 1) {
      for (size_t i = 0; i < cnt / 2; i++) poseidon2_compress(nd + 4 * (off + 2 * i), nd + 4 * (off + 2 * i + 1), nd + 4 * (off + cnt + i), POSEIDON2_RC_HOST);
      off += cnt; cnt /= 2;
    }
    for (int k = 0; k < 4; k++) t.root.v[k] = nd[4 * (n - 2) + k];
    return t;
  }
  DevCommit commit(const DBuf& evals, bool persistent) override {
    DevCommit c; c.nv = dp_ceil_log2(evals.n); c.is_base = !evals.ext; c.evals = evals;
    auto A = [&](size_t n, bool e) { return persistent ? alloc_persistent(n, e) : alloc(n, e); };
    if (c.nv <= 7) { c.bh_evals = evals; c.tree = build_tree(evals, persistent); return c; }
    size_t n = evals.n;
    DBuf co = alloc(n, evals.ext); copy(co, evals);
    for (unsigned i = 1; i <= c.nv; i++) {  // interpolate_over_boolean_hypercube
      size_t chunk = size_t(1) << i, half = chunk >> 1;
      for (size_t s = 0; s < n; s += chunk) for (size_t j = half; j < chunk; j++) { if (co.ext) X(co)[s + j] = ex_sub(X(co)[s + j], X(co)[s + j - half]); else B(co)[s + j] = gl_sub(B(co)[s + j], B(co)[s + j - half]); }
    }
    // RS encode of the bit-reversed coefficient vector on the coset shift*H, |H| = 2n; then bit-reverse the codeword.
    u64 shift = GL_GENERATOR; for (unsigned i = 0; i < full_log_ - c.nv; i++) shift = gl_sqr(shift);
    u64 w = GL_G32; for (unsigned i = c.nv + 1; i < 32; i++) w = gl_sqr(w);
    DBuf cw = A(2 * n, evals.ext);
    // naive-but-exact DFT is O(n^2); use an in-place radix-2 on a scratch copy instead
    std::vector<Ext> a(2 * n, ex_zero());
    for (size_t i = 0; i < n; i++) { size_t j = dp_reverse_bits(i, c.nv); a[j] = ex_mul_base(at(co, i), gl_pow(shift, j)); }
    // iterative DIT FFT over base-field roots
    unsigned lg = c.nv + 1; size_t N = 2 * n;
    for (size_t i = 0; i < N; i++) { size_t j = dp_reverse_bits(i, lg); if (i < j) std::swap(a[i], a[j]); }
    for (unsigned s = 1; s <= lg; s++) {
      size_t m = size_t(1) << s, hm = m >> 1; u64 wm = gl_pow(w, N / m);
      for (size_t k = 0; k < N; k += m) { u64 tw = 1; for (size_t j = 0; j < hm; j++) { Ext t = ex_mul_base(a[k + j + hm], tw), u = a[k + j]; a[k + j] = ex_add(u, t); a[k + j + hm] = ex_sub(u, t); tw = gl_mul(tw, wm); } }
    }
    for (size_t i = 0; i < N; i++) { size_t j = dp_reverse_bits(i, lg); if (cw.ext) X(cw)[j] = a[i]; else B(cw)[j] = a[i].c0; }
    c.bh_evals = A(n, evals.ext); bitrev_copy(c.bh_evals, evals);
    c.tree = build_tree(cw, persistent);
    return c;
  }
  void free_commit(DevCommit& c) override {
    if (c.bh_evals.p && c.bh_evals.p != c.evals.p) free_persistent(c.bh_evals);
    if (c.tree.leaves.p && c.tree.leaves.p != c.evals.p) free_persistent(c.tree.leaves);
    free_persistent(c.tree.nodes); free_persistent(c.evals);
  }
  DevTree merkle_ext(const DBuf& leaves) override { return build_tree(leaves, false); }
  // Dev::batch_tree: row hashes by the host sponge word by word, then the ordinary tree over them
  DevTree batch_tree(const DBuf* cws, int k, bool persistent) override {
    const size_t n = cws[0].n;
    DBuf rows = persistent ? alloc_persistent(2 * n, true) : alloc(2 * n, true);
    for (size_t j = 0; j < n; j++) {
      std::vector<u64> w;
      for (int q = 0; q < k; q++) { if (cws[q].ext) { w.push_back(X(cws[q])[j].c0); w.push_back(X(cws[q])[j].c1); } else w.push_back(B(cws[q])[j]); }
      u64 d[4] = {0, 0, 0, 0};
      if (w.size() <= 4) for (size_t i = 0; i < w.size(); i++) d[i] = w[i];
      else { Challenger ch; for (u64 x : w) ch.observe(x); for (int i = 0; i < 4; i++) d[i] = ch.sample(); }
      X(rows)[2 * j] = ex(d[0], d[1]); X(rows)[2 * j + 1] = ex(d[2], d[3]);
    }
    return build_tree(rows, persistent);
  }
  // factored eq tables (Dev::classic_round): this double factors every polynomial of 3 or more variables and materialises only at
  // two entries, so the batch openings of the CPU suite run their rounds on (lo, hi) pairs of every shape
  unsigned classic_eq_split(unsigned nv) override { return nv >= 3 ? nv / 2 : 0; }
  size_t classic_eq_materialise_n() override { return 2; }
  void eq_outer_many(const EqOuterJob* jobs, size_t n) override {
    for (size_t q = 0; q < n; q++) {
      const EqOuterJob& j = jobs[q];
      DP_REQUIRE(j.out.ext && j.lo.ext && j.hi.ext && j.lo.n && j.out.n == j.lo.n * j.hi.n, DP_ERR_SHAPE, "eq_outer_many: shapes");
      for (size_t i = 0; i < j.out.n; i++) X(j.out)[i] = ex_mul(X(j.lo)[i % j.lo.n], X(j.hi)[i / j.lo.n]);
    }
  }
  void classic_round(DBuf* fs, DBuf* eqs, DBuf* los, int np, const Ext* r, Ext* out) override {
    for (int i = 0; i < np; i++) {
      const bool fac = los && los[i].n;
      DP_REQUIRE(fs[i].n == (fac ? los[i].n : 1) * eqs[i].n, DP_ERR_SHAPE, "classic_round: f/eq shapes");
      if (r && fs[i].n > 1) {
        fs[i] = fold(fs[i], *r);
        if (fac && los[i].n > 1) los[i] = fold(los[i], *r); else eqs[i] = fold(eqs[i], *r);
      }
      auto eq_at = [&](size_t j) { return fac ? ex_mul(at(los[i], j % los[i].n), at(eqs[i], j / los[i].n)) : at(eqs[i], j); };
      Ext c0 = ex_zero(), c2 = ex_zero();
      if (fs[i].n == 1) c0 = ex_mul(at(fs[i], 0), eq_at(0));
      else for (size_t j = 0; j + 1 < fs[i].n; j += 2) {
        Ext l0 = eq_at(j), l1 = eq_at(j + 1), r0 = at(fs[i], j), r1 = at(fs[i], j + 1);
        c0 = ex_add(c0, ex_mul(l0, r0)); c2 = ex_add(c2, ex_mul(ex_sub(l1, l0), ex_sub(r1, r0)));
      }
      out[2 * i] = c0; out[2 * i + 1] = c2;
    }
  }
  // The contract of Dev::commit_tail: the remaining commit-phase rounds (commit_rounds of pcs.h) on a private transcript seeded
  // with the host's sponge (device_commit, DP_DOUBLE_DEVICE_COMMIT=1 in the harness); like a device, only for short oracles
  bool device_commit = false;
  size_t commit_tails = 0;
  bool commit_tail(const CommitTailArgs& a, Challenger& ch, CommitTailOut& out) override {
    if (!device_commit || a.folded.n > 1024) return false;
    commit_tails++;
    std::vector<std::vector<AxpyJob>> merges(1);
    merges.insert(merges.end(), a.merges->begin(), a.merges->end());
    CommitLoopState st;
    st.last.assign(a.last, a.last + 3); st.folded = a.folded; st.eq = a.eq; st.sum_evals = a.sum_evals;
    Transcript t("");
    t.challenger() = ch;
    std::vector<Digest> roots;
    std::vector<DevTree> trees;
    commit_rounds(*this, merges, a.rounds_left + 1, 1, false, st, t, out.msgs, roots, trees, out.final_message);
    out.trees.assign(trees.begin() + 1, trees.end());  // (the first entry is the caller's pending tree, pushed by round "1")
    ch = t.challenger();
    return true;
  }
  // The contract of Dev::eqsum_tail: the eq tables, then the whole sumcheck on a private transcript seeded with the host's sponge
  // (device_eqsum, DP_DOUBLE_DEVICE_EQSUM=1 in the harness)
  bool device_eqsum = false;
  size_t eqsum_tails = 0;
  bool eqsum_tail(const EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned nv, unsigned md,
                  Challenger& ch, EqSumOut& out) override {
    if (!device_eqsum) return false;
    eqsum_tails++;
    for (int j = 0; j < njobs; j++) eq_table(jobs[j].out, jobs[j].pt, jobs[j].k, jobs[j].scale, jobs[j].accumulate);
    DevVP vp(nv);
    vp.tabs.assign(tabs, tabs + ntabs); vp.terms.assign(terms, terms + nterms); vp.coeffs.assign(coeffs, coeffs + nterms); vp.max_degree = md;
    Transcript t("");
    t.challenger() = ch;
    SumcheckOut sc = sumcheck_prove(*this, vp, t);
    out.msgs = sc.proof.proofs; out.point = sc.proof.point; out.finals = sc.finals;
    ch = t.challenger();
    return true;
  }
  // The contract of Dev::dense_tail: bias evaluation, fix_high and the dense sumcheck on a private transcript seeded with the
  // host's sponge (device_dense, DP_DOUBLE_DEVICE_DENSE=1 in the harness)
  bool device_dense = false;
  size_t dense_tails = 0;
  bool dense_tail(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, Challenger& ch, DenseTailOut& out) override {
    if (!device_dense) return false;
    dense_tails++;
    size_t mk = mark();
    mle_eval_batch(&bias, 1, pt, dp_ceil_log2(R), &out.bias_eval);
    DBuf mat = alloc(C, true);
    fix_high(mat, W, R, C, pt);
    DevVP vp(dp_ceil_log2(C));
    vp.add_mle_list({mat, in}, ex_one());
    Transcript t("");
    t.challenger() = ch;
    SumcheckOut sc = sumcheck_prove(*this, vp, t);
    out.msgs = sc.proof.proofs; out.point = sc.proof.point; out.finals[0] = sc.finals[0]; out.finals[1] = sc.finals[1];
    ch = t.challenger();
    release(mk);
    return true;
  }
  // The contract of Dev::classic_tail: the remaining rounds of the batch-opening sumcheck on a private transcript seeded with
  // the host's sponge (device_classic, DP_DOUBLE_DEVICE_CLASSIC=1 in the harness); like a device it only takes over once the
  // tables are small
  bool device_classic = false;
  size_t classic_tails = 0;
  bool classic_tail(const ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) override {
    if (!device_classic) return false;
    size_t maxn = 0; for (int i = 0; i < a.np; i++) maxn = std::max(maxn, a.fs[i].n);
    if (maxn > 512) return false;
    classic_tails++;
    Transcript t("");
    t.challenger() = ch;
    std::vector<Ext> raw(2 * (size_t)a.np);
    Ext sum = a.sum, c = a.r ? *a.r : ex_zero();
    for (unsigned round = a.round; round < a.num_vars; round++) {
      classic_round(a.fs, a.eqs, a.los, a.np, round == a.round ? a.r : &c, raw.data());
      std::vector<Ext> msg = classic_round_message(raw.data(), a.fs, a.eq_xt, (size_t)a.np, a.num_vars, round, sum);
      for (const Ext& e : msg) t.append_ext(e);
      c = t.get_and_append_challenge("sumcheck round");
      challenges.push_back(c);
      sum = ex_add(msg[0], ex_mul(c, ex_add(msg[1], ex_mul(c, msg[2]))));
      msgs.push_back(msg);
    }
    ch = t.challenger();
    return true;
  }
  void axpy_rep(const DBuf& acc, const DBuf& x, Ext coeff, size_t rep) override {
    for (size_t j = 0; j < x.n; j++) { Ext m = ex_mul(at(x, j), coeff); for (size_t q = 0; q < rep; q++) X(acc)[j * rep + q] = ex_add(X(acc)[j * rep + q], m); }
  }
  void bf_round(DBuf& eq, DBuf& f, const Ext* ch, Ext* msg) override {
    if (ch) { eq = fold(eq, *ch); f = fold(f, *ch); }
    if (!msg) return;
    Ext c0 = ex_zero(), c1 = ex_zero(), c2 = ex_zero();
    if (f.n == 1) { msg[0] = msg[1] = msg[2] = X(f)[0]; return; }
    for (size_t j = 0; j + 1 < f.n; j += 2) {
      Ext a = X(f)[j], b = ex_sub(X(f)[j + 1], a), ea = X(eq)[j], eb = ex_sub(X(eq)[j + 1], ea);
      c0 = ex_add(c0, ex_mul(a, ea)); c1 = ex_add(c1, ex_add(ex_mul(b, ea), ex_mul(a, eb))); c2 = ex_add(c2, ex_mul(b, eb));
    }
    msg[0] = c0; msg[1] = c1; msg[2] = c2;
  }
  DBuf fri_fold(const DBuf& o, unsigned level, Ext ch) override {
    DBuf out = alloc(o.n / 2, true);
    u64 g = GL_G32; for (unsigned i = level + 1; i < 32; i++) g = gl_sqr(g);
    u64 gam = GL_GENERATOR; for (unsigned i = 0; i < full_log_ + 1 - level - 1; i++) gam = gl_sqr(gam);
    for (size_t i = 0; i < o.n / 2; i++) {
      u64 x0 = gl_mul(gl_pow(g, dp_reverse_bits(i, level)), gam), w = gl_neg(gl_inv(gl_dbl(x0)));
      Ext y0 = X(o)[2 * i], y1 = X(o)[2 * i + 1];
      X(out)[i] = ex_add(y0, ex_mul(ex_mul(ex_sub(ch, ex_base(x0)), ex_sub(y1, y0)), ex_base(w)));
    }
    return out;
  }
  void query_gather(const QueryDesc* d, size_t nd, std::vector<std::vector<u64>>& out) override {
    out.resize(nd);
    for (size_t i = 0; i < nd; i++) {
      const DevTree& t = *d[i].tree; std::vector<u64>& w = out[i]; w.clear();
      size_t p0 = d[i].p0;
      if (t.leaves.ext) { Ext a = X(t.leaves)[p0], b = X(t.leaves)[p0 + 1]; w = {a.c0, a.c1, b.c0, b.c1}; } else w = {B(t.leaves)[p0], B(t.leaves)[p0 + 1]};
      unsigned h = t.height();
      for (unsigned l = 0; l + 1 < h; l++) { size_t off = t.nleaves - (t.nleaves >> l); size_t idx = (p0 >> (l + 1)) ^ 1; const u64* dg = B(t.nodes) + 4 * (off + idx); for (int k = 0; k < 4; k++) w.push_back(dg[k]); }
    }
  }
};

}  // namespace dp
