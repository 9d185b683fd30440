// Host orchestration of IOPProverState::prove_parallel (sumcheck/src/prover.rs:498-585) over device-resident tables,
// and IOPVerifierState::verify (sumcheck/src/verifier.rs:12-168). The transcript stays on the host (as in the
// reference); per round exactly one device->host message of raw per-term sums and one challenge back.
#pragma once
#include "proof.h"
#include <chrono>

namespace dp {

struct DevVP {  // VirtualPolynomial (multilinear_extensions/src/virtual_poly.rs:50-60) with tables in HBM
  unsigned nv = 0, max_degree = 0;
  std::vector<DBuf> tabs;
  std::vector<Ext> coeffs;
  std::vector<ScTerm> terms;
  explicit DevVP(unsigned nv_) : nv(nv_) {}
  int table_index(const DBuf& b) {
    for (size_t i = 0; i < tabs.size(); i++) if (tabs[i].p == b.p && tabs[i].n == b.n) return (int)i;  // de-dup by identity
    tabs.push_back(b);
    return (int)tabs.size() - 1;
  }
  void add_mle_list(std::initializer_list<DBuf> list, Ext coeff) { add_mle_list(std::vector<DBuf>(list), coeff); }
  void add_mle_list(const std::vector<DBuf>& list, Ext coeff) {
    DP_REQUIRE(list.size() >= 1 && list.size() <= (size_t)SC_MAXK, DP_ERR_SHAPE, "sumcheck term degree must be 1..5");
    ScTerm t; t.k = (int)list.size(); for (int q = 0; q < SC_MAXK; q++) t.t[q] = 0;
    int j = 0;
    for (const DBuf& b : list) {
      // a table may have FEWER variables than the polynomial (virtual_poly.rs:147-180 only asserts num_vars <= max_num_variables);
      // the round-sum body indexes every table of a product with the first one's range (sumcheck_macro/src/lib.rs:228-235),
      // so the tables of one product share their length. A constant is refused as the reference refuses it (prover.rs:663).
      DP_REQUIRE(b.n >= 2 && (b.n & (b.n - 1)) == 0 && b.n <= (size_t(1) << nv), DP_ERR_SHAPE, "sumcheck: table length must be 2^k, 1 <= k <= max_num_variables");
      DP_REQUIRE(b.n == list[0].n, DP_ERR_SHAPE, "sumcheck: the tables of one product must have the same number of variables");
      t.t[j++] = table_index(b);
    }
    if ((unsigned)t.k > max_degree) max_degree = t.k;
    terms.push_back(t);
    coeffs.push_back(coeff);
  }
};

// value at `at` of the polynomial with values evals[i] at i = 0..n-1 (what `extrapolate`, util.rs:101-136, computes)
inline Ext lagrange_eval_small(const Ext* evals, size_t n, Ext at) {
  Ext res = ex_zero();
  for (size_t i = 0; i < n; i++) {
    Ext num = ex_one(), den = ex_one();
    for (size_t j = 0; j < n; j++) {
      if (j == i) continue;
      num = ex_mul(num, ex_sub(at, ex_from_u64(j)));
      den = ex_mul(den, ex_sub(ex_from_u64(i), ex_from_u64(j)));
    }
    res = ex_add(res, ex_mul(evals[i], ex_mul(num, ex_inv(den))));
  }
  return res;
}
// the prover only ever extrapolates from nodes 0..k (k < SC_MAXK) to the integer points k+1..SC_MAXK: those Lagrange
// coefficients are constants, computed once
struct ExtrapolationTable {
  u64 c[SC_MAXK + 1][SC_MAXK + 1][SC_MAXK + 1];
  ExtrapolationTable() {
    for (unsigned kk = 1; kk < (unsigned)SC_MAXK; kk++)
      for (unsigned a = kk + 1; a <= (unsigned)SC_MAXK; a++)
        for (unsigned i = 0; i <= kk; i++) {
          u64 num = 1, den = 1;
          for (unsigned j = 0; j <= kk; j++) { if (j == i) continue; num = gl_mul(num, gl_sub(a, j)); den = gl_mul(den, gl_sub(i, j)); }
          c[kk][a][i] = gl_mul(num, gl_inv(den));
        }
  }
};
inline const u64* extrapolation_coeffs(unsigned k, unsigned at) {  // returns k+1 base-field coefficients
  static const ExtrapolationTable table;  // thread-safe initialisation
  return table.c[k][at];
}
inline Ext extrapolate_small(const Ext* evals, unsigned k, unsigned at) {
  const u64* c = extrapolation_coeffs(k, at);
  Ext r = ex_zero();
  for (unsigned i = 0; i <= k; i++) r = ex_add(r, ex_mul_base(evals[i], c[i]));
  return r;
}

// A table of k < nv variables inside a polynomial of nv variables is f(x_1..x_k), constant in x_(k+1)..x_nv. The reference
// handles it inside the round function: the table is folded like any other while it has variables, a one-element table is a
// constant factor, and the round sums are scaled by 2^(nv - (max(log2 len, 1) + round - 1)) (sumcheck_macro/src/lib.rs:236-247).
// That is exactly the sumcheck of the table TILED to 2^nv entries (index i -> i mod 2^k): while round <= k the tiled sums are
// 2^(nv-k) copies of the short ones, afterwards the folded tile is the constant f(r_1..r_k) summed over the 2^(nv-round)
// remaining points, and the final evaluation is that constant. Field arithmetic is exact, so the messages are bit-identical;
// the device keeps ONE code path (equal-length tables) and pays a copy for a shape zkml's own layers never produce.
inline DBuf tile_to(Dev& dev, const DBuf& b, size_t n) {
  if (b.n == n) return b;
  DBuf out = dev.alloc(n, b.ext);
  dev.copy(out.slice(0, b.n), b);
  for (size_t have = b.n; have < n; have *= 2) dev.copy(out.slice(have, have), out.slice(0, have));
  return out;
}
struct SumcheckOut { IOPProof proof; std::vector<Ext> finals; };
struct ScStats { double dev_ms = 0, host_ms = 0; size_t rounds = 0; };
inline ScStats& sc_stats() { static thread_local ScStats s; return s; }

inline SumcheckOut sumcheck_prove(Dev& dev, DevVP& vp, Transcript& t) {
  SumcheckOut out;
  unsigned nv = vp.nv, md = vp.max_degree;
  DP_REQUIRE(nv > 0, DP_ERR_SHAPE, "sumcheck over a constant");
  size_t mk = dev.mark();
  t.append_usize(nv);
  t.append_usize(md);
  std::vector<DBuf> tabs = vp.tabs;
  for (DBuf& b : tabs) b = tile_to(dev, b, size_t(1) << nv);
  size_t nraw = 0;
  for (auto& tm : vp.terms) nraw += tm.k + 1;
  std::vector<Ext> raw(nraw);
  Ext ch = ex_zero();
  const bool single = vp.terms.size() == 1 && ex_eq(vp.coeffs[0], ex_one()) && (unsigned)vp.terms[0].k == md;
  out.finals.resize(tabs.size());
  bool tail = false;
  for (unsigned round = 0; round < nv; round++) {
    auto tq0 = std::chrono::steady_clock::now();
    // a device that keeps the sponge to itself runs every remaining round (and the final evaluations) in one go
    if (dev.sc_tail(tabs.data(), (int)tabs.size(), round ? &ch : nullptr, vp.terms.data(), vp.coeffs.data(), (int)vp.terms.size(), md, t.challenger(),
                    out.proof.proofs, out.proof.point, out.finals.data())) {
      ScStats& st = sc_stats();
      st.dev_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
      st.rounds += nv - round; tail = true;
      break;
    }
    // one product with coefficient one (the shape of the large standalone sumchecks): the round must sum to the previous
    // round polynomial at its challenge, which lets the device skip one of its evaluation points
    if (single && round) {
      Ext claim = lagrange_eval_small(out.proof.proofs.back().data(), md + 1, ch);
      dev.sc_round_claim(tabs.data(), (int)tabs.size(), &ch, vp.terms.data(), 1, &claim, raw.data());
    } else
      dev.sc_round(tabs.data(), (int)tabs.size(), round ? &ch : nullptr, vp.terms.data(), (int)vp.terms.size(), raw.data());
    auto tq1 = std::chrono::steady_clock::now();
    std::vector<Ext> msg(md + 1, ex_zero());
    size_t off = 0;
    for (size_t ti = 0; ti < vp.terms.size(); ti++) {
      unsigned k = vp.terms[ti].k;
      std::vector<Ext> s(k + 1);
      for (unsigned j = 0; j <= k; j++) s[j] = ex_mul(raw[off + j], vp.coeffs[ti]);
      off += k + 1;
      for (unsigned j = 0; j <= md; j++) {
        Ext v = j <= k ? s[j] : extrapolate_small(s.data(), k, j);
        msg[j] = ex_add(msg[j], v);
      }
    }
    for (const Ext& e : msg) t.append_ext(e);
    out.proof.proofs.push_back(msg);
    ch = t.get_and_append_challenge("Internal round");
    out.proof.point.push_back(ch);
    auto tq2 = std::chrono::steady_clock::now();
    ScStats& st = sc_stats();
    st.dev_ms += std::chrono::duration<double, std::milli>(tq1 - tq0).count();
    st.host_ms += std::chrono::duration<double, std::milli>(tq2 - tq1).count();
    st.rounds++;
  }
  if (!tail) dev.sc_finish(tabs.data(), (int)tabs.size(), ch, out.finals.data());
  dev.release(mk);
  return out;
}

// eq tables (out[idx] (+)= scale * eq(idx, pt), in order) followed by the sumcheck of a virtual polynomial that contains them:
// the shape of the accumulation sumchecks of Requant and same_poly. A device that keeps the sponge to itself does both in one
// go (Dev::eqsum_tail); otherwise the tables are built one by one and sumcheck_prove runs.
struct EqAcc { DBuf out; std::vector<Ext> pt; Ext scale; bool accumulate; };
inline SumcheckOut sumcheck_prove_with_eq(Dev& dev, const std::vector<EqAcc>& eqs, DevVP& vp, Transcript& t) {
  std::vector<Dev::EqAccJob> jobs;
  for (const EqAcc& e : eqs) jobs.push_back({e.out, e.pt.data(), (unsigned)e.pt.size(), e.scale, e.accumulate});
  Dev::EqSumOut eo;
  bool full = true; for (const DBuf& b : vp.tabs) full = full && b.n == (size_t(1) << vp.nv);
  if (full && dev.eqsum_tail(jobs.data(), (int)jobs.size(), vp.tabs.data(), (int)vp.tabs.size(), vp.terms.data(), vp.coeffs.data(), (int)vp.terms.size(), vp.nv, vp.max_degree, t.challenger(), eo)) {
    DP_REQUIRE(eo.msgs.size() == vp.nv && eo.point.size() == vp.nv && eo.finals.size() == vp.tabs.size(), DP_ERR_SHAPE, "eqsum_tail: unexpected result shape");
    SumcheckOut out; out.proof.proofs = eo.msgs; out.proof.point = eo.point; out.finals = eo.finals;
    return out;
  }
  for (const Dev::EqAccJob& j : jobs) dev.eq_table(j.out, j.pt, j.k, j.scale, j.accumulate);
  return sumcheck_prove(dev, vp, t);
}

// interpolate_uni_poly (sumcheck/src/util.rs:148-195) == Lagrange evaluation on nodes 0..len-1
struct SubClaim { std::vector<Ext> point; Ext expected_evaluation; };
inline SubClaim sumcheck_verify(Ext claimed_sum, const IOPProof& proof, unsigned nv, unsigned max_degree, Transcript& t) {
  SubClaim sc;
  if (nv == 0) { sc.expected_evaluation = claimed_sum; return sc; }
  t.append_usize(nv);
  t.append_usize(max_degree);
  DP_REQUIRE(proof.proofs.size() >= nv, DP_ERR_VERIFY, "sumcheck proof is incomplete");
  for (unsigned i = 0; i < nv; i++) {
    for (const Ext& e : proof.proofs[i]) t.append_ext(e);
    sc.point.push_back(t.get_and_append_challenge("Internal round"));
  }
  Ext expected = claimed_sum;
  for (unsigned i = 0; i < nv; i++) {
    const auto& ev = proof.proofs[i];
    DP_REQUIRE(ev.size() == max_degree + 1, DP_ERR_VERIFY, "sumcheck: incorrect number of evaluations");
    DP_REQUIRE(ex_eq(ex_add(ev[0], ev[1]), expected), DP_ERR_VERIFY, "sumcheck: round message inconsistent with the claim");
    expected = lagrange_eval_small(ev.data(), ev.size(), sc.point[i]);
  }
  // IOPProof.point is prover data. The reference verifier ignores it (sumcheck/src/verifier.rs:22-110) while its callers
  // read it back (logup_gkr/verifier.rs:81, convolution.rs:1150-1386, same_poly.rs:175): an honest prover always sends the
  // Fiat-Shamir challenges there, so requiring equality changes no proof byte and closes the gap for every caller at once.
  DP_REQUIRE(proof.point.size() == nv, DP_ERR_VERIFY, "sumcheck: proof point length");
  for (unsigned i = 0; i < nv; i++) DP_REQUIRE(ex_eq(proof.point[i], sc.point[i]), DP_ERR_VERIFY, "sumcheck: proof point differs from the transcript challenges");
  sc.expected_evaluation = expected;
  return sc;
}

// host-side small helpers shared by provers and verifiers
inline std::vector<Ext> host_eq_table(const std::vector<Ext>& r) {  // build_eq_x_r_vec / compute_betas_eval
  std::vector<Ext> buf(size_t(1) << r.size());
  buf[0] = ex_one();
  size_t cur = 1;
  for (size_t t = r.size(); t-- > 0;) {
    for (size_t j = cur; j-- > 0;) {
      Ext prod = ex_mul(r[t], buf[j]);
      buf[2 * j + 1] = prod;
      buf[2 * j] = ex_sub(buf[j], prod);
    }
    cur *= 2;
  }
  return buf;
}
inline Ext eq_eval(const Ext* x, const Ext* y, size_t n) {  // virtual_poly.rs:308-322 / commit/mod.rs:41-53
  Ext res = ex_one();
  for (size_t i = 0; i < n; i++) {
    Ext xy = ex_mul(x[i], y[i]);
    res = ex_mul(res, ex_add(ex_sub(ex_sub(ex_dbl(xy), x[i]), y[i]), ex_one()));
  }
  return res;
}
inline Ext identity_eval(const std::vector<Ext>& a, const std::vector<Ext>& b) {
  size_t n = a.size() < b.size() ? a.size() : b.size();
  return eq_eval(a.data(), b.data(), n);
}
// evaluate a small host-resident MLE (verifier side: model inputs/outputs, trivial openings)
inline Ext host_mle_eval(std::vector<Ext> v, const std::vector<Ext>& pt) {
  DP_REQUIRE(v.size() == (size_t(1) << pt.size()), DP_ERR_SHAPE, "MLE size does not match the point");
  for (const Ext& r : pt) {
    size_t h = v.size() / 2;
    for (size_t i = 0; i < h; i++) v[i] = ex_lerp(v[2 * i], v[2 * i + 1], r);
    v.resize(h);
  }
  return v[0];
}

}  // namespace dp
