// logup-GKR batch prover (zkml/src/lookup/logup_gkr/prover.rs:24-237, circuit.rs:49-270) over device tables, and
// the verifier (verifier.rs:16-211).
#pragma once
#include "sumcheck.h"

namespace dp {

struct LogUpInputDev {
  bool is_table = false;
  std::vector<DBuf> columns;  // base-field columns, all of length n
  DBuf multiplicities;        // base, table variant only
  Ext constant_challenge, column_separation_challenge;
  size_t columns_per_instance = 1;
};

// The layer loop of batch_prove (prover.rs:84-198): for every tree layer, bottom-up from the 2-element outputs, absorb the
// running claim, prove the batched layer sumcheck, draw (batching, alpha, lambda) and form the next claim. Returns the final
// point. Shared by logup_batch_prove and by devices / test doubles that implement Dev::logup_tail.
inline std::vector<Ext> logup_layers(Dev& dev, const std::vector<LogupCircuitDev>& circuits, bool initial_lookup, bool is_table, unsigned total_layers,
                                     Ext batching, Ext alpha, Ext lambda, Ext current_claim, Transcript& t,
                                     std::vector<IOPProof>& sumcheck_proofs, std::vector<std::vector<Ext>>& round_evaluations) {
  std::vector<Ext> point = {batching};
  for (unsigned lv = 1; lv <= total_layers; lv++) {
    t.append_ext(current_claim);
    size_t mk2 = dev.mark();
    size_t half = size_t(1) << lv;
    DBuf eq = dev.alloc(half, true);
    dev.eq_table_lazy(eq, point.data(), lv);
    DevVP vp(lv);
    Ext cur_alpha = ex_one();
    for (auto& c : circuits) {
      size_t li = c.den.size() - 1 - lv;  // layers().iter().rev().skip(1)
      DBuf dlo = c.den[li].slice(0, half), dhi = c.den[li].slice(half, half);
      bool init_lk = initial_lookup && li == 0;
      if (!init_lk) {
        DBuf nlo = c.num[li].slice(0, half), nhi = c.num[li].slice(half, half);
        vp.add_mle_list({eq, nlo, dhi}, cur_alpha);
        vp.add_mle_list({eq, nhi, dlo}, cur_alpha);
        vp.add_mle_list({eq, dlo, dhi}, ex_mul(cur_alpha, lambda));
      } else {
        vp.add_mle_list({eq, dhi}, ex_neg(cur_alpha));
        vp.add_mle_list({eq, dlo}, ex_neg(cur_alpha));
        vp.add_mle_list({eq, dlo, dhi}, ex_mul(cur_alpha, lambda));
      }
      cur_alpha = ex_mul(cur_alpha, alpha);
    }
    SumcheckOut sc = sumcheck_prove(dev, vp, t);
    dev.release(mk2);
    point = sc.proof.point;
    std::vector<Ext> evals(sc.finals.begin() + 1, sc.finals.end());
    batching = t.get_and_append_challenge("logup_batching");
    alpha = t.get_and_append_challenge("logup_alpha");
    lambda = t.get_and_append_challenge("logup_lambda");
    point.push_back(batching);
    sumcheck_proofs.push_back(sc.proof);
    Ext acc = ex_zero(), acomb = ex_one();
    bool lookup_final = (lv == total_layers) && !is_table;  // final_round_claim (prover.rs:201-237)
    if (!lookup_final) {
      for (size_t k = 0; k + 3 < evals.size(); k += 4) {
        const Ext* e = &evals[k];
        Ext a = ex_add(ex_mul(batching, ex_sub(e[2], e[0])), e[0]);
        Ext b = ex_add(ex_mul(batching, ex_sub(e[1], e[3])), e[3]);
        acc = ex_add(acc, ex_mul(acomb, ex_add(a, ex_mul(lambda, b))));
        acomb = ex_mul(acomb, alpha);
      }
    } else {
      for (size_t k = 0; k + 1 < evals.size(); k += 2) {
        const Ext* e = &evals[k];
        acc = ex_add(acc, ex_mul(acomb, ex_add(ex_mul(batching, ex_sub(e[0], e[1])), e[1])));
        acomb = ex_mul(acomb, alpha);
      }
    }
    current_claim = acc;
    round_evaluations.push_back(evals);
  }
  return point;
}

inline LogUpProof logup_batch_prove(Dev& dev, const LogUpInputDev& in, Transcript& t) {
  size_t mk = dev.mark();
  DP_REQUIRE(!in.columns.empty(), DP_ERR_ARG, "logup: no columns");
  size_t n = in.columns[0].n;
  unsigned nvars = dp_ceil_log2(n);
  DP_REQUIRE((size_t(1) << nvars) == n && n >= 4, DP_ERR_SHAPE, "logup: column length must be a power of two >= 4");
  for (auto& c : in.columns) DP_REQUIRE(c.n == n && !c.ext, DP_ERR_SHAPE, "logup: columns must be base field of equal length");
  // ---- build the fractional-sum trees (one per instance); layer j has length n >> j
  size_t cpi = in.is_table ? in.columns.size() : in.columns_per_instance;
  DP_REQUIRE(in.columns.size() % cpi == 0, DP_ERR_SHAPE, "logup: column count must be a multiple of columns_per_instance");
  int ninst = (int)(in.columns.size() / cpi);
  {
    // a device that keeps the sponge to itself proves the whole argument in one go (Dev::logup_full)
    Dev::LogupFullOut fo;
    if (dev.logup_full(in.columns.data(), (int)cpi, ninst, in.is_table ? in.multiplicities : DBuf(), in.constant_challenge, in.column_separation_challenge, t.challenger(), fo)) {
      const size_t nbase = in.columns.size() + (in.is_table ? 1 : 0);
      DP_REQUIRE(fo.outputs.size() == 4 * (size_t)ninst && fo.layer_msgs.size() == nvars - 1 && fo.layer_points.size() == nvars - 1 && fo.round_evals.size() == nvars - 1 &&
                 fo.point.size() == nvars && fo.col_evals.size() == nbase, DP_ERR_SHAPE, "logup_full: unexpected result shape");
      LogUpProof proof; proof.is_table = in.is_table;
      for (int i = 0; i < ninst; i++) proof.circuit_outputs.push_back({fo.outputs[4 * i], fo.outputs[4 * i + 1], fo.outputs[4 * i + 2], fo.outputs[4 * i + 3]});
      for (unsigned l = 0; l + 1 < nvars; l++) { IOPProof ip; ip.point = fo.layer_points[l]; ip.proofs = fo.layer_msgs[l]; proof.sumcheck_proofs.push_back(ip); proof.round_evaluations.push_back(fo.round_evals[l]); }
      for (size_t i = 0; i < nbase; i++) proof.output_claims.push_back({fo.point, fo.col_evals[i]});
      dev.release(mk);
      return proof;
    }
  }
  std::vector<LogupCircuitDev> circuits;
  std::vector<Ext> outs;
  dev.logup_build(in.columns.data(), (int)cpi, ninst, in.is_table ? in.multiplicities : DBuf(), in.constant_challenge,
                  in.column_separation_challenge, circuits, outs);
  const bool initial_lookup = !in.is_table;
  unsigned total_layers = nvars - 1;
  LogUpProof proof; proof.is_table = in.is_table;
  for (int i = 0; i < ninst; i++) proof.circuit_outputs.push_back({outs[4 * i], outs[4 * i + 1], outs[4 * i + 2], outs[4 * i + 3]});
  t.append_field_element(gl_from_u64(circuits.size()));
  for (auto& ev : proof.circuit_outputs) t.append_exts(ev);
  Ext batching = t.get_and_append_challenge("initial_batching");
  Ext alpha = t.get_and_append_challenge("initial_alpha");
  Ext lambda = t.get_and_append_challenge("initial_lambda");
  Ext current_claim = ex_zero(), ac = ex_one();
  for (auto& e : proof.circuit_outputs) {
    Ext a = ex_add(ex_mul(batching, ex_sub(e[1], e[0])), e[0]);
    Ext b = ex_add(ex_mul(batching, ex_sub(e[3], e[2])), e[2]);
    current_claim = ex_add(current_claim, ex_mul(ac, ex_add(a, ex_mul(lambda, b))));
    ac = ex_mul(ac, alpha);
  }
  std::vector<Ext> point;
  {
    // a device that keeps the sponge to itself runs the whole layer loop (Dev::logup_tail); otherwise layer by layer here
    Dev::LogupTailArgs ta{&circuits, initial_lookup, in.is_table, total_layers, batching, alpha, lambda, current_claim};
    std::vector<std::vector<std::vector<Ext>>> lmsgs; std::vector<std::vector<Ext>> lpoints, levals;
    if (dev.logup_tail(ta, t.challenger(), lmsgs, lpoints, levals, point)) {
      DP_REQUIRE(lmsgs.size() == total_layers && lpoints.size() == total_layers && levals.size() == total_layers, DP_ERR_SHAPE, "logup_tail: one entry per layer expected");
      for (unsigned l = 0; l < total_layers; l++) { IOPProof ip; ip.point = lpoints[l]; ip.proofs = lmsgs[l]; proof.sumcheck_proofs.push_back(ip); proof.round_evaluations.push_back(levals[l]); }
    } else
      point = logup_layers(dev, circuits, initial_lookup, in.is_table, total_layers, batching, alpha, lambda, current_claim, t, proof.sumcheck_proofs, proof.round_evaluations);
  }
  std::vector<DBuf> base;
  if (in.is_table) base.push_back(in.multiplicities);
  for (auto& c : in.columns) base.push_back(c);
  std::vector<Ext> ev(base.size());
  dev.mle_eval_batch(base.data(), (int)base.size(), point.data(), (unsigned)point.size(), ev.data());
  for (size_t i = 0; i < base.size(); i++) proof.output_claims.push_back({point, ev[i]});
  dev.release(mk);
  return proof;
}

struct LogUpVerifierClaim { std::vector<Claim> claims; std::vector<Ext> numerators, denominators; };

// verify_logup_proof (verifier.rs:16-211). Throws DpError(DP_ERR_VERIFY) on rejection.
// `expect_table`: 1 / 0 when the caller knows which kind of proof belongs here (a layer's lookup proof vs a table proof:
// the kind is the verifier's knowledge, not the prover's), -1 to take the proof's own tag as the reference does.
inline LogUpVerifierClaim verify_logup_proof(const LogUpProof& proof, size_t num_instances, Ext constant_challenge,
                                             Ext column_separation_challenge, Transcript& t, int expect_table = -1) {
  DP_REQUIRE(expect_table < 0 || proof.is_table == (expect_table != 0), DP_ERR_VERIFY, "logup: proof kind (lookup / table) does not match its place in the proof");
  DP_REQUIRE(num_instances > 0 && proof.circuit_outputs.size() == num_instances, DP_ERR_VERIFY, "logup: wrong number of instances");
  t.append_field_element(gl_from_u64(num_instances));
  LogUpVerifierClaim out;
  for (auto& e : proof.circuit_outputs) {
    DP_REQUIRE(e.size() == 4, DP_ERR_VERIFY, "logup: circuit output must hold 4 values");
    t.append_exts(e);
    out.numerators.push_back(ex_add(ex_mul(e[0], e[3]), ex_mul(e[1], e[2])));  // fractional_outputs (structs.rs:328-339)
    out.denominators.push_back(ex_mul(e[2], e[3]));
  }
  Ext batching = t.get_and_append_challenge("initial_batching");
  Ext alpha = t.get_and_append_challenge("initial_alpha");
  Ext lambda = t.get_and_append_challenge("initial_lambda");
  Ext current_claim = ex_zero(), ac = ex_one();
  for (auto& e : proof.circuit_outputs) {
    Ext a = ex_add(ex_mul(batching, ex_sub(e[1], e[0])), e[0]);
    Ext b = ex_add(ex_mul(batching, ex_sub(e[3], e[2])), e[2]);
    current_claim = ex_add(current_claim, ex_mul(ac, ex_add(a, ex_mul(lambda, b))));
    ac = ex_mul(ac, alpha);
  }
  std::vector<Ext> point = {batching};
  DP_REQUIRE(proof.sumcheck_proofs.size() == proof.round_evaluations.size(), DP_ERR_VERIFY, "logup: proofs/evals length mismatch");
  for (size_t i = 0; i < proof.sumcheck_proofs.size(); i++) {
    const IOPProof& sp = proof.sumcheck_proofs[i];
    const std::vector<Ext>& re = proof.round_evaluations[i];
    t.append_ext(current_claim);
    Ext eq_ev = identity_eval(point, sp.point);
    SubClaim sub = sumcheck_verify(current_claim, sp, (unsigned)(i + 1), 3, t);
    Ext nb = t.get_and_append_challenge("logup_batching");
    Ext next_alpha = t.get_and_append_challenge("logup_alpha");
    Ext next_lambda = t.get_and_append_challenge("logup_lambda");
    size_t per = re.size() / num_instances;
    DP_REQUIRE((per == 4 || per == 2) && per * num_instances == re.size(), DP_ERR_VERIFY, "logup: bad number of round evaluations");
    Ext next_claim = ex_zero(), nac = ex_one(), sc_claim = ex_zero(), pa = ex_one();
    for (size_t k = 0; k < re.size(); k += per) {
      const Ext* e = &re[k];
      if (per == 4) {
        Ext a = ex_add(ex_mul(nb, ex_sub(e[2], e[0])), e[0]);
        Ext b = ex_add(ex_mul(nb, ex_sub(e[1], e[3])), e[3]);
        next_claim = ex_add(next_claim, ex_mul(nac, ex_add(a, ex_mul(next_lambda, b))));
        Ext inner = ex_add(ex_add(ex_mul(e[0], e[1]), ex_mul(e[2], e[3])), ex_mul(ex_mul(lambda, e[3]), e[1]));
        sc_claim = ex_add(sc_claim, ex_mul(pa, ex_mul(eq_ev, inner)));
      } else {
        next_claim = ex_add(next_claim, ex_mul(nac, ex_add(ex_mul(nb, ex_sub(e[0], e[1])), e[1])));
        Ext inner = ex_add(ex_sub(ex_neg(e[1]), e[0]), ex_mul(ex_mul(lambda, e[0]), e[1]));
        sc_claim = ex_add(sc_claim, ex_mul(ex_mul(pa, eq_ev), inner));
      }
      nac = ex_mul(nac, next_alpha);
      pa = ex_mul(pa, alpha);
    }
    DP_REQUIRE(ex_eq(sc_claim, sub.expected_evaluation), DP_ERR_VERIFY, "logup: sumcheck output claim mismatch");
    current_claim = next_claim;
    alpha = next_alpha; lambda = next_lambda;
    point = sub.point;
    point.push_back(nb);
  }
  // calculate_final_eval (verifier.rs:163-211)
  Ext calc;
  if (!proof.is_table) {
    DP_REQUIRE(proof.output_claims.size() % num_instances == 0 && !proof.output_claims.empty(), DP_ERR_VERIFY, "logup: bad output claims");
    size_t per = proof.output_claims.size() / num_instances;
    Ext acc = ex_zero(), acomb = ex_one();
    for (size_t s = 0; s < proof.output_claims.size(); s += per) {
      Ext ce = constant_challenge, csc = ex_one();
      for (size_t j = 0; j < per; j++) { ce = ex_add(ce, ex_mul(proof.output_claims[s + j].eval, csc)); csc = ex_mul(csc, column_separation_challenge); }
      acc = ex_add(acc, ex_mul(ce, acomb));
      acomb = ex_mul(acomb, alpha);
    }
    calc = acc;
  } else {
    DP_REQUIRE(proof.output_claims.size() >= 2, DP_ERR_VERIFY, "logup: table proof needs >= 2 output claims");
    Ext ce = constant_challenge, csc = ex_one();
    for (size_t j = 1; j < proof.output_claims.size(); j++) { ce = ex_add(ce, ex_mul(proof.output_claims[j].eval, csc)); csc = ex_mul(csc, column_separation_challenge); }
    calc = ex_add(proof.output_claims[0].eval, ex_mul(lambda, ce));
  }
  DP_REQUIRE(ex_eq(calc, current_claim), DP_ERR_VERIFY, "logup: final evaluation mismatch");
  // the output claims are claims AT THE VERIFIER'S final point. The reference hands the prover-supplied points on (verifier.rs:160-164) —
  // an honest prover's are this point, so checking costs no proof byte and leaves no word of the stream unbound
  for (const Claim& c : proof.output_claims) {
    DP_REQUIRE(c.point.size() == point.size(), DP_ERR_VERIFY, "logup: output claim at a point other than the verifier's");
    for (size_t i = 0; i < point.size(); i++) DP_REQUIRE(ex_eq(c.point[i], point[i]), DP_ERR_VERIFY, "logup: output claim at a point other than the verifier's");
  }
  out.claims = proof.output_claims;
  return out;
}

}  // namespace dp
