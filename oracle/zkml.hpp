// ORACLE (test infrastructure only).
// Restates the zkml proving pipeline for the Dense / Requant / ReLU graph (BASELINE.json configs 1,2,4):
//   lookup/logup_gkr/{circuit,prover,structs}.rs, lookup/{context,witness}.rs, commit/{context,same_poly,mod}.rs,
//   layers/{dense,requant,activation}.rs, iop/{context,prover}.rs, quantization/mod.rs (constants, Fieldizer).
// The model front-end (ONNX parsing, float quantisation, padding) is out of scope: a model is given as already
// padded, already quantised i64 tensors, laid out like Model::random_with_rng (zkml/src/model/mod.rs:596-665):
// consecutive node ids Dense, Requant, Activation(Relu), Dense, ...
#pragma once
#include "basefold.hpp"
#include "sumcheck.hpp"
#include <map>
#include <unordered_map>
#include <algorithm>
#include <thread>
#include <string>

namespace orc {

constexpr unsigned BIT_LEN = 8;               // quantization/mod.rs:20-25
constexpr int64_t QMIN = -127, QMAX = 127;    // quantization/mod.rs:28-29
constexpr int64_t COLUMN_SEPARATOR = int64_t(1) << 32;  // lookup/context.rs:622

struct Claim { std::vector<E> point; E eval; };

// ------------------------------------------------------------------ logup GKR
struct LogUpInput {
  bool is_table = false;
  std::vector<std::vector<u64>> column_evals;
  std::vector<u64> multiplicities;
  E constant_challenge, column_separation_challenge;
  size_t columns_per_instance = 1;
};
struct LogUpProof {
  std::vector<IOPProof> sumcheck_proofs;
  std::vector<std::vector<E>> round_evaluations;
  std::vector<Claim> output_claims;
  std::vector<std::vector<E>> circuit_outputs;
  bool is_table = false;
};
struct LogUpLayer {  // circuit.rs:15-30
  int kind;  // 0 Generic, 1 InitialTable, 2 InitialLookup
  std::vector<E> num, den;
  unsigned num_vars() const { return ceil_log2(std::max<size_t>(den.size() >> 1, 1)); }
};
static inline bool logup_next_layer(const LogUpLayer& l, LogUpLayer& out) {  // circuit.rs:49-100
  if (l.num_vars() == 0) return false;
  size_t half = size_t(1) << l.num_vars();
  out.kind = 0;
  out.num.resize(half); out.den.resize(half);
  for (size_t i = 0; i < half; i++) {
    E n1 = l.kind == 2 ? eneg(e_one()) : l.num[i], n2 = l.kind == 2 ? eneg(e_one()) : l.num[i + half];
    E d1 = l.den[i], d2 = l.den[i + half];
    out.num[i] = eadd(emul(n1, d2), emul(d1, n2));  // structs.rs:44-54
    out.den[i] = emul(d1, d2);
  }
  return true;
}
static inline std::vector<LogUpLayer> logup_circuit(LogUpLayer init) {
  std::vector<LogUpLayer> layers;
  layers.push_back(std::move(init));
  for (;;) { LogUpLayer nx; if (!logup_next_layer(layers.back(), nx)) break; layers.push_back(std::move(nx)); }
  return layers;
}
static inline std::vector<E> logup_denominators(const std::vector<const std::vector<u64>*>& cols, E c, E chi) {
  std::vector<E> pw; E p = e_one();
  for (size_t j = 0; j < cols.size(); j++) { pw.push_back(p); p = emul(p, chi); }
  size_t n = cols[0]->size();
  std::vector<E> den(n);
  for (size_t i = 0; i < n; i++) { E acc = c; for (size_t j = 0; j < cols.size(); j++) acc = eadd(acc, emul_base(pw[j], (*cols[j])[i])); den[i] = acc; }
  return den;
}
// batch_prove (logup_gkr/prover.rs:24-237)
static inline LogUpProof logup_batch_prove(const LogUpInput& in, Transcript& t) {
  std::vector<std::vector<LogUpLayer>> circuits;
  if (in.is_table) {
    std::vector<const std::vector<u64>*> cols; for (auto& c : in.column_evals) cols.push_back(&c);
    LogUpLayer init; init.kind = 1;
    for (u64 m : in.multiplicities) init.num.push_back(e_from(m));
    init.den = logup_denominators(cols, in.constant_challenge, in.column_separation_challenge);
    circuits.push_back(logup_circuit(std::move(init)));
  } else {
    for (size_t s = 0; s < in.column_evals.size(); s += in.columns_per_instance) {
      std::vector<const std::vector<u64>*> cols;
      for (size_t j = s; j < std::min(s + in.columns_per_instance, in.column_evals.size()); j++) cols.push_back(&in.column_evals[j]);
      LogUpLayer init; init.kind = 2;
      init.den = logup_denominators(cols, in.constant_challenge, in.column_separation_challenge);
      circuits.push_back(logup_circuit(std::move(init)));
    }
  }
  size_t num_instances = circuits.size();
  LogUpProof proof; proof.is_table = in.is_table;
  unsigned total_layers = 0;
  for (auto& c : circuits) {
    total_layers = std::max(total_layers, c[0].num_vars());
    const LogUpLayer& last = c.back();
    std::vector<E> out = last.num; out.insert(out.end(), last.den.begin(), last.den.end());  // flat_evals
    if (last.kind == 2) out = last.den;
    proof.circuit_outputs.push_back(out);
  }
  t.append_field_element(from_u64(num_instances));
  for (auto& ev : proof.circuit_outputs) t.append_exts(ev);
  E batching = t.get_and_append_challenge("initial_batching");
  E alpha = t.get_and_append_challenge("initial_alpha");
  E lambda = t.get_and_append_challenge("initial_lambda");
  E current_claim = e_zero(), ac = e_one();
  for (auto& e : proof.circuit_outputs) {
    current_claim = eadd(current_claim, emul(ac, eadd(eadd(emul(batching, esub(e[1], e[0])), e[0]),
                                                        emul(lambda, eadd(emul(batching, esub(e[3], e[2])), e[2])))));
    ac = emul(ac, alpha);
  }
  std::vector<E> sumcheck_point = {batching};
  for (unsigned lv = 1; lv <= total_layers; lv++) {
    t.append_ext(current_claim);
    MleP eq_poly = mk(Mle::from_ext(compute_betas_eval(sumcheck_point)));
    VirtualPolynomial vp(lv);
    E cur_alpha = e_one();
    for (auto& c : circuits) {
      // layers().iter().rev().skip(1): the lv-th element from the top
      if (c.size() < size_t(lv) + 1) throw std::runtime_error("One of the circuits was not the same size as the others");
      const LogUpLayer& layer = c[c.size() - 1 - lv];
      unsigned nv = layer.num_vars(); size_t half = size_t(1) << nv;
      auto slice = [&](const std::vector<E>& v, bool hi) { return mk(Mle::from_ext(std::vector<E>(v.begin() + (hi ? half : 0), v.begin() + (hi ? 2 * half : half)))); };
      if (layer.kind != 2) {
        MleP nlo = slice(layer.num, false), nhi = slice(layer.num, true), dlo = slice(layer.den, false), dhi = slice(layer.den, true);
        vp.add_mle_list({eq_poly, nlo, dhi}, cur_alpha);
        vp.add_mle_list({eq_poly, nhi, dlo}, cur_alpha);
        vp.add_mle_list({eq_poly, dlo, dhi}, emul(cur_alpha, lambda));
      } else {
        MleP dlo = slice(layer.den, false), dhi = slice(layer.den, true);
        vp.add_mle_list({eq_poly, dhi}, eneg(cur_alpha));
        vp.add_mle_list({eq_poly, dlo}, eneg(cur_alpha));
        vp.add_mle_list({eq_poly, dlo, dhi}, emul(cur_alpha, lambda));
      }
      cur_alpha = emul(cur_alpha, alpha);
    }
    auto [sproof, state] = sumcheck_prove(std::move(vp), t);
    sumcheck_point = sproof.point;
    std::vector<E> fin = state.final_evaluations();
    std::vector<E> evals(fin.begin() + 1, fin.end());
    batching = t.get_and_append_challenge("logup_batching");
    alpha = t.get_and_append_challenge("logup_alpha");
    lambda = t.get_and_append_challenge("logup_lambda");
    sumcheck_point.push_back(batching);
    proof.sumcheck_proofs.push_back(sproof);
    E acc = e_zero(); E acomb = e_one();
    if (lv != total_layers || in.is_table) {
      for (size_t k = 0; k + 3 < evals.size() + 0 && k < evals.size(); k += 4) {
        const E* e = &evals[k];
        acc = eadd(acc, emul(acomb, eadd(eadd(emul(batching, esub(e[2], e[0])), e[0]),
                                           emul(lambda, eadd(emul(batching, esub(e[1], e[3])), e[3])))));
        acomb = emul(acomb, alpha);
      }
    } else {
      for (size_t k = 0; k < evals.size(); k += 2) {
        const E* e = &evals[k];
        acc = eadd(acc, emul(acomb, eadd(emul(batching, esub(e[0], e[1])), e[1])));
        acomb = emul(acomb, alpha);
      }
    }
    current_claim = acc;
    proof.round_evaluations.push_back(evals);
  }
  // output claims on every base column (multiplicities first for tables)
  std::vector<const std::vector<u64>*> base;
  if (in.is_table) base.push_back(&in.multiplicities);
  for (auto& c : in.column_evals) base.push_back(&c);
  for (auto* col : base) proof.output_claims.push_back({sumcheck_point, Mle::from_base(*col).evaluate(sumcheck_point)});
  return proof;
}

// ------------------------------------------------------------------ model description
enum LayerKind { L_DENSE = 0, L_REQUANT = 1, L_RELU = 2 };
struct Layer {
  LayerKind kind;
  size_t nrows = 0, ncols = 0;        // dense (padded to powers of two)
  std::vector<int64_t> weights, bias;  // row major; bias padded to nrows
  unsigned right_shift = 0, fp_scale = 0, intermediate_bit_size = 0;  // requant (requant.rs:46-73)
  int64_t fixed_point_multiplier = 0;
  unsigned shift() const { return fp_scale + right_shift; }
  unsigned clamping_size() const { return intermediate_bit_size + ceil_log2((size_t)fixed_point_multiplier) - shift(); }  // requant.rs:485-488
};
struct Model { size_t input_len = 0; std::vector<Layer> layers; };

struct TableType {  // lookup/context.rs:55-72 (derive Ord: Relu < GELU < Range < Clamping(n) < ...)
  int kind;  // 0 Relu, 2 Range, 3 Clamping
  unsigned size;
  bool operator<(const TableType& o) const { return kind != o.kind ? kind < o.kind : size < o.size; }
  bool operator==(const TableType& o) const { return kind == o.kind && size == o.size; }
  unsigned multiplicity_poly_vars() const { return kind == 3 ? size : BIT_LEN; }
  const char* challenge_label() const { return kind == 0 ? "Relu" : kind == 3 ? "Clamping" : nullptr; }
};
static inline int64_t relu_apply(int64_t x) { return x < 0 ? 0 : x; }
static inline int64_t clamp_q(int64_t x) { return x < QMIN ? QMIN : x > QMAX ? QMAX : x; }
// get_merged_table_column (lookup/context.rs:158-296)
static inline void table_columns(const TableType& tt, std::vector<int64_t>& merged, std::vector<std::vector<u64>>& cols) {
  merged.clear(); cols.clear();
  if (tt.kind == 0) {
    cols.resize(2);
    for (int64_t i = QMIN - 1; i <= QMAX; i++) { int64_t o = relu_apply(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  } else if (tt.kind == 2) {
    cols.resize(1);
    for (int64_t i = 0; i < (int64_t(1) << BIT_LEN); i++) { merged.push_back(i); cols[0].push_back(from_i64(i)); }
  } else {
    cols.resize(2);
    int64_t mx = int64_t(1) << (tt.size - 1);
    for (int64_t i = -mx; i < mx; i++) { int64_t o = clamp_q(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  }
}

// ------------------------------------------------------------------ inference (layers' Evaluate impls)
struct Trace { std::vector<std::vector<int64_t>> in, out; };  // per node input / output tensors
static inline int64_t requant_apply(const Layer& l, int64_t v) {
  unsigned sh = l.shift();
  int64_t tmp = v * l.fixed_point_multiplier + (int64_t(1) << (sh - 1));
  return clamp_q(tmp >> sh);
}
static inline Trace run_model(const Model& m, const std::vector<int64_t>& input) {
  Trace tr; std::vector<int64_t> cur = input;
  if (cur.size() != m.input_len) throw std::runtime_error("input length mismatch");
  for (auto& l : m.layers) {
    tr.in.push_back(cur);
    std::vector<int64_t> o;
    if (l.kind == L_DENSE) {
      if (cur.size() != l.ncols) throw std::runtime_error("dense input size mismatch");
      o.resize(l.nrows);
      for (size_t i = 0; i < l.nrows; i++) { int64_t a = 0; for (size_t j = 0; j < l.ncols; j++) a += l.weights[i * l.ncols + j] * cur[j]; o[i] = a + l.bias[i]; }
    } else if (l.kind == L_REQUANT) {
      for (int64_t v : cur) {
        if (std::llabs(v) > (int64_t(1) << l.intermediate_bit_size)) throw std::runtime_error("requant: value too large");
        o.push_back(requant_apply(l, v));
      }
    } else { for (int64_t v : cur) o.push_back(relu_apply(v)); }
    tr.out.push_back(o); cur = o;
  }
  return tr;
}

// ------------------------------------------------------------------ context (iop/context.rs:109-215, commit/context.rs:59-115)
using ProverCommitment = std::pair<CommitmentWithWitness, Mle>;
struct Context {
  Model model;
  PcsParams pp;
  std::map<size_t, std::map<std::string, ProverCommitment>> model_comms;  // BTreeMap<NodeId, BTreeMap<PolyId,..>>
  std::vector<TableType> tables;                                           // LookupContext (BTreeSet order)
  size_t max_poly_len = 0;
};
static inline size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
static inline Context context_generate(const Model& m) {
  Context ctx; ctx.model = m;
  size_t max_poly_len = m.input_len;
  std::vector<TableType> tset;
  auto add_table = [&](TableType t) { for (auto& x : tset) if (x == t) return; tset.push_back(t); };
  size_t cur_len = m.input_len;
  for (auto& l : m.layers) {
    if (l.kind == L_DENSE) { cur_len = l.nrows; }
    else if (l.kind == L_REQUANT) { add_table({2, 0}); add_table({3, l.clamping_size()}); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }
    else { add_table({0, 0}); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }
  }
  std::sort(tset.begin(), tset.end());
  for (auto& t : tset) max_poly_len = std::max(max_poly_len, size_t(1) << t.multiplicity_poly_vars());
  for (auto& l : m.layers) if (l.kind == L_DENSE) { max_poly_len = std::max(max_poly_len, next_pow2(l.weights.size())); max_poly_len = std::max(max_poly_len, next_pow2(l.bias.size())); }
  max_poly_len = next_pow2(max_poly_len);
  ctx.max_poly_len = max_poly_len;
  ctx.pp = pcs_setup(max_poly_len);
  // commit/context.rs:79-103 commits the model polynomials with `into_par_iter`; the oracle mirrors that with plain
  // threads (one per polynomial) — commitments are independent so the result does not depend on the schedule
  std::vector<std::pair<size_t, const char*>> jobs;
  for (size_t id = 0; id < m.layers.size(); id++) if (m.layers[id].kind == L_DENSE) { jobs.push_back({id, "DenseWeight"}); jobs.push_back({id, "DenseBias"}); }
  for (auto& j : jobs) ctx.model_comms[j.first][j.second];  // create map slots before the threads write into them
  std::vector<std::thread> th;
  for (auto& j : jobs) th.emplace_back([&ctx, &m, j] {
    const Layer& l = m.layers[j.first];
    Mle poly = Mle::from_i64(std::string(j.second) == "DenseWeight" ? l.weights : l.bias);
    ctx.model_comms[j.first][j.second] = {pcs_commit(ctx.pp, poly), poly};
  });
  for (auto& t : th) t.join();
  ctx.tables = tset;  // none of Relu/Range/Clamping has committed columns (lookup/context.rs:492-545)
  return ctx;
}

// ------------------------------------------------------------------ proofs
struct DenseProof { IOPProof sumcheck; E bias_eval; std::vector<E> individual_claims; };
struct SamePolyProof { IOPProof sumcheck; std::vector<E> evals; };
struct ActivationProof { SamePolyProof io_accumulation; LogUpProof lookup; std::vector<Commitment> commits; };
struct RequantProof { IOPProof io_accumulation; std::vector<E> accumulation_evals; LogUpProof clamping_lookup, shifted_lookup; std::vector<Commitment> commitments; };
struct LayerProof { LayerKind kind; DenseProof dense; ActivationProof act; RequantProof req; };
struct TableProof { Commitment multiplicity_commit; LogUpProof lookup; };
struct Proof {
  std::map<size_t, LayerProof> steps;  // canonical order: ascending NodeId
  std::vector<TableProof> table_proofs;
  BasefoldProof batch_proof;
  std::vector<BasefoldProof> trivial_proofs;
};

struct LogUpWitness {  // lookup/witness.rs:18-33
  bool is_table; std::vector<ProverCommitment> commits; std::vector<std::vector<u64>> column_evals;
  size_t columns_per_instance; TableType table_type; std::vector<u64> multiplicity_evals;
};
struct CommitClaim { ProverCommitment comm; Claim claim; };
struct ProverState {
  const Context* ctx;
  Transcript* t;
  std::map<size_t, LayerProof> proofs;
  std::vector<CommitClaim> claims, trivial_claims;
  std::map<size_t, std::vector<LogUpWitness>> lookup_witness;
  std::vector<LogUpWitness> table_witness;
  E constant_challenge; std::map<TableType, E> challenge_map;
  void add_witness_claim(const ProverCommitment& c, Claim cl) {  // commit/context.rs:287-306
    if (c.second.nv <= BASECODE_MSG_SIZE_LOG) trivial_claims.push_back({c, std::move(cl)}); else claims.push_back({c, std::move(cl)});
  }
  LogUpInput logup_input(const LogUpWitness& w) const {  // witness.rs:103-146
    LogUpInput in; in.is_table = w.is_table; in.column_evals = w.column_evals; in.multiplicities = w.multiplicity_evals;
    in.constant_challenge = constant_challenge; in.column_separation_challenge = challenge_map.at(w.table_type);
    in.columns_per_instance = w.columns_per_instance; return in;
  }
};

static inline std::vector<u64> to_base(const std::vector<int64_t>& v) { std::vector<u64> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = from_i64(v[i]); return o; }
static inline void count_into(std::unordered_map<int64_t, u64>& m, int64_t v) { m[v] += 1; }

// generate_lookup_witnesses (lookup/context.rs:631-781) + gen_lookup_witness of requant.rs:208-345, activation.rs:238-318
static inline void instantiate_witness_ctx(ProverState& ps, const Trace& tr) {
  const Context& ctx = *ps.ctx;
  if (ctx.tables.empty()) return;
  std::map<TableType, std::unordered_map<int64_t, u64>> element_count;
  for (size_t id = 0; id < ctx.model.layers.size(); id++) {
    const Layer& l = ctx.model.layers[id];
    if (l.kind == L_REQUANT) {
      unsigned shift = l.shift(); int64_t rounding = int64_t(1) << (shift - 1), mask = (int64_t(1) << shift) - 1;
      std::vector<int64_t> cin, cout, shifted;
      for (int64_t v : tr.in[id]) { int64_t tmp = v * l.fixed_point_multiplier + rounding; int64_t c = tmp >> shift; cin.push_back(c); cout.push_back(clamp_q(c)); shifted.push_back(tmp & mask); }
      unsigned nchunks = shift / BIT_LEN; int64_t rmask = (int64_t(1) << BIT_LEN) - 1;
      std::vector<std::vector<int64_t>> chunks(nchunks);
      for (unsigned j = 0; j < nchunks; j++) for (int64_t s : shifted) chunks[j].push_back((s >> (j * BIT_LEN)) & rmask);
      TableType ct{3, l.clamping_size()}, rt{2, 0};
      for (auto& ch : chunks) for (int64_t v : ch) count_into(element_count[rt], v);
      for (size_t i = 0; i < cin.size(); i++) count_into(element_count[ct], cin[i] + cout[i] * COLUMN_SEPARATOR);
      LogUpWitness wc; wc.is_table = false; wc.columns_per_instance = 2; wc.table_type = ct;
      for (auto* col : {&cin, &cout}) { std::vector<u64> ev = to_base(*col); Mle mle = Mle::from_base(ev); wc.commits.push_back({pcs_commit(ctx.pp, mle), mle}); wc.column_evals.push_back(ev); }
      LogUpWitness ws; ws.is_table = false; ws.columns_per_instance = 1; ws.table_type = rt;
      for (auto& ch : chunks) { std::vector<u64> ev = to_base(ch); Mle mle = Mle::from_base(ev); ws.commits.push_back({pcs_commit(ctx.pp, mle), mle}); ws.column_evals.push_back(ev); }
      ps.lookup_witness[id] = {wc, ws};
    } else if (l.kind == L_RELU) {
      TableType rt{0, 0};
      const auto& a = tr.in[id]; const auto& b = tr.out[id];
      for (size_t i = 0; i < a.size(); i++) count_into(element_count[rt], a[i] + COLUMN_SEPARATOR * b[i]);
      LogUpWitness w; w.is_table = false; w.columns_per_instance = 2; w.table_type = rt;
      for (auto* col : {&a, &b}) { std::vector<u64> ev = to_base(*col); Mle mle = Mle::from_base(ev); w.commits.push_back({pcs_commit(ctx.pp, mle), mle}); w.column_evals.push_back(ev); }
      ps.lookup_witness[id] = {w};
    }
  }
  for (auto& [tt, counts] : element_count) {
    std::vector<int64_t> merged; std::vector<std::vector<u64>> cols;
    table_columns(tt, merged, cols);
    std::map<int64_t, u64> table_count; for (int64_t v : merged) table_count[v] += 1;
    std::vector<u64> mult(merged.size());
    for (size_t i = 0; i < merged.size(); i++) {
      auto it = counts.find(merged[i]);
      if (it == counts.end()) { mult[i] = 0; continue; }
      u64 tc = table_count[merged[i]];
      u64 inv = tc != 1 ? finv(from_u64(tc)) : 1;
      mult[i] = fmul(from_u64(it->second), inv);
    }
    Mle mle = Mle::from_base(mult);
    LogUpWitness w; w.is_table = true; w.table_type = tt; w.multiplicity_evals = mult; w.column_evals = cols; w.columns_per_instance = cols.size();
    w.commits.push_back({pcs_commit(ctx.pp, mle), mle});
    ps.table_witness.push_back(std::move(w));
  }
  // initialise_from_table_set (lookup/context.rs:758-781)
  ps.constant_challenge = ps.t->get_and_append_challenge("table_constant");
  for (auto& [tt, _] : element_count) {
    const char* lab = tt.challenge_label();
    ps.challenge_map[tt] = lab ? ps.t->get_and_append_challenge(lab) : e_one();
  }
}

// Dense::prove_step (layers/dense.rs:423-561)
static inline Claim prove_dense(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& input) {
  if ((size_t(1) << last.point.size()) != l.nrows) throw std::runtime_error("dense: claim point size mismatch");
  E bias_eval = Mle::from_i64(l.bias).evaluate(last.point);
  std::vector<E> w(l.weights.size()); for (size_t i = 0; i < w.size(); i++) w[i] = e_from_i64(l.weights[i]);  // to_2d_mle
  Mle mat = Mle::from_ext(w);
  mat.fix_high_in_place(last.point);
  Mle in = Mle::from_ext(input);
  VirtualPolynomial vp(in.nv);
  vp.add_mle_list({mk(mat), mk(in)}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();
  std::vector<E> point = proof.point; point.insert(point.end(), last.point.begin(), last.point.end());
  // add_common_claims iterates the node's BTreeMap: "DenseBias" then "DenseWeight"
  const auto& comms = ps.ctx->model_comms.at(id);
  ps.add_witness_claim(comms.at("DenseBias"), {last.point, bias_eval});
  ps.add_witness_claim(comms.at("DenseWeight"), {point, fin[0]});
  LayerProof lp; lp.kind = L_DENSE; lp.dense = {proof, bias_eval, fin};
  ps.proofs[id] = lp;
  return {proof.point, fin[1]};
}
// Requant::recombine_claims (requant.rs:499-529)
static inline E recombine_claims(const Layer& l, E clamping_claim, const std::vector<E>& shifted) {
  E full = emul(e_from_u64(u64(1) << l.shift()), clamping_claim); E pw = e_one();
  for (E v : shifted) { full = eadd(full, emul(v, pw)); pw = emul(pw, e_from_u64(u64(1) << BIT_LEN)); }
  E rc = e_from_u64(u64(1) << (l.shift() - 1));
  return emul(esub(full, rc), einv(e_from_i64(l.fixed_point_multiplier)));
}
// Requant::prove_step (requant.rs:531-690)
static inline Claim prove_requant(ProverState& ps, size_t id, const Layer& l, const Claim& last) {
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  const LogUpWitness& clampw = ws[0]; const LogUpWitness& shiftw = ws[1];
  LogUpInput cin = ps.logup_input(clampw), sin = ps.logup_input(shiftw);
  LogUpProof cproof = logup_batch_prove(cin, *ps.t);
  LogUpProof sproof = logup_batch_prove(sin, *ps.t);
  unsigned nv = log2_strict(cin.column_evals[0].size());
  MleP clamp_in = mk(Mle::from_base(cin.column_evals[0])), clamp_out = mk(Mle::from_base(cin.column_evals[1]));
  std::vector<MleP> shifted; for (auto& c : sin.column_evals) shifted.push_back(mk(Mle::from_base(c)));
  MleP clamping_beta = mk(Mle::from_ext(compute_betas_eval(cproof.output_claims[0].point)));
  MleP last_beta = mk(Mle::from_ext(compute_betas_eval(last.point)));
  MleP shifted_beta = mk(Mle::from_ext(compute_betas_eval(sproof.output_claims[0].point)));
  E b = ps.t->get_and_append_challenge("requant_batching");
  VirtualPolynomial vp(nv);
  vp.add_mle_list({clamp_out, last_beta}, e_one());
  vp.add_mle_list({clamp_out, clamping_beta}, b);
  E comb = emul(b, b);
  vp.add_mle_list({clamp_in, clamping_beta}, comb);
  comb = emul(comb, b);
  for (auto& m : shifted) { vp.add_mle_list({shifted_beta, m}, comb); comb = emul(comb, b); }
  auto [acc_proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();
  std::vector<E> point = acc_proof.point;
  E clamping_out_eval = fin[0], clamping_in_eval = fin[3];
  std::vector<E> shifted_evals(fin.begin() + 5, fin.end());
  E combined = recombine_claims(l, clamping_in_eval, shifted_evals);
  RequantProof rp; rp.io_accumulation = acc_proof; rp.clamping_lookup = cproof; rp.shifted_lookup = sproof;
  std::vector<E> evs = {clamping_in_eval, clamping_out_eval}; evs.insert(evs.end(), shifted_evals.begin(), shifted_evals.end());
  std::vector<ProverCommitment> cm = clampw.commits; cm.insert(cm.end(), shiftw.commits.begin(), shiftw.commits.end());
  for (size_t i = 0; i < evs.size(); i++) { rp.commitments.push_back(cm[i].first.pure()); ps.add_witness_claim(cm[i], {point, evs[i]}); rp.accumulation_evals.push_back(evs[i]); }
  LayerProof lp; lp.kind = L_REQUANT; lp.req = rp; ps.proofs[id] = lp;
  return {point, combined};
}
// same_poly::Prover::prove (commit/same_poly.rs:88-122)
static inline SamePolyProof same_poly_prove(const std::vector<Claim>& claims, const Mle& poly, Transcript& t) {
  std::vector<E> ch = t.read_challenges(claims.size());
  std::vector<E> final_beta(size_t(1) << poly.nv, e_zero());
  for (size_t i = 0; i < claims.size(); i++) {
    if (claims[i].point.size() != poly.nv) throw std::runtime_error("same_poly: invalid claim length");
    std::vector<E> be = compute_betas_eval(claims[i].point);
    for (size_t j = 0; j < be.size(); j++) final_beta[j] = eadd(final_beta[j], emul(ch[i], be[j]));
  }
  VirtualPolynomial vp(poly.nv);
  vp.add_mle_list({mk(Mle::from_ext(final_beta)), mk(poly)}, e_one());
  auto [sp, st] = sumcheck_prove(std::move(vp), t);
  return {sp, st.final_evaluations()};
}
// Activation::prove_step (activation.rs:385-456), Relu only
static inline Claim prove_relu(ProverState& ps, size_t id, const Claim& last, const std::vector<E>& output) {
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  LogUpInput in = ps.logup_input(ws[0]);
  LogUpProof lproof = logup_batch_prove(in, *ps.t);
  Claim input_claim = lproof.output_claims[0], output_claim = lproof.output_claims[1];
  SamePolyProof sp = same_poly_prove({last, output_claim}, Mle::from_ext(output), *ps.t);
  ActivationProof ap; ap.io_accumulation = sp; ap.lookup = lproof;
  Claim c2{sp.sumcheck.point, sp.evals[1]};
  ps.add_witness_claim(ws[0].commits[0], input_claim); ap.commits.push_back(ws[0].commits[0].first.pure());
  ps.add_witness_claim(ws[0].commits[1], c2); ap.commits.push_back(ws[0].commits[1].first.pure());
  LayerProof lp; lp.kind = L_RELU; lp.act = ap; ps.proofs[id] = lp;
  return input_claim;
}

// Prover::prove (iop/prover.rs:401-488)
static inline Proof prove(const Context& ctx, const std::vector<int64_t>& input, Transcript& t, Trace* trace_out = nullptr) {
  ProverState ps; ps.ctx = &ctx; ps.t = &t;
  Trace tr = run_model(ctx.model, input);
  if (trace_out) *trace_out = tr;
  // ctx.write_to_transcript: every model commitment root, BTreeMap order (commit/context.rs:181-192)
  for (auto& [id, m] : ctx.model_comms) for (auto& [pid, pc] : m) t.append_digest(pc.first.codeword_tree.root());
  instantiate_witness_ctx(ps, tr);
  auto to_fields = [](const std::vector<int64_t>& v) { std::vector<E> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = e_from_i64(v[i]); return o; };
  const std::vector<int64_t>& out = tr.out.back();
  std::vector<E> r = t.read_challenges(log2_strict(out.size()));
  Claim cur{r, Mle::from_ext(to_fields(out)).evaluate(r)};
  for (size_t id = ctx.model.layers.size(); id-- > 0;) {
    const Layer& l = ctx.model.layers[id];
    if (l.kind == L_DENSE) cur = prove_dense(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_REQUANT) cur = prove_requant(ps, id, l, cur);
    else cur = prove_relu(ps, id, cur, to_fields(tr.out[id]));
  }
  Proof proof;
  // prove_tables (iop/prover.rs:110-157)
  for (auto& tw : ps.table_witness) {
    LogUpInput in = ps.logup_input(tw);
    LogUpProof tp = logup_batch_prove(in, t);
    ps.add_witness_claim(tw.commits[0], tp.output_claims[0]);
    proof.table_proofs.push_back({tw.commits[0].first.pure(), tp});  // Relu/Range/Clamping have no table poly claims
  }
  // CommitmentProver::prove (commit/context.rs:355-418)
  for (auto& c : ps.trivial_claims) proof.trivial_proofs.push_back(pcs_open_trivial(c.comm.second, c.comm.first));
  std::vector<const Mle*> polys; std::vector<const CommitmentWithWitness*> comms; std::vector<std::vector<E>> points; std::vector<Evaluation> evals;
  for (size_t i = 0; i < ps.claims.size(); i++) { polys.push_back(&ps.claims[i].comm.second); comms.push_back(&ps.claims[i].comm.first); points.push_back(ps.claims[i].claim.point); evals.push_back({i, i, ps.claims[i].claim.eval}); }
  proof.batch_proof = pcs_batch_open(ctx.pp, polys, comms, points, evals, t);
  proof.steps = ps.proofs;
  return proof;
}

// ------------------------------------------------------------------ canonical proof stream
struct Writer {
  std::vector<u64> w;
  void u(u64 v) { w.push_back(v); }
  void e(E x) { w.push_back(x.c0); w.push_back(x.c1); }
  void ve(const std::vector<E>& v) { u(v.size()); for (E x : v) e(x); }
  void d(const Digest& x) { for (u64 v : x) u(v); }
  void iop(const IOPProof& p) { ve(p.point); u(p.proofs.size()); for (auto& r : p.proofs) ve(r); }
  void claim(const Claim& c) { ve(c.point); e(c.eval); }
  void logup(const LogUpProof& p) {
    u(p.sumcheck_proofs.size()); for (auto& s : p.sumcheck_proofs) iop(s);
    u(p.round_evaluations.size()); for (auto& r : p.round_evaluations) ve(r);
    u(p.output_claims.size()); for (auto& c : p.output_claims) claim(c);
    u(p.circuit_outputs.size()); for (auto& c : p.circuit_outputs) ve(c);
    u(p.is_table ? 1 : 0);
  }
  void comm(const Commitment& c) { d(c.root); u(c.num_vars); u(c.is_base ? 1 : 0); }
  void cq(const CodewordQuery& q) {
    u(q.is_ext ? 1 : 0);
    if (q.is_ext) { e(q.left); e(q.right); } else { u(q.left.c0); u(q.right.c0); }
    u(q.index); u(q.path.size()); for (auto& x : q.path) d(x);
  }
  void basefold(const BasefoldProof& p) {
    u(p.sumcheck_messages.size()); for (auto& m : p.sumcheck_messages) ve(m);
    u(p.roots.size()); for (auto& r : p.roots) d(r);
    ve(p.final_message);
    u(p.queries.size());
    for (auto& q : p.queries) { u(q.index); u(q.oracle_query.size()); for (auto& c : q.oracle_query) cq(c); u(q.commitments_query.size()); for (auto& c : q.commitments_query) cq(c); }
    u(p.sumcheck_proof.size()); for (auto& m : p.sumcheck_proof) ve(m);
    u(p.trivial_proof.size());
    for (auto& m : p.trivial_proof) { u(m.is_ext ? 1 : 0); u(m.len()); if (m.is_ext) for (E x : m.e) e(x); else for (u64 x : m.b) u(x); }
  }
};
constexpr u64 PROOF_MAGIC = 0x31464F4F52505044ULL;  // "DPPROOF1"
static inline std::vector<u64> serialize_proof(const Proof& p) {
  Writer w; w.u(PROOF_MAGIC); w.u(p.steps.size());
  for (auto& [id, lp] : p.steps) {
    w.u(id); w.u(lp.kind);
    if (lp.kind == L_DENSE) { w.iop(lp.dense.sumcheck); w.e(lp.dense.bias_eval); w.ve(lp.dense.individual_claims); }
    else if (lp.kind == L_REQUANT) {
      w.iop(lp.req.io_accumulation); w.ve(lp.req.accumulation_evals); w.logup(lp.req.clamping_lookup); w.logup(lp.req.shifted_lookup);
      w.u(lp.req.commitments.size()); for (auto& c : lp.req.commitments) w.comm(c);
    } else {
      w.iop(lp.act.io_accumulation.sumcheck); w.ve(lp.act.io_accumulation.evals); w.logup(lp.act.lookup);
      w.u(lp.act.commits.size()); for (auto& c : lp.act.commits) w.comm(c);
    }
  }
  w.u(p.table_proofs.size()); for (auto& tp : p.table_proofs) { w.comm(tp.multiplicity_commit); w.logup(tp.lookup); }
  w.basefold(p.batch_proof);
  w.u(p.trivial_proofs.size()); for (auto& tp : p.trivial_proofs) w.basefold(tp);
  return w.w;
}

}  // namespace orc
