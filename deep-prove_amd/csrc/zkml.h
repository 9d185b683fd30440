// The zkml proving surface for the Dense / Requant / ReLU graph on top of device ops:
//   Context::generate (zkml/src/iop/context.rs:109-215, commit/context.rs:59-115)
//   Prover::prove     (zkml/src/iop/prover.rs:401-488) with Dense::prove_step (layers/dense.rs:423-561),
//                     Requant::prove_step (layers/requant.rs:531-690), Activation::prove_step (layers/activation.rs:385-456),
//                     generate_lookup_witnesses (lookup/context.rs:631-781), prove_tables (iop/prover.rs:110-157),
//                     CommitmentProver::prove (commit/context.rs:355-418)
//   verify            (zkml/src/iop/verifier.rs:72-318) — host only, as in the reference.
// A model is an already padded / quantised chain with node ids Dense, Requant, Relu, Dense, ... as produced by
// Model::random_with_rng (zkml/src/model/mod.rs:596-665); the ONNX / float front-end is out of scope.
#pragma once
#include "logup.h"
#include "pcs.h"
#include <unordered_map>
#include <algorithm>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace dp {

constexpr unsigned Q_BIT_LEN = 8;
constexpr int64_t Q_MIN = -127, Q_MAX = 127;
constexpr int64_t COLUMN_SEPARATOR = int64_t(1) << 32;

struct LayerSpec {
  int kind = L_DENSE;
  size_t nrows = 0, ncols = 0;
  std::vector<int64_t> weights, bias;
  unsigned right_shift = 0, fp_scale = 0, intermediate_bit_size = 0;
  int64_t fixed_point_multiplier = 0;
  unsigned shift() const { return fp_scale + right_shift; }
  unsigned clamping_size() const { return intermediate_bit_size + dp_ceil_log2((size_t)fixed_point_multiplier) - shift(); }
};
struct ModelSpec { size_t input_len = 0; std::vector<LayerSpec> layers; };

struct TableType {
  int kind; unsigned size;  // 0 Relu, 2 Range, 3 Clamping(size)  (derive(Ord) order of lookup/context.rs:55-72)
  bool operator<(const TableType& o) const { return kind != o.kind ? kind < o.kind : size < o.size; }
  bool operator==(const TableType& o) const { return kind == o.kind && size == o.size; }
  unsigned vars() const { return kind == 3 ? size : Q_BIT_LEN; }
  const char* label() const { return kind == 0 ? "Relu" : kind == 3 ? "Clamping" : nullptr; }
};
inline int64_t q_clamp(int64_t x) { return x < Q_MIN ? Q_MIN : x > Q_MAX ? Q_MAX : x; }
inline int64_t q_relu(int64_t x) { return x < 0 ? 0 : x; }
inline void table_columns(const TableType& tt, std::vector<int64_t>& merged, std::vector<std::vector<int64_t>>& cols) {
  merged.clear(); cols.clear();
  if (tt.kind == 0) { cols.resize(2); for (int64_t i = Q_MIN - 1; i <= Q_MAX; i++) { int64_t o = q_relu(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(i); cols[1].push_back(o); } }
  else if (tt.kind == 2) { cols.resize(1); for (int64_t i = 0; i < (int64_t(1) << Q_BIT_LEN); i++) { merged.push_back(i); cols[0].push_back(i); } }
  else { cols.resize(2); int64_t mx = int64_t(1) << (tt.size - 1); for (int64_t i = -mx; i < mx; i++) { int64_t o = q_clamp(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(i); cols[1].push_back(o); } }
}

// ---- inference (the reference's Model::run; CPU pre-processing outside "proving time", zkml/src/bin/bench.rs:341-352)
struct Trace { std::vector<std::vector<int64_t>> in, out; };
inline Trace run_model(const ModelSpec& m, const std::vector<int64_t>& input) {
  Trace tr; std::vector<int64_t> cur = input;
  DP_REQUIRE(cur.size() == m.input_len, DP_ERR_SHAPE, "input length mismatch");
  for (auto& l : m.layers) {
    tr.in.push_back(cur);
    std::vector<int64_t> o;
    if (l.kind == L_DENSE) {
      DP_REQUIRE(cur.size() == l.ncols, DP_ERR_SHAPE, "dense input size mismatch");
      o.resize(l.nrows);
      for (size_t i = 0; i < l.nrows; i++) { int64_t a = 0; const int64_t* w = &l.weights[i * l.ncols]; for (size_t j = 0; j < l.ncols; j++) a += w[j] * cur[j]; o[i] = a + l.bias[i]; }
    } else if (l.kind == L_REQUANT) {
      unsigned sh = l.shift();
      for (int64_t v : cur) {
        DP_REQUIRE((v < 0 ? -v : v) <= (int64_t(1) << l.intermediate_bit_size), DP_ERR_ARG, "requant: value exceeds intermediate bit size");
        o.push_back(q_clamp((v * l.fixed_point_multiplier + (int64_t(1) << (sh - 1))) >> sh));
      }
    } else for (int64_t v : cur) o.push_back(q_relu(v));
    tr.out.push_back(o); cur = o;
  }
  return tr;
}

inline size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

// ---- what the verifier needs (the serialisable part of Context)
struct VerifierContext {
  ModelSpec shape;  // weights/bias vectors left empty
  unsigned full_log = 0;
  std::map<size_t, std::map<std::string, Commitment>> model_comms;
  std::vector<TableType> tables;
};

struct Context {
  Dev* dev = nullptr;
  ModelSpec model;
  unsigned full_log = 0;
  size_t max_poly_len = 0;
  std::map<size_t, std::map<std::string, DevCommit>> model_comms;  // BTreeMap<NodeId, BTreeMap<PolyId, ..>>
  std::map<size_t, DBuf> weights_dev;                              // base-field weight matrices kept for K2
  std::vector<TableType> tables;
  VerifierContext verifier_ctx() const {
    VerifierContext v; v.full_log = full_log; v.tables = tables; v.shape.input_len = model.input_len;
    for (auto& l : model.layers) { LayerSpec s = l; s.weights.clear(); s.bias.clear(); v.shape.layers.push_back(s); }
    for (auto& kv : model_comms) for (auto& pc : kv.second) v.model_comms[kv.first][pc.first] = pure_commitment(pc.second);
    return v;
  }
  ~Context() {
    if (!dev) return;
    for (auto& kv : model_comms) for (auto& pc : kv.second) dev->free_commit(pc.second);
  }
};

inline void validate_model(const ModelSpec& m) {
  DP_REQUIRE(is_pow2(m.input_len) && !m.layers.empty(), DP_ERR_SHAPE, "model: input length must be a power of two");
  size_t cur = m.input_len;
  for (auto& l : m.layers) {
    if (l.kind == L_DENSE) {
      DP_REQUIRE(is_pow2(l.nrows) && is_pow2(l.ncols) && l.nrows >= 2 && l.ncols >= 2, DP_ERR_SHAPE, "dense: padded dimensions must be powers of two >= 2");
      DP_REQUIRE(l.ncols == cur && l.weights.size() == l.nrows * l.ncols && l.bias.size() == l.nrows, DP_ERR_SHAPE, "dense: tensor sizes");
      cur = l.nrows;
    } else if (l.kind == L_REQUANT) {
      DP_REQUIRE(l.fixed_point_multiplier > 0 && l.shift() % Q_BIT_LEN == 0 && l.shift() >= Q_BIT_LEN && l.shift() < 63, DP_ERR_ARG, "requant: shift must be a positive multiple of BIT_LEN");
      DP_REQUIRE(l.intermediate_bit_size + l.fp_scale <= 63, DP_ERR_ARG, "requant: intermediate_bit_size + fp_scale > 63");
      unsigned cs = l.clamping_size();
      DP_REQUIRE(cs >= 1 && cs <= 24 && cur >= 4, DP_ERR_ARG, "requant: unsupported clamping table size / tensor length");
    } else if (l.kind == L_RELU) { DP_REQUIRE(cur >= 4, DP_ERR_SHAPE, "relu: tensor length must be >= 4"); }
    else DP_REQUIRE(false, DP_ERR_ARG, "unknown layer kind");
  }
}

inline std::unique_ptr<Context> context_generate(Dev& dev, const ModelSpec& m) {
  validate_model(m);
  std::unique_ptr<Context> ctx(new Context());
  ctx->dev = &dev; ctx->model = m;
  size_t mpl = m.input_len, cur = m.input_len;
  std::vector<TableType> ts;
  auto add = [&](TableType t) { for (auto& x : ts) if (x == t) return; ts.push_back(t); };
  for (auto& l : m.layers) {
    if (l.kind == L_DENSE) cur = l.nrows;
    else if (l.kind == L_REQUANT) { add({2, 0}); add({3, l.clamping_size()}); mpl = std::max(mpl, next_pow2(cur)); }
    else { add({0, 0}); mpl = std::max(mpl, next_pow2(cur)); }
  }
  std::sort(ts.begin(), ts.end());
  for (auto& t : ts) mpl = std::max(mpl, size_t(1) << t.vars());
  for (auto& l : m.layers) if (l.kind == L_DENSE) mpl = std::max(mpl, std::max(next_pow2(l.weights.size()), next_pow2(l.bias.size())));
  mpl = next_pow2(mpl);
  ctx->max_poly_len = mpl; ctx->full_log = dp_ceil_log2(mpl); ctx->tables = ts;
  dev.pcs_init(ctx->full_log);
  for (size_t id = 0; id < m.layers.size(); id++) {
    const LayerSpec& l = m.layers[id];
    if (l.kind != L_DENSE) continue;
    DBuf w = dev.alloc_persistent(l.weights.size(), false), b = dev.alloc_persistent(l.bias.size(), false);
    dev.upload_i64(w, l.weights.data()); dev.upload_i64(b, l.bias.data());
    ctx->model_comms[id]["DenseWeight"] = dev.commit(w, true);
    ctx->model_comms[id]["DenseBias"] = dev.commit(b, true);
    ctx->weights_dev[id] = w;
  }
  return ctx;
}

// ---- prover
struct LogUpWitness {
  bool is_table = false;
  std::vector<DevCommit> commits;
  std::vector<DBuf> columns;
  size_t columns_per_instance = 1;
  TableType table_type{0, 0};
  DBuf multiplicities;
};
struct CommitClaim { DevCommit comm; Claim claim; };
struct ProverState {
  Context* ctx; Dev* dev; Transcript* t;
  std::map<size_t, LayerProof> proofs;
  std::vector<CommitClaim> claims, trivial_claims;
  std::map<size_t, std::vector<LogUpWitness>> lookup_witness;
  std::vector<LogUpWitness> table_witness;
  Ext constant_challenge; std::map<TableType, Ext> challenge_map;
  void add_witness_claim(const DevCommit& c, Claim cl) {
    if (c.nv <= PCS_BASECODE_LOG) trivial_claims.push_back({c, std::move(cl)}); else claims.push_back({c, std::move(cl)});
  }
  LogUpInputDev logup_input(const LogUpWitness& w) const {
    LogUpInputDev in; in.is_table = w.is_table; in.columns = w.columns; in.multiplicities = w.multiplicities;
    in.constant_challenge = constant_challenge; in.column_separation_challenge = challenge_map.at(w.table_type);
    in.columns_per_instance = w.columns_per_instance; return in;
  }
};

struct PhaseTimer {  // DP_TIMING=1 prints the host-side wall time of each proving phase to stderr
  bool on; std::chrono::steady_clock::time_point t0; const char* name;
  PhaseTimer() : on(getenv("DP_TIMING") && atoi(getenv("DP_TIMING"))), t0(std::chrono::steady_clock::now()), name(nullptr) {}
  void lap(const char* what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[dp timing] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
// generate_lookup_witnesses (lookup/context.rs:631-781) with gen_lookup_witness of requant.rs:208-345 / activation.rs:238-318.
// All witness columns of the inference are produced on the host (tiny integer work on activation vectors), shipped to
// the device in ONE upload and committed in ONE batched call (the reference commits them one by one on rayon threads).
inline void instantiate_witness_ctx(ProverState& ps, const Trace& tr) {
  Context& ctx = *ps.ctx; Dev& dev = *ps.dev;
  if (ctx.tables.empty()) return;
  PhaseTimer wt;
  std::map<TableType, std::unordered_map<int64_t, u64>> counts;
  struct Col { std::vector<int64_t> v; };
  std::vector<Col> cols;                       // every i64 column that goes to the device, in commit order first
  struct Pending { size_t node; int which; std::vector<size_t> col_ids; size_t cpi; TableType tt; };
  std::vector<Pending> pend;
  for (size_t id = 0; id < ctx.model.layers.size(); id++) {
    const LayerSpec& l = ctx.model.layers[id];
    if (l.kind == L_REQUANT) {
      unsigned shift = l.shift(); int64_t rounding = int64_t(1) << (shift - 1), mask = (int64_t(1) << shift) - 1;
      std::vector<int64_t> cin, cout, shifted;
      for (int64_t v : tr.in[id]) { int64_t tmp = v * l.fixed_point_multiplier + rounding; int64_t c = tmp >> shift; cin.push_back(c); cout.push_back(q_clamp(c)); shifted.push_back(tmp & mask); }
      unsigned nchunks = shift / Q_BIT_LEN; int64_t rmask = (int64_t(1) << Q_BIT_LEN) - 1;
      TableType ct{3, l.clamping_size()}, rt{2, 0};
      int64_t cmax = int64_t(1) << (ct.size - 1);
      for (size_t i = 0; i < cin.size(); i++) {
        DP_REQUIRE(cin[i] >= -cmax && cin[i] < cmax, DP_ERR_ARG, "requant: value falls outside the clamping table");
        counts[ct][cin[i] + cout[i] * COLUMN_SEPARATOR] += 1;
      }
      Pending pc{id, 0, {}, 2, ct}, pr{id, 1, {}, 1, rt};
      pc.col_ids = {cols.size(), cols.size() + 1};
      cols.push_back({cin}); cols.push_back({cout});
      for (unsigned j = 0; j < nchunks; j++) {
        std::vector<int64_t> ch; ch.reserve(shifted.size());
        for (int64_t sft : shifted) { int64_t v = (sft >> (j * Q_BIT_LEN)) & rmask; ch.push_back(v); counts[rt][v] += 1; }
        pr.col_ids.push_back(cols.size()); cols.push_back({std::move(ch)});
      }
      pend.push_back(pc); pend.push_back(pr);
    } else if (l.kind == L_RELU) {
      TableType rt{0, 0};
      const auto& a = tr.in[id]; const auto& b = tr.out[id];
      for (size_t i = 0; i < a.size(); i++) counts[rt][a[i] + COLUMN_SEPARATOR * b[i]] += 1;
      Pending p{id, 0, {cols.size(), cols.size() + 1}, 2, rt};
      cols.push_back({a}); cols.push_back({b});
      pend.push_back(p);
    }
  }
  size_t n_witness_cols = cols.size();
  wt.lap("  witness: host columns");
  // table columns (not committed) ride in the same upload
  struct TabInfo { TableType tt; std::vector<size_t> col_ids; std::vector<u64> mult; };
  std::vector<TabInfo> tabs;
  for (auto& kv : counts) {
    const TableType& tt = kv.first;
    std::vector<int64_t> merged; std::vector<std::vector<int64_t>> tc;
    table_columns(tt, merged, tc);
    std::unordered_map<int64_t, u64> cnt; for (int64_t v : merged) cnt[v] += 1;
    TabInfo ti; ti.tt = tt; ti.mult.resize(merged.size());
    for (size_t i = 0; i < merged.size(); i++) {
      auto it = kv.second.find(merged[i]);
      if (it == kv.second.end()) { ti.mult[i] = 0; continue; }
      u64 c = cnt[merged[i]];
      ti.mult[i] = gl_mul(gl_from_u64(it->second), c != 1 ? gl_inv(gl_from_u64(c)) : 1);
    }
    for (auto& c : tc) { ti.col_ids.push_back(cols.size()); cols.push_back({std::move(c)}); }
    tabs.push_back(std::move(ti));
  }
  wt.lap("  witness: multiplicities");
  // one upload of all i64 columns, one of all multiplicity vectors
  size_t total = 0; for (auto& c : cols) total += c.v.size();
  std::vector<int64_t> flat; flat.reserve(total);
  std::vector<size_t> offs;
  for (auto& c : cols) { offs.push_back(flat.size()); flat.insert(flat.end(), c.v.begin(), c.v.end()); }
  DBuf big = dev.alloc(total, false);
  dev.upload_i64(big, flat.data());
  std::vector<DBuf> dcol;
  for (size_t i = 0; i < cols.size(); i++) dcol.push_back(big.slice(offs[i], cols[i].v.size()));
  size_t mtotal = 0; for (auto& t : tabs) mtotal += t.mult.size();
  std::vector<u64> mflat; mflat.reserve(mtotal);
  std::vector<size_t> moffs;
  for (auto& t : tabs) { moffs.push_back(mflat.size()); mflat.insert(mflat.end(), t.mult.begin(), t.mult.end()); }
  DBuf mbig = dev.alloc(mtotal, false);
  dev.upload(mbig, mflat.data());
  wt.lap("  witness: uploads");
  // one batched commit: witness columns in order, then the table multiplicities
  std::vector<DBuf> to_commit(dcol.begin(), dcol.begin() + n_witness_cols);
  for (size_t i = 0; i < tabs.size(); i++) to_commit.push_back(mbig.slice(moffs[i], tabs[i].mult.size()));
  std::vector<DevCommit> comms = dev.commit_many(to_commit, false);
  wt.lap("  witness: commit_many");
  for (auto& p : pend) {
    LogUpWitness w; w.columns_per_instance = p.cpi; w.table_type = p.tt;
    for (size_t cid : p.col_ids) { w.columns.push_back(dcol[cid]); w.commits.push_back(comms[cid]); }
    ps.lookup_witness[p.node].push_back(std::move(w));
  }
  for (size_t i = 0; i < tabs.size(); i++) {
    LogUpWitness w; w.is_table = true; w.table_type = tabs[i].tt; w.columns_per_instance = tabs[i].col_ids.size();
    w.multiplicities = to_commit[n_witness_cols + i];
    for (size_t cid : tabs[i].col_ids) w.columns.push_back(dcol[cid]);
    w.commits.push_back(comms[n_witness_cols + i]);
    ps.table_witness.push_back(std::move(w));
  }
  ps.constant_challenge = ps.t->get_and_append_challenge("table_constant");
  for (auto& kv : counts) ps.challenge_map[kv.first] = kv.first.label() ? ps.t->get_and_append_challenge(kv.first.label()) : ex_one();
}

inline std::vector<u64> ext_words_from_i64(const std::vector<int64_t>& v) { std::vector<u64> w(2 * v.size()); for (size_t i = 0; i < v.size(); i++) { w[2 * i] = gl_from_i64(v[i]); w[2 * i + 1] = 0; } return w; }

inline Claim prove_dense(ProverState& ps, size_t id, const LayerSpec& l, const Claim& last, const std::vector<int64_t>& input) {
  Dev& dev = *ps.dev;
  DP_REQUIRE((size_t(1) << last.point.size()) == l.nrows, DP_ERR_SHAPE, "dense: claim point length");
  size_t mk = dev.mark();
  const auto& comms = ps.ctx->model_comms.at(id);
  Ext bias_eval;
  dev.mle_eval_batch(&comms.at("DenseBias").evals, 1, last.point.data(), (unsigned)last.point.size(), &bias_eval);
  DBuf mat = dev.alloc(l.ncols, true);
  dev.fix_high(mat, ps.ctx->weights_dev.at(id), l.nrows, l.ncols, last.point.data());
  DBuf in = dev.alloc(input.size(), true);  // trace.into_fields(): i64 -> Ext (model/trace.rs:50-92)
  { std::vector<u64> w = ext_words_from_i64(input); dev.upload(in, w.data()); }
  DevVP vp(dp_ceil_log2(l.ncols));
  vp.add_mle_list({mat, in}, ex_one());
  SumcheckOut sc = sumcheck_prove(dev, vp, *ps.t);
  std::vector<Ext> point = sc.proof.point; point.insert(point.end(), last.point.begin(), last.point.end());
  ps.add_witness_claim(comms.at("DenseBias"), {last.point, bias_eval});   // BTreeMap order: "DenseBias" < "DenseWeight"
  ps.add_witness_claim(comms.at("DenseWeight"), {point, sc.finals[0]});
  LayerProof lp; lp.kind = L_DENSE; lp.dense.sumcheck = sc.proof; lp.dense.bias_eval = bias_eval; lp.dense.individual_claims = sc.finals;
  ps.proofs[id] = lp;
  dev.release(mk);
  return {sc.proof.point, sc.finals[1]};
}
inline Ext recombine_claims(const LayerSpec& l, Ext clamping_claim, const Ext* shifted, size_t ns) {
  Ext full = ex_mul(ex_from_u64(u64(1) << l.shift()), clamping_claim), pw = ex_one();
  for (size_t i = 0; i < ns; i++) { full = ex_add(full, ex_mul(shifted[i], pw)); pw = ex_mul(pw, ex_from_u64(u64(1) << Q_BIT_LEN)); }
  return ex_mul(ex_sub(full, ex_from_u64(u64(1) << (l.shift() - 1))), ex_inv(ex_from_i64(l.fixed_point_multiplier)));
}
inline Claim prove_requant(ProverState& ps, size_t id, const LayerSpec& l, const Claim& last) {
  Dev& dev = *ps.dev;
  std::vector<LogUpWitness>& ws = ps.lookup_witness.at(id);
  LogUpWitness& cw = ws[0]; LogUpWitness& sw = ws[1];
  LogUpProof cproof = logup_batch_prove(dev, ps.logup_input(cw), *ps.t);
  LogUpProof sproof = logup_batch_prove(dev, ps.logup_input(sw), *ps.t);
  size_t mk = dev.mark();
  size_t n = cw.columns[0].n; unsigned nv = dp_ceil_log2(n);
  DBuf clamp_in = cw.columns[0], clamp_out = cw.columns[1];
  DBuf cbeta = dev.alloc(n, true), lbeta = dev.alloc(n, true), sbeta = dev.alloc(n, true);
  DP_REQUIRE(last.point.size() == nv, DP_ERR_SHAPE, "requant: claim point length");
  dev.eq_table(cbeta, cproof.output_claims[0].point.data(), nv, ex_one(), false);
  dev.eq_table(lbeta, last.point.data(), nv, ex_one(), false);
  dev.eq_table(sbeta, sproof.output_claims[0].point.data(), nv, ex_one(), false);
  Ext b = ps.t->get_and_append_challenge("requant_batching");
  DevVP vp(nv);
  vp.add_mle_list({clamp_out, lbeta}, ex_one());
  vp.add_mle_list({clamp_out, cbeta}, b);
  Ext comb = ex_mul(b, b);
  vp.add_mle_list({clamp_in, cbeta}, comb);
  comb = ex_mul(comb, b);
  for (auto& m : sw.columns) { vp.add_mle_list({sbeta, m}, comb); comb = ex_mul(comb, b); }
  SumcheckOut sc = sumcheck_prove(dev, vp, *ps.t);
  dev.release(mk);
  const std::vector<Ext>& fin = sc.finals;
  Ext cout_eval = fin[0], cin_eval = fin[3];
  size_t ns = fin.size() - 5;
  Ext combined = recombine_claims(l, cin_eval, &fin[5], ns);
  RequantProof rp; rp.io_accumulation = sc.proof; rp.clamping_lookup = cproof; rp.shifted_lookup = sproof;
  std::vector<Ext> evs = {cin_eval, cout_eval}; evs.insert(evs.end(), fin.begin() + 5, fin.end());
  std::vector<DevCommit> cm = cw.commits; cm.insert(cm.end(), sw.commits.begin(), sw.commits.end());
  for (size_t i = 0; i < evs.size(); i++) { rp.commitments.push_back(pure_commitment(cm[i])); ps.add_witness_claim(cm[i], {sc.proof.point, evs[i]}); rp.accumulation_evals.push_back(evs[i]); }
  LayerProof lp; lp.kind = L_REQUANT; lp.req = rp; ps.proofs[id] = lp;
  return {sc.proof.point, combined};
}
// same_poly::Prover::prove (commit/same_poly.rs:88-122)
inline SamePolyProof same_poly_prove(Dev& dev, const std::vector<Claim>& claims, const DBuf& poly, Transcript& t) {
  size_t mk = dev.mark();
  unsigned nv = dp_ceil_log2(poly.n);
  std::vector<Ext> a = t.read_challenges(claims.size());
  DBuf beta = dev.alloc(poly.n, true);
  for (size_t i = 0; i < claims.size(); i++) {
    DP_REQUIRE(claims[i].point.size() == nv, DP_ERR_SHAPE, "same_poly: claim point length");
    dev.eq_table(beta, claims[i].point.data(), nv, a[i], i > 0);
  }
  DevVP vp(nv);
  vp.add_mle_list({beta, poly}, ex_one());
  SumcheckOut sc = sumcheck_prove(dev, vp, t);
  dev.release(mk);
  return {sc.proof, sc.finals};
}
inline Claim prove_relu(ProverState& ps, size_t id, const Claim& last, const std::vector<int64_t>& output) {
  Dev& dev = *ps.dev;
  LogUpWitness& w = ps.lookup_witness.at(id)[0];
  LogUpProof lproof = logup_batch_prove(dev, ps.logup_input(w), *ps.t);
  Claim input_claim = lproof.output_claims[0], output_claim = lproof.output_claims[1];
  size_t mk = dev.mark();
  DBuf out = dev.alloc(output.size(), true);
  { std::vector<u64> ww = ext_words_from_i64(output); dev.upload(out, ww.data()); }
  SamePolyProof sp = same_poly_prove(dev, {last, output_claim}, out, *ps.t);
  dev.release(mk);
  ActivationProof ap; ap.io_accumulation = sp; ap.lookup = lproof;
  ps.add_witness_claim(w.commits[0], input_claim); ap.commits.push_back(pure_commitment(w.commits[0]));
  ps.add_witness_claim(w.commits[1], {sp.sumcheck.point, sp.evals[1]}); ap.commits.push_back(pure_commitment(w.commits[1]));
  LayerProof lp; lp.kind = L_RELU; lp.act = ap; ps.proofs[id] = lp;
  return input_claim;
}

// Prover::prove(trace). `tr` comes from run_model (inference is not part of proving time in the reference either).
// `dev` may be any device context on the GPU that holds `ctx` (the model commitments are only read), so several proofs
// can be in flight at once, each on its own stream/arena (dp_model_prove_batch).
inline Proof prove(Context& ctx, Dev& dev, const Trace& tr, Transcript& t) {
  PhaseTimer pt;
  sc_stats() = ScStats();
  size_t mk = dev.mark();
  ProverState ps; ps.ctx = &ctx; ps.dev = &dev; ps.t = &t;
  for (auto& kv : ctx.model_comms) for (auto& pc : kv.second) t.append_digest(pc.second.tree.root);
  instantiate_witness_ctx(ps, tr);
  pt.lap("witness columns + commits");
  const std::vector<int64_t>& out = tr.out.back();
  std::vector<Ext> r = t.read_challenges(dp_ceil_log2(out.size()));
  Claim cur; cur.point = r;
  { std::vector<Ext> ov(out.size()); for (size_t i = 0; i < out.size(); i++) ov[i] = ex_from_i64(out[i]); cur.eval = host_mle_eval(ov, r); }
  for (size_t id = ctx.model.layers.size(); id-- > 0;) {
    const LayerSpec& l = ctx.model.layers[id];
    if (l.kind == L_DENSE) cur = prove_dense(ps, id, l, cur, tr.in[id]);
    else if (l.kind == L_REQUANT) cur = prove_requant(ps, id, l, cur);
    else cur = prove_relu(ps, id, cur, tr.out[id]);
    if (pt.on) { char b[64]; snprintf(b, sizeof b, "layer %zu (%s)", id, l.kind == L_DENSE ? "dense" : l.kind == L_REQUANT ? "requant" : "relu"); pt.lap(b); }
  }
  Proof proof;
  for (auto& tw : ps.table_witness) {
    LogUpProof tp = logup_batch_prove(dev, ps.logup_input(tw), t);
    ps.add_witness_claim(tw.commits[0], tp.output_claims[0]);
    proof.table_proofs.push_back({pure_commitment(tw.commits[0]), tp});
  }
  pt.lap("table proofs");
  for (auto& c : ps.trivial_claims) proof.trivial_proofs.push_back(pcs_open_trivial(dev, c.comm));
  pt.lap("trivial openings");
  std::vector<OpenClaim> oc;
  for (auto& c : ps.claims) oc.push_back({&c.comm, c.claim.point, c.claim.eval});
  proof.batch_proof = pcs_batch_open(dev, ctx.full_log, oc, t);
  pt.lap("batch_open");
  if (pt.on) fprintf(stderr, "[dp timing] sumcheck rounds %zu: device wait %.3f ms, host transcript+algebra %.3f ms\n", sc_stats().rounds, sc_stats().dev_ms, sc_stats().host_ms);
  proof.steps = ps.proofs;
  dev.release(mk);
  return proof;
}

inline Proof prove(Context& ctx, const Trace& tr, Transcript& t) { return prove(ctx, *ctx.dev, tr, t); }

// ---- verifier (zkml/src/iop/verifier.rs:72-318; layers' verify fns; commit/context.rs:424-599)
struct IO { std::vector<int64_t> input, output; };
inline void verify(const VerifierContext& vc, const Proof& proof, const IO& io, Transcript& t) {
  const ModelSpec& m = vc.shape;
  for (auto& kv : vc.model_comms) for (auto& pc : kv.second) t.append_digest(pc.second.root);
  Ext constant_challenge = ex_zero(); std::map<TableType, Ext> chmap;
  if (!vc.tables.empty()) {
    constant_challenge = t.get_and_append_challenge("table_constant");
    for (auto& tt : vc.tables) chmap[tt] = tt.label() ? t.get_and_append_challenge(tt.label()) : ex_one();
  }
  std::vector<Ext> nums, dens;
  auto add_fracs = [&](const LogUpProof& p) {
    for (auto& e : p.circuit_outputs) { DP_REQUIRE(e.size() == 4, DP_ERR_VERIFY, "circuit outputs"); nums.push_back(ex_add(ex_mul(e[0], e[3]), ex_mul(e[1], e[2]))); dens.push_back(ex_mul(e[2], e[3])); }
  };
  for (size_t id = 0; id < m.layers.size(); id++) {
    auto it = proof.steps.find(id);
    DP_REQUIRE(it != proof.steps.end() && it->second.kind == m.layers[id].kind, DP_ERR_VERIFY, "missing or mistyped layer proof");
    if (it->second.kind == L_RELU) add_fracs(it->second.act.lookup);
    if (it->second.kind == L_REQUANT) { add_fracs(it->second.req.clamping_lookup); add_fracs(it->second.req.shifted_lookup); }
  }
  DP_REQUIRE(proof.steps.size() == m.layers.size(), DP_ERR_VERIFY, "unexpected layer proofs");
  for (auto& tp : proof.table_proofs) add_fracs(tp.lookup);
  // output claim
  DP_REQUIRE(is_pow2(io.output.size()) && io.input.size() == m.input_len, DP_ERR_VERIFY, "io shapes");
  std::vector<Ext> r = t.read_challenges(dp_ceil_log2(io.output.size()));
  Claim cur; cur.point = r;
  { std::vector<Ext> ov(io.output.size()); for (size_t i = 0; i < ov.size(); i++) ov[i] = ex_from_i64(io.output[i]); cur.eval = host_mle_eval(ov, r); }
  std::vector<VerifyClaim> claims, trivial_claims;
  auto add_claim = [&](const Commitment& c, const Claim& cl) {
    VerifyClaim v{c, cl.point, cl.eval};
    if (cl.point.size() <= PCS_BASECODE_LOG) trivial_claims.push_back(v); else claims.push_back(v);
  };
  std::map<size_t, std::map<std::string, Commitment>> unused = vc.model_comms;
  size_t cur_len = io.output.size();
  for (size_t id = m.layers.size(); id-- > 0;) {
    const LayerSpec& l = m.layers[id];
    const LayerProof& lp = proof.steps.at(id);
    if (l.kind == L_DENSE) {  // DenseCtx::verify_dense (dense.rs:576-643)
      const DenseProof& dpf = lp.dense;
      DP_REQUIRE(cur.point.size() == dp_ceil_log2(l.nrows) && dpf.individual_claims.size() == 2, DP_ERR_VERIFY, "dense: shapes");
      Ext eval_no_bias = ex_sub(cur.eval, dpf.bias_eval);
      SubClaim sub = sumcheck_verify(eval_no_bias, dpf.sumcheck, dp_ceil_log2(l.ncols), 2, t);
      std::vector<Ext> pt = sub.point; pt.insert(pt.end(), cur.point.begin(), cur.point.end());
      auto nit = unused.find(id);
      DP_REQUIRE(nit != unused.end(), DP_ERR_VERIFY, "dense: no commitments for node");
      add_claim(nit->second.at("DenseBias"), {cur.point, dpf.bias_eval});
      add_claim(nit->second.at("DenseWeight"), {pt, dpf.individual_claims[0]});
      unused.erase(nit);
      DP_REQUIRE(ex_eq(ex_mul(dpf.individual_claims[0], dpf.individual_claims[1]), sub.expected_evaluation), DP_ERR_VERIFY, "dense: sumcheck claim failed");
      cur = {sub.point, dpf.individual_claims[1]};
      cur_len = l.ncols;
    } else if (l.kind == L_REQUANT) {  // verify_requant (requant.rs:692-817)
      const RequantProof& rp = lp.req;
      TableType ct{3, l.clamping_size()};
      DP_REQUIRE(chmap.count(ct), DP_ERR_VERIFY, "requant: no challenge for clamping table");
      size_t inst = l.shift() / Q_BIT_LEN;
      LogUpVerifierClaim cc = verify_logup_proof(rp.clamping_lookup, 1, constant_challenge, chmap[ct], t);
      LogUpVerifierClaim scl = verify_logup_proof(rp.shifted_lookup, inst, constant_challenge, ex_one(), t);
      Ext b = t.get_and_append_challenge("requant_batching");
      DP_REQUIRE(cc.claims.size() == 2 && scl.claims.size() == inst && rp.accumulation_evals.size() == 2 + inst && rp.commitments.size() == 2 + inst, DP_ERR_VERIFY, "requant: shapes");
      const std::vector<Ext>& cpt = cc.claims[0].point; const std::vector<Ext>& spt = scl.claims[0].point;
      Ext init = ex_zero(), chal = ex_one();
      std::vector<Ext> vals = {cur.eval, cc.claims[1].eval, cc.claims[0].eval};
      for (auto& c : scl.claims) vals.push_back(c.eval);
      for (const Ext& v : vals) { init = ex_add(init, ex_mul(chal, v)); chal = ex_mul(chal, b); }
      SubClaim sub = sumcheck_verify(init, rp.io_accumulation, (unsigned)cpt.size(), 2, t);
      DP_REQUIRE(cur.point.size() == sub.point.size() && cpt.size() == sub.point.size() && spt.size() == sub.point.size(), DP_ERR_VERIFY, "requant: point sizes");
      Ext lb = eq_eval(cur.point.data(), sub.point.data(), sub.point.size());
      Ext cb = eq_eval(cpt.data(), sub.point.data(), sub.point.size());
      Ext sb = eq_eval(spt.data(), sub.point.data(), sub.point.size());
      const std::vector<Ext>& ae = rp.accumulation_evals;
      Ext calc = ex_mul(ex_add(lb, ex_mul(b, cb)), ae[1]);
      Ext comb = ex_mul(b, b);
      calc = ex_add(calc, ex_mul(ex_mul(comb, cb), ae[0]));
      comb = ex_mul(comb, b);
      for (size_t i = 2; i < ae.size(); i++) { calc = ex_add(calc, ex_mul(ex_mul(ae[i], sb), comb)); comb = ex_mul(comb, b); }
      DP_REQUIRE(ex_eq(calc, sub.expected_evaluation), DP_ERR_VERIFY, "requant: accumulation claim mismatch");
      Ext next = recombine_claims(l, ae[0], &ae[2], ae.size() - 2);
      for (size_t i = 0; i < ae.size(); i++) add_claim(rp.commitments[i], {sub.point, ae[i]});
      cur = {sub.point, next};
    } else {  // verify_activation (activation.rs:459-517)
      const ActivationProof& ap = lp.act;
      TableType rt{0, 0};
      DP_REQUIRE(chmap.count(rt), DP_ERR_VERIFY, "relu: no challenge for table");
      LogUpVerifierClaim vcl = verify_logup_proof(ap.lookup, 1, constant_challenge, chmap[rt], t);
      DP_REQUIRE(vcl.claims.size() == 2 && ap.commits.size() == 2 && ap.io_accumulation.evals.size() == 2, DP_ERR_VERIFY, "relu: shapes");
      unsigned nv = dp_ceil_log2(cur_len);
      std::vector<Claim> sp_claims = {cur, vcl.claims[1]};
      for (auto& c : sp_claims) DP_REQUIRE(c.point.size() == nv, DP_ERR_VERIFY, "same_poly: invalid claim length");
      std::vector<Ext> a = t.read_challenges(sp_claims.size());
      Ext y = ex_zero();
      for (size_t i = 0; i < a.size(); i++) y = ex_add(y, ex_mul(sp_claims[i].eval, a[i]));
      SubClaim sub = sumcheck_verify(y, ap.io_accumulation.sumcheck, nv, 2, t);
      Ext computed = ex_zero();
      for (size_t i = 0; i < a.size(); i++) computed = ex_add(computed, ex_mul(a[i], identity_eval(sp_claims[i].point, ap.io_accumulation.sumcheck.point)));
      DP_REQUIRE(ex_eq(computed, ap.io_accumulation.evals[0]), DP_ERR_VERIFY, "same_poly: beta evaluation mismatch");
      DP_REQUIRE(ex_eq(ex_mul(ap.io_accumulation.evals[0], ap.io_accumulation.evals[1]), sub.expected_evaluation), DP_ERR_VERIFY, "same_poly: final evals invalid");
      Claim new_out{ap.io_accumulation.sumcheck.point, ap.io_accumulation.evals[1]};
      add_claim(ap.commits[0], vcl.claims[0]);
      add_claim(ap.commits[1], new_out);
      cur = vcl.claims[0];
    }
  }
  // table proofs (verifier.rs:320-383)
  DP_REQUIRE(proof.table_proofs.size() == vc.tables.size(), DP_ERR_VERIFY, "wrong number of table proofs");
  for (size_t i = 0; i < vc.tables.size(); i++) {
    const TableType& tt = vc.tables[i];
    const TableProof& tp = proof.table_proofs[i];
    LogUpVerifierClaim v = verify_logup_proof(tp.lookup, 1, constant_challenge, chmap[tt], t);
    add_claim(tp.multiplicity_commit, v.claims[0]);
    const std::vector<Ext>& pt = v.claims[0].point;
    DP_REQUIRE(pt.size() == tt.vars(), DP_ERR_VERIFY, "table: point size");
    std::vector<Ext> expect;  // evaluate_table_columns (lookup/context.rs:302-462)
    Ext idx = ex_zero();
    for (size_t k = 0; k < pt.size(); k++) idx = ex_add(idx, ex_mul(pt[k], ex_from_u64(u64(1) << k)));
    if (tt.kind == 2) expect = {idx};
    else if (tt.kind == 0) {
      Ext second = ex_zero();
      for (size_t k = 0; k + 1 < pt.size(); k++) second = ex_add(second, ex_mul(ex_mul(pt[k], ex_from_u64(u64(1) << k)), pt.back()));
      expect = {ex_sub(idx, ex_from_u64(u64(1) << (Q_BIT_LEN - 1))), second};
    } else {
      int64_t mx = int64_t(1) << (tt.size - 1);
      std::vector<Ext> col; for (int64_t x = -mx; x < mx; x++) col.push_back(ex_from_i64(q_clamp(x)));
      expect = {ex_sub(idx, ex_from_u64(u64(1) << (tt.size - 1))), host_mle_eval(col, pt)};
    }
    DP_REQUIRE(expect.size() + 1 == v.claims.size(), DP_ERR_VERIFY, "table: number of column claims");
    for (size_t k = 0; k < expect.size(); k++) DP_REQUIRE(ex_eq(v.claims[k + 1].eval, expect[k]), DP_ERR_VERIFY, "table: claimed column evaluation is wrong");
  }
  // input claim (provable/mod.rs:542-565)
  { std::vector<Ext> iv(io.input.size()); for (size_t i = 0; i < iv.size(); i++) iv[i] = ex_from_i64(io.input[i]);
    DP_REQUIRE(cur.point.size() == dp_ceil_log2(iv.size()) && ex_eq(host_mle_eval(iv, cur.point), cur.eval), DP_ERR_VERIFY, "input claim is incorrect"); }
  // commitment openings (commit/context.rs:520-598)
  DP_REQUIRE(unused.empty(), DP_ERR_VERIFY, "not all model commitments have been used");
  DP_REQUIRE(trivial_claims.size() == proof.trivial_proofs.size(), DP_ERR_VERIFY, "number of trivial proofs");
  for (size_t i = 0; i < trivial_claims.size(); i++) pcs_verify_trivial(trivial_claims[i].comm, trivial_claims[i].point, trivial_claims[i].eval, proof.trivial_proofs[i]);
  VerifierParams vp; vp.full_log = vc.full_log;
  pcs_batch_verify(vp, claims, proof.batch_proof, t);
  // global logup check (verifier.rs:273-291)
  Ext fn = ex_zero(), fd = ex_one();
  for (size_t i = 0; i < nums.size(); i++) { fn = ex_add(ex_mul(fn, dens[i]), ex_mul(nums[i], fd)); fd = ex_mul(fd, dens[i]); }
  DP_REQUIRE(ex_is_zero(fn), DP_ERR_VERIFY, "final logup numerator is non-zero");
  DP_REQUIRE(!ex_is_zero(fd), DP_ERR_VERIFY, "final logup denominator is zero");
}

}  // namespace dp
