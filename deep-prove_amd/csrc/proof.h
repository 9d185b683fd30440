// Proof objects of the reference (zkml/src/iop/mod.rs:21-44, layers/*.rs, lookup/logup_gkr/structs.rs:313-319,
// mpcs/src/basefold/structure.rs:334-345, sumcheck/src/structs.rs:42-61) and their canonical u64 stream
// (ascending NodeId, struct fields in declaration order — SURVEY.md A.12 / F4). The oracle emits the same stream.
#pragma once
#include "dev.h"
#include <map>

namespace dp {

struct Claim { std::vector<Ext> point; Ext eval; };
struct IOPProof { std::vector<Ext> point; std::vector<std::vector<Ext>> proofs; };
struct LogUpProof {
  std::vector<IOPProof> sumcheck_proofs;
  std::vector<std::vector<Ext>> round_evaluations;
  std::vector<Claim> output_claims;
  std::vector<std::vector<Ext>> circuit_outputs;
  bool is_table = false;
};
struct Commitment { Digest root; unsigned num_vars = 0; bool is_base = true; };
struct FieldVec { bool is_ext = false; std::vector<u64> w; size_t len() const { return is_ext ? w.size() / 2 : w.size(); } };
struct CodewordQuery { bool is_ext = false; Ext left, right; size_t index = 0; std::vector<Digest> path; };
struct BatchedQuery { size_t index = 0; std::vector<CodewordQuery> oracle_query, commitments_query; };
// A block of words WITHOUT value-initialisation whose storage is recycled per host thread: the query section of a batch opening is 5.8 MB that the device's image overwrites
// entirely — as a std::vector it was zero-filled on every resize and came from fresh zero pages on every proof (a page fault per 4 KB), ~0.6 ms of the proving thread.
struct RawWords {
  u64* p = nullptr; size_t n = 0, cap = 0;
  RawWords() = default;
  RawWords(const RawWords& o) { if (o.n) { acquire(o.n); memcpy(p, o.p, o.n * 8); } }
  RawWords(RawWords&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
  RawWords& operator=(RawWords o) noexcept { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); return *this; }
  ~RawWords() { release(); }
  static std::vector<std::pair<u64*, size_t>>& pool() { static thread_local struct P { std::vector<std::pair<u64*, size_t>> v; ~P() { for (auto& b : v) free(b.first); } } pl; return pl.v; }
  void acquire(size_t words) {
    release();
    auto& pl = pool();
    for (size_t i = 0; i < pl.size(); i++) if (pl[i].second >= words) { p = pl[i].first; cap = pl[i].second; pl.erase(pl.begin() + (long)i); n = words; return; }
    p = (u64*)malloc(std::max<size_t>(words, 1) * 8);
    if (!p) throw std::bad_alloc();
    cap = words; n = words;
  }
  void release() {
    if (!p) return;
    auto& pl = pool();
    if (pl.size() < 3) pl.push_back({p, cap}); else free(p);  // (a thread serialises a proof before its next member finishes one: a few blocks cover it)
    p = nullptr; n = cap = 0;
  }
  u64* data() { return p; }
  const u64* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};
struct BasefoldProof {
  std::vector<std::vector<Ext>> sumcheck_messages;
  std::vector<Digest> roots;
  std::vector<Ext> final_message;
  std::vector<BatchedQuery> queries;
  // The query section as its stream words (everything Writer::basefold emits for `queries`, from the count on), when the prover assembled it straight from the
  // device's gather buffer (pcs_batch_open_evals): `queries` is then empty. A Dense-4M batch opening has 200 x 56 opened pairs with a Merkle path each — 5.8 MB,
  // nine tenths of the proof: as vectors of CodewordQuery that was 11 200 heap allocations, a copy into them, a copy out of them into the stream and as many
  // frees per proof, 1.3 ms of the proving thread that the ~20 other members of its cohort wait behind (profiles/r06_host_work_by_launch.txt).
  RawWords queries_ser;
  std::vector<std::vector<Ext>> sumcheck_proof;
  std::vector<FieldVec> trivial_proof;
  bool is_trivial() const { return sumcheck_messages.empty() && queries.empty() && queries_ser.empty() && sumcheck_proof.empty(); }
};
enum LayerKind { L_DENSE = 0, L_REQUANT = 1, L_RELU = 2, L_CONV = 3, L_MAXPOOL = 4, L_FLATTEN = 5, L_MATMUL = 6, L_ADD = 7, L_EMBED = 8, L_POSITIONAL = 9,
                 L_MATMUL2 = 10, L_ADD2 = 11, L_CONCAT_MATMUL = 12, L_QKV = 13, L_LAYERNORM = 14, L_SOFTMAX = 15, L_MHA = 16, L_GELU = 17 };  // the two-input forms of MatMul / Add, ConcatMatMul, QKV (nodes of a model GRAPH)
struct DenseProof { IOPProof sumcheck; Ext bias_eval; std::vector<Ext> individual_claims; };
struct AddProof { Ext left_eval = ex_zero(), right_eval = ex_zero(); };  // layers/add.rs:59-63
struct PositionalProof { std::vector<Ext> sub_matrix_evals; AddProof add_proof; };  // SinglePositionalProof (transformer/positional.rs:45-55), one input
struct MatMulProof { IOPProof sumcheck; std::vector<Ext> individual_claims; bool has_bias = false; Ext bias_eval = ex_zero(); };  // layers/matrix_mul.rs:153-161 (bias_eval: Option<E>)
struct SamePolyProof { IOPProof sumcheck; std::vector<Ext> evals; };
struct ConcatMatMulProof { IOPProof sumcheck; std::vector<Ext> individual_claims; };  // layers/concat_matmul.rs:365-373: beta, left, right
// layers/transformer/qkv.rs:63-83 (declaration order); individual_claims: (input, weight) evaluations of Q, of K, of V
struct QKVProof { IOPProof sumcheck; SamePolyProof aggregation; std::vector<Ext> pre_bias_evals; std::vector<Ext> individual_claims; };
struct ActivationProof { SamePolyProof io_accumulation; LogUpProof lookup; std::vector<Commitment> commits; };
struct RequantProof { IOPProof io_accumulation; std::vector<Ext> accumulation_evals; LogUpProof clamping_lookup, shifted_lookup; std::vector<Commitment> commitments; };
struct HadamardProof { IOPProof sumcheck; std::vector<Ext> individual_claim; };  // layers/hadamard.rs:51-56
struct ConvProof {  // layers/convolution.rs:98-127, fields in declaration order
  IOPProof fft_proof, fft_proof_weights;
  std::vector<IOPProof> fft_delegation_proof, fft_delegation_proof_weights;
  IOPProof ifft_proof;
  std::vector<IOPProof> ifft_delegation_proof;
  IOPProof hadamard_proof;
  std::vector<Ext> fft_claims, fft_weight_claims, ifft_claims;
  std::vector<std::vector<Ext>> fft_delegation_claims, fft_delegation_weights_claims, ifft_delegation_claims;
  std::vector<Ext> partial_evals, hadamard_clams;
  Ext bias_claim;
  HadamardProof clearing_proof;
};
struct PoolingProof { IOPProof sumcheck; LogUpProof lookup; std::vector<Ext> zerocheck_evals; size_t variable_gap = 0; std::vector<Commitment> commitments; };  // layers/pooling.rs:60-76
// LayerNormProof (layers/transformer/layernorm.rs:644-667), fields in declaration order: the two lookups (inverse square root, range), the
// commitments of their columns, the three sumchecks, the evaluations that tie them together
struct LayerNormProof { std::vector<LogUpProof> logup_proofs; std::vector<Commitment> commitments; IOPProof accumulation_proof, io_proof, input_proof; std::vector<Ext> acc_evals, evaluations; Ext gamma_eval, beta_eval; };
// SoftmaxProof (layers/transformer/softmax.rs:102-117): lookups (exponential, range, error, zero table if any), commitments, the accumulation and
// the mask sumcheck, the evaluations exp_in, exp_out, low, high, shift, (zero_in, zero_out)*
struct SoftmaxProof { std::vector<LogUpProof> logup_proofs; std::vector<Commitment> commitments; IOPProof accumulation_proof, mask_proof; std::vector<Ext> evaluations; };
struct LayerProof { int kind = 0; DenseProof dense; MatMulProof matmul; AddProof add; PositionalProof pos; ActivationProof act; RequantProof req; ConvProof conv; PoolingProof pool; ConcatMatMulProof cmm; QKVProof qkv; LayerNormProof ln; SoftmaxProof sm;
                    ConcatMatMulProof mha_final, mha_qk; };  // MhaProof {final_mul_proof, softmax_proof, qk_proof} (transformer/mha.rs:122-128) = {mha_final, sm, mha_qk}
struct TableProof { Commitment multiplicity_commit; LogUpProof lookup; };
struct Proof {
  std::map<size_t, LayerProof> steps;
  std::vector<TableProof> table_proofs;
  BasefoldProof batch_proof;
  std::vector<BasefoldProof> trivial_proofs;
};

constexpr u64 PROOF_MAGIC = 0x31464F4F52505044ULL;  // "DPPROOF1"

// (the bulk of a proof is Merkle paths and round messages: vectors of digests / extension elements, which ARE their words in memory — appended as blocks.
// 2.1 -> ~0.5 ms for the 5.9 MB of a Dense-4M proof: the members of a cohort serialise their proofs one after the other on the cohort's thread)
static_assert(sizeof(Ext) == 16 && sizeof(Digest) == 32, "proof.h: Ext / Digest are their words");
struct Writer {
  std::vector<u64> w;
  void u(u64 v) { w.push_back(v); }
  void e(Ext x) { w.push_back(x.c0); w.push_back(x.c1); }
  void words(const u64* p, size_t n) { w.insert(w.end(), p, p + n); }
  void ve(const std::vector<Ext>& v) { u(v.size()); words((const u64*)v.data(), 2 * v.size()); }
  void d(const Digest& x) { words(x.v, 4); }
  void iop(const IOPProof& p) { ve(p.point); u(p.proofs.size()); for (auto& r : p.proofs) ve(r); }
  void claim(const Claim& c) { ve(c.point); e(c.eval); }
  void viop(const std::vector<IOPProof>& v) { u(v.size()); for (auto& x : v) iop(x); }
  void vve(const std::vector<std::vector<Ext>>& v) { u(v.size()); for (auto& x : v) ve(x); }
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
    u(q.index); u(q.path.size()); words((const u64*)q.path.data(), 4 * q.path.size());
  }
  void basefold(const BasefoldProof& p) {
    u(p.sumcheck_messages.size()); for (auto& m : p.sumcheck_messages) ve(m);
    u(p.roots.size()); words((const u64*)p.roots.data(), 4 * p.roots.size());
    ve(p.final_message);
    if (!p.queries_ser.empty()) words(p.queries_ser.data(), p.queries_ser.size());
    else {
      u(p.queries.size());
      for (auto& q : p.queries) {
        u(q.index);
        u(q.oracle_query.size()); for (auto& c : q.oracle_query) cq(c);
        u(q.commitments_query.size()); for (auto& c : q.commitments_query) cq(c);
      }
    }
    u(p.sumcheck_proof.size()); for (auto& m : p.sumcheck_proof) ve(m);
    u(p.trivial_proof.size());
    for (auto& m : p.trivial_proof) { u(m.is_ext ? 1 : 0); u(m.len()); words(m.w.data(), m.w.size()); }
  }
};
// into `out` (its capacity is reused: a caller that serialises proof after proof keeps one buffer per thread and pays no 8 MB of fresh pages per proof)
inline void serialize_proof_to(const Proof& p, std::vector<u64>& out) {
  Writer w; w.w.swap(out); w.w.clear(); w.w.reserve(size_t(1) << 20); w.u(PROOF_MAGIC); w.u(p.steps.size());
  for (auto& kv : p.steps) {
    const LayerProof& lp = kv.second;
    w.u(kv.first); w.u((u64)lp.kind);
    if (lp.kind == L_DENSE) { w.iop(lp.dense.sumcheck); w.e(lp.dense.bias_eval); w.ve(lp.dense.individual_claims); }
    else if (lp.kind == L_ADD) { w.e(lp.add.left_eval); w.e(lp.add.right_eval); }
    else if (lp.kind == L_POSITIONAL) { w.u(1); w.ve(lp.pos.sub_matrix_evals); w.e(lp.pos.add_proof.left_eval); w.e(lp.pos.add_proof.right_eval); }  // PositionalProof {proofs: one per input}
    else if (lp.kind == L_EMBED) { w.iop(lp.matmul.sumcheck); w.ve(lp.matmul.individual_claims); }  // EmbeddingsProof {sumcheck, individual_claims} (embeddings.rs:60-67)
    else if (lp.kind == L_ADD2) { w.e(lp.add.left_eval); w.e(lp.add.right_eval); }
    else if (lp.kind == L_MATMUL || lp.kind == L_MATMUL2) { w.iop(lp.matmul.sumcheck); w.ve(lp.matmul.individual_claims); w.u(lp.matmul.has_bias ? 1 : 0); if (lp.matmul.has_bias) w.e(lp.matmul.bias_eval); }
    else if (lp.kind == L_CONCAT_MATMUL) { w.iop(lp.cmm.sumcheck); w.ve(lp.cmm.individual_claims); }
    else if (lp.kind == L_QKV) { w.iop(lp.qkv.sumcheck); w.iop(lp.qkv.aggregation.sumcheck); w.ve(lp.qkv.aggregation.evals); w.ve(lp.qkv.pre_bias_evals); w.ve(lp.qkv.individual_claims); }
    else if (lp.kind == L_LAYERNORM) {
      const LayerNormProof& q = lp.ln;
      w.u(q.logup_proofs.size()); for (auto& x : q.logup_proofs) w.logup(x);
      w.u(q.commitments.size()); for (auto& c : q.commitments) w.comm(c);
      w.iop(q.accumulation_proof); w.iop(q.io_proof); w.iop(q.input_proof); w.ve(q.acc_evals); w.ve(q.evaluations); w.e(q.gamma_eval); w.e(q.beta_eval);
    }
    else if (lp.kind == L_SOFTMAX || lp.kind == L_MHA) {
      if (lp.kind == L_MHA) { w.iop(lp.mha_final.sumcheck); w.ve(lp.mha_final.individual_claims); }
      const SoftmaxProof& q = lp.sm;
      w.u(q.logup_proofs.size()); for (auto& x : q.logup_proofs) w.logup(x);
      w.u(q.commitments.size()); for (auto& c : q.commitments) w.comm(c);
      w.iop(q.accumulation_proof); w.iop(q.mask_proof); w.ve(q.evaluations);
      if (lp.kind == L_MHA) { w.iop(lp.mha_qk.sumcheck); w.ve(lp.mha_qk.individual_claims); }
    }
    else if (lp.kind == L_REQUANT) {
      w.iop(lp.req.io_accumulation); w.ve(lp.req.accumulation_evals); w.logup(lp.req.clamping_lookup); w.logup(lp.req.shifted_lookup);
      w.u(lp.req.commitments.size()); for (auto& c : lp.req.commitments) w.comm(c);
    } else if (lp.kind == L_RELU || lp.kind == L_GELU) {
      w.iop(lp.act.io_accumulation.sumcheck); w.ve(lp.act.io_accumulation.evals); w.logup(lp.act.lookup);
      w.u(lp.act.commits.size()); for (auto& c : lp.act.commits) w.comm(c);
    } else if (lp.kind == L_CONV) {
      const ConvProof& c = lp.conv;
      w.iop(c.fft_proof); w.iop(c.fft_proof_weights); w.viop(c.fft_delegation_proof); w.viop(c.fft_delegation_proof_weights);
      w.iop(c.ifft_proof); w.viop(c.ifft_delegation_proof); w.iop(c.hadamard_proof);
      w.ve(c.fft_claims); w.ve(c.fft_weight_claims); w.ve(c.ifft_claims);
      w.vve(c.fft_delegation_claims); w.vve(c.fft_delegation_weights_claims); w.vve(c.ifft_delegation_claims);
      w.ve(c.partial_evals); w.ve(c.hadamard_clams); w.e(c.bias_claim);
      w.iop(c.clearing_proof.sumcheck); w.ve(c.clearing_proof.individual_claim);
    } else if (lp.kind == L_MAXPOOL) {
      w.iop(lp.pool.sumcheck); w.logup(lp.pool.lookup); w.ve(lp.pool.zerocheck_evals); w.u(lp.pool.variable_gap);
      w.u(lp.pool.commitments.size()); for (auto& c : lp.pool.commitments) w.comm(c);
    } else DP_REQUIRE(false, DP_ERR_ARG, "serialize_proof: unknown layer kind");
  }
  w.u(p.table_proofs.size()); for (auto& tp : p.table_proofs) { w.comm(tp.multiplicity_commit); w.logup(tp.lookup); }
  w.basefold(p.batch_proof);
  w.u(p.trivial_proofs.size()); for (auto& tp : p.trivial_proofs) w.basefold(tp);
  out.swap(w.w);
}

struct Reader {
  const u64* p; size_t n, pos = 0;
  Reader(const u64* p_, size_t n_) : p(p_), n(n_) {}
  u64 u() { DP_REQUIRE(pos < n, DP_ERR_ARG, "proof stream truncated"); return p[pos++]; }
  u64 small(u64 max) { u64 v = u(); DP_REQUIRE(v <= max, DP_ERR_ARG, "proof stream: value out of range"); return v; }  // (values that are narrowed after reading)
  bool flag() { u64 v = u(); DP_REQUIRE(v <= 1, DP_ERR_ARG, "proof stream: a boolean is 0 or 1"); return v != 0; }  // (one encoding per proof: no malleable streams)
  // a length prefix is bounded by what is left of the stream (division: `v * unit_words` wraps for v >= 2^62)
  size_t len(size_t unit_words = 1) { u64 v = u(); DP_REQUIRE(v <= (n - pos) / (unit_words ? unit_words : 1), DP_ERR_ARG, "proof stream: bad length"); return (size_t)v; }
  u64 fe() { u64 v = u(); DP_REQUIRE(v < GL_P, DP_ERR_ARG, "proof stream: non-canonical field element"); return v; }
  Ext e() { u64 a = fe(); u64 b = fe(); return ex(a, b); }
  std::vector<Ext> ve() { size_t k = len(2); std::vector<Ext> v(k); for (auto& x : v) x = e(); return v; }
  Digest d() { Digest x; for (int i = 0; i < 4; i++) x.v[i] = fe(); return x; }
  IOPProof iop() { IOPProof q; q.point = ve(); size_t k = len(); q.proofs.resize(k); for (auto& r : q.proofs) r = ve(); return q; }
  Claim claim() { Claim c; c.point = ve(); c.eval = e(); return c; }
  std::vector<IOPProof> viop() { size_t k = len(); std::vector<IOPProof> v(k); for (auto& x : v) x = iop(); return v; }
  std::vector<std::vector<Ext>> vve() { size_t k = len(); std::vector<std::vector<Ext>> v(k); for (auto& x : v) x = ve(); return v; }
  LogUpProof logup() {
    LogUpProof q; size_t k = len(); q.sumcheck_proofs.resize(k); for (auto& s : q.sumcheck_proofs) s = iop();
    k = len(); q.round_evaluations.resize(k); for (auto& r : q.round_evaluations) r = ve();
    k = len(); q.output_claims.resize(k); for (auto& c : q.output_claims) c = claim();
    k = len(); q.circuit_outputs.resize(k); for (auto& c : q.circuit_outputs) c = ve();
    q.is_table = flag(); return q;
  }
  Commitment comm() { Commitment c; c.root = d(); c.num_vars = (unsigned)small(64); c.is_base = flag(); return c; }
  CodewordQuery cq() {
    CodewordQuery q; q.is_ext = flag();
    if (q.is_ext) { q.left = e(); q.right = e(); } else { q.left = ex(fe(), 0); q.right = ex(fe(), 0); }
    q.index = (size_t)u(); size_t k = len(4); q.path.resize(k); for (auto& x : q.path) x = d(); return q;
  }
  BasefoldProof basefold() {
    BasefoldProof b; size_t k = len(); b.sumcheck_messages.resize(k); for (auto& m : b.sumcheck_messages) m = ve();
    k = len(4); b.roots.resize(k); for (auto& r : b.roots) r = d();
    b.final_message = ve();
    k = len(); b.queries.resize(k);
    for (auto& q : b.queries) {
      q.index = (size_t)u();
      size_t a = len(); q.oracle_query.resize(a); for (auto& c : q.oracle_query) c = cq();
      a = len(); q.commitments_query.resize(a); for (auto& c : q.commitments_query) c = cq();
    }
    k = len(); b.sumcheck_proof.resize(k); for (auto& m : b.sumcheck_proof) m = ve();
    k = len(); b.trivial_proof.resize(k);
    for (auto& m : b.trivial_proof) { m.is_ext = flag(); size_t l = len(m.is_ext ? 2 : 1); m.w.resize(l * (m.is_ext ? 2 : 1)); for (auto& x : m.w) x = fe(); }
    return b;
  }
};
inline std::vector<u64> serialize_proof(const Proof& p) { std::vector<u64> out; serialize_proof_to(p, out); return out; }
inline Proof deserialize_proof(const u64* words, size_t n) {
  Reader r(words, n); Proof p;
  DP_REQUIRE(r.u() == PROOF_MAGIC, DP_ERR_ARG, "bad proof magic");
  size_t ns = r.len();
  for (size_t i = 0; i < ns; i++) {
    size_t id = (size_t)r.u(); LayerProof lp; lp.kind = (int)r.small(255);
    if (lp.kind == L_DENSE) { lp.dense.sumcheck = r.iop(); lp.dense.bias_eval = r.e(); lp.dense.individual_claims = r.ve(); }
    else if (lp.kind == L_ADD) { lp.add.left_eval = r.e(); lp.add.right_eval = r.e(); }
    else if (lp.kind == L_EMBED) { lp.matmul.sumcheck = r.iop(); lp.matmul.individual_claims = r.ve(); }
    else if (lp.kind == L_POSITIONAL) { DP_REQUIRE(r.u() == 1, DP_ERR_ARG, "proof stream: positional proofs per node"); lp.pos.sub_matrix_evals = r.ve(); lp.pos.add_proof.left_eval = r.e(); lp.pos.add_proof.right_eval = r.e(); }
    else if (lp.kind == L_ADD2) { lp.add.left_eval = r.e(); lp.add.right_eval = r.e(); }
    else if (lp.kind == L_MATMUL || lp.kind == L_MATMUL2) { lp.matmul.sumcheck = r.iop(); lp.matmul.individual_claims = r.ve(); u64 hb = r.u(); DP_REQUIRE(hb <= (lp.kind == L_MATMUL ? 1u : 0u), DP_ERR_ARG, "proof stream: matmul bias flag"); lp.matmul.has_bias = hb != 0; if (hb) lp.matmul.bias_eval = r.e(); }
    else if (lp.kind == L_CONCAT_MATMUL) { lp.cmm.sumcheck = r.iop(); lp.cmm.individual_claims = r.ve(); }
    else if (lp.kind == L_QKV) { lp.qkv.sumcheck = r.iop(); lp.qkv.aggregation.sumcheck = r.iop(); lp.qkv.aggregation.evals = r.ve(); lp.qkv.pre_bias_evals = r.ve(); lp.qkv.individual_claims = r.ve(); }
    else if (lp.kind == L_LAYERNORM) {
      LayerNormProof& q = lp.ln;
      size_t k = r.len(); q.logup_proofs.resize(k); for (auto& x : q.logup_proofs) x = r.logup();
      k = r.len(); q.commitments.resize(k); for (auto& c : q.commitments) c = r.comm();
      q.accumulation_proof = r.iop(); q.io_proof = r.iop(); q.input_proof = r.iop(); q.acc_evals = r.ve(); q.evaluations = r.ve(); q.gamma_eval = r.e(); q.beta_eval = r.e();
    }
    else if (lp.kind == L_SOFTMAX || lp.kind == L_MHA) {
      if (lp.kind == L_MHA) { lp.mha_final.sumcheck = r.iop(); lp.mha_final.individual_claims = r.ve(); }
      SoftmaxProof& q = lp.sm;
      size_t k = r.len(); q.logup_proofs.resize(k); for (auto& x : q.logup_proofs) x = r.logup();
      k = r.len(); q.commitments.resize(k); for (auto& c : q.commitments) c = r.comm();
      q.accumulation_proof = r.iop(); q.mask_proof = r.iop(); q.evaluations = r.ve();
      if (lp.kind == L_MHA) { lp.mha_qk.sumcheck = r.iop(); lp.mha_qk.individual_claims = r.ve(); }
    }
    else if (lp.kind == L_REQUANT) {
      lp.req.io_accumulation = r.iop(); lp.req.accumulation_evals = r.ve(); lp.req.clamping_lookup = r.logup(); lp.req.shifted_lookup = r.logup();
      size_t k = r.len(); lp.req.commitments.resize(k); for (auto& c : lp.req.commitments) c = r.comm();
    } else if (lp.kind == L_RELU || lp.kind == L_GELU) {
      lp.act.io_accumulation.sumcheck = r.iop(); lp.act.io_accumulation.evals = r.ve(); lp.act.lookup = r.logup();
      size_t k = r.len(); lp.act.commits.resize(k); for (auto& c : lp.act.commits) c = r.comm();
    } else if (lp.kind == L_CONV) {
      ConvProof& c = lp.conv;
      c.fft_proof = r.iop(); c.fft_proof_weights = r.iop(); c.fft_delegation_proof = r.viop(); c.fft_delegation_proof_weights = r.viop();
      c.ifft_proof = r.iop(); c.ifft_delegation_proof = r.viop(); c.hadamard_proof = r.iop();
      c.fft_claims = r.ve(); c.fft_weight_claims = r.ve(); c.ifft_claims = r.ve();
      c.fft_delegation_claims = r.vve(); c.fft_delegation_weights_claims = r.vve(); c.ifft_delegation_claims = r.vve();
      c.partial_evals = r.ve(); c.hadamard_clams = r.ve(); c.bias_claim = r.e();
      c.clearing_proof.sumcheck = r.iop(); c.clearing_proof.individual_claim = r.ve();
    } else if (lp.kind == L_MAXPOOL) {
      lp.pool.sumcheck = r.iop(); lp.pool.lookup = r.logup(); lp.pool.zerocheck_evals = r.ve(); lp.pool.variable_gap = (size_t)r.u();
      size_t k = r.len(); lp.pool.commitments.resize(k); for (auto& c : lp.pool.commitments) c = r.comm();
    } else DP_REQUIRE(false, DP_ERR_ARG, "proof stream: unknown layer kind");
    p.steps[id] = lp;
  }
  size_t nt = r.len(); p.table_proofs.resize(nt);
  for (auto& tp : p.table_proofs) { tp.multiplicity_commit = r.comm(); tp.lookup = r.logup(); }
  p.batch_proof = r.basefold();
  size_t ntr = r.len(); p.trivial_proofs.resize(ntr); for (auto& tp : p.trivial_proofs) tp = r.basefold();
  DP_REQUIRE(r.pos == n, DP_ERR_ARG, "proof stream: trailing words");
  return p;
}

}  // namespace dp
