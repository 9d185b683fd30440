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
#include <cstring>
#include <cmath>

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
enum LayerKind { L_DENSE = 0, L_REQUANT = 1, L_RELU = 2, L_CONV = 3, L_MAXPOOL = 4, L_FLATTEN = 5, L_MATMUL = 6, L_ADD = 7, L_EMBED = 8, L_POSITIONAL = 9,
                 L_MATMUL2 = 10, L_ADD2 = 11, L_CONCAT_MATMUL = 12, L_QKV = 13, L_LAYERNORM = 14, L_SOFTMAX = 15, L_MHA = 16, L_GELU = 17 };
// An edge of the model graph (layers/provable/mod.rs:195-229, Edge): output `index` of node `node`, or input tensor `index` of the model (node < 0)
struct Wire { int node = -1; int index = 0; };
struct Layer {
  LayerKind kind;
  // where the node's inputs come from (Node::inputs, provable/mod.rs:204-211); empty = the chain form: output 0 of the previous node, the
  // model's input for node 0. A node only reads nodes with smaller ids.
  std::vector<Wire> inputs;
  // matmul2 (layers/matrix_mul.rs, MatMul::new(Input, Input)): left [s][nrows] times right [nrows][ncols] ([ncols][nrows] with transpose_b), no bias
  // add2 (layers/add.rs, Add::new()): out = add_left * a + add_right * b for two inputs of one length
  // qkv (layers/transformer/qkv.rs): weights = W_q | W_k | W_v, each [nrows][ncols] row major, bias = b_q | b_k | b_v, each [ncols]; the input a
  // [s][nrows] matrix, the three outputs X W + b
  // concat matmul (layers/concat_matmul.rs): two rank-3 inputs of shapes cm_a / cm_b; cm_left / cm_right = (concat, mat_mul, output) dimension
  // of each (InputMatrixDimensions); cm_perm = the optional permutation of the [concat][rows][cols] result (empty: none)
  size_t cm_a[3] = {0, 0, 0}, cm_b[3] = {0, 0, 0};
  int cm_left[3] = {0, 2, 1}, cm_right[3] = {0, 1, 2};
  std::vector<int> cm_perm;
  // dense (padded to powers of two). matmul (layers/matrix_mul.rs, MatMul::new_constant: Input x Weight [+ bias]): the constant RIGHT
  // matrix is [nrows][ncols] row major, the input a row-major [s][nrows] matrix, bias [ncols] or empty
  size_t nrows = 0, ncols = 0;
  // positional (layers/transformer/positional.rs, Positional::Learned): weights = the [nrows = positions][ncols = embedding size] table,
  // out = add_left * x + add_right * table[0 .. tokens) for a [tokens][ncols] input, tokens <= nrows
  // embeddings (layers/transformer/embeddings.rs): the FIRST layer of a model; the input is a vector of token ids, weights the
  // [nrows = vocabulary][ncols = embedding size] table, the output [tokens][ncols]
  // add (layers/add.rs, Add::new_with(operand)): out = add_left * x + add_right * operand, the operand (a constant tensor as long as the
  // input, e.g. learned positional embeddings) in `weights`; the multipliers are QuantInfo::left/right_multiplier (add.rs:271-283)
  int64_t add_left = 1, add_right = 1;
  bool transpose_b = false;  // matmul, Config::TransposeB (matrix_mul.rs:36-39): the constant matrix is stored as [ncols][nrows] and used transposed
  std::vector<int64_t> weights, bias;  // dense: row major, bias padded to nrows; conv: filter [kw][kx][real_nw][real_nw], bias [kw]
  // conv (layers/convolution.rs:52-83, tensor.rs:409-431): padded filter count kw, padded input channels kx, padded
  // kernel side real_nw, padded input side nw (= n_x of fft_conv); unp_out = conv2d_shape of the UNPADDED tensors
  size_t kw = 0, kx = 0, real_nw = 0, nw = 0;
  size_t unp_out[3] = {0, 0, 0};
  // maxpool (layers/pooling.rs): padded input shape [c, h, w]
  size_t pin[3] = {0, 0, 0};
  size_t filter_size() const { return nw * nw; }
  // layernorm (layers/transformer/layernorm.rs:74-101, QuantisedLayerNormData): gamma in `weights`, beta in `bias`, both as long as the (padded)
  // normalisation dimension; dim_size = N, the multiplier of the inverse-square-root input, the f32 bits of the rescaled epsilon, the bits that
  // are shifted away and range checked, log2 of the scalar of their most significant chunk
  size_t ln_dim_size = 0; int64_t ln_multiplier = 0; uint32_t ln_eps_bits = 0; unsigned ln_range_check_bits = 0, ln_top_chunk_scalar_log = 0;
  // softmax (layers/transformer/softmax.rs:66-99, QuantisedSoftmaxData / SoftmaxCtx :1153-1169) over the last dimension of a padded
  // [sm_shape[0]][sm_shape[1]][sm_shape[2]] tensor with a causal mask (the last two dimensions are equal): the multiplier that brings the input to
  // the scale 2^24, the f32 bits of 1 / temperature and of the input scale (for the row shifts the prover computes in floating point), the
  // exponential table (2^sm_table_size entries, zero from sm_bkm on), the zero tables of the bits above it, the allowable error of a row sum
  int64_t sm_scalar = 0, sm_bkm = 0, sm_allowable_error = 0; uint32_t sm_temp_bits = 0, sm_in_scale_bits = 0; unsigned sm_table_size = 0, sm_zero_chunks = 0, sm_zero_vars = 0;
  size_t sm_shape[3] = {0, 0, 0};
  // mha (layers/transformer/mha.rs:133-186, Mha::new): ONE node over the three inputs Q, K, V, each a padded [seq][heads * head_dim] matrix read as
  // [seq][heads][head_dim] (inputs_reshape); qk = ConcatMatMul (1,2,0) x (1,2,0) -> [heads][seq][seq]; the Softmax above (sm_* fields, sm_shape
  // unused: it is [heads][seq][seq]) directly on the products; final_mul = ConcatMatMul (0,2,1) x (1,0,2) permuted (1,0,2) -> [seq][heads][head_dim]
  size_t mha_shape[3] = {0, 0, 0};  // seq, heads, head_dim (padded)
  // gelu (layers/activation.rs:559-572, GELUQuantData): the integer the input is multiplied by before the table is consulted,
  // round(2^12 * input scale) (GELU::quantize, :629-659); the table runs over [-2^(7 + ceil_log2(m)), 2^(7 + ceil_log2(m)))
  int64_t gelu_multiplier = 0;
  unsigned right_shift = 0, fp_scale = 0, intermediate_bit_size = 0;  // requant (requant.rs:46-73)
  int64_t fixed_point_multiplier = 0;
  unsigned shift() const { return fp_scale + right_shift; }
  unsigned clamping_size() const { return intermediate_bit_size + ceil_log2((size_t)fixed_point_multiplier) - shift(); }  // requant.rs:485-488
};
// input_lens: the model's input tensors (empty: one of input_len; otherwise input_len is their sum and the input vector their concatenation);
// outputs: the model's output tensors (empty: output 0 of the last node), concatenated in this order
struct Model { size_t input_len = 0; std::vector<Layer> layers; std::vector<size_t> input_lens; std::vector<Wire> outputs; };
static inline size_t n_outputs(const Layer& l) { return l.kind == L_QKV ? 3 : 1; }
static inline size_t n_inputs(const Layer& l) { return l.kind == L_MHA ? 3 : (l.kind == L_MATMUL2 || l.kind == L_ADD2 || l.kind == L_CONCAT_MATMUL) ? 2 : 1; }
// the three sub-layers of an Mha node (Mha::new, mha.rs:147-186)
static inline Layer mha_qk_layer(const Layer& l) {
  Layer s; s.kind = L_CONCAT_MATMUL;
  for (int d = 0; d < 3; d++) s.cm_a[d] = s.cm_b[d] = l.mha_shape[d];
  const int dims[3] = {1, 2, 0}; for (int d = 0; d < 3; d++) s.cm_left[d] = s.cm_right[d] = dims[d];
  return s;
}
static inline Layer mha_softmax_layer(const Layer& l) {
  Layer s; s.kind = L_SOFTMAX;
  s.sm_scalar = l.sm_scalar; s.sm_bkm = l.sm_bkm; s.sm_allowable_error = l.sm_allowable_error; s.sm_temp_bits = l.sm_temp_bits; s.sm_in_scale_bits = l.sm_in_scale_bits;
  s.sm_table_size = l.sm_table_size; s.sm_zero_chunks = l.sm_zero_chunks; s.sm_zero_vars = l.sm_zero_vars;
  s.sm_shape[0] = l.mha_shape[1]; s.sm_shape[1] = s.sm_shape[2] = l.mha_shape[0];
  return s;
}
static inline Layer mha_final_layer(const Layer& l) {
  Layer s; s.kind = L_CONCAT_MATMUL;
  s.cm_a[0] = l.mha_shape[1]; s.cm_a[1] = s.cm_a[2] = l.mha_shape[0];
  for (int d = 0; d < 3; d++) s.cm_b[d] = l.mha_shape[d];
  const int dl[3] = {0, 2, 1}, dr[3] = {1, 0, 2}; for (int d = 0; d < 3; d++) { s.cm_left[d] = dl[d]; s.cm_right[d] = dr[d]; }
  s.cm_perm = {1, 0, 2};
  return s;
}
static inline std::vector<Wire> node_inputs(const Model& m, size_t id) {
  if (!m.layers[id].inputs.empty()) return m.layers[id].inputs;
  Wire w; if (id > 0) { w.node = (int)id - 1; w.index = 0; }
  return {w};
}
static inline std::vector<Wire> model_outputs(const Model& m) {
  if (!m.outputs.empty()) return m.outputs;
  Wire w; w.node = (int)m.layers.size() - 1; w.index = 0;
  return {w};
}
static inline std::vector<size_t> model_input_lens(const Model& m) { return m.input_lens.empty() ? std::vector<size_t>{m.input_len} : m.input_lens; }
// the single consumer of an output wire (claims_for_node, provable/mod.rs:235-270: "only one edge per output wire"): input `pos` of node `node`,
// or output `pos` of the model (node < 0)
struct WireUse { int node = -1; int pos = 0; bool found = false; };
static inline WireUse wire_use(const Model& m, int node, int index) {
  WireUse u; size_t uses = 0;
  for (size_t id = 0; id < m.layers.size(); id++) { std::vector<Wire> in = node_inputs(m, id); for (size_t q = 0; q < in.size(); q++) if (in[q].node == node && in[q].index == index) { u.node = (int)id; u.pos = (int)q; u.found = true; uses++; } }
  std::vector<Wire> outs = model_outputs(m);
  for (size_t q = 0; q < outs.size(); q++) if (outs[q].node == node && outs[q].index == index) { u.node = -1; u.pos = (int)q; u.found = true; uses++; }
  if (uses != 1) throw std::runtime_error("model graph: every output wire needs exactly one consumer");
  return u;
}
// NodeIterator<_, false> (model/iterator.rs:152-185): the smallest unvisited id all of whose consumers have been visited, again and again
static inline std::vector<size_t> backward_order(const Model& m) {
  std::vector<bool> done(m.layers.size(), false);
  std::vector<size_t> order;
  for (size_t step = 0; step < m.layers.size(); step++) {
    bool found = false;
    for (size_t id = 0; id < m.layers.size() && !found; id++) {
      if (done[id]) continue;
      bool ready = true;
      for (size_t j = 0; j < n_outputs(m.layers[id]) && ready; j++) { WireUse u = wire_use(m, (int)id, (int)j); if (u.node >= 0 && !done[(size_t)u.node]) ready = false; }
      if (ready) { done[id] = true; order.push_back(id); found = true; }
    }
    if (!found) throw std::runtime_error("model graph: cycle");
  }
  return order;
}
// Tensor::permute3d (tensor.rs:1769-1800): dimension d of the result is dimension order[d] of the input
template <class T> static inline std::vector<T> permute3d(const std::vector<T>& x, const size_t shape[3], const int order[3], size_t out_shape[3]) {
  for (int d = 0; d < 3; d++) out_shape[d] = shape[order[d]];
  std::vector<T> o(x.size());
  for (size_t i = 0; i < shape[0]; i++) for (size_t j = 0; j < shape[1]; j++) for (size_t k = 0; k < shape[2]; k++) {
    const size_t pos[3] = {i, j, k};
    o[(pos[order[0]] * out_shape[1] + pos[order[1]]) * out_shape[2] + pos[order[2]]] = x[(i * shape[1] + j) * shape[2] + k];
  }
  return o;
}
// InputMatrixDimensions::compute_permutation (concat_matmul.rs:101-114): the order that brings (concat, mat_mul, output) = dims to `expected`;
// false: already there
static inline bool cm_permutation(const int dims[3], const int expected[3], int order[3]) {
  if (dims[0] == expected[0] && dims[1] == expected[1] && dims[2] == expected[2]) return false;
  order[expected[0]] = dims[0]; order[expected[2]] = dims[2]; order[expected[1]] = dims[1];
  return true;
}
constexpr int CM_EXPECTED_LEFT[3] = {0, 2, 1}, CM_EXPECTED_RIGHT[3] = {0, 1, 2};  // (concat, mat_mul, output) the per-chunk products need (concat_matmul.rs:443-465)
// shape of ConcatMatMul's output (MatrixPermutations::output_shapes, concat_matmul.rs:206-237)
static inline void cm_output_shape(const Layer& l, size_t out[3]) {
  size_t r[3] = {l.cm_a[l.cm_left[0]], l.cm_a[l.cm_left[2]], l.cm_b[l.cm_right[2]]};
  for (int d = 0; d < 3; d++) out[d] = l.cm_perm.empty() ? r[d] : r[l.cm_perm[d]];
}

struct TableType {  // lookup/context.rs:55-72 (derive Ord: Relu < GELU < Range < Clamping(n) < Softmax < ErrorTable < ZeroTable < InverseSQRT)
  int kind;  // 0 Relu, 1 GELU (aux2 = GELUQuantData::multiplier, size = log2 of the table length; min / max follow from the multiplier, so derive(Ord) on
             // (multiplier, min, max) is the order of the multipliers), 2 Range, 3 Clamping, 4 Softmax, 5 ErrorTable, 6 ZeroTable, 7 InverseSQRT
  unsigned size;      // Clamping: bits; Softmax: table_size; ZeroTable: bits; InverseSQRT: range_check_bits
  uint32_t aux = 0;   // InverseSQRT: eps_bits (InverseSQRTTableData derives Ord on (eps_bits, range_check_bits)); Softmax: float_bits
  int64_t aux2 = 0;   // Softmax: bkm (SoftmaxTableData orders by (float_bits, table_size, bkm)); ErrorTable(4096, allowable_error): the error
  bool operator<(const TableType& o) const { return kind != o.kind ? kind < o.kind : aux != o.aux ? aux < o.aux : size != o.size ? size < o.size : aux2 < o.aux2; }
  bool operator==(const TableType& o) const { return kind == o.kind && size == o.size && aux == o.aux && aux2 == o.aux2; }
  unsigned multiplicity_poly_vars() const { return kind == 1 || kind == 3 || kind == 4 || kind == 6 ? size : kind == 5 ? ceil_log2((size_t)(2 * aux2)) : kind == 7 ? 2 * (BIT_LEN - 1) + 1 : BIT_LEN; }  // context.rs:481-492
  const char* challenge_label() const { return kind == 0 ? "Relu" : kind == 1 ? "GELU" : kind == 3 ? "Clamping" : kind == 4 ? "Softmax" : kind == 6 ? "Zero" : kind == 7 ? "InverseSQRT" : nullptr; }
  bool has_committed_column() const { return kind == 1 || kind == 7 || kind == 4 || kind == 5; }  // committed_columns (context.rs:495-545): the output column (ErrorTable: its only column)
};
constexpr unsigned SM_LOG_SCALE = 24; constexpr int64_t SM_OUT_ONE = 1 << 12;  // softmax.rs:56-60
// SoftmaxTableData::table_output (lookup/context.rs:111-122), f32 exp as there
static inline int64_t softmax_table_output(uint32_t temp_bits, int64_t bkm, int64_t j) {
  float temp; std::memcpy(&temp, &temp_bits, 4);
  const int64_t prod = (int64_t(1) << (SM_LOG_SCALE - 8)) * j;
  if (prod >= bkm) return 0;
  const float e = std::exp((float)(-prod) / ((float)(1u << SM_LOG_SCALE) * temp));
  return (int64_t)std::round(e * (float)SM_OUT_ONE);
}
constexpr unsigned LOG_LAYERNORM_SCALE_FACTOR = 24, LOG_LAYERNORM_OUTPUT_SCALE_FACTOR = 10;  // layernorm.rs:61-65
// InverseSQRTTableData::table_output (lookup/context.rs:147-157): f32 arithmetic, `as Element` of a NaN (negative argument) is 0
static inline int64_t inv_sqrt_table_output(uint32_t eps_bits, unsigned range_check_bits, int64_t j) {
  float eps; std::memcpy(&eps, &eps_bits, 4);
  const int64_t shifted = j * (int64_t(1) << range_check_bits);
  const float arg = (float)shifted / (float)(1u << LOG_LAYERNORM_SCALE_FACTOR) + eps;
  const float out = 1.0f / std::sqrt(arg);
  const float r = std::round(out * (float)(1u << LOG_LAYERNORM_OUTPUT_SCALE_FACTOR));
  if (std::isnan(r)) return 0;
  if (r >= 9.2e18f) return INT64_MAX;
  if (r <= -9.2e18f) return INT64_MIN;
  return (int64_t)r;
}
static inline int64_t relu_apply(int64_t x) { return x < 0 ? 0 : x; }
// gelu_float (activation.rs:623-627) and GELUQuantData::table_output (:582-588), single precision throughout. Each product and sum is its own statement on a
// float the optimiser has to store: the Rust code rounds after every operation, a fused multiply-add here would not.
static inline float gelu_f32(float x) {
  volatile float cube = x * x; cube = cube * x;
  volatile float poly = 0.044715f * cube; poly = x + poly;
  volatile float root = 2.0f / 3.14159265358979323846f; root = std::sqrt((float)root);
  volatile float arg = root * poly;
  volatile float t = std::tanh((float)arg); t = 1.0f + t;
  volatile float half = 0.5f * x;
  return half * t;
}
static inline int64_t gelu_table_output(int64_t scaled_input) {
  const float fin = (float)scaled_input / 4096.0f;  // GELU_SCALE_FACTOR = 1 << 12 (:42-43)
  volatile float prod = gelu_f32(fin) * (float)QMAX;
  return (int64_t)std::round((float)prod);
}
static inline unsigned gelu_table_log2(int64_t multiplier) { return 7 + ceil_log2((size_t)multiplier) + 1; }
static inline int64_t gelu_apply(int64_t multiplier, int64_t x) {  // GELU::apply (:661-671); scaled == max is let through there although the table ends one before it
  const int64_t scaled = x * multiplier, edge = int64_t(1) << (gelu_table_log2(multiplier) - 1);
  if (scaled < -edge || scaled >= edge) throw std::runtime_error("gelu: Input out of range");
  return gelu_table_output(scaled);
}
static inline int64_t clamp_q(int64_t x) { return x < QMIN ? QMIN : x > QMAX ? QMAX : x; }
// get_merged_table_column (lookup/context.rs:158-296)
static inline void table_columns(const TableType& tt, std::vector<int64_t>& merged, std::vector<std::vector<u64>>& cols) {
  merged.clear(); cols.clear();
  if (tt.kind == 0) {
    cols.resize(2);
    for (int64_t i = QMIN - 1; i <= QMAX; i++) { int64_t o = relu_apply(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  } else if (tt.kind == 1) {  // context.rs:163-182, GELUQuantData::table (:579-581): the rows min .. max - 1
    cols.resize(2);
    const int64_t edge = int64_t(1) << (tt.size - 1);
    for (int64_t i = -edge; i < edge; i++) { const int64_t o = gelu_table_output(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  } else if (tt.kind == 2) {
    cols.resize(1);
    for (int64_t i = 0; i < (int64_t(1) << BIT_LEN); i++) { merged.push_back(i); cols[0].push_back(from_i64(i)); }
  } else if (tt.kind == 4) {  // context.rs:232-247
    cols.resize(2);
    for (int64_t j = 0; j < (int64_t(1) << tt.size); j++) { int64_t o = softmax_table_output(tt.aux, tt.aux2, j); merged.push_back(j + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(j)); cols[1].push_back(from_i64(o)); }
  } else if (tt.kind == 5) {  // context.rs:248-264: quant_one - error ..= quant_one + error, cut / zero padded to 2^ceil_log2(2 error) entries
    cols.resize(1);
    const size_t n = size_t(1) << ceil_log2((size_t)(2 * tt.aux2));
    for (int64_t v = SM_OUT_ONE - tt.aux2; v <= SM_OUT_ONE + tt.aux2 && merged.size() < n; v++) { merged.push_back(v); cols[0].push_back(from_i64(v)); }
    while (merged.size() < n) { merged.push_back(0); cols[0].push_back(0); }
  } else if (tt.kind == 6) {  // context.rs:265-279
    cols.resize(2);
    for (int64_t i = 0; i < (int64_t(1) << tt.size); i++) { int64_t o = i != 0 ? 0 : 1; merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  } else if (tt.kind == 7) {  // context.rs:280-294
    cols.resize(2);
    int64_t mx = int64_t(1) << (2 * (BIT_LEN - 1));
    for (int64_t i = -mx; i < mx; i++) { int64_t o = inv_sqrt_table_output(tt.aux, tt.size, i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  } else {
    cols.resize(2);
    int64_t mx = int64_t(1) << (tt.size - 1);
    for (int64_t i = -mx; i < mx; i++) { int64_t o = clamp_q(i); merged.push_back(i + o * COLUMN_SEPARATOR); cols[0].push_back(from_i64(i)); cols[1].push_back(from_i64(o)); }
  }
}
static inline TableType layernorm_table(const Layer& l) { TableType t{7, l.ln_range_check_bits}; t.aux = l.ln_eps_bits; return t; }

// ------------------------------------------------------------------ inference (layers' Evaluate impls)
// ConvData (tensor.rs:326-372): everything the FFT convolution computes on the way, kept for the prover
struct ConvData {
  std::vector<std::vector<E>> input, input_fft, prod, output;
  std::vector<int64_t> output_as_element;  // conv output AFTER the bias, BEFORE clearing the garbage (convolution.rs:311-316)
};
// per node input / output tensors; in2 = the second input of a two-input node, out_more = the outputs after the first (QKV: K, V)
// MhaData (mha.rs:45-52): the products Q K^T (the input of the softmax) and the probabilities (the left input of final_mul); the output of final_mul is
// the node's output (final_reshape does not move anything)
struct MhaData { std::vector<int64_t> softmax_in, softmax_out; };
struct Trace {
  std::vector<std::vector<int64_t>> in, out, in2, in3; std::vector<std::vector<std::vector<int64_t>>> out_more; std::vector<ConvData> conv; std::map<size_t, MhaData> mha;
  const std::vector<int64_t>& output(size_t node, size_t index) const { return index == 0 ? out[node] : out_more[node][index - 1]; }
};

// get_root_of_unity (tensor.rs:220-231)
static inline E get_root_of_unity(unsigned n) {
  u64 rou = two_adic_generator(32);
  for (unsigned i = 0; i < 32 - n; i++) rou = fmul(rou, rou);
  return e_from(rou);
}
// fft (tensor.rs:261-323): flag false -> FFT, true -> iFFT
static inline void fft(std::vector<E>& v, bool flag) {
  size_t n = v.size(); unsigned logn = ceil_log2(n);
  std::vector<size_t> rev(n, 0); std::vector<E> w(n, e_zero());
  for (size_t i = 1; i < n; i++) rev[i] = (rev[i >> 1] >> 1) | ((i & 1) << (logn - 1));
  w[0] = e_one();
  if (n > 1) { w[1] = get_root_of_unity(logn); if (flag) w[1] = einv(w[1]); }
  for (size_t i = 2; i < n; i++) w[i] = emul(w[i - 1], w[1]);
  for (size_t i = 0; i < n; i++) if (rev[i] < i) std::swap(v[i], v[rev[i]]);
  for (size_t i = 2; i <= n; i <<= 1) {
    size_t half = i >> 1;
    for (size_t c = 0; c < n; c += i)
      for (size_t k = 0; k < half; k++) { E u = v[c + k], l = emul(v[c + k + half], w[n / i * k]); v[c + k] = eadd(u, l); v[c + k + half] = esub(u, l); }
  }
  if (flag) { E ilen = einv(e_from_u64(n)); for (auto& x : v) x = emul(x, ilen); }
}
// index_w / index_wf (tensor.rs:236-254, convolution.rs:1535-1550)
static inline std::vector<E> index_wf(const std::vector<E>& w, size_t n_real, size_t n, size_t output_len) {
  std::vector<E> o(output_len, e_zero());
  for (size_t idx = 0; idx < output_len; idx++) { size_t i = idx / n, j = idx % n; if (i < n_real && j < n_real) o[idx] = w[i * n_real + j]; }
  return o;
}
// IntoElement::to_element (quantization/mod.rs:225-242)
static inline int64_t to_element(E x) { u64 e = x.c0; return e <= (P >> 1) ? (int64_t)e : -(int64_t)(P - e); }
// Tensor::fft_conv (tensor.rs:458-523) + Convolution::op (convolution.rs:303-336)
static inline std::vector<int64_t> conv_op(const Layer& l, const std::vector<int64_t>& x, ConvData& cd) {
  size_t n_x = l.nw, nn = n_x * n_x, new_n = 2 * nn, fsz = l.real_nw * l.real_nw;
  if (x.size() != l.kx * nn) throw std::runtime_error("conv: input size mismatch");
  cd = ConvData();
  for (size_t j = 0; j < l.kx; j++) {
    std::vector<E> xin(nn);
    for (size_t t = 0; t < nn; t++) xin[t] = e_from_i64(x[j * nn + nn - 1 - t]);  // chunk reversed
    std::vector<E> xf = xin; xf.resize(new_n, e_zero());
    fft(xf, false);
    cd.input.push_back(xin); cd.input_fft.push_back(xf);
  }
  std::vector<std::vector<E>> out(l.kw, std::vector<E>(2 * l.nw * l.nw, e_zero()));
  for (size_t i = 0; i < l.kw; i++)
    for (size_t j = 0; j < l.kx; j++) {
      std::vector<E> wr(fsz);
      for (size_t k = 0; k < fsz; k++) wr[k] = e_from_i64(l.weights[i * l.kx * fsz + j * fsz + k]);
      std::vector<E> wf = index_wf(wr, l.real_nw, l.nw, 2 * l.nw * l.nw);
      fft(wf, false);
      for (size_t k = 0; k < out[i].size(); k++) out[i][k] = eadd(out[i][k], emul(cd.input_fft[j][k], wf[k]));
    }
  cd.prod = out;
  for (auto& e : out) fft(e, true);
  cd.output = out;
  std::vector<int64_t> o(l.kw * nn);
  for (size_t i = 0; i < l.kw; i++) for (size_t p = 0; p < nn; p++) o[i * nn + p] = to_element(out[i][nn - 1 - p]);  // index_u
  for (size_t i = 0; i < l.kw; i++) for (size_t p = 0; p < nn; p++) o[i * nn + p] += l.bias[i];                       // add_bias
  cd.output_as_element = o;
  std::vector<int64_t> cleared = o;  // clear_garbage (convolution.rs:1484-1506)
  for (size_t i = 0; i < l.kw; i++) for (size_t j = 0; j < n_x; j++) for (size_t k = 0; k < n_x; k++)
    if (!(i < l.unp_out[0] && j < l.unp_out[1] && k < l.unp_out[2])) cleared[i * nn + j * n_x + k] = 0;
  return cleared;
}
// new_clearing_tensor (convolution.rs:1508-1529)
static inline std::vector<int64_t> new_clearing_tensor(const size_t og[3], const size_t padded[3]) {
  std::vector<int64_t> d(padded[0] * padded[1] * padded[2], 0);
  for (size_t i = 0; i < padded[0]; i++) for (size_t j = 0; j < padded[1]; j++) for (size_t k = 0; k < padded[2]; k++)
    if (i < og[0] && j < og[1] && k < og[2]) d[i * padded[1] * padded[2] + j * padded[2] + k] = 1;
  return d;
}
// Tensor::maxpool2d with kernel = stride = 2 (tensor.rs:1335-1383)
static inline std::vector<int64_t> maxpool_op(const Layer& l, const std::vector<int64_t>& x) {
  size_t c = l.pin[0], h = l.pin[1], w = l.pin[2], oh = h / 2, ow = w / 2;
  if (x.size() != c * h * w) throw std::runtime_error("maxpool: input size mismatch");
  std::vector<int64_t> o(c * oh * ow);
  for (size_t n = 0; n < c; n++) for (size_t i = 0; i < oh; i++) for (size_t j = 0; j < ow; j++) {
    int64_t m = x[n * h * w + 2 * i * w + 2 * j];
    for (size_t ki = 0; ki < 2; ki++) for (size_t kj = 0; kj < 2; kj++) m = std::max(m, x[n * h * w + (2 * i + ki) * w + 2 * j + kj]);
    o[n * oh * ow + i * ow + j] = m;
  }
  return o;
}
// Maxpool2D::compute_polys (pooling.rs:686-767): output - input at the four kernel offsets, in the order
// (dy,dx) = (0,0), (1,0), (0,1), (1,1), each laid out like the pooled output
static inline std::vector<std::vector<int64_t>> maxpool_diff_polys(const Layer& l, const std::vector<int64_t>& x, const std::vector<int64_t>& out) {
  size_t c = l.pin[0], h = l.pin[1], w = l.pin[2], oh = h / 2, ow = w / 2;
  std::vector<std::vector<int64_t>> cols(4, std::vector<int64_t>(out.size()));
  static const size_t dy[4] = {0, 1, 0, 1}, dx[4] = {0, 0, 1, 1};
  for (int q = 0; q < 4; q++)
    for (size_t n = 0; n < c; n++) for (size_t i = 0; i < oh; i++) for (size_t j = 0; j < ow; j++) {
      size_t oi = n * oh * ow + i * ow + j;
      cols[q][oi] = out[oi] - x[n * h * w + (2 * i + dy[q]) * w + 2 * j + dx[q]];
    }
  return cols;
}
// LayerNorm::evaluate on Elements (layernorm.rs:394-470); LayerNormData = what the prover needs later
struct LayerNormData { std::vector<int64_t> lookup_input, lookup_output, range_check; };
static inline std::vector<int64_t> layernorm_op(const Layer& l, const std::vector<int64_t>& x, LayerNormData* d) {
  const size_t fd = l.weights.size();
  if (!fd || (fd & (fd - 1)) || l.bias.size() != fd || x.size() % fd || !l.ln_dim_size || l.ln_dim_size > fd) throw std::runtime_error("layernorm: shapes");
  const int64_t n = (int64_t)l.ln_dim_size, mask = (int64_t(1) << l.ln_range_check_bits) - 1, tmax = int64_t(1) << (2 * (BIT_LEN - 1));
  std::vector<int64_t> o(x.size());
  for (size_t c = 0; c < x.size() / fd; c++) {
    int64_t sq = 0, sum = 0;
    for (size_t i = 0; i < fd; i++) { sq += x[c * fd + i] * x[c * fd + i]; sum += x[c * fd + i]; }
    const int64_t full = n * l.ln_multiplier * sq - l.ln_multiplier * sum * sum;
    const int64_t in = full >> l.ln_range_check_bits;
    if (in < -tmax || in >= tmax) throw std::runtime_error("layernorm: the inverse square root input leaves its table");
    const int64_t inv = inv_sqrt_table_output(l.ln_eps_bits, l.ln_range_check_bits, in);
    if (d) { d->lookup_input.push_back(in); d->lookup_output.push_back(inv); d->range_check.push_back(full & mask); }
    for (size_t i = 0; i < fd; i++) o[c * fd + i] = l.weights[i] * (n * x[c * fd + i] - sum) * inv + l.bias[i];
  }
  return o;
}
static inline TableType softmax_table(const Layer& l) { TableType t{4, l.sm_table_size}; t.aux = l.sm_temp_bits; t.aux2 = l.sm_bkm; return t; }
static inline TableType softmax_error_table(const Layer& l) { TableType t{5, 0}; t.aux2 = l.sm_allowable_error; return t; }
// Softmax::evaluate on Elements (softmax.rs:455-566) with calculate_shift_data (:250-320) and the causal AttentionMask (:1590-1750)
struct SoftmaxData {
  std::vector<int64_t> shift, shifted_input, tril, bias, low, high, exp_in, exp_out;
  std::vector<std::vector<int64_t>> zero_in, zero_out;
};
static inline std::vector<int64_t> softmax_op(const Layer& l, const std::vector<int64_t>& x, SoftmaxData* out) {
  const size_t C = l.sm_shape[0], R = l.sm_shape[1], K = l.sm_shape[2];
  if (!C || !R || R != K || x.size() != C * R * K) throw std::runtime_error("softmax: shapes");
  float inv_temp, in_scale; std::memcpy(&inv_temp, &l.sm_temp_bits, 4); std::memcpy(&in_scale, &l.sm_in_scale_bits, 4);
  SoftmaxData d;
  const int64_t neg_inf = -(((l.sm_bkm >> 16) + 1) << 16);
  for (size_t i = 0; i < C * R; i++) {  // the shift of every row: -ln(sum of the exponentials of its unmasked entries), in the scale 2^24
    const int64_t* row = &x[i * K]; const size_t take = i % R + 1;
    if (i % R == 0) { d.shift.push_back(-row[0] * l.sm_scalar); continue; }
    int64_t mx = row[0]; for (size_t j = 1; j < take; j++) mx = std::max(mx, row[j]);
    float sum = 0.0f;
    for (size_t j = 0; j < take; j++) sum += std::exp(((float)(row[j] - mx) * in_scale) / inv_temp);
    const float log_sum = std::log(sum);
    d.shift.push_back(-(int64_t)std::round((float)(1u << SM_LOG_SCALE) * inv_temp * log_sum) - mx * l.sm_scalar);
  }
  d.tril.resize(x.size()); d.bias.resize(x.size()); d.shifted_input.resize(x.size());
  for (size_t i = 0; i < C * R; i++) for (size_t j = 0; j < K; j++) {
    const bool keep = j <= i % R;
    d.tril[i * K + j] = keep ? 1 : 0; d.bias[i * K + j] = keep ? 0 : neg_inf;
    d.shifted_input[i * K + j] = x[i * K + j] * l.sm_scalar + d.shift[i];
  }
  const unsigned tv = ceil_log2((size_t)(l.sm_bkm >> 16));
  if (tv != l.sm_table_size) throw std::runtime_error("softmax: table size and bkm disagree");
  const int64_t tmask = (int64_t(1) << tv) - 1, zmask = (int64_t(1) << l.sm_zero_vars) - 1;
  d.zero_in.resize(l.sm_zero_chunks); d.zero_out.resize(l.sm_zero_chunks);
  std::vector<int64_t> o;
  for (size_t q = 0; q < x.size(); q++) {
    const int64_t masked = d.shifted_input[q] * d.tril[q] + d.bias[q];
    int64_t r = masked < 0 ? -masked : masked;
    d.low.push_back(r & 255); r >>= 8; d.high.push_back(r & 255); r >>= 8;
    const int64_t lk = r & tmask, ev = softmax_table_output(l.sm_temp_bits, l.sm_bkm, lk);
    d.exp_in.push_back(lk); d.exp_out.push_back(ev); r >>= tv;
    int64_t acc = ev;
    for (unsigned z = 0; z < l.sm_zero_chunks; z++) { const int64_t zi = r & zmask, zo = zi != 0 ? 0 : 1; d.zero_in[z].push_back(zi); d.zero_out[z].push_back(zo); r >>= l.sm_zero_vars; acc *= zo; }
    o.push_back(acc);
  }
  if (out) *out = std::move(d);
  return o;
}
// ConcatMatMul::evaluate (concat_matmul.rs:568-616): chunk c of the result = chunk c of A times chunk c of B
static inline std::vector<int64_t> concat_matmul_op(const Layer& l, const std::vector<int64_t>& cur, const std::vector<int64_t>& b0) {
  std::vector<int64_t> o;
      if (cur.size() != l.cm_a[0] * l.cm_a[1] * l.cm_a[2] || b0.size() != l.cm_b[0] * l.cm_b[1] * l.cm_b[2]) throw std::runtime_error("concat matmul: input shapes");
      int order[3]; size_t sa[3], sb[3];
      std::vector<int64_t> a = cur, b = b0;
      for (int d = 0; d < 3; d++) { sa[d] = l.cm_a[d]; sb[d] = l.cm_b[d]; }
      if (cm_permutation(l.cm_left, CM_EXPECTED_LEFT, order)) { size_t t[3]; a = permute3d(cur, l.cm_a, order, t); for (int d = 0; d < 3; d++) sa[d] = t[d]; }
      if (cm_permutation(l.cm_right, CM_EXPECTED_RIGHT, order)) { size_t t[3]; b = permute3d(b0, l.cm_b, order, t); for (int d = 0; d < 3; d++) sb[d] = t[d]; }
      if (sa[0] != sb[0] || sa[2] != sb[1]) throw std::runtime_error("concat matmul: chunk shapes");
      const size_t C = sa[0], R = sa[1], M = sa[2], N = sb[2];
      std::vector<int64_t> r(C * R * N, 0);
      for (size_t c = 0; c < C; c++) for (size_t i = 0; i < R; i++) for (size_t j = 0; j < N; j++) { int64_t acc = 0; for (size_t q = 0; q < M; q++) acc += a[(c * R + i) * M + q] * b[(c * M + q) * N + j]; r[(c * R + i) * N + j] = acc; }
      if (l.cm_perm.empty()) o = r;
      else { const size_t rs[3] = {C, R, N}; const int po[3] = {l.cm_perm[0], l.cm_perm[1], l.cm_perm[2]}; size_t t[3]; o = permute3d(r, rs, po, t); }
  return o;
}
static inline int64_t requant_apply(const Layer& l, int64_t v) {
  unsigned sh = l.shift();
  int64_t tmp = v * l.fixed_point_multiplier + (int64_t(1) << (sh - 1));
  return clamp_q(tmp >> sh);
}
// the concatenated output tensors of the model (ModelSpec::outputs order)
static inline std::vector<int64_t> model_output(const Model& m, const Trace& tr) {
  std::vector<int64_t> o;
  for (const Wire& w : model_outputs(m)) { const std::vector<int64_t>& v = tr.output((size_t)w.node, (size_t)w.index); o.insert(o.end(), v.begin(), v.end()); }
  return o;
}
static inline Trace run_model(const Model& m, const std::vector<int64_t>& input) {
  Trace tr;
  if (input.size() != m.input_len) throw std::runtime_error("input length mismatch");
  const std::vector<size_t> in_lens = model_input_lens(m);
  { size_t tot = 0; for (size_t n : in_lens) tot += n; if (tot != m.input_len) throw std::runtime_error("input tensors do not add up to input_len"); }
  auto value = [&](const Wire& w) -> std::vector<int64_t> {
    if (w.node >= 0) { if ((size_t)w.node >= tr.out.size()) throw std::runtime_error("model graph: a node reads a later node"); return tr.output((size_t)w.node, (size_t)w.index); }
    size_t off = 0; for (int q = 0; q < w.index; q++) off += in_lens.at((size_t)q);
    return std::vector<int64_t>(input.begin() + off, input.begin() + off + in_lens.at((size_t)w.index));
  };
  tr.in2.resize(m.layers.size()); tr.in3.resize(m.layers.size()); tr.out_more.resize(m.layers.size());
  for (size_t id = 0; id < m.layers.size(); id++) {
    const Layer& l = m.layers[id];
    const std::vector<Wire> wires = node_inputs(m, id);
    std::vector<int64_t> cur = value(wires[0]);
    tr.in.push_back(cur);
    if (wires.size() > 1) tr.in2[id] = value(wires[1]);
    if (wires.size() > 2) tr.in3[id] = value(wires[2]);
    if (wires.size() != n_inputs(l)) throw std::runtime_error("model graph: wrong number of inputs for a node");
    std::vector<int64_t> o;
    if (l.kind == L_MATMUL2) {  // MatMul::op (matrix_mul.rs:230-311) on two input tensors
      const std::vector<int64_t>& b = tr.in2[id];
      const size_t k = l.nrows, n = l.ncols;
      if (!k || cur.size() % k || b.size() != k * n) throw std::runtime_error("matmul2: input shapes");
      const size_t s_ = cur.size() / k;
      o.assign(s_ * n, 0);
      for (size_t i = 0; i < s_; i++) for (size_t j = 0; j < n; j++) { int64_t a = 0; for (size_t q = 0; q < k; q++) a += cur[i * k + q] * (l.transpose_b ? b[j * k + q] : b[q * n + j]); o[i * n + j] = a; }
    } else if (l.kind == L_ADD2) {  // Add::evaluate (add.rs:184-210), no operand
      const std::vector<int64_t>& b = tr.in2[id];
      if (cur.size() != b.size()) throw std::runtime_error("add2: inputs of different lengths");
      o.resize(cur.size());
      for (size_t i = 0; i < cur.size(); i++) o[i] = l.add_left * cur[i] + l.add_right * b[i];
    } else if (l.kind == L_QKV) {  // QKV::evaluate (qkv.rs:275-340, no cache): X W_q + b_q, X W_k + b_k, X W_v + b_v
      const size_t k = l.nrows, n = l.ncols;
      if (!k || cur.size() % k || l.weights.size() != 3 * k * n || l.bias.size() != 3 * n) throw std::runtime_error("qkv: shapes");
      const size_t s_ = cur.size() / k;
      for (int w = 0; w < 3; w++) {
        std::vector<int64_t> y(s_ * n);
        for (size_t i = 0; i < s_; i++) for (size_t j = 0; j < n; j++) { int64_t a = 0; for (size_t q = 0; q < k; q++) a += cur[i * k + q] * l.weights[(w * k + q) * n + j]; y[i * n + j] = a + l.bias[w * n + j]; }
        if (w == 0) o = y; else tr.out_more[id].push_back(y);
      }
    } else if (l.kind == L_CONCAT_MATMUL) o = concat_matmul_op(l, cur, tr.in2[id]);
    else if (l.kind == L_MHA) {  // Mha::evaluate_with_intermediate_outputs (mha.rs:216-300): qk, softmax, final_mul on the reshaped inputs
      const size_t S = l.mha_shape[0], H = l.mha_shape[1], D = l.mha_shape[2];
      if (cur.size() != S * H * D || tr.in2[id].size() != cur.size() || tr.in3[id].size() != cur.size()) throw std::runtime_error("mha: input shapes");
      MhaData d;
      d.softmax_in = concat_matmul_op(mha_qk_layer(l), cur, tr.in2[id]);
      d.softmax_out = softmax_op(mha_softmax_layer(l), d.softmax_in, nullptr);
      o = concat_matmul_op(mha_final_layer(l), d.softmax_out, tr.in3[id]);
      tr.mha[id] = d;
    } else
    if (l.kind == L_DENSE) {
      if (cur.size() != l.ncols) throw std::runtime_error("dense input size mismatch");
      o.resize(l.nrows);
      for (size_t i = 0; i < l.nrows; i++) { int64_t a = 0; for (size_t j = 0; j < l.ncols; j++) a += l.weights[i * l.ncols + j] * cur[j]; o[i] = a + l.bias[i]; }
    } else if (l.kind == L_MATMUL) {  // MatMul::op (matrix_mul.rs:230-311): input [s][k] times the constant [k][n], bias added to every row
      const size_t k = l.nrows, n = l.ncols;
      if (cur.size() % k) throw std::runtime_error("matmul input size mismatch");
      const size_t s_ = cur.size() / k;
      o.assign(s_ * n, 0);
      for (size_t i = 0; i < s_; i++) for (size_t j = 0; j < n; j++) { int64_t a = 0; for (size_t q = 0; q < k; q++) a += cur[i * k + q] * (l.transpose_b ? l.weights[j * k + q] : l.weights[q * n + j]); o[i * n + j] = a + (l.bias.empty() ? 0 : l.bias[j]); }
    } else if (l.kind == L_EMBED) {  // Embeddings::evaluate (embeddings.rs:197-236): row x[i] of the table for every token
      o.resize(cur.size() * l.ncols);
      for (size_t i = 0; i < cur.size(); i++) {
        if (cur[i] < 0 || (size_t)cur[i] >= l.nrows) throw std::runtime_error("embeddings: token outside the vocabulary");
        for (size_t j = 0; j < l.ncols; j++) o[i * l.ncols + j] = l.weights[(size_t)cur[i] * l.ncols + j];
      }
    } else if (l.kind == L_POSITIONAL) {  // Positional::evaluate (positional.rs:157-185): the Add layer on (x, the first rows of the table)
      if (cur.size() % l.ncols || cur.size() > l.weights.size()) throw std::runtime_error("positional: input shape");
      o.resize(cur.size());
      for (size_t i = 0; i < cur.size(); i++) o[i] = l.add_left * cur[i] + l.add_right * l.weights[i];
    } else if (l.kind == L_ADD) {  // Add::evaluate (add.rs:184-210)
      if (cur.size() != l.weights.size()) throw std::runtime_error("add: operand size mismatch");
      o.resize(cur.size());
      for (size_t i = 0; i < cur.size(); i++) o[i] = l.add_left * cur[i] + l.add_right * l.weights[i];
    } else if (l.kind == L_REQUANT) {
      for (int64_t v : cur) {
        if (std::llabs(v) > (int64_t(1) << l.intermediate_bit_size)) throw std::runtime_error("requant: value too large");
        o.push_back(requant_apply(l, v));
      }
    } else if (l.kind == L_RELU) { for (int64_t v : cur) o.push_back(relu_apply(v)); }
    else if (l.kind == L_GELU) { for (int64_t v : cur) o.push_back(gelu_apply(l.gelu_multiplier, v)); }
    else if (l.kind == L_LAYERNORM) o = layernorm_op(l, cur, nullptr);
    else if (l.kind == L_SOFTMAX) o = softmax_op(l, cur, nullptr);
    else if (l.kind == L_CONV) { tr.conv.resize(m.layers.size()); o = conv_op(l, cur, tr.conv[tr.in.size() - 1]); }
    else if (l.kind == L_MAXPOOL) o = maxpool_op(l, cur);
    else if (l.kind == L_FLATTEN) o = cur;
    else throw std::runtime_error("unknown layer kind");
    tr.out.push_back(o);
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
  std::map<TableType, ProverCommitment> table_comms;                       // committed table columns (commit/context.rs:105-107)
  size_t max_poly_len = 0;
};
static inline size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
// lengths of every node's output tensors (the shape propagation of run_model)
static inline std::vector<std::vector<size_t>> node_output_lens(const Model& m) {
  const std::vector<size_t> in_lens = model_input_lens(m);
  std::vector<std::vector<size_t>> lens(m.layers.size());
  auto len_of = [&](const Wire& w) { return w.node < 0 ? in_lens.at((size_t)w.index) : lens.at((size_t)w.node).at((size_t)w.index); };
  for (size_t id = 0; id < m.layers.size(); id++) {
    const Layer& l = m.layers[id];
    const std::vector<Wire> wires = node_inputs(m, id);
    for (const Wire& w : wires) if (w.node >= (int)id) throw std::runtime_error("model graph: a node reads a later node");
    size_t cur = len_of(wires[0]);
    if (l.kind == L_DENSE) cur = l.nrows;
    else if (l.kind == L_MATMUL || l.kind == L_MATMUL2 || l.kind == L_QKV) cur = cur / l.nrows * l.ncols;
    else if (l.kind == L_EMBED) cur = cur * l.ncols;
    else if (l.kind == L_CONV) cur = l.kw * l.nw * l.nw;
    else if (l.kind == L_MAXPOOL) cur = l.pin[0] * (l.pin[1] / 2) * (l.pin[2] / 2);
    else if (l.kind == L_CONCAT_MATMUL) { size_t o[3]; cm_output_shape(l, o); cur = o[0] * o[1] * o[2]; }
    lens[id].assign(n_outputs(l), cur);
  }
  return lens;
}
static inline Context context_generate(const Model& m) {
  Context ctx; ctx.model = m;
  size_t max_poly_len = 0;
  for (size_t n : model_input_lens(m)) max_poly_len = std::max(max_poly_len, n);
  std::vector<TableType> tset;
  auto add_table = [&](TableType t) { for (auto& x : tset) if (x == t) return; tset.push_back(t); };
  const std::vector<std::vector<size_t>> out_lens = node_output_lens(m);
  for (size_t id_ = 0; id_ < m.layers.size(); id_++) {
    // an Mha node brings the tables of its softmax (Mha::step_info, mha.rs:432-503: qk, softmax and final_mul step_info one after the other; only
    // the softmax has tables and witness polynomials, as long as the [heads][seq][seq] products)
    const Layer& l0 = m.layers[id_];
    const Layer sub = l0.kind == L_MHA ? mha_softmax_layer(l0) : Layer();
    const Layer& l = l0.kind == L_MHA ? sub : l0;
    size_t cur_len = l0.kind == L_MHA ? sub.sm_shape[0] * sub.sm_shape[1] * sub.sm_shape[2] : out_lens[id_][0];  // (Requant / Relu keep the length of their input)
    if (l.kind == L_REQUANT) { add_table({2, 0}); add_table({3, l.clamping_size()}); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }
    else if (l.kind == L_RELU) { add_table({0, 0}); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }
    else if (l.kind == L_GELU) { TableType g{1, gelu_table_log2(l.gelu_multiplier)}; g.aux2 = l.gelu_multiplier; add_table(g); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }  // activation.rs:163-167
    else if (l.kind == L_LAYERNORM) { add_table({2, 0}); add_table(layernorm_table(l)); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }  // layernorm.rs:587-618
    else if (l.kind == L_SOFTMAX) {  // softmax.rs:1205-1245
      add_table({2, 0}); add_table(softmax_table(l)); add_table(softmax_error_table(l));
      if (l.sm_zero_vars) add_table({6, l.sm_zero_vars});
      max_poly_len = std::max(max_poly_len, next_pow2(cur_len));
    }
    else if (l.kind == L_CONV) { cur_len = l.kw * l.nw * l.nw; }                                                       // convolution.rs:506-511
    else if (l.kind == L_MAXPOOL) { add_table({2, 0}); cur_len = l.pin[0] * (l.pin[1] / 2) * (l.pin[2] / 2); max_poly_len = std::max(max_poly_len, next_pow2(cur_len)); }  // pooling.rs:131-166
  }
  std::sort(tset.begin(), tset.end());
  for (auto& t : tset) max_poly_len = std::max(max_poly_len, size_t(1) << t.multiplicity_poly_vars());
  for (auto& l : m.layers) if (l.kind == L_QKV) { max_poly_len = std::max(max_poly_len, next_pow2(l.weights.size() / 3)); max_poly_len = std::max(max_poly_len, next_pow2(l.bias.size() / 3)); }
  for (auto& l : m.layers) if (l.kind == L_DENSE || l.kind == L_CONV || l.kind == L_MATMUL || l.kind == L_ADD || l.kind == L_EMBED || l.kind == L_POSITIONAL || l.kind == L_LAYERNORM) { max_poly_len = std::max(max_poly_len, next_pow2(l.weights.size())); max_poly_len = std::max(max_poly_len, next_pow2(l.bias.size())); }
  max_poly_len = next_pow2(max_poly_len);
  ctx.max_poly_len = max_poly_len;
  ctx.pp = pcs_setup(max_poly_len);
  // commit/context.rs:79-103 commits the model polynomials with `into_par_iter`; the oracle mirrors that with plain
  // threads (one per polynomial) — commitments are independent so the result does not depend on the schedule
  std::vector<std::pair<size_t, const char*>> jobs;
  for (size_t id = 0; id < m.layers.size(); id++) {
    if (m.layers[id].kind == L_DENSE) { jobs.push_back({id, "DenseWeight"}); jobs.push_back({id, "DenseBias"}); }
    if (m.layers[id].kind == L_CONV) { jobs.push_back({id, "ConvFilter"}); jobs.push_back({id, "ConvBias"}); }  // convolution.rs:452-453,546-553
    if (m.layers[id].kind == L_POSITIONAL) jobs.push_back({id, "PositionalMatrix"});  // positional.rs:229,243-251
    if (m.layers[id].kind == L_EMBED) jobs.push_back({id, "EmbeddingMat"});  // embeddings.rs:271,284-291
    if (m.layers[id].kind == L_ADD) jobs.push_back({id, "255"});  // OPERAND_POLY_ID = 0xff, to_string() (add.rs:32,520)
    if (m.layers[id].kind == L_MATMUL) { jobs.push_back({id, "MatMulWeight"}); if (!m.layers[id].bias.empty()) jobs.push_back({id, "MatMulBias"}); }  // matrix_mul.rs:947-963
    if (m.layers[id].kind == L_LAYERNORM) { jobs.push_back({id, "LayerNormGamma"}); jobs.push_back({id, "LayerNormBeta"}); }  // layernorm.rs:67-68,603-613
    if (m.layers[id].kind == L_QKV) for (const char* pid : {"WeightQ", "WeightK", "WeightV", "BiasQ", "BiasK", "BiasV"}) jobs.push_back({id, pid});  // qkv.rs:388-418
  }
  for (auto& j : jobs) ctx.model_comms[j.first][j.second];  // create map slots before the threads write into them
  std::vector<std::thread> th;
  for (auto& j : jobs) th.emplace_back([&ctx, &m, j] {
    const Layer& l = m.layers[j.first];
    std::string pid = j.second;
    Mle poly;
    if (l.kind == L_QKV) {  // the three weight matrices and bias vectors are committed one by one
      const bool wgt = pid[0] == 'W'; const size_t which = pid.back() == 'Q' ? 0 : pid.back() == 'K' ? 1 : 2;
      const std::vector<int64_t>& src = wgt ? l.weights : l.bias; const size_t n = src.size() / 3;
      poly = Mle::from_i64(std::vector<int64_t>(src.begin() + which * n, src.begin() + (which + 1) * n));
    } else poly = Mle::from_i64(pid == "DenseWeight" || pid == "ConvFilter" || pid == "MatMulWeight" || pid == "255" || pid == "EmbeddingMat" || pid == "PositionalMatrix" || pid == "LayerNormGamma" ? l.weights : l.bias);
    ctx.model_comms[j.first][j.second] = {pcs_commit(ctx.pp, poly), poly};
  });
  for (auto& t : th) t.join();
  ctx.tables = tset;
  // Relu / Range / Clamping have no committed columns; InverseSQRT commits its output column (lookup/context.rs:492-545, commit/context.rs:105-107)
  for (auto& t : tset) if (t.has_committed_column()) {
    std::vector<int64_t> merged; std::vector<std::vector<u64>> cols;
    table_columns(t, merged, cols);
    Mle poly = Mle::from_base(cols.back());  // (the output column; the only column of an ErrorTable)
    ctx.table_comms[t] = {pcs_commit(ctx.pp, poly), poly};
  }
  return ctx;
}

// ------------------------------------------------------------------ proofs
struct DenseProof { IOPProof sumcheck; E bias_eval; std::vector<E> individual_claims; };
struct AddProof { E left_eval, right_eval; };  // add.rs:59-63
struct PositionalProof { std::vector<E> sub_matrix_evals; AddProof add_proof; };  // SinglePositionalProof (positional.rs:45-55); one input
struct MatMulProof { IOPProof sumcheck; std::vector<E> individual_claims; bool has_bias = false; E bias_eval{}; };  // matrix_mul.rs:153-161 (bias_eval: Option<E>)
struct SamePolyProof { IOPProof sumcheck; std::vector<E> evals; };
struct ConcatMatMulProof { IOPProof sumcheck; std::vector<E> individual_claims; };  // concat_matmul.rs:365-373
// qkv.rs:63-83, fields in declaration order; individual_claims = the (input, weight) evaluation pair of Q, K, V
struct QKVProof { IOPProof sumcheck; SamePolyProof aggregation_proof; std::vector<E> pre_bias_evals; std::vector<E> individual_claims; };
struct ActivationProof { SamePolyProof io_accumulation; LogUpProof lookup; std::vector<Commitment> commits; };
struct RequantProof { IOPProof io_accumulation; std::vector<E> accumulation_evals; LogUpProof clamping_lookup, shifted_lookup; std::vector<Commitment> commitments; };
struct HadamardProof { IOPProof sumcheck; std::vector<E> individual_claim; };  // hadamard.rs:51-56
struct ConvProof {  // convolution.rs:98-127, fields in declaration order
  IOPProof fft_proof, fft_proof_weights;
  std::vector<IOPProof> fft_delegation_proof, fft_delegation_proof_weights;
  IOPProof ifft_proof;
  std::vector<IOPProof> ifft_delegation_proof;
  IOPProof hadamard_proof;
  std::vector<E> fft_claims, fft_weight_claims, ifft_claims;
  std::vector<std::vector<E>> fft_delegation_claims, fft_delegation_weights_claims, ifft_delegation_claims;
  std::vector<E> partial_evals, hadamard_clams;
  E bias_claim;
  HadamardProof clearing_proof;
};
struct PoolingProof { IOPProof sumcheck; LogUpProof lookup; std::vector<E> zerocheck_evals; size_t variable_gap = 0; std::vector<Commitment> commitments; };  // pooling.rs:60-76
// LayerNormProof (layernorm.rs:644-667), fields in declaration order
struct LayerNormProof { std::vector<LogUpProof> logup_proofs; std::vector<Commitment> commitments; IOPProof accumulation_proof, io_proof, input_proof; std::vector<E> acc_evals, evaluations; E gamma_eval, beta_eval; };
// SoftmaxProof (softmax.rs:102-117)
struct SoftmaxProof { std::vector<LogUpProof> logup_proofs; std::vector<Commitment> commitments; IOPProof accumulation_proof, mask_proof; std::vector<E> evaluations; };
struct LayerProof { LayerKind kind; DenseProof dense; MatMulProof matmul; AddProof add; PositionalProof pos; ActivationProof act; RequantProof req; ConvProof conv; PoolingProof pool; ConcatMatMulProof cmm; QKVProof qkv; LayerNormProof ln; SoftmaxProof sm; ConcatMatMulProof mha_final, mha_qk; };  // MhaProof (mha.rs:122-128) = {mha_final, sm, mha_qk}
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
    // Mha::gen_lookup_witness (mha.rs:706-719): the witness of its softmax on the products, under the node's own id
    const Layer& l0 = ctx.model.layers[id];
    const Layer sub = l0.kind == L_MHA ? mha_softmax_layer(l0) : Layer();
    const Layer& l = l0.kind == L_MHA ? sub : l0;
    const std::vector<int64_t>& softmax_input = l0.kind == L_MHA ? tr.mha.at(id).softmax_in : tr.in[id];
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
    } else if (l.kind == L_RELU || l.kind == L_GELU) {
      TableType rt{0, 0};
      if (l.kind == L_GELU) { rt = TableType{1, gelu_table_log2(l.gelu_multiplier)}; rt.aux2 = l.gelu_multiplier; }
      std::vector<int64_t> a = tr.in[id]; const auto& b = tr.out[id];
      if (l.kind == L_GELU) for (auto& v : a) v *= l.gelu_multiplier;  // (activation.rs:268-276: the looked-up element is the scaled input)
      for (size_t i = 0; i < a.size(); i++) count_into(element_count[rt], a[i] + COLUMN_SEPARATOR * b[i]);
      LogUpWitness w; w.is_table = false; w.columns_per_instance = 2; w.table_type = rt;
      for (const std::vector<int64_t>* col : {(const std::vector<int64_t>*)&a, &b}) { std::vector<u64> ev = to_base(*col); Mle mle = Mle::from_base(ev); w.commits.push_back({pcs_commit(ctx.pp, mle), mle}); w.column_evals.push_back(ev); }
      ps.lookup_witness[id] = {w};
    } else if (l.kind == L_LAYERNORM) {  // LayerNorm::lookup_witness (layernorm.rs:1103-1218)
      LayerNormData d; layernorm_op(l, tr.in[id], &d);
      const unsigned nrc = (l.ln_range_check_bits - 1) / BIT_LEN + 1;
      const int64_t rmask = (int64_t(1) << BIT_LEN) - 1, top = int64_t(1) << l.ln_top_chunk_scalar_log;
      std::vector<std::vector<int64_t>> chunks(nrc);
      for (unsigned j = 0; j < nrc; j++) for (int64_t v : d.range_check) chunks[j].push_back(((v >> (j * BIT_LEN)) & rmask) * (j + 1 == nrc ? top : 1));
      TableType it = layernorm_table(l), rt{2, 0};
      for (auto& ch : chunks) for (int64_t v : ch) count_into(element_count[rt], v);
      for (size_t i = 0; i < d.lookup_input.size(); i++) count_into(element_count[it], d.lookup_input[i] + d.lookup_output[i] * COLUMN_SEPARATOR);
      LogUpWitness wi; wi.is_table = false; wi.columns_per_instance = 2; wi.table_type = it;
      for (auto* col : {&d.lookup_input, &d.lookup_output}) { std::vector<u64> ev = to_base(*col); Mle mle = Mle::from_base(ev); wi.commits.push_back({pcs_commit(ctx.pp, mle), mle}); wi.column_evals.push_back(ev); }
      LogUpWitness wr; wr.is_table = false; wr.columns_per_instance = 1; wr.table_type = rt;
      for (auto& ch : chunks) { std::vector<u64> ev = to_base(ch); Mle mle = Mle::from_base(ev); wr.commits.push_back({pcs_commit(ctx.pp, mle), mle}); wr.column_evals.push_back(ev); }
      ps.lookup_witness[id] = {wi, wr};
    } else if (l.kind == L_SOFTMAX) {  // Softmax::lookup_witness (softmax.rs:890-1066)
      SoftmaxData d; std::vector<int64_t> out = softmax_op(l, softmax_input, &d);
      const size_t K = l.sm_shape[2];
      std::vector<int64_t> row_sums;
      for (size_t i = 0; i < out.size() / K; i++) { int64_t a = 0; for (size_t j = 0; j < K; j++) a += out[i * K + j]; row_sums.push_back(a); }
      TableType st = softmax_table(l), rt{2, 0}, et = softmax_error_table(l), zt{6, l.sm_zero_vars};
      for (int64_t v : d.low) count_into(element_count[rt], v);
      for (int64_t v : d.high) count_into(element_count[rt], v);
      for (size_t i = 0; i < d.exp_in.size(); i++) count_into(element_count[st], d.exp_in[i] + d.exp_out[i] * COLUMN_SEPARATOR);
      for (int64_t v : row_sums) count_into(element_count[et], v);
      auto witness = [&](const std::vector<const std::vector<int64_t>*>& cols, size_t cpi, TableType tt) {
        LogUpWitness w; w.is_table = false; w.columns_per_instance = cpi; w.table_type = tt;
        for (auto* col : cols) { std::vector<u64> ev = to_base(*col); Mle mle = Mle::from_base(ev); w.commits.push_back({pcs_commit(ctx.pp, mle), mle}); w.column_evals.push_back(ev); }
        return w;
      };
      std::vector<LogUpWitness> ws = {witness({&d.exp_in, &d.exp_out}, 2, st), witness({&d.low, &d.high}, 1, rt)};
      { LogUpWitness w; w.is_table = false; w.columns_per_instance = 1; w.table_type = et;  // the row sums are looked up, the SHIFT polynomial is what is committed here
        Mle shift = Mle::from_base(to_base(d.shift)); w.commits.push_back({pcs_commit(ctx.pp, shift), shift}); w.column_evals.push_back(to_base(row_sums)); ws.push_back(w); }
      if (l.sm_zero_chunks) {
        std::vector<const std::vector<int64_t>*> cols;
        for (unsigned z = 0; z < l.sm_zero_chunks; z++) { cols.push_back(&d.zero_in[z]); cols.push_back(&d.zero_out[z]); for (size_t i = 0; i < d.zero_in[z].size(); i++) count_into(element_count[zt], d.zero_in[z][i] + d.zero_out[z][i] * COLUMN_SEPARATOR); }
        ws.push_back(witness(cols, 2, zt));
      }
      ps.lookup_witness[id] = ws;
    } else if (l.kind == L_MAXPOOL) {  // Pooling::gen_lookup_witness (pooling.rs:206-262)
      TableType rt{2, 0};
      std::vector<std::vector<int64_t>> diffs = maxpool_diff_polys(l, tr.in[id], tr.out[id]);
      for (auto& d : diffs) for (int64_t v : d) count_into(element_count[rt], v);
      LogUpWitness w; w.is_table = false; w.columns_per_instance = 1; w.table_type = rt;
      for (auto& d : diffs) { std::vector<u64> ev = to_base(d); Mle mle = Mle::from_base(ev); w.commits.push_back({pcs_commit(ctx.pp, mle), mle}); w.column_evals.push_back(ev); }
      { Mle mle = Mle::from_base(to_base(tr.out[id])); w.commits.push_back({pcs_commit(ctx.pp, mle), mle}); }  // the output poly is committed too
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
// Positional::prove, Learned (layers/transformer/positional.rs:327-452): the Add layer on (input, the first `tokens` rows of the table) gives
// the evaluations of both at the claim's point (add.rs:81-145, two inputs: no transcript traffic); the claim on the slice is then lifted to
// one on the WHOLE committed table: the output claim and the slice claim are absorbed, one extra coordinate per doubling is drawn, the
// prover sends the evaluation of every "upper half" sub-matrix (rows [tokens 2^k, tokens 2^(k+1))) at the prefix of the point, and
// table(point | extras) = fold_k (acc (1 - c_k) + sub_k c_k) (compute_positional_matrix_claim, :106-126).
static inline Claim prove_positional(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& input) {
  const size_t n = input.size();
  if ((size_t(1) << last.point.size()) != n) throw std::runtime_error("positional: claim point size mismatch");
  std::vector<E> sub(n); for (size_t i = 0; i < n; i++) sub[i] = e_from_i64(l.weights[i]);
  E left_eval = Mle::from_ext(input).evaluate(last.point), right_eval = Mle::from_ext(sub).evaluate(last.point);
  const unsigned nv_all = log2_strict(l.weights.size()), nv_sub = (unsigned)last.point.size(), diff = nv_all - nv_sub;
  Transcript& t = *ps.t;
  t.append_exts(last.point); t.append_ext(last.eval); t.append_exts(last.point); t.append_ext(right_eval);  // sample_random_coordinates (:80-99)
  std::vector<E> point = last.point;
  for (unsigned k = 0; k < diff; k++) point.push_back(t.read_challenge());
  PositionalProof pp; pp.add_proof = {left_eval, right_eval};
  E acc = right_eval;
  for (unsigned k = 0; k < diff; k++) {
    const size_t len = n << k;  // rows [tokens 2^k, tokens 2^(k+1)) of the table, as long as everything below them
    std::vector<E> sm(len); for (size_t i = 0; i < len; i++) sm[i] = e_from_i64(l.weights[len + i]);
    std::vector<E> pfx(point.begin(), point.begin() + nv_sub + k);
    E ev = Mle::from_ext(sm).evaluate(pfx);
    pp.sub_matrix_evals.push_back(ev);
    E c = point[nv_sub + k];
    acc = eadd(emul(acc, esub(e_one(), c)), emul(ev, c));
  }
  ps.add_witness_claim(ps.ctx->model_comms.at(id).at("PositionalMatrix"), {point, acc});
  LayerProof lp; lp.kind = L_POSITIONAL; lp.pos = pp;
  ps.proofs[id] = lp;
  return {last.point, left_eval};
}
// Embeddings::prove (layers/transformer/embeddings.rs:359-462): the matmul protocol on (one-hot(tokens), table) without ever building
// the one-hot matrix — its row variables fixed at the row part of the claim give reduced[x[i]] += beta(i, row part) (:382-401); the
// table's column variables are fixed at the column part; a degree-2 sumcheck over the vocabulary; the table claim
// [column part | sumcheck point] goes to its commitment, the one-hot claim [sumcheck point | row part] is what the verifier checks
// against the public tokens (verify_input_claim, :530-571).
static inline Claim prove_embeddings(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<int64_t>& tokens) {
  const unsigned nvc = log2_strict(l.ncols), nvr = log2_strict(tokens.size());
  if (last.point.size() != nvc + nvr) throw std::runtime_error("embeddings: claim point size mismatch");
  std::vector<E> col_pt(last.point.begin(), last.point.begin() + nvc), row_pt(last.point.begin() + nvc, last.point.end());
  std::vector<E> beta = build_eq_x_r_vec(row_pt);
  std::vector<E> reduced(l.nrows, e_zero());
  for (size_t i = 0; i < tokens.size(); i++) reduced[(size_t)tokens[i]] = eadd(reduced[(size_t)tokens[i]], beta[i]);
  std::vector<E> w(l.weights.size()); for (size_t i = 0; i < w.size(); i++) w[i] = e_from_i64(l.weights[i]);
  Mle table = Mle::from_ext(w);
  table.fix_low_in_place(col_pt);
  Mle in = Mle::from_ext(reduced);
  VirtualPolynomial vp(in.nv);
  vp.add_mle_list({mk(in), mk(table)}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();
  std::vector<E> one_hot_pt = proof.point; one_hot_pt.insert(one_hot_pt.end(), row_pt.begin(), row_pt.end());
  std::vector<E> table_pt = col_pt; table_pt.insert(table_pt.end(), proof.point.begin(), proof.point.end());
  ps.add_witness_claim(ps.ctx->model_comms.at(id).at("EmbeddingMat"), {table_pt, fin[1]});
  LayerProof lp; lp.kind = L_EMBED; lp.matmul.sumcheck = proof; lp.matmul.individual_claims = fin;  // EmbeddingsProof {sumcheck, individual_claims} (:60-67)
  ps.proofs[id] = lp;
  return {one_hot_pt, fin[0]};
}
// Add::prove_step with a static operand (layers/add.rs:81-145): no sumcheck and no transcript traffic — the prover evaluates the input at the
// claim's point, solves out(r) = M1 x(r) + M2 c(r) for the operand's evaluation and hands that claim to the operand's commitment
static inline Claim prove_add(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& input) {
  E left_eval = Mle::from_ext(input).evaluate(last.point);
  E right_eval = emul(esub(last.eval, emul(left_eval, e_from_i64(l.add_left))), einv(e_from_i64(l.add_right)));
  ps.add_witness_claim(ps.ctx->model_comms.at(id).at("255"), {last.point, right_eval});
  LayerProof lp; lp.kind = L_ADD; lp.add = {left_eval, right_eval};
  ps.proofs[id] = lp;
  return {last.point, left_eval};
}
// MatMul::prove_step (layers/matrix_mul.rs:701-873) for the (Input, Weight) arrangement, right matrix not transposed:
// split_claim (:339-356): the low variables of the output point address columns -> the right matrix, the high ones rows -> the left;
// the bias (one value per column) is evaluated on the column part and leaves the claim; left.fix_high(rows), right.fix_low(cols);
// one degree-2 sumcheck over the inner dimension; full_points (:364-383): left at [sumcheck point | row part] (the claim handed to the
// previous layer), right at [column part | sumcheck point] (opened against the weight commitment).
static inline Claim prove_matmul(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& input) {
  const size_t k = l.nrows, n = l.ncols, s_ = input.size() / k;
  const unsigned nvc = log2_strict(n), nvr = log2_strict(s_);
  if (last.point.size() != nvc + nvr) throw std::runtime_error("matmul: claim point size mismatch");
  std::vector<E> pt_right(last.point.begin(), last.point.begin() + nvc), pt_left(last.point.begin() + nvc, last.point.end());
  const bool hb = !l.bias.empty();
  E bias_eval = e_zero();
  if (hb) bias_eval = Mle::from_i64(l.bias).evaluate(pt_right);  // (last_claim.eval -= bias_eval only matters to the verifier)
  Mle left = Mle::from_ext(input);
  left.fix_high_in_place(pt_left);
  std::vector<E> w(l.weights.size()); for (size_t i = 0; i < w.size(); i++) w[i] = e_from_i64(l.weights[i]);
  Mle right = Mle::from_ext(w);
  // not transposed: [k][n], the column variables are the LOW ones; transposed: stored [n][k], the variables of its rows are the HIGH ones (:815-824)
  if (l.transpose_b) right.fix_high_in_place(pt_right); else right.fix_low_in_place(pt_right);
  if (left.nv != right.nv) throw std::runtime_error("matmul: inner dimensions differ");
  VirtualPolynomial vp(left.nv);
  vp.add_mle_list({mk(left), mk(right)}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();
  std::vector<E> point_left = proof.point; point_left.insert(point_left.end(), pt_left.begin(), pt_left.end());
  std::vector<E> point_right = l.transpose_b ? proof.point : pt_right;
  if (l.transpose_b) point_right.insert(point_right.end(), pt_right.begin(), pt_right.end()); else point_right.insert(point_right.end(), proof.point.begin(), proof.point.end());
  // add_common_claims iterates the node's BTreeMap: "MatMulBias" then "MatMulWeight"
  const auto& comms = ps.ctx->model_comms.at(id);
  if (hb) ps.add_witness_claim(comms.at("MatMulBias"), {pt_right, bias_eval});
  ps.add_witness_claim(comms.at("MatMulWeight"), {point_right, fin[1]});
  LayerProof lp; lp.kind = L_MATMUL; lp.matmul.sumcheck = proof; lp.matmul.individual_claims = fin; lp.matmul.has_bias = hb; lp.matmul.bias_eval = bias_eval;
  ps.proofs[id] = lp;
  return {point_left, fin[0]};
}
// MatMul::prove_step (layers/matrix_mul.rs:701-873) with BOTH matrices inputs: the same sumcheck, no bias, no commitment — both final
// evaluations leave as claims, the left input's first (the order of the node's inputs)
static inline std::vector<Claim> prove_matmul2(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& a, const std::vector<E>& b) {
  const size_t k = l.nrows, n = l.ncols, s_ = a.size() / k;
  const unsigned nvc = log2_strict(n), nvr = log2_strict(s_);
  if (last.point.size() != nvc + nvr) throw std::runtime_error("matmul2: claim point size mismatch");
  std::vector<E> pt_right(last.point.begin(), last.point.begin() + nvc), pt_left(last.point.begin() + nvc, last.point.end());
  Mle left = Mle::from_ext(a), right = Mle::from_ext(b);
  left.fix_high_in_place(pt_left);
  if (l.transpose_b) right.fix_high_in_place(pt_right); else right.fix_low_in_place(pt_right);
  if (left.nv != right.nv) throw std::runtime_error("matmul2: inner dimensions differ");
  VirtualPolynomial vp(left.nv);
  vp.add_mle_list({mk(left), mk(right)}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();
  std::vector<E> point_left = proof.point; point_left.insert(point_left.end(), pt_left.begin(), pt_left.end());
  std::vector<E> point_right = l.transpose_b ? proof.point : pt_right;
  if (l.transpose_b) point_right.insert(point_right.end(), pt_right.begin(), pt_right.end()); else point_right.insert(point_right.end(), proof.point.begin(), proof.point.end());
  LayerProof lp; lp.kind = L_MATMUL2; lp.matmul.sumcheck = proof; lp.matmul.individual_claims = fin; lp.matmul.has_bias = false;
  ps.proofs[id] = lp;
  return {{point_left, fin[0]}, {point_right, fin[1]}};
}
// Add::prove_step without operand (layers/add.rs:81-145): both inputs evaluated at the output claim's point, two claims out, no transcript traffic
static inline std::vector<Claim> prove_add2(ProverState& ps, size_t id, const Claim& last, const std::vector<E>& a, const std::vector<E>& b) {
  E left_eval = Mle::from_ext(a).evaluate(last.point), right_eval = Mle::from_ext(b).evaluate(last.point);
  LayerProof lp; lp.kind = L_ADD2; lp.add = {left_eval, right_eval};
  ps.proofs[id] = lp;
  return {{last.point, left_eval}, {last.point, right_eval}};
}
// MatrixPermutations::split_output_claim_point (concat_matmul.rs:295-343): the point of the output claim cut into the coordinates of its
// three dimensions (the last dimension = the low variables), handed back as (concat, rows, columns) of the un-permuted result
static inline void cm_split_output_point(const Layer& l, const std::vector<E>& point, std::vector<E>& p_concat, std::vector<E>& p_row, std::vector<E>& p_col) {
  size_t os[3]; cm_output_shape(l, os);
  std::vector<E> parts[3]; size_t hi = point.size();
  size_t total = 0; for (int d = 0; d < 3; d++) total += log2_strict(os[d]);
  if (total != point.size()) throw std::runtime_error("concat matmul: point length does not match the output shape");
  for (int d = 0; d < 3; d++) { size_t nv = log2_strict(os[d]); parts[d].assign(point.begin() + (hi - nv), point.begin() + hi); hi -= nv; }
  int where[3] = {0, 1, 2};  // where[source dimension] = its position in the (permuted) output
  if (!l.cm_perm.empty()) for (int i = 0; i < 3; i++) where[l.cm_perm[i]] = i;
  p_concat = parts[where[0]]; p_row = parts[where[1]]; p_col = parts[where[2]];
}
// InputMatrixDimensions::input_mle_for_proving (concat_matmul.rs:134-166): the input with the coordinates of its output dimension fixed; what is
// left runs over (concat, mat_mul) with the mat_mul variables low
static inline Mle cm_input_mle(const std::vector<E>& x, const size_t shape[3], const int dims[3], const std::vector<E>& partial) {
  const int concat = dims[0], mm = dims[1], out = dims[2];
  if (concat > mm || out == 1) {
    const int order[3] = {concat, mm, out}; size_t t[3];
    Mle m = Mle::from_ext(permute3d(x, shape, order, t));
    m.fix_low_in_place(partial);
    return m;
  }
  Mle m = Mle::from_ext(x);
  if (out == 0) m.fix_high_in_place(partial); else m.fix_low_in_place(partial);
  return m;
}
// InputMatrixDimensions::build_point_for_input (concat_matmul.rs:116-132): the three sub-points in the order of the input's dimensions, last first
static inline std::vector<E> cm_build_point(const int dims[3], const std::vector<E>& p_concat, const std::vector<E>& p_mm, const std::vector<E>& p_out) {
  const std::vector<E>* by_dim[3] = {nullptr, nullptr, nullptr};
  by_dim[dims[0]] = &p_concat; by_dim[dims[1]] = &p_mm; by_dim[dims[2]] = &p_out;
  std::vector<E> pt;
  for (int d = 2; d >= 0; d--) pt.insert(pt.end(), by_dim[d]->begin(), by_dim[d]->end());
  return pt;
}
// ConcatMatMul::prove_step (concat_matmul.rs:467-566): sum over (chunk c, inner index j) of beta(c) * A_c[row point][j] * B_c[j][column point]
static inline std::vector<Claim> prove_concat_matmul(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& a, const std::vector<E>& b) {
  std::vector<E> p_concat, p_row, p_col;
  cm_split_output_point(l, last.point, p_concat, p_row, p_col);
  Mle left = cm_input_mle(a, l.cm_a, l.cm_left, p_row), right = cm_input_mle(b, l.cm_b, l.cm_right, p_col);
  if (left.nv != right.nv) throw std::runtime_error("concat matmul: left and right MLEs differ in size");
  const size_t M = l.cm_a[l.cm_left[1]];
  if (M != l.cm_b[l.cm_right[1]]) throw std::runtime_error("concat matmul: mat_mul dimensions differ");
  std::vector<E> betas = compute_betas_eval(p_concat), beta_evals;
  for (E e : betas) for (size_t j = 0; j < M; j++) beta_evals.push_back(e);
  Mle beta = Mle::from_ext(beta_evals);
  if (beta.nv != left.nv) throw std::runtime_error("concat matmul: beta vector of the wrong size");
  VirtualPolynomial vp(left.nv);
  vp.add_mle_list({mk(beta), mk(left), mk(right)}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> evals = st.final_evaluations();
  const unsigned nvm = log2_strict(M);  // split_sumcheck_point (:256-281): the mat_mul coordinates come first
  std::vector<E> s_mm(proof.point.begin(), proof.point.begin() + nvm), s_concat(proof.point.begin() + nvm, proof.point.end());
  LayerProof lp; lp.kind = L_CONCAT_MATMUL; lp.cmm.sumcheck = proof; lp.cmm.individual_claims = evals;
  ps.proofs[id] = lp;
  return {{cm_build_point(l.cm_left, s_concat, s_mm, p_row), evals[1]}, {cm_build_point(l.cm_right, s_concat, s_mm, p_col), evals[2]}};
}
static inline SamePolyProof same_poly_prove(const std::vector<Claim>& claims, const Mle& poly, Transcript& t);
// QKV::prove (layers/transformer/qkv.rs:462-630): the three products X W_q, X W_k, X W_v in ONE sumcheck batched with two challenges, then the
// three claims on X merged into one (same_poly)
static inline std::vector<Claim> prove_qkv(ProverState& ps, size_t id, const Layer& l, const std::vector<Claim>& last, const std::vector<E>& input) {
  const size_t k = l.nrows, n = l.ncols, s_ = input.size() / k;
  const unsigned nvc = log2_strict(n), nvr = log2_strict(s_);
  if (last.size() != 3) throw std::runtime_error("qkv: three output claims expected");
  std::vector<std::vector<E>> p_row(3), p_col(3);
  std::vector<E> bias_evals(3), pre_bias(3);
  for (int w = 0; w < 3; w++) {
    if (last[w].point.size() != nvc + nvr) throw std::runtime_error("qkv: claim point size mismatch");
    p_col[w].assign(last[w].point.begin(), last[w].point.begin() + nvc); p_row[w].assign(last[w].point.begin() + nvc, last[w].point.end());  // split_claim_point (:173-184)
    bias_evals[w] = Mle::from_i64(std::vector<int64_t>(l.bias.begin() + w * n, l.bias.begin() + (w + 1) * n)).evaluate(p_col[w]);
    pre_bias[w] = esub(last[w].eval, bias_evals[w]);
  }
  // challenges_for_batched_sumcheck (:210-232): points and bias-free evaluations enter the transcript; coefficients 1, c1, c2
  for (int w = 0; w < 3; w++) { ps.t->append_exts(last[w].point); ps.t->append_ext(pre_bias[w]); }
  std::vector<E> coeff = {e_one(), ps.t->read_challenge(), ps.t->read_challenge()};
  VirtualPolynomial vp(log2_strict(k));
  for (int w = 0; w < 3; w++) {
    Mle x = Mle::from_ext(input); x.fix_high_in_place(p_row[w]);
    Mle wm = Mle::from_i64(std::vector<int64_t>(l.weights.begin() + w * k * n, l.weights.begin() + (w + 1) * k * n)); wm.fix_low_in_place(p_col[w]);
    vp.add_mle_list({mk(x), mk(wm)}, coeff[w]);
  }
  auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
  std::vector<E> fin = st.final_evaluations();  // (input, weight) of Q, of K, of V
  if (fin.size() != 6) throw std::runtime_error("qkv: six final evaluations expected");
  std::vector<Claim> input_claims, weight_claims;
  for (int w = 0; w < 3; w++) {  // build_points (:193-204)
    std::vector<E> pi = proof.point; pi.insert(pi.end(), p_row[w].begin(), p_row[w].end());
    std::vector<E> pw = p_col[w]; pw.insert(pw.end(), proof.point.begin(), proof.point.end());
    input_claims.push_back({pi, fin[2 * w]}); weight_claims.push_back({pw, fin[2 * w + 1]});
  }
  // add_common_claims walks the node's polynomials in BTreeMap order: BiasK, BiasQ, BiasV, WeightK, WeightQ, WeightV
  const auto& comms = ps.ctx->model_comms.at(id);
  ps.add_witness_claim(comms.at("BiasK"), {p_col[1], bias_evals[1]}); ps.add_witness_claim(comms.at("BiasQ"), {p_col[0], bias_evals[0]}); ps.add_witness_claim(comms.at("BiasV"), {p_col[2], bias_evals[2]});
  ps.add_witness_claim(comms.at("WeightK"), weight_claims[1]); ps.add_witness_claim(comms.at("WeightQ"), weight_claims[0]); ps.add_witness_claim(comms.at("WeightV"), weight_claims[2]);
  SamePolyProof agg = same_poly_prove(input_claims, Mle::from_ext(input), *ps.t);
  LayerProof lp; lp.kind = L_QKV; lp.qkv.sumcheck = proof; lp.qkv.aggregation_proof = agg; lp.qkv.pre_bias_evals = pre_bias; lp.qkv.individual_claims = fin;
  ps.proofs[id] = lp;
  return {{agg.sumcheck.point, agg.evals[1]}};
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
// Activation::prove_step (activation.rs:385-456). For a GELU the claim that leaves the step is the lookup's claim on the scaled column divided by the multiplier
// (:405-413) — and, because that line re-binds `input_claim`, it is ALSO the claim the reference files with the commitment of the scaled column (:419-430),
// where the verifier files the lookup's own claim (:495-505). g_gelu_files_lookup_claim = true files the lookup's claim instead (what a verifier can
// check); false is the reference to the letter.
static bool g_gelu_files_lookup_claim = false;
static inline Claim prove_relu(ProverState& ps, size_t id, const Claim& last, const std::vector<E>& output, int64_t gelu_multiplier = 0) {
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  LogUpInput in = ps.logup_input(ws[0]);
  LogUpProof lproof = logup_batch_prove(in, *ps.t);
  Claim input_claim = lproof.output_claims[0], output_claim = lproof.output_claims[1];
  const Claim lookup_claim = input_claim;
  if (gelu_multiplier) input_claim.eval = emul(input_claim.eval, einv(e_from_i64(gelu_multiplier)));
  SamePolyProof sp = same_poly_prove({last, output_claim}, Mle::from_ext(output), *ps.t);
  ActivationProof ap; ap.io_accumulation = sp; ap.lookup = lproof;
  Claim c2{sp.sumcheck.point, sp.evals[1]};
  ps.add_witness_claim(ws[0].commits[0], gelu_multiplier && g_gelu_files_lookup_claim ? lookup_claim : input_claim); ap.commits.push_back(ws[0].commits[0].first.pure());
  ps.add_witness_claim(ws[0].commits[1], c2); ap.commits.push_back(ws[0].commits[1].first.pure());
  LayerProof lp; lp.kind = gelu_multiplier ? L_GELU : L_RELU; lp.act = ap; ps.proofs[id] = lp;
  return input_claim;
}

// LayerNorm::prove + prove_step (layernorm.rs:729-1100). Three sumchecks after the two lookups: (1) every lookup claim moved to ONE point;
// (2) at that point the inverse-square-root input (recombined with the range-checked chunks) must be multiplier (N sum x^2 - (sum x)^2), the
// output claim must be gamma (N x - sum x) inv_sqrt + beta, and the inv_sqrt column used in both is the committed one — batched with two
// challenges; sums over the normalisation dimension are evaluations with 1/2 in that dimension's coordinates times 2^k; (3) `mean` really is
// the row sum of the input.
static inline Claim prove_layernorm(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& input) {
  const size_t fd = l.weights.size();
  const unsigned sum_dim_vars = ceil_log2(fd);
  std::vector<E> mean(input.size());
  for (size_t c = 0; c < input.size() / fd; c++) { E sum = e_zero(); for (size_t i = 0; i < fd; i++) sum = eadd(sum, input[c * fd + i]); for (size_t i = 0; i < fd; i++) mean[c * fd + i] = sum; }
  MleP input_poly = mk(Mle::from_ext(input)), mean_poly = mk(Mle::from_ext(mean));
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  if (ws.size() != 2) throw std::runtime_error("LayerNorm requires two lookups");
  LayerNormProof pr;
  for (auto& w : ws) pr.logup_proofs.push_back(logup_batch_prove(ps.logup_input(w), *ps.t));
  const std::vector<Claim>& inv_claims = pr.logup_proofs[0].output_claims; const std::vector<Claim>& range_claims = pr.logup_proofs[1].output_claims;
  const unsigned nv = (unsigned)inv_claims[0].point.size();
  std::vector<E> bc; for (unsigned q = 0; q < ceil_log2(inv_claims.size() + range_claims.size()); q++) bc.push_back(ps.t->get_and_append_challenge("batching"));
  std::vector<E> rlc = compute_betas_eval(bc);
  MleP sqrt_eq = mk(Mle::from_ext(compute_betas_eval(inv_claims[0].point))), range_eq = mk(Mle::from_ext(compute_betas_eval(range_claims[0].point)));
  std::vector<ProverCommitment> commits = ws[0].commits; commits.insert(commits.end(), ws[1].commits.begin(), ws[1].commits.end());
  {
    VirtualPolynomial vp(nv);
    for (size_t q = 0; q < commits.size(); q++) vp.add_mle_list({mk(commits[q].second), q < 2 ? sqrt_eq : range_eq}, rlc.at(q));
    auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
    pr.accumulation_proof = proof;
    std::vector<E> fin = st.final_evaluations();  // sqrt_in, sqrt_eq, sqrt_out, range_0, range_eq, range_1, ...
    pr.acc_evals = {fin[0], fin[2], fin[3]}; pr.acc_evals.insert(pr.acc_evals.end(), fin.begin() + 5, fin.end());
  }
  const std::vector<E> sc_point = pr.accumulation_proof.point;
  const E two_inv = einv(e_from_u64(2)), two_mul = e_from_u64(u64(1) << sum_dim_vars);
  const E c1 = ps.t->get_and_append_challenge("batching"), c2 = ps.t->get_and_append_challenge("batching");
  const E first = emul(esub(e_one(), c1), esub(e_one(), c2)), second = emul(c1, esub(e_one(), c2)), third = emul(esub(e_one(), c1), c2);
  std::vector<E> full_point(sum_dim_vars, two_inv); full_point.insert(full_point.end(), sc_point.begin(), sc_point.end());
  if (full_point.size() != last.point.size()) throw std::runtime_error("layernorm: claim point size mismatch");
  MleP input_eq = mk(Mle::from_ext(compute_betas_eval(full_point)));
  const E n_f = e_from_u64(l.ln_dim_size), mult_f = e_from_i64(l.ln_multiplier);
  const size_t repeats = size_t(1) << (last.point.size() - sum_dim_vars);
  std::vector<E> g(input.size()), b(input.size()), inv(input.size());
  for (size_t c = 0; c < repeats; c++) for (size_t i = 0; i < fd; i++) { g[c * fd + i] = e_from_i64(l.weights[i]); b[c * fd + i] = e_from_i64(l.bias[i]); inv[c * fd + i] = e_from_u64(ws[0].column_evals[1][c]); }
  MleP gamma_poly = mk(Mle::from_ext(g)), beta_poly = mk(Mle::from_ext(b)), inv_poly = mk(Mle::from_ext(inv)), last_eq = mk(Mle::from_ext(compute_betas_eval(last.point)));
  E input_eval, mean_eval, inv_eval;
  {
    VirtualPolynomial vp((unsigned)full_point.size());
    vp.add_mle_list({input_eq, input_poly, input_poly}, emul(emul(first, mult_f), emul(n_f, two_mul)));
    vp.add_mle_list({input_eq, mean_poly, mean_poly}, eneg(emul(first, mult_f)));
    vp.add_mle_list({last_eq, gamma_poly, input_poly, inv_poly}, emul(second, n_f));
    vp.add_mle_list({last_eq, gamma_poly, mean_poly, inv_poly}, eneg(second));
    vp.add_mle_list({last_eq, beta_poly}, second);
    vp.add_mle_list({input_eq, inv_poly}, third);
    auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
    pr.io_proof = proof;
    std::vector<E> fin = st.final_evaluations();  // input_eq, input, mean, last_eq, gamma, inv_sqrt_out, beta
    input_eval = fin[1]; mean_eval = fin[2]; pr.gamma_eval = fin[4]; inv_eval = fin[5]; pr.beta_eval = fin[6];
  }
  const std::vector<E> io_point = pr.io_proof.point;
  const E ic = ps.t->get_and_append_challenge("batching");
  Claim input_claim;
  {
    std::vector<E> sum_io(sum_dim_vars, two_inv); sum_io.insert(sum_io.end(), io_point.begin() + sum_dim_vars, io_point.end());
    VirtualPolynomial vp((unsigned)io_point.size());
    vp.add_mle_list({input_poly, mk(Mle::from_ext(compute_betas_eval(io_point)))}, esub(e_one(), ic));
    vp.add_mle_list({input_poly, mk(Mle::from_ext(compute_betas_eval(sum_io)))}, emul(ic, two_mul));
    auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t);
    pr.input_proof = proof;
    input_claim = {proof.point, st.final_evaluations()[0]};
  }
  // witness claims: the lookup input at the accumulation point, the lookup output at the tail of the io point, the range chunks at the accumulation point
  std::vector<Claim> cl = {{sc_point, pr.acc_evals[0]}, {std::vector<E>(io_point.begin() + sum_dim_vars, io_point.end()), inv_eval}};
  for (size_t q = 2; q < pr.acc_evals.size(); q++) cl.push_back({sc_point, pr.acc_evals[q]});
  for (size_t q = 0; q < cl.size(); q++) { pr.commitments.push_back(commits[q].first.pure()); pr.evaluations.push_back(cl[q].eval); ps.add_witness_claim(commits[q], cl[q]); }
  pr.evaluations.push_back(input_eval); pr.evaluations.push_back(mean_eval);
  // add_common_claims walks the node's BTreeMap: "LayerNormBeta" then "LayerNormGamma"
  const std::vector<E> gp(io_point.begin(), io_point.begin() + sum_dim_vars);
  const auto& comms = ps.ctx->model_comms.at(id);
  ps.add_witness_claim(comms.at("LayerNormBeta"), {gp, pr.beta_eval});
  ps.add_witness_claim(comms.at("LayerNormGamma"), {gp, pr.gamma_eval});
  LayerProof lp; lp.kind = L_LAYERNORM; lp.ln = pr; ps.proofs[id] = lp;
  return input_claim;
}

// Softmax::prove_step (softmax.rs:573-888). Lookups: (input, output) of the exponential table, the two low bytes, the row sums against the table
// of values within the allowable error of one, the bits above the exponential's input against the zero table. One sumcheck brings every
// lookup claim to a single point and ties the output claim to exp_out * prod zero_out and the row sums to the same product (1/2 in the
// coordinates of the normalisation dimension, times 2^k); a second one shows the masked input is shifted_input * tril + bias.
// The claim handed on is the one the VERIFIER derives, (shifted - shift) / scalar (:1541-1543); the reference's prover returns the same value
// without the division (:748-749), which no later prover reads.
static inline Claim prove_softmax(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<int64_t>& input) {
  SoftmaxData d; softmax_op(l, input, &d);
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  const bool zero = l.sm_zero_chunks != 0;
  if (ws.size() != (zero ? 4u : 3u)) throw std::runtime_error("softmax: lookup witnesses");
  SoftmaxProof pr;
  for (auto& w : ws) pr.logup_proofs.push_back(logup_batch_prove(ps.logup_input(w), *ps.t));
  const std::vector<E> exp_point = pr.logup_proofs[0].output_claims.at(0).point, range_point = pr.logup_proofs[1].output_claims.at(0).point, error_point = pr.logup_proofs[2].output_claims.at(0).point;
  const size_t extra = exp_point.size() - error_point.size();
  const E two_inv = einv(e_from_u64(2)), two_mult = e_from_u64(u64(1) << extra);
  std::vector<E> full_error(extra, two_inv); full_error.insert(full_error.end(), error_point.begin(), error_point.end());
  const E alpha = ps.t->get_and_append_challenge("batching_challenge");
  MleP exp_beta = mk(Mle::from_ext(compute_betas_eval(exp_point))), range_beta = mk(Mle::from_ext(compute_betas_eval(range_point)));
  MleP error_beta = mk(Mle::from_ext(compute_betas_eval(full_error))), last_beta = mk(Mle::from_ext(compute_betas_eval(last.point)));
  VirtualPolynomial vp((unsigned)exp_point.size());
  E bc = e_one();
  for (auto& c : ws[0].commits) { vp.add_mle_list({mk(c.second), exp_beta}, bc); bc = emul(bc, alpha); }
  for (auto& c : ws[1].commits) { vp.add_mle_list({mk(c.second), range_beta}, bc); bc = emul(bc, alpha); }
  const Mle& exp_output = ws[0].commits[1].second;
  if (zero) {
    MleP zbeta = mk(Mle::from_ext(compute_betas_eval(pr.logup_proofs[3].output_claims.at(0).point)));
    for (auto& c : ws[3].commits) { vp.add_mle_list({zbeta, mk(c.second)}, bc); bc = emul(bc, alpha); }
    std::vector<MleP> prod; for (size_t q = 1; q < ws[3].commits.size(); q += 2) prod.push_back(mk(ws[3].commits[q].second));
    prod.push_back(mk(exp_output));
    std::vector<MleP> err = prod, outp = prod; err.push_back(error_beta); outp.push_back(last_beta);
    vp.add_mle_list(err, emul(bc, two_mult));
    vp.add_mle_list(outp, emul(bc, alpha));
  } else {
    vp.add_mle_list({mk(exp_output), error_beta}, emul(bc, two_mult));
    vp.add_mle_list({mk(exp_output), last_beta}, emul(bc, alpha));
  }
  std::vector<E> all;
  { auto [proof, st] = sumcheck_prove(std::move(vp), *ps.t); pr.accumulation_proof = proof; all = st.final_evaluations(); }
  const std::vector<E> sc_point = pr.accumulation_proof.point;
  // the mask: eq(sc_point, x) (shifted_input(x) tril(x) + bias(x))
  E shifted_eval;
  {
    MleP meq = mk(Mle::from_ext(compute_betas_eval(sc_point)));
    VirtualPolynomial mv((unsigned)sc_point.size());
    mv.add_mle_list({mk(Mle::from_i64(d.shifted_input)), mk(Mle::from_i64(d.tril)), meq}, e_one());
    mv.add_mle_list({mk(Mle::from_i64(d.bias)), meq}, e_one());
    auto [proof, st] = sumcheck_prove(std::move(mv), *ps.t);
    pr.mask_proof = proof; shifted_eval = st.final_evaluations()[0];
  }
  const ProverCommitment& shift_c = ws[2].commits[0];
  const std::vector<E> shift_point(pr.mask_proof.point.begin() + (pr.mask_proof.point.size() - shift_c.second.nv), pr.mask_proof.point.end());
  const E shift_eval = shift_c.second.evaluate(shift_point);
  const std::vector<E> evs4 = {all[0], all[2], all[3], all[5]};
  for (size_t q = 0; q < 4; q++) { const ProverCommitment& c = q < 2 ? ws[0].commits[q] : ws[1].commits[q - 2]; ps.add_witness_claim(c, {sc_point, evs4[q]}); pr.commitments.push_back(c.first.pure()); pr.evaluations.push_back(evs4[q]); }
  ps.add_witness_claim(shift_c, {shift_point, shift_eval}); pr.commitments.push_back(shift_c.first.pure()); pr.evaluations.push_back(shift_eval);
  if (zero) for (size_t q = 0; q < ws[3].commits.size(); q++) { ps.add_witness_claim(ws[3].commits[q], {sc_point, all[7 + q]}); pr.commitments.push_back(ws[3].commits[q].first.pure()); pr.evaluations.push_back(all[7 + q]); }
  const std::vector<E> mask_point = pr.mask_proof.point;
  LayerProof lp; lp.kind = L_SOFTMAX; lp.sm = pr; ps.proofs[id] = lp;
  return {mask_point, emul(esub(shifted_eval, shift_eval), einv(e_from_i64(l.sm_scalar)))};
}

// ------------------------------------------------------------------ convolution (zkCNN FFT protocol)
// phi_pow_init (iop/prover.rs:214-227): powers of the 2^n-th root of unity (inverted when `is_fft`)
static inline std::vector<E> phi_pow_init(unsigned n, bool is_fft) {
  size_t length = size_t(1) << n;
  E phi = get_root_of_unity(n);
  if (is_fft) phi = einv(phi);
  std::vector<E> pm(length); pm[0] = e_one();
  for (size_t i = 1; i < length; i++) pm[i] = emul(pm[i - 1], phi);
  return pm;
}
// phi_g_init (iop/prover.rs:231-284): the FFT / iFFT matrix reduced at rx, with the intermediate tables
static inline void phi_g_init(std::vector<E>& phi_g, std::vector<std::vector<E>>& mid, const std::vector<E>& rx, E scale, unsigned n, bool is_fft) {
  std::vector<E> phi_mul = phi_pow_init(n, is_fft);
  if (is_fft) {
    phi_g[0] = scale; phi_g[1] = scale;
    for (unsigned i = 1; i < n + 1; i++) {
      for (size_t b = 0; b < (size_t(1) << (i - 1)); b++) {
        size_t l = b, r = b ^ (size_t(1) << (i - 1)); unsigned m = n - i;
        E tmp1 = esub(e_one(), rx[m]), tmp2 = emul(rx[m], phi_mul[b << m]);
        phi_g[r] = emul(phi_g[l], esub(tmp1, tmp2));
        phi_g[l] = emul(phi_g[l], eadd(tmp1, tmp2));
      }
      if (i < n) mid[i - 1].assign(phi_g.begin(), phi_g.begin() + (size_t(1) << i));
    }
  } else {
    phi_g[0] = scale;
    for (unsigned i = 1; i < n; i++) {
      for (size_t b = 0; b < (size_t(1) << (i - 1)); b++) {
        size_t l = b, r = b ^ (size_t(1) << (i - 1)); unsigned m = n - i;
        E tmp1 = esub(e_one(), rx[m]), tmp2 = emul(rx[m], phi_mul[b << m]);
        phi_g[r] = emul(phi_g[l], esub(tmp1, tmp2));
        phi_g[l] = emul(phi_g[l], eadd(tmp1, tmp2));
      }
      mid[i - 1].assign(phi_g.begin(), phi_g.begin() + (size_t(1) << i));
    }
    for (size_t b = 0; b < (size_t(1) << (n - 1)); b++) {
      E tmp1 = esub(e_one(), rx[0]), tmp2 = emul(rx[0], phi_mul[b]);
      phi_g[b] = emul(phi_g[b], eadd(tmp1, tmp2));
    }
  }
}
struct MatrixEval { std::vector<IOPProof> proofs; std::vector<std::vector<E>> claims; };
// delegate_matrix_evaluation (iop/prover.rs:164-211)
static inline MatrixEval delegate_matrix_evaluation(Transcript& t, std::vector<std::vector<E>>& f_middle, const std::vector<E>& r1, std::vector<E> r2, bool is_fft) {
  std::vector<E> omegas = phi_pow_init((unsigned)r1.size(), is_fft);
  MatrixEval me;
  size_t fm = f_middle.size();
  for (size_t l = r1.size() - 1; l-- > 0;) {
    std::vector<E> phi(f_middle[l].size());
    std::vector<E> beta = compute_betas_eval(std::vector<E>(r2.begin(), r2.end() - 1));
    E r2l = r2.back(), r1e = r1[(fm - 1) - l];
    for (size_t i = 0; i < phi.size(); i++) {
      E om = omegas[i << ((fm - 1) - l)];
      if (!is_fft && l == fm - 1) phi[i] = emul(esub(e_one(), r2l), eadd(esub(e_one(), r1e), emul(r1e, om)));
      else phi[i] = eadd(esub(e_one(), r1e), emul(emul(esub(e_one(), emul(e_from_u64(2), r2l)), r1e), om));
    }
    MleP f1 = mk(Mle::from_ext(beta)), f2 = mk(Mle::from_ext(phi)), f3 = mk(Mle::from_ext(f_middle[l]));
    VirtualPolynomial vp(f1->nv);
    vp.add_mle_list({f1, f2, f3}, e_one());
    auto [proof, st] = sumcheck_prove(std::move(vp), t);
    r2 = proof.point;
    me.proofs.push_back(proof); me.claims.push_back(st.final_evaluations());
  }
  return me;
}
struct BatchFFTProof { IOPProof proof; std::vector<E> claims; MatrixEval matrix_eval; std::vector<E> partial_evals; };
static inline std::vector<E> flatten2(const std::vector<std::vector<E>>& x) { std::vector<E> o; for (auto& v : x) o.insert(o.end(), v.begin(), v.end()); return o; }
// prove_batch_fft (iop/prover.rs:290-338)
static inline BatchFFTProof prove_batch_fft(Transcript& t, const std::vector<E>& r, std::vector<std::vector<E>> x) {
  size_t padded_rows = 2 * x[0].size();
  for (auto& it : x) it.resize(padded_rows, e_zero());
  unsigned l1 = log2_strict(x[0].size()), l2 = log2_strict(x.size());
  std::vector<E> r1(r.begin(), r.begin() + l1), r2(r.begin() + l1, r.begin() + l1 + l2);
  std::vector<E> w_red(x[0].size(), e_zero()); std::vector<std::vector<E>> f_middle(l1 - 1);
  phi_g_init(w_red, f_middle, r1, e_one(), l1, false);
  Mle f_m = Mle::from_ext(flatten2(x));
  f_m.fix_high_in_place(r2);
  MleP fm = mk(f_m), fr = mk(Mle::from_ext(w_red));
  VirtualPolynomial vp(fm->nv);
  vp.add_mle_list({fm, fr}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), t);
  BatchFFTProof out; out.proof = proof; out.claims = st.final_evaluations();
  out.matrix_eval = delegate_matrix_evaluation(t, f_middle, r1, proof.point, false);
  return out;
}
// prove_batch_ifft (iop/prover.rs:340-398)
static inline BatchFFTProof prove_batch_ifft(Transcript& t, const std::vector<E>& r, const std::vector<std::vector<E>>& prod) {
  E scale = einv(e_from_u64(prod[0].size()));
  unsigned l1 = log2_strict(prod[0].size()), l2 = log2_strict(prod.size());
  std::vector<E> r1(r.begin(), r.begin() + l1), r2(r.begin() + l1, r.begin() + l1 + l2);
  if (!e_is_zero(r1[l1 - 1])) throw std::runtime_error("Error in randomness init batch ifft");
  std::vector<E> w_red(prod[0].size(), e_zero()); std::vector<std::vector<E>> f_middle(l1 - 1);
  phi_g_init(w_red, f_middle, r1, scale, l1, true);
  MleP fr = mk(Mle::from_ext(w_red));
  Mle f_m = Mle::from_ext(flatten2(prod));
  f_m.fix_high_in_place(r2);
  MleP fm = mk(f_m);
  VirtualPolynomial vp(fm->nv);
  vp.add_mle_list({fm, fr}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), t);
  BatchFFTProof out; out.proof = proof; out.claims = st.final_evaluations();
  out.matrix_eval = delegate_matrix_evaluation(t, f_middle, r1, proof.point, true);
  return out;
}
// Convolution::prove_batch_fft_weights (convolution.rs:358-443)
static inline BatchFFTProof prove_batch_fft_weights(Transcript& t, const Layer& l, const std::vector<E>& r) {
  size_t padded_rows = 2 * l.nw * l.nw, fsz = l.real_nw * l.real_nw;
  unsigned l1 = log2_strict(padded_rows);
  std::vector<E> r1(r.begin(), r.begin() + l1), r2(r.begin() + l1, r.end());
  std::vector<E> w_red(padded_rows, e_zero()); std::vector<std::vector<E>> f_middle(l1 - 1);
  std::vector<E> beta = compute_betas_eval(r2);
  phi_g_init(w_red, f_middle, r1, e_one(), l1, false);
  std::vector<E> w1_reduced(fsz, e_zero());
  for (size_t i = 0; i < l.kw; i++) for (size_t j = 0; j < l.kx; j++) for (size_t k = 0; k < fsz; k++)
    w1_reduced[k] = eadd(w1_reduced[k], emul(beta[i * l.kx + j], e_from_i64(l.weights[i * fsz * l.kx + j * fsz + k])));
  BatchFFTProof out; out.partial_evals = w1_reduced;
  MleP fm = mk(Mle::from_ext(index_wf(w1_reduced, l.real_nw, l.nw, padded_rows))), fr = mk(Mle::from_ext(w_red));
  VirtualPolynomial vp(fm->nv);
  vp.add_mle_list({fm, fr}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), t);
  out.proof = proof; out.claims = st.final_evaluations();
  out.matrix_eval = delegate_matrix_evaluation(t, f_middle, r1, proof.point, false);
  return out;
}
// hadamard::prove (hadamard.rs:83-130)
static inline HadamardProof hadamard_prove(Transcript& t, const Claim& output_claim, const std::vector<int64_t>& v1, const std::vector<int64_t>& v2) {
  if (output_claim.point.size() != log2_strict(v1.size()) || v1.size() != v2.size()) throw std::runtime_error("hadamard: shapes");
  auto to_e = [](const std::vector<int64_t>& v) { std::vector<E> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = e_from_i64(v[i]); return o; };
  MleP beta = mk(Mle::from_ext(compute_betas_eval(output_claim.point))), m1 = mk(Mle::from_ext(to_e(v1))), m2 = mk(Mle::from_ext(to_e(v2)));
  VirtualPolynomial vp(m1->nv);
  vp.add_mle_list({m1, m2, beta}, e_one());
  auto [proof, st] = sumcheck_prove(std::move(vp), t);
  std::vector<E> fin = st.final_evaluations();
  return {proof, {fin[0], fin[1]}};
}
// Convolution::prove_convolution_step (convolution.rs:697-1080)
static inline Claim prove_conv(ProverState& ps, size_t id, const Layer& l, const Claim& last_in, const ConvData& pd) {
  Transcript& t = *ps.t;
  size_t padded_out[3] = {l.kw, l.nw, l.nw};
  std::vector<int64_t> clearing = new_clearing_tensor(l.unp_out, padded_out);
  HadamardProof clearing_proof = hadamard_prove(t, last_in, pd.output_as_element, clearing);
  Claim last{clearing_proof.sumcheck.point, clearing_proof.individual_claim[0]};
  unsigned lfs = log2_strict(l.filter_size()), lkw = log2_strict(l.kw);
  if (lfs + lkw != last.point.size()) throw std::runtime_error("conv: inconsistent random point size");
  std::vector<E> r(last.point.size() + 1, e_zero()), bias_point(lkw, e_zero());
  for (unsigned i = 0; i < lfs; i++) r[i] = esub(e_one(), last.point[i]);
  for (unsigned i = 0; i < lkw; i++) { r[i + lfs + 1] = last.point[i + lfs]; bias_point[i] = last.point[i + lfs]; }
  E bias_eval = e_zero();
  if (!bias_point.empty()) bias_eval = Mle::from_i64(l.bias).evaluate(bias_point);
  else if (l.bias.size() == 1) bias_eval = e_from_i64(l.bias[0]);
  BatchFFTProof ifft = prove_batch_ifft(t, r, pd.prod);
  if (ifft.proof.point.size() != lfs + 1) throw std::runtime_error("Error in ifft sumcheck");
  std::vector<E> r_ifft = ifft.proof.point;
  unsigned lo = log2_strict(pd.output[0].size());
  for (size_t i = lo; i < r.size(); i++) r_ifft.push_back(r[i]);
  std::vector<E> r1(r_ifft.begin() + lo, r_ifft.end()), r2(r_ifft.begin(), r_ifft.begin() + lo);
  std::vector<E> beta1 = compute_betas_eval(r1), beta2 = compute_betas_eval(r2);
  std::vector<E> beta_acc; for (size_t i = 0; i < l.kx; i++) beta_acc.insert(beta_acc.end(), beta2.begin(), beta2.end());
  size_t fsz = l.real_nw * l.real_nw;
  std::vector<std::vector<E>> agg(l.kx, std::vector<E>(fsz, e_zero()));
  for (size_t i = 0; i < l.kx; i++) {
    for (size_t j = 0; j < l.kw; j++) for (size_t k = 0; k < fsz; k++)
      agg[i][k] = eadd(agg[i][k], emul(beta1[j], e_from_i64(l.weights[j * l.kx * fsz + i * fsz + k])));
    agg[i] = index_wf(agg[i], l.real_nw, l.nw, 2 * l.nw * l.nw);
    fft(agg[i], false);
  }
  MleP f1 = mk(Mle::from_ext(flatten2(agg))), f2 = mk(Mle::from_ext(flatten2(pd.input_fft))), f3 = mk(Mle::from_ext(beta_acc));
  VirtualPolynomial vp(f1->nv);
  vp.add_mle_list({f1, f2, f3}, e_one());
  auto [hadamard_proof, hst] = sumcheck_prove(std::move(vp), t);
  std::vector<E> hadamard_claims = hst.final_evaluations();
  std::vector<E> point = hadamard_proof.point; point.insert(point.end(), r1.begin(), r1.end());
  BatchFFTProof fftp = prove_batch_fft(t, hadamard_proof.point, pd.input);
  BatchFFTProof wp = prove_batch_fft_weights(t, l, point);
  std::vector<E> weights_rand = t.read_challenges(log2_strict(fsz));
  unsigned l2n = log2_strict(2 * l.nw * l.nw);
  Claim bias_claim{bias_point, bias_eval};
  Claim filter_claim; filter_claim.point = weights_rand; filter_claim.point.insert(filter_claim.point.end(), point.begin() + l2n, point.end());
  filter_claim.eval = Mle::from_ext(wp.partial_evals).evaluate(weights_rand);
  const auto& comms = ps.ctx->model_comms.at(id);  // add_common_claims: BTreeMap order "ConvBias" < "ConvFilter"
  ps.add_witness_claim(comms.at("ConvBias"), bias_claim);
  ps.add_witness_claim(comms.at("ConvFilter"), filter_claim);
  ConvProof cp;
  cp.fft_proof = fftp.proof; cp.fft_claims = fftp.claims; cp.fft_proof_weights = wp.proof; cp.ifft_proof = ifft.proof;
  cp.fft_delegation_proof = fftp.matrix_eval.proofs; cp.fft_delegation_proof_weights = wp.matrix_eval.proofs; cp.ifft_delegation_proof = ifft.matrix_eval.proofs;
  cp.hadamard_proof = hadamard_proof; cp.ifft_claims = ifft.claims; cp.fft_weight_claims = wp.claims;
  cp.fft_delegation_claims = fftp.matrix_eval.claims; cp.fft_delegation_weights_claims = wp.matrix_eval.claims; cp.ifft_delegation_claims = ifft.matrix_eval.claims;
  cp.hadamard_clams = hadamard_claims; cp.bias_claim = bias_eval; cp.partial_evals = wp.partial_evals; cp.clearing_proof = clearing_proof;
  LayerProof lp; lp.kind = L_CONV; lp.conv = cp; ps.proofs[id] = lp;
  std::vector<E> input_point = fftp.proof.point;
  E v = input_point.back(); input_point.pop_back();
  v = einv(esub(e_one(), v));
  for (auto& ip : input_point) ip = esub(e_one(), ip);
  Claim fin; fin.point = input_point;
  fin.point.insert(fin.point.end(), hadamard_proof.point.begin() + log2_strict(l.filter_size() * 2), hadamard_proof.point.end());
  fin.eval = emul(fftp.claims[0], v);
  return fin;
}
// Pooling::prove_pooling (pooling.rs:342-520)
static inline Claim prove_pooling(ProverState& ps, size_t id, const Layer& l, const Claim& last, const std::vector<E>& output) {
  Transcript& t = *ps.t;
  std::vector<LogUpWitness> ws = ps.lookup_witness.at(id);
  if (ws.size() != 1) throw std::runtime_error("pooling: one lookup witness expected");
  LogUpInput in = ps.logup_input(ws[0]);
  LogUpProof lproof = logup_batch_prove(in, t);
  unsigned nv = log2_strict(output.size());
  std::vector<MleP> diff; for (auto& c : in.column_evals) diff.push_back(mk(Mle::from_base(c)));
  VirtualPolynomial vp(nv);
  const std::vector<E>& lookup_point = lproof.output_claims[0].point;
  E batch = t.get_and_append_challenge("batch_pooling");
  MleP beta = mk(Mle::from_ext(compute_betas_eval(lookup_point))), last_beta = mk(Mle::from_ext(compute_betas_eval(last.point)));
  E comb = batch;
  std::vector<std::pair<std::vector<MleP>, E>> parts;
  for (auto& d : diff) { parts.push_back({{d, beta}, comb}); comb = emul(comb, batch); }
  std::vector<MleP> all = diff; all.push_back(beta);
  vp.add_mle_list(all, e_one());
  for (auto& pr : parts) vp.add_mle_list(pr.first, pr.second);
  MleP out_mle = mk(Mle::from_ext(output));
  vp.add_mle_list({out_mle, last_beta}, comb);
  auto [proof, st] = sumcheck_prove(std::move(vp), t);
  std::vector<E> evals = st.final_evaluations();
  size_t ks = 4;  // kernel_size^2
  E output_eval = evals[ks + 1];
  PoolingProof pp; pp.sumcheck = proof; pp.lookup = lproof;
  for (size_t i = 0; i <= ks; i++) {
    E ev = i < ks ? evals[i] : output_eval;
    pp.commitments.push_back(ws[0].commits[i].first.pure());
    ps.add_witness_claim(ws[0].commits[i], {proof.point, ev});
  }
  unsigned row_log = ceil_log2(l.pin[2]);
  E r1 = t.get_and_append_challenge("input_batching"), r2 = r1;  // [x; 2]: one challenge, used twice (pooling.rs:463-466)
  E om1 = esub(e_one(), r1), om2 = esub(e_one(), r2);
  E mult[4] = {emul(om1, om2), emul(om1, r2), emul(r1, om2), emul(r1, r2)};
  E zc = e_zero();
  for (size_t i = 0; i < ks; i++) zc = eadd(zc, emul(mult[i], esub(output_eval, evals[i])));
  Claim next; next.point.push_back(r1);
  next.point.insert(next.point.end(), proof.point.begin(), proof.point.begin() + (row_log - 1));
  next.point.push_back(r2);
  next.point.insert(next.point.end(), proof.point.begin() + (row_log - 1), proof.point.end());
  next.eval = zc;
  pp.zerocheck_evals.assign(evals.begin(), evals.begin() + ks); pp.zerocheck_evals.push_back(output_eval);
  pp.variable_gap = row_log - 1;
  LayerProof lp; lp.kind = L_MAXPOOL; lp.pool = pp; ps.proofs[id] = lp;
  return next;
}

// Prover::prove (iop/prover.rs:401-488)
static inline Proof prove(const Context& ctx, const Trace& tr, Transcript& t) {
  ProverState ps; ps.ctx = &ctx; ps.t = &t;
  // ctx.write_to_transcript: every model commitment root, BTreeMap order (commit/context.rs:181-192)
  for (auto& [id, m] : ctx.model_comms) for (auto& [pid, pc] : m) t.append_digest(pc.first.codeword_tree.root());
  instantiate_witness_ctx(ps, tr);
  auto to_fields = [](const std::vector<int64_t>& v) { std::vector<E> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = e_from_i64(v[i]); return o; };
  // one claim per output tensor of the model (iop/prover.rs:419-435), then the nodes in backward order, each handing one claim per input to
  // whoever produces that input (claims_for_node, provable/mod.rs:235-270)
  const Model& mdl = ctx.model;
  std::vector<Claim> out_claims;
  for (const Wire& w : model_outputs(mdl)) {
    const std::vector<int64_t>& out = tr.output((size_t)w.node, (size_t)w.index);
    std::vector<E> r = t.read_challenges(log2_strict(out.size()));
    out_claims.push_back({r, Mle::from_ext(to_fields(out)).evaluate(r)});
  }
  std::map<size_t, std::vector<Claim>> claims_by_node;
  for (size_t id : backward_order(mdl)) {
    const Layer& l = mdl.layers[id];
    std::vector<Claim> last;
    for (size_t j = 0; j < n_outputs(l); j++) { WireUse u = wire_use(mdl, (int)id, (int)j); last.push_back(u.node < 0 ? out_claims.at((size_t)u.pos) : claims_by_node.at((size_t)u.node).at((size_t)u.pos)); }
    Claim cur = last[0];
    if (l.kind == L_MATMUL2) { claims_by_node[id] = prove_matmul2(ps, id, l, cur, to_fields(tr.in[id]), to_fields(tr.in2[id])); continue; }
    if (l.kind == L_ADD2) { claims_by_node[id] = prove_add2(ps, id, cur, to_fields(tr.in[id]), to_fields(tr.in2[id])); continue; }
    if (l.kind == L_CONCAT_MATMUL) { claims_by_node[id] = prove_concat_matmul(ps, id, l, cur, to_fields(tr.in[id]), to_fields(tr.in2[id])); continue; }
    if (l.kind == L_QKV) { claims_by_node[id] = prove_qkv(ps, id, l, last, to_fields(tr.in[id])); continue; }
    if (l.kind == L_MHA) {  // Mha::prove (mha.rs:633-704): final_mul on (softmax_out, V), the softmax, qk on (Q, K); the claims on Q, K, V in this order
      const MhaData& d = tr.mha.at(id);
      std::vector<Claim> fm = prove_concat_matmul(ps, id, mha_final_layer(l), cur, to_fields(d.softmax_out), to_fields(tr.in3[id]));
      LayerProof lp; lp.kind = L_MHA; lp.mha_final = ps.proofs.at(id).cmm;
      const Claim sc = prove_softmax(ps, id, mha_softmax_layer(l), fm[0], d.softmax_in);
      lp.sm = ps.proofs.at(id).sm;
      std::vector<Claim> qk = prove_concat_matmul(ps, id, mha_qk_layer(l), sc, to_fields(tr.in[id]), to_fields(tr.in2[id]));
      lp.mha_qk = ps.proofs.at(id).cmm;
      ps.proofs[id] = lp;
      claims_by_node[id] = {qk[0], qk[1], fm[1]};
      continue;
    }
    if (l.kind == L_DENSE) cur = prove_dense(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_MATMUL) cur = prove_matmul(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_ADD) cur = prove_add(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_EMBED) cur = prove_embeddings(ps, id, l, cur, tr.in[id]);
    else if (l.kind == L_POSITIONAL) cur = prove_positional(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_REQUANT) cur = prove_requant(ps, id, l, cur);
    else if (l.kind == L_RELU) cur = prove_relu(ps, id, cur, to_fields(tr.out[id]));
    else if (l.kind == L_GELU) cur = prove_relu(ps, id, cur, to_fields(tr.out[id]), l.gelu_multiplier);
    else if (l.kind == L_LAYERNORM) cur = prove_layernorm(ps, id, l, cur, to_fields(tr.in[id]));
    else if (l.kind == L_SOFTMAX) cur = prove_softmax(ps, id, l, cur, tr.in[id]);
    else if (l.kind == L_CONV) cur = prove_conv(ps, id, l, cur, tr.conv[id]);
    else if (l.kind == L_MAXPOOL) cur = prove_pooling(ps, id, l, cur, to_fields(tr.out[id]));
    // L_FLATTEN is not provable: the claim is propagated unchanged (iop/prover.rs:449-456)
    claims_by_node[id] = {cur};
  }
  Proof proof;
  // prove_tables (iop/prover.rs:110-157)
  for (auto& tw : ps.table_witness) {
    LogUpInput in = ps.logup_input(tw);
    LogUpProof tp = logup_batch_prove(in, t);
    ps.add_witness_claim(tw.commits[0], tp.output_claims[0]);
    // table_claims (lookup/context.rs:548-563): InverseSQRT hands the claim on its committed output column (the last one) to the opening
    if (tw.table_type.has_committed_column()) ps.add_witness_claim(ctx.table_comms.at(tw.table_type), tp.output_claims.back());
    proof.table_proofs.push_back({tw.commits[0].first.pure(), tp});
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
    else if (lp.kind == L_ADD) { w.e(lp.add.left_eval); w.e(lp.add.right_eval); }
    else if (lp.kind == L_EMBED) { w.iop(lp.matmul.sumcheck); w.ve(lp.matmul.individual_claims); }
    else if (lp.kind == L_POSITIONAL) { w.u(1); w.ve(lp.pos.sub_matrix_evals); w.e(lp.pos.add_proof.left_eval); w.e(lp.pos.add_proof.right_eval); }  // PositionalProof {proofs: [one per input]}
    else if (lp.kind == L_ADD2) { w.e(lp.add.left_eval); w.e(lp.add.right_eval); }
    else if (lp.kind == L_MATMUL2) { w.iop(lp.matmul.sumcheck); w.ve(lp.matmul.individual_claims); w.u(0); }  // (bias_eval: None)
    else if (lp.kind == L_CONCAT_MATMUL) { w.iop(lp.cmm.sumcheck); w.ve(lp.cmm.individual_claims); }
    else if (lp.kind == L_QKV) { w.iop(lp.qkv.sumcheck); w.iop(lp.qkv.aggregation_proof.sumcheck); w.ve(lp.qkv.aggregation_proof.evals); w.ve(lp.qkv.pre_bias_evals); w.ve(lp.qkv.individual_claims); }
    else if (lp.kind == L_MATMUL) { w.iop(lp.matmul.sumcheck); w.ve(lp.matmul.individual_claims); w.u(lp.matmul.has_bias ? 1 : 0); if (lp.matmul.has_bias) w.e(lp.matmul.bias_eval); }
    else if (lp.kind == L_REQUANT) {
      w.iop(lp.req.io_accumulation); w.ve(lp.req.accumulation_evals); w.logup(lp.req.clamping_lookup); w.logup(lp.req.shifted_lookup);
      w.u(lp.req.commitments.size()); for (auto& c : lp.req.commitments) w.comm(c);
    } else if (lp.kind == L_RELU || lp.kind == L_GELU) {
      w.iop(lp.act.io_accumulation.sumcheck); w.ve(lp.act.io_accumulation.evals); w.logup(lp.act.lookup);
      w.u(lp.act.commits.size()); for (auto& c : lp.act.commits) w.comm(c);
    } else if (lp.kind == L_CONV) {
      const ConvProof& c = lp.conv;
      auto viop = [&](const std::vector<IOPProof>& v) { w.u(v.size()); for (auto& x : v) w.iop(x); };
      auto vve = [&](const std::vector<std::vector<E>>& v) { w.u(v.size()); for (auto& x : v) w.ve(x); };
      w.iop(c.fft_proof); w.iop(c.fft_proof_weights); viop(c.fft_delegation_proof); viop(c.fft_delegation_proof_weights);
      w.iop(c.ifft_proof); viop(c.ifft_delegation_proof); w.iop(c.hadamard_proof);
      w.ve(c.fft_claims); w.ve(c.fft_weight_claims); w.ve(c.ifft_claims);
      vve(c.fft_delegation_claims); vve(c.fft_delegation_weights_claims); vve(c.ifft_delegation_claims);
      w.ve(c.partial_evals); w.ve(c.hadamard_clams); w.e(c.bias_claim);
      w.iop(c.clearing_proof.sumcheck); w.ve(c.clearing_proof.individual_claim);
    } else if (lp.kind == L_LAYERNORM) {
      const LayerNormProof& q = lp.ln;
      w.u(q.logup_proofs.size()); for (auto& x : q.logup_proofs) w.logup(x);
      w.u(q.commitments.size()); for (auto& c : q.commitments) w.comm(c);
      w.iop(q.accumulation_proof); w.iop(q.io_proof); w.iop(q.input_proof); w.ve(q.acc_evals); w.ve(q.evaluations); w.e(q.gamma_eval); w.e(q.beta_eval);
    } else if (lp.kind == L_SOFTMAX) {
      const SoftmaxProof& q = lp.sm;
      w.u(q.logup_proofs.size()); for (auto& x : q.logup_proofs) w.logup(x);
      w.u(q.commitments.size()); for (auto& c : q.commitments) w.comm(c);
      w.iop(q.accumulation_proof); w.iop(q.mask_proof); w.ve(q.evaluations);
    } else if (lp.kind == L_MHA) {  // MhaProof {final_mul_proof, softmax_proof, qk_proof}
      w.iop(lp.mha_final.sumcheck); w.ve(lp.mha_final.individual_claims);
      const SoftmaxProof& q = lp.sm;
      w.u(q.logup_proofs.size()); for (auto& x : q.logup_proofs) w.logup(x);
      w.u(q.commitments.size()); for (auto& c : q.commitments) w.comm(c);
      w.iop(q.accumulation_proof); w.iop(q.mask_proof); w.ve(q.evaluations);
      w.iop(lp.mha_qk.sumcheck); w.ve(lp.mha_qk.individual_claims);
    } else if (lp.kind == L_MAXPOOL) {
      w.iop(lp.pool.sumcheck); w.logup(lp.pool.lookup); w.ve(lp.pool.zerocheck_evals); w.u(lp.pool.variable_gap);
      w.u(lp.pool.commitments.size()); for (auto& c : lp.pool.commitments) w.comm(c);
    }
  }
  w.u(p.table_proofs.size()); for (auto& tp : p.table_proofs) { w.comm(tp.multiplicity_commit); w.logup(tp.lookup); }
  w.basefold(p.batch_proof);
  w.u(p.trivial_proofs.size()); for (auto& tp : p.trivial_proofs) w.basefold(tp);
  return w.w;
}

}  // namespace orc
