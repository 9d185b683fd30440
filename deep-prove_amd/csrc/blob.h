// The model blob of dp_model_setup (include/deep_prove_hip.h "model blob") -> ModelSpec. Plain host C++ (no HIP): the C ABI (capi.cpp) and the
// CPU test harness (tests/support/hostlogic_check.cpp, mode `blob`) read blobs through this one parser.
#pragma once
#include "zkml.h"

namespace dp {

inline ModelSpec parse_model(const int64_t* b, size_t n) {
  size_t pos = 0;
  auto rd = [&]() { DP_REQUIRE(pos < n, DP_ERR_ARG, "model blob truncated"); return b[pos++]; };
  ModelSpec m; m.input_len = (size_t)rd();
  // graph form: a NEGATIVE node count, then [#input tensors, their lengths] [#outputs, (node, slot) each] and, in front of every node's
  // parameters, [#inputs, (node, slot) each] with node = -1 for an input tensor of the model
  const int64_t nl_raw = rd(); const bool graph = nl_raw < 0; const size_t nl = (size_t)(graph ? -nl_raw : nl_raw);
  DP_REQUIRE(nl > 0 && nl < 4096, DP_ERR_ARG, "model blob: bad layer count");
  auto rd_edge = [&]() { Edge e; const int64_t f = rd(), sl = rd(); DP_REQUIRE(f >= -1 && f < (int64_t)nl && sl >= 0 && sl < 4096, DP_ERR_ARG, "model blob: edge"); e.from = (int)f; e.slot = (int)sl; return e; };
  if (graph) {
    const size_t ni = (size_t)rd(); DP_REQUIRE(ni > 0 && ni < 4096, DP_ERR_ARG, "model blob: input tensor count");
    for (size_t i = 0; i < ni; i++) { const int64_t len = rd(); DP_REQUIRE(len >= 1 && len <= (int64_t(1) << 40), DP_ERR_ARG, "model blob: input tensor length"); m.input_lens.push_back((size_t)len); }
    const size_t no = (size_t)rd(); DP_REQUIRE(no > 0 && no < 4096, DP_ERR_ARG, "model blob: output tensor count");
    for (size_t i = 0; i < no; i++) m.outputs.push_back(rd_edge());
  }
  for (size_t i = 0; i < nl; i++) {
    LayerSpec l; l.kind = (int)rd();
    if (graph) { const size_t k = (size_t)rd(); DP_REQUIRE(k >= 1 && k <= 3, DP_ERR_ARG, "model blob: a node has one to three inputs"); for (size_t q = 0; q < k; q++) l.inputs.push_back(rd_edge()); }
    if (l.kind == L_MATMUL2) {  // [10, inner dimension k, output columns n, flags (2 = Config::TransposeB)]
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd(); const size_t fl = (size_t)rd();
      DP_REQUIRE((fl & ~size_t(2)) == 0, DP_ERR_ARG, "model blob: matmul2 flags"); l.mm_transpose = fl != 0;
    } else if (l.kind == L_ADD2) { l.add_left = rd(); l.add_right = rd(); }  // [11, left multiplier, right multiplier]
    else if (l.kind == L_CONCAT_MATMUL) {  // [12, shape of A (3), shape of B (3), (concat, mat_mul, output) axis of A (3), of B (3), 0 | 1 + output permutation (3)]
      for (int d = 0; d < 3; d++) l.cm_a[d] = (size_t)rd();
      for (int d = 0; d < 3; d++) l.cm_b[d] = (size_t)rd();
      for (int d = 0; d < 3; d++) { const int64_t x = rd(); DP_REQUIRE(x >= 0 && x < 3, DP_ERR_ARG, "model blob: concat matmul axes"); l.cm_left[d] = (int)x; }
      for (int d = 0; d < 3; d++) { const int64_t x = rd(); DP_REQUIRE(x >= 0 && x < 3, DP_ERR_ARG, "model blob: concat matmul axes"); l.cm_right[d] = (int)x; }
      for (int d = 0; d < 3; d++) DP_REQUIRE(l.cm_a[d] && l.cm_b[d] && l.cm_a[d] <= (size_t(1) << 24) && l.cm_b[d] <= (size_t(1) << 24), DP_ERR_ARG, "model blob: concat matmul shapes");
      const int64_t hp = rd(); DP_REQUIRE(hp == 0 || hp == 1, DP_ERR_ARG, "model blob: concat matmul permutation flag");
      if (hp) for (int d = 0; d < 3; d++) { const int64_t x = rd(); DP_REQUIRE(x >= 0 && x < 3, DP_ERR_ARG, "model blob: concat matmul permutation"); l.cm_perm.push_back((int)x); }
    } else if (l.kind == L_QKV) {  // [13, k, n, W_q | W_k | W_v ([k][n] each), b_q | b_k | b_v ([n] each)]
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      size_t nw = 0, nw3 = 0, nb3 = 0, tot = 0;
      DP_REQUIRE(l.nrows && l.ncols && !__builtin_mul_overflow(l.nrows, l.ncols, &nw) && !__builtin_mul_overflow(nw, (size_t)3, &nw3) && !__builtin_mul_overflow(l.ncols, (size_t)3, &nb3) && !__builtin_add_overflow(nw3, nb3, &tot) && tot <= n - pos, DP_ERR_ARG, "model blob: qkv tensor sizes");
      l.weights.assign(b + pos, b + pos + nw3); pos += nw3;
      l.bias.assign(b + pos, b + pos + nb3); pos += nb3;
    } else
    if (l.kind == L_DENSE) {
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      size_t nw = 0, tot = 0;  // overflow-checked: a wrapped product would also satisfy validate_model's size equality
      DP_REQUIRE(l.nrows && l.ncols && !__builtin_mul_overflow(l.nrows, l.ncols, &nw) && !__builtin_add_overflow(nw, l.nrows, &tot) && tot <= n - pos, DP_ERR_ARG, "model blob: dense tensor sizes");
      l.weights.assign(b + pos, b + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols;
      l.bias.assign(b + pos, b + pos + l.nrows); pos += l.nrows;
    } else if (l.kind == L_POSITIONAL) {  // [9, left multiplier, right multiplier, positions (padded), embedding size (padded), table row major]
      l.add_left = rd(); l.add_right = rd(); l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      size_t nw = 0;
      DP_REQUIRE(l.nrows && l.ncols && !__builtin_mul_overflow(l.nrows, l.ncols, &nw) && nw <= n - pos, DP_ERR_ARG, "model blob: positional table size");
      l.weights.assign(b + pos, b + pos + nw); pos += nw;
    } else if (l.kind == L_EMBED) {  // [8, vocabulary (padded), embedding size (padded), table row major]
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      size_t nw = 0;
      DP_REQUIRE(l.nrows && l.ncols && !__builtin_mul_overflow(l.nrows, l.ncols, &nw) && nw <= n - pos, DP_ERR_ARG, "model blob: embedding table size");
      l.weights.assign(b + pos, b + pos + nw); pos += nw;
    } else if (l.kind == L_ADD) {  // [7, left multiplier, right multiplier, n, operand[n]]
      l.add_left = rd(); l.add_right = rd(); const size_t cnt = (size_t)rd();
      DP_REQUIRE(cnt && cnt <= n - pos, DP_ERR_ARG, "model blob: add operand size");
      l.weights.assign(b + pos, b + pos + cnt); pos += cnt;
    } else if (l.kind == L_MATMUL) {  // [6, inner dimension k, output columns n, flags, weights ([k][n] row major; [n][k] with TransposeB), bias]
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd(); const size_t fl = (size_t)rd(), hb = fl & 1;  // flags: 1 = bias, 2 = Config::TransposeB
      l.mm_transpose = (fl & 2) != 0;
      size_t nw = 0, tot = 0;
      DP_REQUIRE(l.nrows && l.ncols && fl <= 3 && !__builtin_mul_overflow(l.nrows, l.ncols, &nw) && !__builtin_add_overflow(nw, hb ? l.ncols : 0, &tot) && tot <= n - pos, DP_ERR_ARG, "model blob: matmul tensor sizes");
      l.weights.assign(b + pos, b + pos + nw); pos += nw;
      if (hb) { l.bias.assign(b + pos, b + pos + l.ncols); pos += l.ncols; }
    } else if (l.kind == L_REQUANT) { l.right_shift = (unsigned)rd(); l.fp_scale = (unsigned)rd(); l.fixed_point_multiplier = rd(); l.intermediate_bit_size = (unsigned)rd(); }
    else if (l.kind == L_CONV) {
      l.kw = (size_t)rd(); l.kx = (size_t)rd(); l.real_nw = (size_t)rd(); l.nw = (size_t)rd(); for (int k = 0; k < 3; k++) l.unp_out[k] = (size_t)rd();
      DP_REQUIRE(l.kw && l.kx && l.real_nw && l.kw < (1u << 16) && l.kx < (1u << 16) && l.real_nw < (1u << 12), DP_ERR_ARG, "model blob: conv dimensions");
      DP_REQUIRE(l.nw >= 1 && l.nw <= (size_t(1) << 16), DP_ERR_ARG, "model blob: conv padded size");
      for (int k = 0; k < 3; k++) DP_REQUIRE(l.unp_out[k] >= 1 && l.unp_out[k] <= (size_t(1) << 24), DP_ERR_ARG, "model blob: conv output shape");
      size_t nf = l.kw * l.kx * l.real_nw * l.real_nw;
      DP_REQUIRE(nf <= n - pos && l.kw <= n - pos - nf, DP_ERR_ARG, "model blob: conv tensor sizes");
      l.weights.assign(b + pos, b + pos + nf); pos += nf;
      l.bias.assign(b + pos, b + pos + l.kw); pos += l.kw;
    } else if (l.kind == L_MAXPOOL) { for (int k = 0; k < 3; k++) { l.pin[k] = (size_t)rd(); DP_REQUIRE(l.pin[k] >= 1 && l.pin[k] <= (size_t(1) << 24), DP_ERR_ARG, "model blob: maxpool input shape"); } }
    else if (l.kind == L_LAYERNORM) {  // [14, padded dimension, N, multiplier, epsilon bits (f32), range check bits, log2 of the top chunk scalar, gamma[dim], beta[dim]]
      const size_t dim = (size_t)rd(); l.nrows = dim; l.ln_dim_size = (size_t)rd(); l.ln_multiplier = rd();
      const int64_t eb = rd(), rcb = rd(), tcs = rd();
      DP_REQUIRE(dim && dim <= (n - pos) / 2 && eb >= 0 && eb <= 0xFFFFFFFFll && rcb >= 1 && rcb <= 40 && tcs >= 0 && tcs < 8, DP_ERR_ARG, "model blob: layernorm parameters");
      l.ln_eps_bits = (uint32_t)eb; l.ln_range_check_bits = (unsigned)rcb; l.ln_top_chunk_scalar_log = (unsigned)tcs;
      l.weights.assign(b + pos, b + pos + dim); pos += dim; l.bias.assign(b + pos, b + pos + dim); pos += dim;
    }
    else if (l.kind == L_SOFTMAX) {  // [15, shape[3], multiplier, 1 / temperature bits, input scale bits, table size, bkm, zero chunks, zero table vars, allowable error]
      for (int k = 0; k < 3; k++) l.sm_shape[k] = (size_t)rd();
      l.sm_scalar = rd(); const int64_t tb = rd(), sb = rd(), ts = rd(); l.sm_bkm = rd(); const int64_t zc = rd(), zv = rd(); l.sm_allowable_error = rd();
      DP_REQUIRE(tb >= 0 && tb <= 0xFFFFFFFFll && sb >= 0 && sb <= 0xFFFFFFFFll && ts >= 1 && ts <= 22 && zc >= 0 && zc <= 3 && zv >= 0 && zv <= 22, DP_ERR_ARG, "model blob: softmax parameters");
      l.sm_temp_bits = (uint32_t)tb; l.sm_in_scale_bits = (uint32_t)sb; l.sm_table_size = (unsigned)ts; l.sm_zero_chunks = (unsigned)zc; l.sm_zero_vars = (unsigned)zv;
    }
    else if (l.kind == L_MHA) {  // [16, seq, heads, head_dim, then the parameters of its softmax as in kind 15 after the shape]
      for (int k = 0; k < 3; k++) { const int64_t x = rd(); DP_REQUIRE(x >= 1 && x <= (1 << 12), DP_ERR_ARG, "model blob: mha shape"); l.mha_shape[k] = (size_t)x; }
      l.sm_scalar = rd(); const int64_t tb = rd(), sb = rd(), ts = rd(); l.sm_bkm = rd(); const int64_t zc = rd(), zv = rd(); l.sm_allowable_error = rd();
      DP_REQUIRE(tb >= 0 && tb <= 0xFFFFFFFFll && sb >= 0 && sb <= 0xFFFFFFFFll && ts >= 1 && ts <= 22 && zc >= 0 && zc <= 3 && zv >= 0 && zv <= 22, DP_ERR_ARG, "model blob: mha softmax parameters");
      l.sm_temp_bits = (uint32_t)tb; l.sm_in_scale_bits = (uint32_t)sb; l.sm_table_size = (unsigned)ts; l.sm_zero_chunks = (unsigned)zc; l.sm_zero_vars = (unsigned)zv;
    }
    else if (l.kind == L_GELU) { l.fixed_point_multiplier = rd(); DP_REQUIRE(l.fixed_point_multiplier >= 1 && l.fixed_point_multiplier <= (int64_t(1) << 12), DP_ERR_ARG, "model blob: gelu multiplier"); }  // [17, multiplier = round(2^12 * input scale)]
    else DP_REQUIRE(l.kind == L_RELU || l.kind == L_FLATTEN, DP_ERR_ARG, "model blob: unknown layer kind");
    m.layers.push_back(std::move(l));
  }
  DP_REQUIRE(pos == n, DP_ERR_ARG, "model blob: trailing words");
  return m;
}

}  // namespace dp
