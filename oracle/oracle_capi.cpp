// ORACLE (test infrastructure only): C ABI over the CPU restatement so that tests/ and bench.py's cpu_baseline leg can
// drive it through ctypes. Mirrors the argument conventions of include/deep_prove_hip.h (canonical u64 words).
#include "zkml.hpp"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace orc;
static thread_local std::string g_err;
template <class F> static int guard(F f) { try { f(); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } }
static uint64_t* copy_out(const std::vector<u64>& w) { uint64_t* p = (uint64_t*)malloc(std::max<size_t>(w.size(), 1) * 8); memcpy(p, w.data(), w.size() * 8); return p; }
static std::vector<E> rd_pt(const uint64_t* w, size_t k) { std::vector<E> p(k); for (size_t i = 0; i < k; i++) p[i] = {w[2 * i], w[2 * i + 1]}; return p; }
static Mle rd_mle(const uint64_t* w, size_t n, int is_ext) {
  if (is_ext) { std::vector<E> e(n); for (size_t i = 0; i < n; i++) e[i] = {w[2 * i], w[2 * i + 1]}; return Mle::from_ext(e); }
  return Mle::from_base(std::vector<u64>(w, w + n));
}
static Model parse_model(const int64_t* b, size_t n) {
  size_t pos = 0; auto rd = [&]() { if (pos >= n) throw std::runtime_error("model blob truncated"); return b[pos++]; };
  Model m; m.input_len = (size_t)rd();
  // graph form (include/deep_prove_hip.h): a NEGATIVE layer count, then the input tensors, the output wires, and every node's input wires
  int64_t nl_raw = rd(); const bool graph = nl_raw < 0; size_t nl = (size_t)(graph ? -nl_raw : nl_raw);
  auto rd_wire = [&]() { Wire w; w.node = (int)rd(); w.index = (int)rd(); return w; };
  if (graph) {
    size_t ni = (size_t)rd(); if (ni == 0 || ni > n) throw std::runtime_error("model blob: input count");
    for (size_t i = 0; i < ni; i++) m.input_lens.push_back((size_t)rd());
    size_t no = (size_t)rd(); if (no == 0 || no > n) throw std::runtime_error("model blob: output count");
    for (size_t i = 0; i < no; i++) m.outputs.push_back(rd_wire());
  }
  for (size_t i = 0; i < nl; i++) {
    Layer l; l.kind = (LayerKind)rd();
    if (graph) { size_t nin = (size_t)rd(); if (nin == 0 || nin > 3) throw std::runtime_error("model blob: a node has one to three inputs"); for (size_t q = 0; q < nin; q++) l.inputs.push_back(rd_wire()); }
    if (l.kind == L_MATMUL2) { l.nrows = (size_t)rd(); l.ncols = (size_t)rd(); l.transpose_b = (rd() & 2) != 0; }
    else if (l.kind == L_ADD2) { l.add_left = rd(); l.add_right = rd(); }
    else if (l.kind == L_CONCAT_MATMUL) {
      for (int d = 0; d < 3; d++) l.cm_a[d] = (size_t)rd();
      for (int d = 0; d < 3; d++) l.cm_b[d] = (size_t)rd();
      for (int d = 0; d < 3; d++) l.cm_left[d] = (int)rd();
      for (int d = 0; d < 3; d++) l.cm_right[d] = (int)rd();
      if (rd()) for (int d = 0; d < 3; d++) l.cm_perm.push_back((int)rd());
    }
    else if (l.kind == L_QKV) {
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      if (l.nrows == 0 || l.ncols == 0 || l.nrows > n || l.ncols > n || pos + 3 * l.nrows * l.ncols + 3 * l.ncols > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + 3 * l.nrows * l.ncols); pos += 3 * l.nrows * l.ncols;
      l.bias.assign(b + pos, b + pos + 3 * l.ncols); pos += 3 * l.ncols;
    }
    else if (l.kind == L_DENSE) { l.nrows = (size_t)rd(); l.ncols = (size_t)rd(); l.weights.assign(b + pos, b + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols; l.bias.assign(b + pos, b + pos + l.nrows); pos += l.nrows; }
    else if (l.kind == L_POSITIONAL) {
      l.add_left = rd(); l.add_right = rd(); l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      if (l.nrows == 0 || l.ncols == 0 || l.nrows > n || l.ncols > n || pos + l.nrows * l.ncols > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols;
    }
    else if (l.kind == L_EMBED) {
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd();
      if (l.nrows == 0 || l.ncols == 0 || l.nrows > n || l.ncols > n || pos + l.nrows * l.ncols > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols;
    }
    else if (l.kind == L_ADD) {
      l.add_left = rd(); l.add_right = rd(); size_t cnt = (size_t)rd();
      if (cnt > n || pos + cnt > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + cnt); pos += cnt;
    }
    else if (l.kind == L_MATMUL) {
      l.nrows = (size_t)rd(); l.ncols = (size_t)rd(); size_t fl = (size_t)rd(); const size_t hb = fl & 1; l.transpose_b = (fl & 2) != 0;
      if (l.nrows == 0 || l.ncols == 0 || l.nrows > n || l.ncols > n || pos + l.nrows * l.ncols + (hb ? l.ncols : 0) > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols;
      if (hb) { l.bias.assign(b + pos, b + pos + l.ncols); pos += l.ncols; }
    }
    else if (l.kind == L_REQUANT) { l.right_shift = (unsigned)rd(); l.fp_scale = (unsigned)rd(); l.fixed_point_multiplier = rd(); l.intermediate_bit_size = (unsigned)rd(); }
    else if (l.kind == L_CONV) {
      l.kw = (size_t)rd(); l.kx = (size_t)rd(); l.real_nw = (size_t)rd(); l.nw = (size_t)rd(); for (int k = 0; k < 3; k++) l.unp_out[k] = (size_t)rd();
      size_t nf = l.kw * l.kx * l.real_nw * l.real_nw;
      if (pos + nf + l.kw > n) throw std::runtime_error("model blob truncated");
      l.weights.assign(b + pos, b + pos + nf); pos += nf; l.bias.assign(b + pos, b + pos + l.kw); pos += l.kw;
    } else if (l.kind == L_MHA) {  // [16, seq, heads, head_dim, then the softmax parameters as in kind 15 after the shape]
      for (int k = 0; k < 3; k++) l.mha_shape[k] = (size_t)rd();
      l.sm_scalar = rd(); l.sm_temp_bits = (uint32_t)rd(); l.sm_in_scale_bits = (uint32_t)rd(); l.sm_table_size = (unsigned)rd(); l.sm_bkm = rd();
      l.sm_zero_chunks = (unsigned)rd(); l.sm_zero_vars = (unsigned)rd(); l.sm_allowable_error = rd();
      if (l.sm_table_size < 1 || l.sm_table_size > 22 || l.sm_zero_chunks > 3 || l.sm_zero_vars > 22 || !l.mha_shape[0] || !l.mha_shape[1] || !l.mha_shape[2]) throw std::runtime_error("model blob: mha parameters");
    } else if (l.kind == L_MAXPOOL) { for (int k = 0; k < 3; k++) l.pin[k] = (size_t)rd(); }
    else if (l.kind == L_LAYERNORM) {  // dim, N, multiplier, eps bits, range check bits, log2 of the top chunk scalar, gamma[dim], beta[dim]
      size_t dim = (size_t)rd(); l.ln_dim_size = (size_t)rd(); l.ln_multiplier = rd(); l.ln_eps_bits = (uint32_t)rd(); l.ln_range_check_bits = (unsigned)rd(); l.ln_top_chunk_scalar_log = (unsigned)rd();
      if (dim == 0 || dim > n || pos + 2 * dim > n) throw std::runtime_error("model blob truncated");
      if (l.ln_range_check_bits == 0 || l.ln_range_check_bits > 40 || l.ln_top_chunk_scalar_log >= 8) throw std::runtime_error("layernorm: range check parameters");
      l.weights.assign(b + pos, b + pos + dim); pos += dim; l.bias.assign(b + pos, b + pos + dim); pos += dim;
    }
    else if (l.kind == L_SOFTMAX) {  // shape[3], scalar, 1/temperature bits, input scale bits, table size, bkm, zero chunks, zero table vars, allowable error
      for (int k = 0; k < 3; k++) l.sm_shape[k] = (size_t)rd();
      l.sm_scalar = rd(); l.sm_temp_bits = (uint32_t)rd(); l.sm_in_scale_bits = (uint32_t)rd(); l.sm_table_size = (unsigned)rd(); l.sm_bkm = rd(); l.sm_zero_chunks = (unsigned)rd(); l.sm_zero_vars = (unsigned)rd(); l.sm_allowable_error = rd();
      if (l.sm_table_size == 0 || l.sm_table_size > 22 || l.sm_zero_chunks > 3 || l.sm_zero_vars > 22 || l.sm_allowable_error < 1 || l.sm_allowable_error > (1 << 20) || l.sm_scalar < 1 || l.sm_bkm < (1 << 17)) throw std::runtime_error("softmax: parameters");
    }
    else if (l.kind == L_GELU) { l.gelu_multiplier = rd(); if (l.gelu_multiplier < 1 || l.gelu_multiplier > 4096) throw std::runtime_error("model blob: gelu multiplier"); }
    else if (l.kind == L_RELU || l.kind == L_FLATTEN) {}
    else throw std::runtime_error("model blob: unknown layer kind");
    m.layers.push_back(std::move(l));
  }
  return m;
}
struct orc_transcript { Transcript t; };
struct orc_model { Context ctx; };

extern "C" {
const char* orc_last_error(void) { return g_err.c_str(); }
// which claim a GELU's prover files with the commitment of its scaled input column (zkml.hpp prove_relu): 0 = the reference to the letter (activation.rs:419-430,
// verifiable only when the column is opened by showing it), 1 = the lookup's own claim, the one verify_activation checks (:495-505)
void orc_set_gelu_files_lookup_claim(int on) { g_gelu_files_lookup_claim = on != 0; }
void orc_free(void* p) { free(p); }

// constants + algebra self checks (SURVEY.md Appendix B fingerprints, A.1 generators). 0 = ok, else a failing check id.
int orc_selftest(void) {
  const Poseidon2Consts& C = poseidon2_consts();
  static const u64 e0[8] = {0xdd5743e7f2a5a5d9ULL, 0xcb3a864e58ada44bULL, 0xffa2449ed32f8cdcULL, 0x42025f65d6bd13eeULL, 0x7889175e25506323ULL, 0x34b98bb03d24b737ULL, 0xbdcc535ecc4faa2aULL, 0x5b20ad869fc0d033ULL};
  static const u64 dg[8] = {0xa98811a1fed4e3a5ULL, 0x1cc48b54f377e2a0ULL, 0xe40cd4f6c5609a26ULL, 0x11de79ebca97a4a3ULL, 0x9177c73d8b7e929cULL, 0x2a6fe8085797e791ULL, 0x3de6e93329f8d5adULL, 0x3f7af9125da962feULL};
  for (int i = 0; i < 8; i++) { if (C.ext_init[0][i] != e0[i]) return 1; if (C.diag_m1[i] != dg[i]) return 2; }
  if (C.internal[0] != 0x488897d85ff51f56ULL || C.internal[1] != 0x1140737ccb162218ULL || C.internal[2] != 0xa7eeb9215866ed35ULL) return 3;
  if (C.ext_term[0][0] != 0x014ef1197d341346ULL || C.ext_term[0][1] != 0x9725e20825d07394ULL || C.ext_term[3][7] != 0x95f2394459fbc25eULL) return 4;
  if (fpow(GENERATOR, (P - 1) >> 32) != G32) return 5;
  for (u64 q : {2ULL, 3ULL, 5ULL, 17ULL, 257ULL, 65537ULL}) if (fpow(GENERATOR, (P - 1) / q) == 1) return 6;  // 7 generates F_p^*
  if (fpow(W, (P - 1) / 2) == 1) return 7;                                                                          // 7 is a non-residue
  u64 x = 0x243F6A8885A308D3ULL;
  for (int i = 0; i < 20000; i++) {
    x = x * 6364136223846793005ULL + 1442695040888963407ULL; u64 a = x % P; x = x * 6364136223846793005ULL + 1442695040888963407ULL; u64 b = x % P;
    if (fmul(a, b) != (u64)(((u128)a * b) % P)) return 8;
    if (fadd(a, b) != (u64)(((u128)a + b) % P)) return 9;
    if (fsub(a, b) != (u64)(((u128)a + P - b) % P)) return 10;
    if (a && fmul(a, finv(a)) != 1) return 11;
    E e{a, b}; if (!e_is_zero(e) && emul(e, einv(e)) != e_one()) return 12;
  }
  return 0;
}
// ports of the reference's own RS-code property tests (mpcs/src/basefold/encoding/rs.rs:559-624 FFT vs naive Horner,
// encoding.rs:174-238 fold(encode(msg)) == encode(fold(msg)), hypercube interpolation round trip). 0 = ok.
int orc_rs_selftest(uint64_t seed) {
  u64 s = seed; auto rnd = [&]() { s += 0x9E3779B97F4A7C15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return from_u64(z ^ (z >> 31)); };
  PcsParams pp = pcs_setup(1 << 12);
  std::vector<u64> c(1 << 8); for (auto& x : c) x = rnd();
  Mle cw = rs_encode(pp, Mle::from_base(c));
  u64 shift = GENERATOR; for (int i = 0; i < 12 - 8; i++) shift = fmul(shift, shift);
  u64 w = two_adic_generator(9);
  for (size_t k = 0; k < 512; k += 7) { u64 x = fmul(shift, fpow(w, k)), acc = 0; for (size_t i = c.size(); i-- > 0;) acc = fadd(fmul(acc, x), c[i]); if (acc != cw.b[k]) return 1; }
  std::vector<E> m(1 << 8); for (auto& x : m) x = {rnd(), rnd()};
  Mle cwe = rs_encode(pp, Mle::from_ext(m)); reverse_bits_mle(cwe);
  E ch = {rnd(), rnd()};
  std::vector<E> folded = basefold_fold(pp, log2_strict(cwe.len()) - 1, cwe.e, ch);
  std::vector<E> m2(1 << 7); for (size_t i = 0; i < m2.size(); i++) m2[i] = eadd(m[2 * i], emul(ch, m[2 * i + 1]));
  Mle cw2 = rs_encode(pp, Mle::from_ext(m2)); reverse_bits_mle(cw2);
  if (cw2.e != folded) return 2;
  // multilinear coefficients reproduce the evaluations: f(x) = sum_S c_S prod_{i in S} x_i
  std::vector<u64> ev(1 << 6); for (auto& x : ev) x = rnd();
  Mle co = Mle::from_base(ev); interpolate_over_boolean_hypercube(co);
  for (size_t x = 0; x < ev.size(); x++) { u64 acc = 0; for (size_t S = 0; S < ev.size(); S++) if ((S & x) == S) acc = fadd(acc, co.b[S]); if (acc != ev[x]) return 3; }
  return 0;
}
void orc_rc_table(uint64_t out[94]) { const Poseidon2Consts& C = poseidon2_consts(); memcpy(out, &C.ext_init[0][0], 32 * 8); memcpy(out + 32, C.internal, 22 * 8); memcpy(out + 54, &C.ext_term[0][0], 32 * 8); memcpy(out + 86, C.diag_m1, 64); }
void orc_poseidon2_permute(uint64_t s[8]) { poseidon2_permute(s); }
void orc_compress(const uint64_t x[4], const uint64_t y[4], uint64_t out[4]) { Digest a, b; for (int i = 0; i < 4; i++) { a[i] = x[i]; b[i] = y[i]; } Digest d = compress(a, b); for (int i = 0; i < 4; i++) out[i] = d[i]; }

orc_transcript* orc_transcript_new(const char* label) { orc_transcript* t = new orc_transcript(); if (label) t->t.append_message(label); return t; }
void orc_transcript_free(orc_transcript* t) { delete t; }
int orc_transcript_append_elements(orc_transcript* t, const uint64_t* e, size_t n) { t->t.append_field_elements(e, n); return 0; }
int orc_transcript_append_message(orc_transcript* t, const uint8_t* b, size_t n) { t->t.append_message(b, n); return 0; }
int orc_transcript_challenge(orc_transcript* t, const char* label, uint64_t out[2]) { E c = label ? t->t.get_and_append_challenge(label) : t->t.read_challenge(); out[0] = c.c0; out[1] = c.c1; return 0; }

int orc_eq_table(const uint64_t* point, uint32_t k, uint64_t* out) { return guard([&] { auto v = build_eq_x_r_vec(rd_pt(point, k)); auto v2 = compute_betas_eval(rd_pt(point, k)); if (v != v2) throw std::runtime_error("eq tables disagree"); for (size_t i = 0; i < v.size(); i++) { out[2 * i] = v[i].c0; out[2 * i + 1] = v[i].c1; } }); }
int orc_mle_eval(const uint64_t* words, size_t n, int is_ext, const uint64_t* point, uint32_t k, uint64_t out[2]) { return guard([&] { E r = rd_mle(words, n, is_ext).evaluate(rd_pt(point, k)); out[0] = r.c0; out[1] = r.c1; }); }
// fix_high_variables of a rows x cols table (base words) at a log2(rows) point -> cols ext values
int orc_fix_high(const uint64_t* words, size_t rows, size_t cols, const uint64_t* point, uint64_t* out) {
  return guard([&] { Mle m = Mle::from_base(std::vector<u64>(words, words + rows * cols)); m.fix_high_in_place(rd_pt(point, log2_strict(rows))); for (size_t i = 0; i < cols; i++) { E v = m.at(i); out[2 * i] = v.c0; out[2 * i + 1] = v.c1; } });
}
// table_nv[i]: number of variables of table i (<= nv; NULL: every table has nv); term_tables: the terms' table lists back to back
int orc_sumcheck_prove(uint32_t nv, const uint64_t* const* tables, const int32_t* is_ext, const uint32_t* table_nv, int32_t ntables, const int32_t* term_degree, const int32_t* term_tables,
                       const uint64_t* term_coeffs, int32_t nterms, orc_transcript* t, uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals) {
  return guard([&] {
    std::vector<MleP> ms; for (int i = 0; i < ntables; i++) ms.push_back(mk(rd_mle(tables[i], size_t(1) << (table_nv ? table_nv[i] : nv), is_ext[i])));
    VirtualPolynomial vp(nv);
    for (int i = 0; i < ntables; i++) vp.flattened.push_back(ms[i]);  // keep the caller's table order for `finals`
    size_t off = 0;
    for (int i = 0; i < nterms; i++) { std::vector<MleP> l; for (int j = 0; j < term_degree[i]; j++) l.push_back(ms[term_tables[off + j]]); off += term_degree[i]; vp.add_mle_list(l, E{term_coeffs[2 * i], term_coeffs[2 * i + 1]}); }
    auto res = sumcheck_prove(std::move(vp), t->t);
    Writer w; w.iop(res.first);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
    if (finals) { auto f = res.second.final_evaluations(); for (int i = 0; i < ntables; i++) { finals[2 * i] = f[i].c0; finals[2 * i + 1] = f[i].c1; } }
  });
}
int orc_logup_prove(const uint64_t* const* columns, int32_t ncols, size_t n, int32_t cpi, const uint64_t* mult, const uint64_t cc[2], const uint64_t csc[2],
                    orc_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    LogUpInput in; in.is_table = mult != nullptr; in.columns_per_instance = cpi;
    for (int i = 0; i < ncols; i++) in.column_evals.push_back(std::vector<u64>(columns[i], columns[i] + n));
    if (mult) in.multiplicities.assign(mult, mult + n);
    in.constant_challenge = {cc[0], cc[1]}; in.column_separation_challenge = {csc[0], csc[1]};
    LogUpProof p = logup_batch_prove(in, t->t);
    Writer w; w.logup(p); *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
int orc_pcs_commit_root(size_t max_poly_size, const uint64_t* words, size_t n, int is_ext, uint64_t root[4]) {
  return guard([&] { PcsParams pp = pcs_setup(max_poly_size); CommitmentWithWitness c = pcs_commit(pp, rd_mle(words, n, is_ext)); for (int i = 0; i < 4; i++) root[i] = c.codeword_tree.root()[i]; });
}
int orc_pcs_open(size_t max_poly_size, const uint64_t* words, size_t n, int is_ext, const uint64_t* point, orc_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    PcsParams pp = pcs_setup(max_poly_size);
    Mle poly = rd_mle(words, n, is_ext);
    CommitmentWithWitness c = pcs_commit(pp, poly);
    Transcript scratch = default_transcript();
    BasefoldProof p = pcs_open(pp, poly, c, rd_pt(point, poly.nv), t ? t->t : scratch);
    Writer w; w.basefold(p); *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
int orc_pcs_batch_open(size_t max_poly_size, const uint64_t* const* polys, const size_t* lens, const int32_t* is_ext, int32_t n, const uint64_t* points_flat, const uint64_t* evals,
                       orc_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    PcsParams pp = pcs_setup(max_poly_size);
    std::vector<Mle> ps; std::vector<CommitmentWithWitness> cs;
    for (int i = 0; i < n; i++) { ps.push_back(rd_mle(polys[i], lens[i], is_ext[i])); cs.push_back(pcs_commit(pp, ps.back())); }
    std::vector<const Mle*> pp_; std::vector<const CommitmentWithWitness*> cc; std::vector<std::vector<E>> pts; std::vector<Evaluation> evs; size_t off = 0;
    for (int i = 0; i < n; i++) { pp_.push_back(&ps[i]); cc.push_back(&cs[i]); pts.push_back(rd_pt(points_flat + off, ps[i].nv)); off += 2 * ps[i].nv; evs.push_back({(size_t)i, (size_t)i, E{evals[2 * i], evals[2 * i + 1]}}); }
    BasefoldProof p = pcs_batch_open(pp, pp_, cc, pts, evs, t->t);
    Writer w; w.basefold(p); *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
// batch_open over a general Evaluation list (basefold.rs:546-770): polynomial i committed on its own; evals = (poly, point, value)
int orc_pcs_batch_open_evals(size_t max_poly_size, const uint64_t* const* polys, const size_t* lens, const int32_t* is_ext, int32_t n_polys, const uint64_t* points_flat, const uint32_t* point_num_vars,
                             int32_t n_points, const uint32_t* eval_poly, const uint32_t* eval_point, const uint64_t* eval_values, int32_t n_evals, orc_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    PcsParams pp = pcs_setup(max_poly_size);
    std::vector<Mle> ps; std::vector<CommitmentWithWitness> cs;
    for (int i = 0; i < n_polys; i++) { ps.push_back(rd_mle(polys[i], lens[i], is_ext[i])); cs.push_back(pcs_commit(pp, ps.back())); }
    std::vector<const Mle*> pp_; std::vector<const CommitmentWithWitness*> cc;
    for (int i = 0; i < n_polys; i++) { pp_.push_back(&ps[i]); cc.push_back(&cs[i]); }
    std::vector<std::vector<E>> pts; size_t off = 0;
    for (int i = 0; i < n_points; i++) { pts.push_back(rd_pt(points_flat + off, point_num_vars[i])); off += 2 * (size_t)point_num_vars[i]; }
    std::vector<Evaluation> evs;
    for (int i = 0; i < n_evals; i++) evs.push_back({(size_t)eval_poly[i], (size_t)eval_point[i], E{eval_values[2 * i], eval_values[2 * i + 1]}});
    BasefoldProof p = pcs_batch_open(pp, pp_, cc, pts, evs, t->t);
    Writer w; w.basefold(p); *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
// batch_commit + simple_batch_open of k polynomials of `n` elements each (basefold.rs:356-446, 777-861): root of the common tree and the proof stream
int orc_pcs_simple_batch_open(size_t max_poly_size, const uint64_t* const* polys, size_t n, int32_t k, int is_ext, const uint64_t* point, orc_transcript* t, uint64_t root[4],
                              uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    PcsParams pp = pcs_setup(max_poly_size);
    std::vector<Mle> ps;
    for (int i = 0; i < k; i++) ps.push_back(rd_mle(polys[i], n, is_ext));
    BatchCommitmentWithWitness c = pcs_batch_commit(pp, ps);
    for (int i = 0; i < 4; i++) root[i] = c.root()[i];
    if (!proof_words) return;
    Transcript scratch = default_transcript();
    BasefoldProof p = pcs_simple_batch_open(pp, c, rd_pt(point, c.num_vars), t ? t->t : scratch);
    Writer w; w.basefold(p); *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
int orc_model_setup(const int64_t* blob, size_t nwords, orc_model** out) { return guard([&] { Model m = parse_model(blob, nwords); orc_model* om = new orc_model{context_generate(m)}; *out = om; }); }
void orc_model_free(orc_model* m) { delete m; }
// the sponge traffic of every transcript created on this thread between begin and take: pairs (0, absorbed) / (1, squeezed)
static thread_local std::vector<u64> g_trace_store;
int orc_trace_begin() { return guard([&] { g_trace_store.clear(); g_transcript_trace = &g_trace_store; }); }
int orc_trace_take(uint64_t** words, size_t* nwords) { return guard([&] { g_transcript_trace = nullptr; *words = copy_out(g_trace_store); *nwords = g_trace_store.size(); g_trace_store.clear(); }); }
int orc_hash_or_noop(const uint64_t* in, size_t n, uint64_t out[4]) { return guard([&] { Digest d = hash_or_noop(in, n); for (int i = 0; i < 4; i++) out[i] = d[i]; }); }
int orc_model_prove(orc_model* m, const int64_t* input, size_t ninput, uint64_t** proof_words, size_t* proof_nwords, int64_t* output, size_t* noutput, double* prove_ms) {
  return guard([&] {
    Transcript t = default_transcript();
    Trace tr = run_model(m->ctx.model, std::vector<int64_t>(input, input + ninput));  // inference is not "proving time" (zkml/src/bin/bench.rs:341-408)
    auto t0 = std::chrono::steady_clock::now();
    Proof p = prove(m->ctx, tr, t);
    auto t1 = std::chrono::steady_clock::now();
    if (prove_ms) *prove_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    std::vector<u64> w = serialize_proof(p);
    *proof_words = copy_out(w); *proof_nwords = w.size();
    if (output && noutput) { const std::vector<int64_t> o = model_output(m->ctx.model, tr); if (*noutput < o.size()) throw std::runtime_error("output buffer too small"); memcpy(output, o.data(), o.size() * 8); *noutput = o.size(); }
  });
}
// CPU throughput baseline: `threads` host threads each prove `per_thread` independent proofs of the same input (the CPU
// analogue of the product's proofs in flight; the reference parallelises inside one proof with rayon, which a scalar
// restatement does not, so replicas are how it fills the cores). Inference is done once, outside the timed region.
// digest = wrapping sum of the words of every proof stream (all replicas must produce the same proof).
int orc_model_prove_many(orc_model* m, const int64_t* input, size_t ninput, int32_t threads, int32_t per_thread, double* wall_ms, uint64_t* digest) {
  return guard([&] {
    if (threads < 1 || per_thread < 1) throw std::runtime_error("threads and per_thread must be positive");
    Trace tr = run_model(m->ctx.model, std::vector<int64_t>(input, input + ninput));
    std::vector<u64> dg(threads, 0); std::vector<std::string> errs(threads);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < threads; i++) th.emplace_back([&, i] {
      try {
        for (int j = 0; j < per_thread; j++) { Transcript t = default_transcript(); Proof p = prove(m->ctx, tr, t); u64 d = 0; for (u64 w : serialize_proof(p)) d += w; dg[i] += d; }
      } catch (const std::exception& e) { errs[i] = e.what(); }
    });
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    u64 d = 0; for (u64 x : dg) d += x; if (digest) *digest = d;
  });
}
// The "port-mt" CPU baseline: ONE proof on `threads` cores — the oracle's O(n) loops (sumcheck round sums and folds, Merkle
// layers, RS butterflies, the batch-opening sums and merges) chunked over a fork-join pool with rayon's with_min_len(64), the
// role rayon plays in the reference (par.hpp). prove() only, as zkml/src/bin/bench.rs:390-408 times it; the digest (wrapping
// word sum of the proof stream) must equal the single-threaded proof's.
int orc_model_prove_mt(orc_model* m, const int64_t* input, size_t ninput, int32_t threads, double* wall_ms, uint64_t* digest) {
  return guard([&] {
    if (threads < 1) throw std::runtime_error("threads must be positive");
    Trace tr = run_model(m->ctx.model, std::vector<int64_t>(input, input + ninput));
    ParScope scope(threads);
    auto t0 = std::chrono::steady_clock::now();
    Transcript t = default_transcript();
    Proof p = prove(m->ctx, tr, t);
    auto t1 = std::chrono::steady_clock::now();
    *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    u64 d = 0; for (u64 w : serialize_proof(p)) d += w;
    if (digest) *digest = d;
  });
}
// CPU baseline for the standalone sumcheck bench (config 5 shape): one product of k base tables of 2^nv SplitMix64-derived
// canonical elements, label "test". Returns wall seconds of prove only.
int orc_bench_sumcheck(uint32_t nv, int32_t k, uint64_t seed, double* seconds, uint64_t digest[2]) {
  return guard([&] {
    u64 s = seed; auto rnd = [&]() { s += 0x9E3779B97F4A7C15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    VirtualPolynomial vp(nv); std::vector<MleP> l;
    for (int j = 0; j < k; j++) { std::vector<u64> v(size_t(1) << nv); for (auto& x : v) x = from_u64(rnd()); l.push_back(mk(Mle::from_base(std::move(v)))); }
    vp.add_mle_list(l, e_one());
    Transcript t("test");
    auto t0 = std::chrono::steady_clock::now();
    auto res = sumcheck_prove(std::move(vp), t);
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    E c = t.read_challenge(); digest[0] = c.c0; digest[1] = c.c1;
  });
}
}
