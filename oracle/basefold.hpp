// ORACLE (test infrastructure only).
// Restates the Basefold PCS prover exactly as zkml instantiates it:
//   Basefold<GoldilocksExt2, BasefoldRSParams<PoseidonHasher>>  (zkml/src/bin/bench.rs:26, zkml/src/testing.rs:8)
//   RS code rate 1/2, 200 queries, base-case message 2^7 (mpcs/src/basefold/encoding/rs.rs:194-215).
// Files followed: mpcs/src/basefold.rs:86-154,304-354,466-770; basefold/commit_phase.rs:187-359,511-526;
// basefold/sumcheck.rs:6-152; basefold/query_phase.rs:67-102,419-472; basefold/encoding/rs.rs:31-68,92-189,
// 275-410,458-501; util/merkle_tree.rs:23-153,261-329; util/hash.rs:25-63; util/arithmetic.rs:120-132;
// util/arithmetic/hypercube.rs:16-37; sum_check/classic.rs:100-285; sum_check/classic/coeff.rs:198-345.
#pragma once
#include "par.hpp"
#include "mle.hpp"
#include "poseidon2.hpp"

namespace orc {

constexpr unsigned RATE_LOG = 1;
constexpr unsigned BASECODE_MSG_SIZE_LOG = 7;
constexpr unsigned NUM_QUERIES = 200;

// ---------------------------------------------------------------- RS code parameters
struct PcsParams {
  unsigned full_message_size_log = 0;
  std::vector<std::vector<u64>> fft_root_table;  // rs.rs:31-68
  std::vector<u64> gamma_powers, gamma_powers_inv_div_two;
};
static inline std::vector<std::vector<u64>> fft_root_table(unsigned lg_n) {
  std::vector<u64> bases;
  u64 base = two_adic_generator(lg_n);
  bases.push_back(base);
  for (unsigned i = 1; i < lg_n; i++) { base = fmul(base, base); bases.push_back(base); }
  std::vector<std::vector<u64>> table;
  for (unsigned lg_m = 1; lg_m <= lg_n; lg_m++) {
    size_t half_m = size_t(1) << (lg_m - 1);
    u64 b = bases[lg_n - lg_m];
    size_t cnt = std::max<size_t>(half_m, 2);
    std::vector<u64> row(cnt);
    row[0] = 1;
    for (size_t i = 1; i < cnt; i++) row[i] = fmul(row[i - 1], b);
    table.push_back(std::move(row));
  }
  return table;
}
// PCS::setup + trim (basefold.rs:278-303, rs.rs:275-348); poly_size must be a power of two
static inline PcsParams pcs_setup(size_t poly_size) {
  PcsParams pp;
  unsigned L = log2_strict(poly_size);
  pp.full_message_size_log = L;
  if (L < BASECODE_MSG_SIZE_LOG) return pp;
  pp.fft_root_table = fft_root_table(L + RATE_LOG);
  pp.gamma_powers.push_back(GENERATOR);
  std::vector<u64> ginv;
  ginv.push_back(finv(GENERATOR));
  for (unsigned i = 1; i < L + RATE_LOG; i++) {
    pp.gamma_powers.push_back(fmul(pp.gamma_powers[i - 1], pp.gamma_powers[i - 1]));
    ginv.push_back(fmul(ginv[i - 1], ginv[i - 1]));
  }
  u64 inv2 = finv(2);
  for (auto& x : ginv) x = fmul(x, inv2);
  pp.gamma_powers_inv_div_two = ginv;
  return pp;
}

// ---------------------------------------------------------------- generic helpers over base/ext vectors
static inline void interpolate_over_boolean_hypercube(Mle& m) {  // hypercube.rs:16-37
  size_t n = m.len();
  unsigned lg = log2_strict(n);
  for (unsigned i = 1; i <= lg; i++) {
    size_t chunk = size_t(1) << i, half = chunk >> 1;
    par_for(n / 2, [&](size_t lo, size_t hi) {  // butterfly q: chunk q / half, offset q % half
      for (size_t q = lo; q < hi; q++) {
        size_t c = (q / half) * chunk, j = half + (q % half);
        if (m.is_ext) m.e[c + j] = esub(m.e[c + j], m.e[c + j - half]);
        else m.b[c + j] = fsub(m.b[c + j], m.b[c + j - half]);
      }
    });
  }
}
static inline void reverse_bits_mle(Mle& m) {
  if (m.is_ext) reverse_index_bits_in_place(m.e); else reverse_index_bits_in_place(m.b);
}
// fft (rs.rs:129-173) with zero_factor r, then fft_classic_inner (rs.rs:92-121); twiddles are base field
static inline void rs_fft(Mle& v, unsigned r, const std::vector<std::vector<u64>>& root_table) {
  reverse_bits_mle(v);
  size_t n = v.len();
  unsigned lg_n = log2_strict(n);
  if (root_table.size() != lg_n) throw std::runtime_error("fft: root table length mismatch");
  if (r > 0) {
    size_t mask = ~((size_t(1) << r) - 1);
    for (size_t i = 0; i < n; i++) { if (v.is_ext) v.e[i] = v.e[i & mask]; else v.b[i] = v.b[i & mask]; }
  }
  for (unsigned lg_half_m = r; lg_half_m < lg_n; lg_half_m++) {
    size_t half_m = size_t(1) << lg_half_m, m = half_m * 2;
    const std::vector<u64>& om = root_table[lg_half_m];
    par_for(n / 2, [&](size_t lo, size_t hi) {  // butterfly q: block q / half_m, offset q % half_m
      for (size_t q = lo; q < hi; q++) {
        size_t k = (q / half_m) * m, j = q % half_m;
        if (v.is_ext) {
          E t = emul_base(v.e[k + half_m + j], om[j]); E u = v.e[k + j];
          v.e[k + j] = eadd(u, t); v.e[k + half_m + j] = esub(u, t);
        } else {
          u64 t = fmul(v.b[k + half_m + j], om[j]); u64 u = v.b[k + j];
          v.b[k + j] = fadd(u, t); v.b[k + half_m + j] = fsub(u, t);
        }
      }
    });
  }
}
// RSCode::encode -> encode_internal -> coset_fft (rs.rs:350-354,458-501,175-189)
static inline Mle rs_encode(const PcsParams& pp, const Mle& coeffs) {
  unsigned lg_m = log2_strict(coeffs.len());
  assert(lg_m >= BASECODE_MSG_SIZE_LOG && lg_m <= pp.full_message_size_log);
  Mle ret = coeffs;
  size_t n = coeffs.len();
  if (ret.is_ext) ret.e.resize(2 * n, e_zero()); else ret.b.resize(2 * n, 0);
  ret.nv = lg_m + 1;
  u64 shift = GENERATOR;
  for (unsigned i = 0; i < pp.full_message_size_log - lg_m; i++) shift = fmul(shift, shift);  // exp_power_of_2
  u64 sp = 1;
  for (size_t i = 0; i < 2 * n; i++) {
    if (ret.is_ext) ret.e[i] = emul_base(ret.e[i], sp); else ret.b[i] = fmul(ret.b[i], sp);
    sp = fmul(sp, shift);
  }
  std::vector<std::vector<u64>> table(pp.fft_root_table.begin(), pp.fft_root_table.begin() + lg_m + RATE_LOG);
  rs_fft(ret, RATE_LOG, table);
  return ret;
}
// prover_folding_coeffs (rs.rs:377-410)
static inline void prover_folding_coeffs(const PcsParams& pp, unsigned level, size_t index, u64& x0, u64& x1, u64& w) {
  index = reverse_bits(index, level);
  size_t half = size_t(1) << level;
  u64 root = index < half ? pp.fft_root_table[level][index] : fneg(pp.fft_root_table[level][index - half]);
  unsigned gi = pp.full_message_size_log + RATE_LOG - level - 1;
  x0 = fmul(root, pp.gamma_powers[gi]);
  x1 = fneg(x0);
  u64 f;
  if (index == 0) f = 1;
  else if (index < half) f = fneg(pp.fft_root_table[level][half - index]);
  else if (index == half) f = fneg(1);
  else f = pp.fft_root_table[level][(half << 1) - index];
  w = fmul(fneg(pp.gamma_powers_inv_div_two[gi]), f);
}
// interpolate2_weights (arithmetic.rs:120-132)
static inline E interpolate2_weights(E a0, E a1, E b0, E b1, E w, E x) {
  (void)b0;
  return eadd(a1, emul(emul(esub(x, a0), esub(b1, a1)), w));
}
// basefold_one_round_by_interpolation_weights (commit_phase.rs:511-526)
static inline std::vector<E> basefold_fold(const PcsParams& pp, unsigned level, const std::vector<E>& vals, E ch) {
  std::vector<E> out(vals.size() / 2);
  par_for(out.size(), [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      u64 x0, x1, w;
      prover_folding_coeffs(pp, level, i, x0, x1, w);
      out[i] = interpolate2_weights(e_from(x0), vals[2 * i], e_from(x1), vals[2 * i + 1], e_from(w), ch);
    }
  });
  return out;
}

// ---------------------------------------------------------------- Merkle tree (merkle_tree.rs, hash.rs)
struct MerkleTree {
  std::vector<std::vector<Digest>> inner;
  Mle leaves;
  const Digest& root() const { return inner.back()[0]; }
  size_t height() const { return inner.size(); }
  // merkle_path_without_leaf_sibling_or_root (merkle_tree.rs:139-152)
  std::vector<Digest> path(size_t leaf_index) const {
    std::vector<Digest> p;
    for (size_t l = 0; l + 1 < inner.size(); l++) p.push_back(inner[l][(leaf_index >> (l + 1)) ^ 1]);
    return p;
  }
};
static inline std::vector<std::vector<Digest>> merkelize(const Mle& v) {  // merkle_tree.rs:261-329
  unsigned log_v = log2_strict(v.len());
  std::vector<std::vector<Digest>> tree;
  std::vector<Digest> h(v.len() >> 1);
  par_for(h.size(), [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      if (v.is_ext) { u64 in[4] = {v.e[2 * i].c0, v.e[2 * i].c1, v.e[2 * i + 1].c0, v.e[2 * i + 1].c1}; h[i] = hash_or_noop(in, 4); }
      else { u64 in[2] = {v.b[2 * i], v.b[2 * i + 1]}; h[i] = hash_or_noop(in, 2); }
    }
  });
  tree.push_back(std::move(h));
  for (unsigned i = 1; i < log_v; i++) {
    const auto& prev = tree[i - 1];
    std::vector<Digest> nx(prev.size() / 2);
    par_for(nx.size(), [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) nx[j] = compress(prev[2 * j], prev[2 * j + 1]); });
    tree.push_back(std::move(nx));
  }
  return tree;
}
static inline MerkleTree merkle_from_leaves(Mle leaves) {
  MerkleTree t;
  t.inner = merkelize(leaves);
  t.leaves = std::move(leaves);
  return t;
}

// ---------------------------------------------------------------- commitments (structure.rs:63-166)
struct Commitment {
  Digest root;
  unsigned num_vars;
  bool is_base;
};
struct CommitmentWithWitness {
  MerkleTree codeword_tree;  // leaves = bit-reversed codeword (or the raw evaluations when trivial)
  Mle bh_evals;              // bit-reversed evaluations (raw when trivial)
  unsigned num_vars = 0;
  bool is_base = true;
  bool is_trivial() const { return num_vars <= BASECODE_MSG_SIZE_LOG; }
  size_t codeword_size() const { return codeword_tree.leaves.len(); }
  unsigned codeword_size_log() const { return codeword_tree.height(); }
  Commitment pure() const { return {codeword_tree.root(), num_vars, is_base}; }
};
// Basefold::commit (basefold.rs:304-354) + get_poly_bh_evals_and_codeword (basefold.rs:86-154)
static inline CommitmentWithWitness pcs_commit(const PcsParams& pp, const Mle& poly) {
  CommitmentWithWitness c;
  c.num_vars = poly.nv;
  c.is_base = !poly.is_ext;
  if (poly.nv > pp.full_message_size_log) throw std::runtime_error("PolynomialTooLarge");
  if (poly.nv <= BASECODE_MSG_SIZE_LOG) {
    c.codeword_tree = merkle_from_leaves(poly);
    c.bh_evals = poly;
    return c;
  }
  Mle coeffs = poly;
  interpolate_over_boolean_hypercube(coeffs);
  reverse_bits_mle(coeffs);  // message_is_even_and_odd_folding() == true for RS
  Mle codeword = rs_encode(pp, coeffs);
  Mle bh = poly;
  reverse_bits_mle(bh);
  reverse_bits_mle(codeword);
  c.codeword_tree = merkle_from_leaves(std::move(codeword));
  c.bh_evals = std::move(bh);
  return c;
}

// ---------------------------------------------------------------- proof objects (structure.rs:334-363, query_phase.rs)
struct CodewordQuery {
  bool is_ext;
  E left, right;  // base values stored in c0 when !is_ext
  size_t index;   // p0
  std::vector<Digest> path;
};
struct BatchedQuery {
  size_t index;
  std::vector<CodewordQuery> oracle_query;
  std::vector<CodewordQuery> commitments_query;
};
struct BasefoldProof {
  bool trivial = false;
  std::vector<Mle> trivial_proof;                 // BasefoldProof::trivial(vec![poly.evaluations])
  std::vector<std::vector<E>> sumcheck_messages;  // commit phase: rounds x 3 coefficients
  std::vector<Digest> roots;
  std::vector<E> final_message;
  std::vector<BatchedQuery> queries;
  std::vector<std::vector<E>> sumcheck_proof;     // classic sumcheck: num_vars x 3 coefficients
};

// Basefold::open -- only the trivial branch is reachable from zkml (basefold.rs:466-483; commit/context.rs:295)
static inline BasefoldProof pcs_open_trivial(const Mle& poly, const CommitmentWithWitness& comm) {
  if (!comm.is_trivial()) throw std::runtime_error("pcs_open: only the trivial path is used by zkml");
  BasefoldProof p;
  p.trivial = true;
  p.trivial_proof.push_back(poly);
  return p;
}

// basefold/sumcheck.rs helpers on (eq, bh) both ext
static inline void one_level_interp_hc(std::vector<E>& v) {
  if (v.size() == 1) return;
  for (size_t i = 0; i + 1 < v.size(); i += 2) v[i + 1] = esub(v[i + 1], v[i]);
}
static inline void one_level_eval_hc(std::vector<E>& v, E ch) {
  std::vector<E> out(v.size() / 2);
  par_for(out.size(), [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) out[i] = eadd(v[2 * i], emul(ch, v[2 * i + 1])); });
  v = std::move(out);
}
static inline std::vector<E> parallel_pi(const std::vector<E>& evals, const std::vector<E>& eq) {
  if (evals.size() == 1) return {evals[0], evals[0], evals[0]};
  struct C3 { E c1, c2, c3; };
  C3 r = par_reduce<C3>(evals.size() / 2, C3{e_zero(), e_zero(), e_zero()}, [&](size_t lo, size_t hi) {
    C3 a{e_zero(), e_zero(), e_zero()};
    for (size_t q = lo; q < hi; q++) {
      size_t i = 2 * q;
      a.c1 = eadd(a.c1, emul(evals[i], eq[i]));
      a.c2 = eadd(a.c2, eadd(emul(evals[i + 1], eq[i]), emul(evals[i], eq[i + 1])));
      a.c3 = eadd(a.c3, emul(evals[i + 1], eq[i + 1]));
    }
    return a;
  }, [](C3 a, C3 b) { return C3{eadd(a.c1, b.c1), eadd(a.c2, b.c2), eadd(a.c3, b.c3)}; });
  return {r.c1, r.c2, r.c3};
}

// Basefold::open of ONE committed polynomial (basefold.rs:466-544; zkml itself only reaches the trivial branch, commit/context.rs:295):
//   commit_phase (commit_phase.rs:30-185): running_oracle = the committed codeword read as extension elements, running_evals =
//   the bit-reversed hypercube evaluations kept by commit(), eq = bit-reversed build_eq_x_r_vec(point); first message by
//   sum_check_first_round; per round: absorb the message, draw "commit round", fold the oracle, (i > 0) keep the tree of the
//   previous oracle; not the last round: sum_check_challenge_round -> next message, Merkle tree of the folded oracle, root to the
//   transcript; last round: sum_check_last_round, the bit-reversed evaluations are the final message.
//   prover_query_phase + basefold_get_query (query_phase.rs:31-66, 373-417): 200 x "query indices" mod the codeword size; the
//   commitment pair at (x | 1) - 1, then one pair per oracle tree at ((x >> 1) | 1) - 1, ((x >> 2) | 1) - 1, ...
//   The proof has no batch sumcheck (sumcheck_proof: None) and one commitment query per index (..::Single).
// The commit phase shared by open (commit_phase.rs:30-185) and simple_batch_open (commit_phase.rs:363-503): the two differ only in
// what the first oracle / the summed evaluations are. Returns the trees of the folded oracles 1..num_rounds-1.
static inline std::vector<MerkleTree> single_commit_phase(const PcsParams& pp, std::vector<E> running_oracle, std::vector<E> running_evals, const std::vector<E>& point,
                                                          unsigned num_rounds, Transcript& t, BasefoldProof& proof) {
  std::vector<E> eq = build_eq_x_r_vec(point);
  reverse_index_bits_in_place(eq);
  one_level_interp_hc(eq); one_level_interp_hc(running_evals);
  std::vector<E> last_msg = parallel_pi(running_evals, eq);
  std::vector<MerkleTree> trees;
  std::vector<std::vector<Digest>> running_tree_inner;
  for (unsigned i = 0; i < num_rounds; i++) {
    for (E e : last_msg) t.append_ext(e);
    proof.sumcheck_messages.push_back(last_msg);
    E ch = t.get_and_append_challenge("commit round");
    std::vector<E> new_running_oracle = basefold_fold(pp, log2_strict(running_oracle.size()) - 1, running_oracle, ch);
    if (i > 0) { MerkleTree rt; rt.inner = running_tree_inner; rt.leaves = Mle::from_ext(running_oracle); trees.push_back(std::move(rt)); }
    if (i < num_rounds - 1) {
      one_level_eval_hc(running_evals, ch); one_level_eval_hc(eq, ch);
      one_level_interp_hc(eq); one_level_interp_hc(running_evals);
      last_msg = parallel_pi(running_evals, eq);
      running_tree_inner = merkelize(Mle::from_ext(new_running_oracle));
      Digest root = running_tree_inner.back()[0];
      t.append_digest(root);
      proof.roots.push_back(root);
      running_oracle = std::move(new_running_oracle);
    } else {
      one_level_eval_hc(running_evals, ch); one_level_eval_hc(eq, ch);
      reverse_index_bits_in_place(running_evals);
      t.append_exts(running_evals);
      proof.final_message = running_evals;
    }
  }
  return trees;
}
static inline void oracle_tree_queries(const std::vector<MerkleTree>& trees, size_t x_index, BatchedQuery& bq) {  // query_phase.rs:399-416 / 511-528
  size_t index = x_index >> 1;
  for (auto& tree : trees) {
    size_t p1 = index | 1, p0 = p1 - 1;
    CodewordQuery cq; cq.is_ext = true; cq.left = tree.leaves.at(p0); cq.right = tree.leaves.at(p1); cq.index = p0; cq.path = tree.path(p0);
    bq.oracle_query.push_back(std::move(cq));
    index >>= 1;
  }
}
static inline BasefoldProof pcs_open(const PcsParams& pp, const Mle& poly, const CommitmentWithWitness& comm, const std::vector<E>& point, Transcript& t) {
  if (comm.is_trivial()) return pcs_open_trivial(poly, comm);
  if (point.size() != poly.nv) throw std::runtime_error("open: point/poly mismatch");
  if (poly.nv > pp.full_message_size_log) throw std::runtime_error("open: PolynomialTooLarge");
  BasefoldProof proof;
  const unsigned num_vars = poly.nv, num_rounds = num_vars - BASECODE_MSG_SIZE_LOG;
  const Mle& cw = comm.codeword_tree.leaves;
  std::vector<E> running_oracle(cw.len());
  for (size_t j = 0; j < running_oracle.size(); j++) running_oracle[j] = cw.at(j);
  std::vector<E> running_evals(comm.bh_evals.len());
  for (size_t j = 0; j < running_evals.size(); j++) running_evals[j] = comm.bh_evals.at(j);
  std::vector<MerkleTree> trees = single_commit_phase(pp, std::move(running_oracle), std::move(running_evals), point, num_rounds, t, proof);
  const size_t codeword_size = comm.codeword_size();
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % codeword_size));
  for (size_t x_index : qidx) {
    BatchedQuery bq; bq.index = x_index;
    { size_t p1 = x_index | 1, p0 = p1 - 1;
      CodewordQuery cq; cq.is_ext = cw.is_ext; cq.left = cw.at(p0); cq.right = cw.at(p1); cq.index = p0; cq.path = comm.codeword_tree.path(p0);
      bq.commitments_query.push_back(std::move(cq)); }
    oracle_tree_queries(trees, x_index, bq);
    proof.queries.push_back(std::move(bq));
  }
  return proof;
}

// ---------------------------------------------------------------- batch_commit / simple_batch_open (several polynomials of one size in ONE tree)
// MerkleTree::from_batch_leaves (merkle_tree.rs:68-74, 261-329): pair i of the first layer is
//   hash_two_leaves_batch(a, b) = hash_two_digests(hash(a), hash(b)),  a = [v_k[2i]]_k, b = [v_k[2i+1]]_k  (util/hash.rs:32-41),
// hash = hash_or_noop over the base words of the row (<= 4 words: the words themselves, zero padded; poseidon_hash.rs:22-28);
// with ONE polynomial the ordinary tree (hash_two_leaves). Upper layers as always.
struct BatchCommitmentWithWitness {
  std::vector<std::vector<Digest>> inner;
  std::vector<Mle> codewords;  // bit-reversed RS codewords (the raw evaluations when trivial)
  std::vector<Mle> bh_evals;
  unsigned num_vars = 0;
  bool is_base = true;
  const Digest& root() const { return inner.back()[0]; }
  bool is_trivial() const { return num_vars <= BASECODE_MSG_SIZE_LOG; }
  size_t codeword_size() const { return codewords[0].len(); }
  std::vector<Digest> path(size_t leaf_index) const {
    std::vector<Digest> p;
    for (size_t l = 0; l + 1 < inner.size(); l++) p.push_back(inner[l][(leaf_index >> (l + 1)) ^ 1]);
    return p;
  }
};
static inline Digest batch_row_hash(const std::vector<Mle>& v, size_t j) {
  std::vector<u64> w;
  for (const Mle& m : v) { if (m.is_ext) { w.push_back(m.e[j].c0); w.push_back(m.e[j].c1); } else w.push_back(m.b[j]); }
  return hash_or_noop(w.data(), w.size());
}
static inline std::vector<std::vector<Digest>> merkelize_batch(const std::vector<Mle>& v) {
  if (v.size() == 1) return merkelize(v[0]);
  unsigned log_v = log2_strict(v[0].len());
  std::vector<std::vector<Digest>> tree;
  std::vector<Digest> h(v[0].len() >> 1);
  for (size_t i = 0; i < h.size(); i++) h[i] = compress(batch_row_hash(v, 2 * i), batch_row_hash(v, 2 * i + 1));
  tree.push_back(std::move(h));
  for (unsigned i = 1; i < log_v; i++) {
    const auto& prev = tree[i - 1];
    std::vector<Digest> nx(prev.size() / 2);
    par_for(nx.size(), [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) nx[j] = compress(prev[2 * j], prev[2 * j + 1]); });
    tree.push_back(std::move(nx));
  }
  return tree;
}
// Basefold::batch_commit (basefold.rs:356-446)
static inline BatchCommitmentWithWitness pcs_batch_commit(const PcsParams& pp, const std::vector<Mle>& polys) {
  if (polys.empty()) throw std::runtime_error("cannot batch commit to zero polynomials");
  BatchCommitmentWithWitness c;
  c.num_vars = polys[0].nv; c.is_base = !polys[0].is_ext;
  for (const Mle& p : polys) {
    if (p.nv != c.num_vars) throw std::runtime_error("cannot batch commit to polynomials with different number of variables");
    if (p.is_ext != polys[0].is_ext) throw std::runtime_error("batch commit: all polynomials must be in the same field");
    CommitmentWithWitness one = pcs_commit(pp, p);  // get_poly_bh_evals_and_codeword; its own tree is not used
    c.codewords.push_back(std::move(one.codeword_tree.leaves));
    c.bh_evals.push_back(std::move(one.bh_evals));
  }
  if (c.codewords[0].len() < 2) throw std::runtime_error("batch commit: a tree needs at least one pair of leaves");
  c.inner = merkelize_batch(c.codewords);
  return c;
}
// Basefold::simple_batch_open (basefold.rs:777-861) + simple_batch_commit_phase + simple_batch_prover_query_phase.
// Stream form of ..::SimpleBatched: per query `commitments_query` holds one entry per polynomial (its pair of the row pair at p0),
// all with index p0; the Merkle path of the row pair travels with entry 0, the other entries have an empty path.
static inline BasefoldProof pcs_simple_batch_open(const PcsParams& pp, const BatchCommitmentWithWitness& comm, const std::vector<E>& point, Transcript& t) {
  BasefoldProof proof;
  if (comm.is_trivial()) { proof.trivial = true; proof.trivial_proof = comm.bh_evals; return proof; }  // the transcript is not touched
  if (point.size() != comm.num_vars) throw std::runtime_error("simple_batch_open: point/poly mismatch");
  const size_t k = comm.codewords.size();
  unsigned batch_size_log = 0; while ((size_t(1) << batch_size_log) < k) batch_size_log++;
  std::vector<E> tt;
  for (unsigned i = 0; i < batch_size_log; i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
  std::vector<E> eq_xt = build_eq_x_r_vec(tt);
  eq_xt.resize(k);
  const unsigned num_rounds = comm.num_vars - BASECODE_MSG_SIZE_LOG;
  std::vector<E> running_oracle(comm.codeword_size(), e_zero()), running_evals(size_t(1) << comm.num_vars, e_zero());
  for (size_t q = 0; q < k; q++) {
    for (size_t j = 0; j < running_oracle.size(); j++) running_oracle[j] = eadd(running_oracle[j], emul(comm.codewords[q].at(j), eq_xt[q]));
    for (size_t j = 0; j < running_evals.size(); j++) running_evals[j] = eadd(running_evals[j], emul(comm.bh_evals[q].at(j), eq_xt[q]));
  }
  std::vector<MerkleTree> trees = single_commit_phase(pp, std::move(running_oracle), std::move(running_evals), point, num_rounds, t, proof);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % comm.codeword_size()));
  for (size_t x_index : qidx) {
    BatchedQuery bq; bq.index = x_index;
    const size_t p1 = x_index | 1, p0 = p1 - 1;
    for (size_t q = 0; q < k; q++) {
      CodewordQuery cq; cq.is_ext = comm.codewords[q].is_ext; cq.left = comm.codewords[q].at(p0); cq.right = comm.codewords[q].at(p1); cq.index = p0;
      if (q == 0) cq.path = comm.path(p0);
      bq.commitments_query.push_back(std::move(cq));
    }
    oracle_tree_queries(trees, x_index, bq);
    proof.queries.push_back(std::move(bq));
  }
  return proof;
}

struct Evaluation { size_t poly, point; E value; };

// Basefold::batch_open (basefold.rs:546-770) for the zkml call shape (one point per polynomial is NOT assumed
// here: the general (poly, point) indexing of the reference is kept).
static inline BasefoldProof pcs_batch_open(const PcsParams& pp, const std::vector<const Mle*>& polys,
                                           const std::vector<const CommitmentWithWitness*>& comms,
                                           const std::vector<std::vector<E>>& points,
                                           const std::vector<Evaluation>& evals, Transcript& t) {
  BasefoldProof proof;
  if (polys.empty() && comms.empty() && points.empty() && evals.empty()) { proof.trivial = true; return proof; }
  unsigned num_vars = 0, min_nv = ~0u;
  for (auto* p : polys) { num_vars = std::max(num_vars, p->nv); min_nv = std::min(min_nv, p->nv); }
  if (min_nv <= BASECODE_MSG_SIZE_LOG) throw std::runtime_error("batch_open: polynomial too small");
  for (auto* c : comms) if (c->is_trivial()) throw std::runtime_error("batch_open: trivial commitment");
  // validate_input: points match poly sizes, poly <= max
  for (auto& ev : evals) if (points[ev.point].size() != polys[ev.poly]->nv) throw std::runtime_error("batch_open: point/poly mismatch");
  if (num_vars > pp.full_message_size_log) throw std::runtime_error("batch_open: PolynomialTooLarge");

  size_t bs = 1; unsigned batch_size_log = 0;
  while (bs < evals.size()) { bs <<= 1; batch_size_log++; }
  std::vector<E> tt;
  for (unsigned i = 0; i < batch_size_log; i++) tt.push_back(t.get_and_append_challenge("batch coeffs"));
  std::vector<E> eq_xt = build_eq_x_r_vec(tt);
  E target_sum = e_zero();
  for (size_t i = 0; i < evals.size(); i++) {
    E sc = e_from_u64(u64(1) << (num_vars - points[evals[i].point].size()));
    target_sum = eadd(target_sum, emul(emul(evals[i].value, sc), eq_xt[i]));
  }
  // merged polys: one (scalar, polynomial) per point (basefold.rs:607-643)
  struct Merged { E scalar; Mle poly; bool empty = true; };
  std::vector<Merged> merged(points.size());
  for (size_t i = 0; i < evals.size(); i++) {
    Merged& m = merged[evals[i].point];
    const Mle& src = *polys[evals[i].poly];
    if (m.empty) { m.scalar = eq_xt[i]; m.poly = src; m.empty = false; }
    else {
      // force scalar to one then add poly * coeff (polys of smaller size are repeated: add_polynomial_with_coeff)
      if (m.scalar != e_one()) {
        std::vector<E> v(m.poly.len());
        for (size_t j = 0; j < v.size(); j++) v[j] = emul(m.poly.at(j), m.scalar);
        m.poly = Mle::from_ext(v); m.scalar = e_one();
      }
      if (!m.poly.is_ext) { std::vector<E> v(m.poly.len()); for (size_t j = 0; j < v.size(); j++) v[j] = m.poly.at(j); m.poly = Mle::from_ext(v); }
      if (src.nv != m.poly.nv) throw std::runtime_error("batch_open: mixed sizes at one point unsupported by oracle");
      for (size_t j = 0; j < m.poly.len(); j++) m.poly.e[j] = eadd(m.poly.e[j], emul(src.at(j), eq_xt[i]));
    }
  }
  // classic sumcheck over sum_i scalar_i * eq(x, z_i) * f_i(x) (classic.rs:232-285; coeff.rs:198-345)
  std::vector<Mle> eq_xys;
  for (auto& pt : points) eq_xys.push_back(Mle::from_ext(build_eq_x_r_vec(pt)));
  std::vector<Mle> fs;
  for (auto& m : merged) fs.push_back(m.poly);
  std::vector<E> challenges;
  E sum = target_sum;
  for (unsigned round = 0; round < num_vars; round++) {
    size_t size = size_t(1) << (num_vars - round - 1);
    E h0 = e_zero(), h2 = e_zero();
    for (size_t i = 0; i < fs.size(); i++) {
      const Mle& lhs = eq_xys[i]; const Mle& rhs = fs[i];
      size_t poly_len = size_t(1) << lhs.nv;
      E c0 = e_zero(), c2 = e_zero();
      if (poly_len == 1) {
        c0 = emul(emul(lhs.at(0), rhs.at(0)), e_from_u64(size));
      } else {
        size_t poly_size, multiple;
        if (size < poly_len || size == 1) { poly_size = size; multiple = 1; }
        else if (size == poly_len) { poly_size = poly_len >> 1; multiple = 2; }
        else { poly_size = poly_len >> 1; multiple = poly_size ? size / poly_size : 1; }
        struct C2 { E c0, c2; };
        C2 cc = par_reduce<C2>(poly_size, C2{e_zero(), e_zero()}, [&](size_t lo, size_t hi) {
          C2 a{e_zero(), e_zero()};
          for (size_t j = lo; j < hi; j++) {
            E l0 = lhs.at(2 * j), l1 = lhs.at(2 * j + 1), r0 = rhs.at(2 * j), r1 = rhs.at(2 * j + 1);
            a.c0 = eadd(a.c0, emul(l0, r0));
            a.c2 = eadd(a.c2, emul(esub(l1, l0), esub(r1, r0)));
          }
          return a;
        }, [](C2 a, C2 b) { return C2{eadd(a.c0, b.c0), eadd(a.c2, b.c2)}; });
        c0 = cc.c0; c2 = cc.c2;
        if (multiple != 1) { E mf = e_from_u64(multiple); c0 = emul(c0, mf); c2 = emul(c2, mf); }
      }
      E sc = merged[i].scalar;
      if (sc == e_one()) { h0 = eadd(h0, c0); h2 = eadd(h2, c2); }
      else if (!e_is_zero(sc)) { h0 = eadd(h0, emul(sc, c0)); h2 = eadd(h2, emul(sc, c2)); }
    }
    E h1 = esub(esub(sum, edbl(h0)), h2);
    std::vector<E> msg = {h0, h1, h2};
    for (E e : msg) t.append_ext(e);
    E ch = t.get_and_append_challenge("sumcheck round");
    challenges.push_back(ch);
    sum = eadd(h0, emul(ch, eadd(h1, emul(ch, h2))));  // horner
    for (auto& q : eq_xys) if (q.nv > 0) q.fix_low_in_place(ch);
    for (auto& f : fs) if (f.nv > 0) f.fix_low_in_place(ch);
    proof.sumcheck_proof.push_back(msg);
  }
  // coeffs (basefold.rs:690-701)
  std::vector<E> eq_xy_evals;
  for (auto& pt : points) {
    std::vector<E> c(challenges.begin(), challenges.begin() + pt.size());
    eq_xy_evals.push_back(eq_eval(c, pt));
  }
  std::vector<E> coeffs(comms.size(), e_zero());
  for (size_t i = 0; i < evals.size(); i++) coeffs[evals[i].poly] = eadd(coeffs[evals[i].poly], emul(eq_xy_evals[evals[i].point], eq_xt[i]));

  // ---- batch_commit_phase (commit_phase.rs:187-359)
  const std::vector<E>& point = challenges;
  unsigned num_rounds = num_vars - BASECODE_MSG_SIZE_LOG;
  std::vector<MerkleTree> trees;
  std::vector<E> running_oracle(size_t(1) << (num_vars + RATE_LOG), e_zero());
  for (size_t k = 0; k < comms.size(); k++)
    if (comms[k]->codeword_size() == running_oracle.size())
      par_for(running_oracle.size(), [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) running_oracle[j] = eadd(running_oracle[j], emul(comms[k]->codeword_tree.leaves.at(j), coeffs[k])); });
  std::vector<E> sum_evals(size_t(1) << num_vars, e_zero());
  for (size_t k = 0; k < comms.size(); k++) {
    const Mle& bh = comms[k]->bh_evals;
    size_t rep = size_t(1) << (num_vars - log2_strict(bh.len()));
    par_for(bh.len(), [&](size_t lo, size_t hi) {
      for (size_t j = lo; j < hi; j++) {
        E mul = emul(bh.at(j), coeffs[k]);
        for (size_t q = 0; q < rep; q++) sum_evals[j * rep + q] = eadd(sum_evals[j * rep + q], mul);
      }
    });
  }
  std::vector<E> eq = build_eq_x_r_vec(point);
  reverse_index_bits_in_place(eq);
  one_level_interp_hc(eq);
  one_level_interp_hc(sum_evals);
  std::vector<E> last_msg = parallel_pi(sum_evals, eq);
  proof.sumcheck_messages.push_back(last_msg);
  std::vector<std::vector<Digest>> running_tree_inner;
  std::vector<E> new_running_oracle;
  for (unsigned i = 0; i < num_rounds; i++) {
    for (E e : last_msg) t.append_ext(e);
    E ch = t.get_and_append_challenge("commit round");
    if (i > 0) {
      MerkleTree rt; rt.inner = running_tree_inner; rt.leaves = Mle::from_ext(new_running_oracle);
      trees.push_back(std::move(rt));
      for (size_t k = 0; k < comms.size(); k++)
        if (comms[k]->codeword_size() == new_running_oracle.size())
          par_for(new_running_oracle.size(), [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) new_running_oracle[j] = eadd(new_running_oracle[j], emul(comms[k]->codeword_tree.leaves.at(j), coeffs[k])); });
      running_oracle = new_running_oracle;
    }
    new_running_oracle = basefold_fold(pp, log2_strict(running_oracle.size()) - 1, running_oracle, ch);
    if (i < num_rounds - 1) {
      one_level_eval_hc(sum_evals, ch); one_level_eval_hc(eq, ch);
      one_level_interp_hc(eq); one_level_interp_hc(sum_evals);
      last_msg = parallel_pi(sum_evals, eq);
      proof.sumcheck_messages.push_back(last_msg);
      running_tree_inner = merkelize(Mle::from_ext(new_running_oracle));
      Digest root = running_tree_inner.back()[0];
      t.append_digest(root);
      proof.roots.push_back(root);
    } else {
      one_level_eval_hc(sum_evals, ch); one_level_eval_hc(eq, ch);
      reverse_index_bits_in_place(sum_evals);
      t.append_exts(sum_evals);
      proof.final_message = sum_evals;
    }
  }
  // ---- batch_prover_query_phase (query_phase.rs:67-102, 419-472) + merkle paths (:1062-1087)
  size_t codeword_size = size_t(1) << (num_vars + RATE_LOG);
  std::vector<size_t> qidx;
  for (unsigned q = 0; q < NUM_QUERIES; q++) qidx.push_back((size_t)(t.get_and_append_challenge("query indices").c0 % codeword_size));
  for (size_t x_index : qidx) {
    BatchedQuery bq; bq.index = x_index;
    size_t index = x_index >> 1;
    for (auto& tree : trees) {
      size_t p1 = index | 1, p0 = p1 - 1;
      CodewordQuery cq; cq.is_ext = true; cq.left = tree.leaves.at(p0); cq.right = tree.leaves.at(p1); cq.index = p0; cq.path = tree.path(p0);
      bq.oracle_query.push_back(std::move(cq));
      index >>= 1;
    }
    for (auto* comm : comms) {
      size_t xi = x_index >> (log2_strict(codeword_size) - comm->codeword_size_log());
      size_t p1 = xi | 1, p0 = p1 - 1;
      const Mle& cw = comm->codeword_tree.leaves;
      CodewordQuery cq; cq.is_ext = cw.is_ext; cq.left = cw.at(p0); cq.right = cw.at(p1); cq.index = p0; cq.path = comm->codeword_tree.path(p0);
      bq.commitments_query.push_back(std::move(cq));
    }
    proof.queries.push_back(std::move(bq));
  }
  return proof;
}

}  // namespace orc
