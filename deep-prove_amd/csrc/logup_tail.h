// Host <-> device interface of k_logup_tail (hip_dev.hip): the descriptor the kernel reads, the layout of the message it
// publishes, and the host code on either side of the launch. Kept apart from hip_dev.hip so that the kernel-emulation test
// (tests/support/kernel_emul) drives the kernel source with exactly the host code the product uses.
#pragma once
#include "dev.h"
#include <cstdlib>
#include <cstring>

namespace dp {

constexpr int LT_MAXI = 7;        // instances of one batch proof: 1 + 4 * 7 tables <= LT_MAX_TABS, 3 * 7 terms <= MAX_TERMS
constexpr int LT_MAXL = 16;       // tree layers (columns of at most 2^16 rows; logup_tail_accepts stops far below)
constexpr int LT_MAX_TABS = 32;   // == MAX_TABS of hip_dev.hip
constexpr size_t LOGUP_TAIL_MAX_N = 65536;  // = 2^LT_MAXL: one workgroup still walks a 2^15-row table in well under a millisecond per layer
// DP_LOGUP_TAIL_MAX_N (a power of two, 4 .. 65536; default 65536): lookups with longer columns run layer by layer through chip-wide kernels
// (logup_layers) instead of in one workgroup. Same transcript, same proof. The transformer layer at 64 x 256 spends 72 % of its kernel time in
// one-workgroup tails over 2^14 .. 2^16-row columns with only 64 proofs in flight (profiles/r03_transformer_layer.txt): the knob is there to
// measure where the crossover lies.
inline size_t logup_tail_max_n() {
  static const size_t v = [] {
    const char* e = getenv("DP_LOGUP_TAIL_MAX_N");
    size_t x = e ? (size_t)strtoull(e, nullptr, 10) : LOGUP_TAIL_MAX_N;
    if (x < 4) x = 4;
    if (x > LOGUP_TAIL_MAX_N) x = LOGUP_TAIL_MAX_N;
    return x;
  }();
  return v;
}

struct LogupTailDesc {
  const void* num[LT_MAXI][LT_MAXL];  // numerators of tree layer li: extension; layer 0 of a table instance: the base-field
                                      // multiplicities; layer 0 of a lookup instance: unused (all numerators are -1)
  const Ext* den[LT_MAXI][LT_MAXL];
  Ext* eq; Ext* bufA[LT_MAX_TABS]; Ext* bufB[LT_MAX_TABS];
  int ninst, nlayers, total_layers, is_table;
  Ext batching, alpha, lambda, claim;
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h): mapped request / reply areas of this proof and the last sequence number served; null: the sponge runs on the device from `state`
  u64 lab_round[2], lab_batching[2], lab_alpha[2], lab_lambda[2];
  // full mode (Dev::logup_full): the kernel also builds the trees, absorbs the outputs, draws the initial challenges and
  // evaluates the columns at the final point; num / den above are then derived from den_all / num_all by the kernel
  int full, cpi;
  size_t n;                                     // rows of every column
  const u64* col[LT_MAXI][8];                   // base-field columns of every instance
  const u64* mult;                              // table: base-field multiplicities (one instance), else null
  Ext c, chi;                                   // constant / column-separation challenge of the denominators
  Ext* den_all[LT_MAXI]; Ext* num_all[LT_MAXI]; // tree storage: den layer li at 2n - (2n >> li) (2n values), num layer li >= 1 at n - (2n >> li)
  Ext* eqn;                                     // n values: eq(final point, .)
  u64 lab_ibatching[2], lab_ialpha[2], lab_ilambda[2];
  unsigned lds_ext, pad_;                       // extension values of dynamic LDS the launch brings (the layer tables live there once they fit)
};

// Dynamic LDS of a k_logup_tail launch, in bytes: the kernel keeps table slots of S values — 1 + 4 * ninst tables and the second half of the eq table's
// double buffer — and takes the tables of a layer into LDS from the first round in which they are at most S long. S = n / 2 (the widest layer from its
// first round on) when that fits into DP_LOGUP_LDS_KB (default 64, at most 80), else the largest power of two that does. 80: the launch also brings the
// message (up to MSG_LDS_MAX = 48 KB, hip_dev.hip) in dynamic LDS and the kernel's attribute is 128 KB (DP_SET_LDS_ONE; with the 6.6 KB static frame still
// inside the CU's 160 KB) — a larger setting made the LAUNCH fail instead of the kernel declining (advisor, round 5).
constexpr size_t LOGUP_TAIL_LDS_CAP_KB = 80;
inline size_t logup_tail_lds_bytes(size_t n, int ninst) {
  static const size_t cap = [] { const char* e = getenv("DP_LOGUP_LDS_KB"); size_t kb = e ? (size_t)strtoull(e, nullptr, 10) : 64; if (kb < 4) kb = 4; if (kb > LOGUP_TAIL_LDS_CAP_KB) kb = LOGUP_TAIL_LDS_CAP_KB; return kb << 10; }();
  const size_t slots = 2 + 4 * (size_t)ninst;
  size_t S = 2;
  while (2 * S <= n / 2 && 2 * S * slots * 16 <= cap) S *= 2;
  return S * slots * 16;
}

// a transcript label as the (at most two) field elements append_message makes of it (poseidon2.h Transcript)
inline void logup_tail_label(const char* lab, u64 out[2]) {
  size_t n = strlen(lab);
  out[0] = out[1] = 0;
  for (size_t i = 0, q = 0; i < n && q < 2; i += 8, q++) {
    u64 v = 0;
    size_t m = n - i < 8 ? n - i : 8;
    for (size_t b = 0; b < m; b++) v |= (u64)(uint8_t)lab[i + b] << (8 * b);
    out[q] = gl_from_u64(v);
  }
}

// the shapes the kernel was written for (everything else runs layer by layer through logup_layers)
inline bool logup_tail_accepts(const Dev::LogupTailArgs& a) {
  const std::vector<LogupCircuitDev>& cs = *a.circuits;
  const int ninst = (int)cs.size();
  if (ninst < 1 || ninst > LT_MAXI) return false;
  const size_t nlayers = cs[0].den.size();
  if (nlayers < 2 || nlayers > (size_t)LT_MAXL || a.total_layers != nlayers - 1 || a.initial_lookup == a.is_table) return false;
  const size_t n = cs[0].den[0].n;
  if (n > logup_tail_max_n() || n != (size_t(1) << nlayers)) return false;
  for (const LogupCircuitDev& c : cs) {
    if (c.den.size() != nlayers || c.num.size() != nlayers) return false;
    for (size_t li = 0; li < nlayers; li++) {
      if (c.den[li].n != (n >> li) || !c.den[li].ext || c.den[li].null()) return false;
      if (li > 0 && (c.num[li].n != (n >> li) || !c.num[li].ext || c.num[li].null())) return false;
    }
    if (a.is_table && (c.num[0].null() || c.num[0].ext || c.num[0].n != n)) return false;
  }
  return true;
}

// Message layout, in words: one block per layer lv = 1..L — [lv x 4 message values][lv challenges][batching][final
// evaluations without eq: 4 (2 in the last layer of a lookup) per instance] — then the sponge [8 state, 4 input buffer,
// in_len, out_len]. The tag is mix(seq) + sum over blocks of sum_i (i + 1) * word_i with i relative to the block.
inline std::vector<size_t> logup_tail_blocks(const Dev::LogupTailArgs& a) {
  std::vector<size_t> blocks;
  const size_t ninst = a.circuits->size();
  for (unsigned lv = 1; lv <= a.total_layers; lv++) {
    const bool lookup_final = lv == a.total_layers && !a.is_table;
    blocks.push_back(((size_t)lv * 5 + 1 + ninst * (lookup_final ? 2 : 4)) * 2);
  }
  blocks.push_back(14);
  return blocks;
}

// scratch the kernel needs, allocated by the caller: eq (n / 2), per table bufA (n / 4) and bufB (n / 8)
inline void logup_tail_fill(LogupTailDesc* d, const Dev::LogupTailArgs& a, const Challenger& ch, Dev& dev) {
  const std::vector<LogupCircuitDev>& cs = *a.circuits;
  const int ninst = (int)cs.size();
  const size_t nlayers = cs[0].den.size(), n = cs[0].den[0].n, half_max = n / 2;
  memset((void*)d, 0, sizeof(LogupTailDesc));
  for (int i = 0; i < ninst; i++)
    for (size_t li = 0; li < nlayers; li++) { d->num[i][li] = cs[i].num[li].p; d->den[i][li] = (const Ext*)cs[i].den[li].p; }
  d->eq = (Ext*)dev.alloc(half_max, true).p;
  for (int t = 0; t < 1 + 4 * ninst; t++) {
    d->bufA[t] = (Ext*)dev.alloc(std::max<size_t>(half_max / 2, 1), true).p;
    d->bufB[t] = (Ext*)dev.alloc(std::max<size_t>(half_max / 4, 1), true).p;
  }
  d->ninst = ninst; d->nlayers = (int)nlayers; d->total_layers = (int)a.total_layers; d->is_table = a.is_table ? 1 : 0;
  d->batching = a.batching; d->alpha = a.alpha; d->lambda = a.lambda; d->claim = a.claim;
  for (int i = 0; i < 8; i++) d->state[i] = ch.state[i];
  for (int i = 0; i < 4; i++) d->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
  d->in_len = ch.in_len; d->out_len = ch.out_len;
  logup_tail_label("Internal round", d->lab_round); logup_tail_label("logup_batching", d->lab_batching);
  logup_tail_label("logup_alpha", d->lab_alpha); logup_tail_label("logup_lambda", d->lab_lambda);
  d->lds_ext = (unsigned)(logup_tail_lds_bytes(n, ninst) / 16);
}

inline unsigned long long logup_tail_checksum(const volatile u64* w, const std::vector<size_t>& blocks) {
  unsigned long long cs = 0;
  size_t o = 0;
  for (size_t bw : blocks) { for (size_t i = 0; i < bw; i++) cs += (unsigned long long)(i + 1) * w[o + i]; o += bw; }
  return cs;
}

inline void logup_tail_parse(const u64* w, const Dev::LogupTailArgs& a, const std::vector<size_t>& blocks, Challenger& ch,
                             std::vector<std::vector<std::vector<Ext>>>& layer_msgs, std::vector<std::vector<Ext>>& layer_points,
                             std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) {
  size_t o = 0;
  for (unsigned lv = 1; lv <= a.total_layers; lv++) {
    std::vector<std::vector<Ext>> msgs;
    for (unsigned q = 0; q < lv; q++) {
      std::vector<Ext> m(4);
      for (unsigned j = 0; j < 4; j++) { size_t x = o + ((size_t)q * 4 + j) * 2; m[j] = ex(w[x], w[x + 1]); }
      msgs.push_back(std::move(m));
    }
    std::vector<Ext> pts;
    for (unsigned q = 0; q < lv; q++) { size_t x = o + ((size_t)lv * 4 + q) * 2; pts.push_back(ex(w[x], w[x + 1])); }
    const size_t xb = o + (size_t)lv * 10;
    const Ext batching = ex(w[xb], w[xb + 1]);
    const size_t nev = blocks[lv - 1] / 2 - ((size_t)lv * 5 + 1);
    std::vector<Ext> ev;
    for (size_t e = 0; e < nev; e++) ev.push_back(ex(w[xb + 2 + 2 * e], w[xb + 3 + 2 * e]));
    if (lv == a.total_layers) { point = pts; point.push_back(batching); }
    layer_msgs.push_back(std::move(msgs)); layer_points.push_back(std::move(pts)); round_evals.push_back(std::move(ev));
    o += blocks[lv - 1];
  }
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

// ---------------------------------------------------------------------------------------------------- full mode
// Message: [4 outputs per instance] [layer blocks as above] [evaluations of [mult,] columns at the final point] [sponge].
inline bool logup_full_accepts(const DBuf* cols, int cpi, int ninst, const DBuf& mult, size_t* n_out) {
  if (ninst < 1 || ninst > LT_MAXI || cpi < 1 || cpi > 8) return false;
  const size_t n = cols[0].n;
  if (n < 4 || n > logup_tail_max_n() || (n & (n - 1))) return false;
  for (int i = 0; i < ninst * cpi; i++) if (cols[i].null() || cols[i].ext || cols[i].n != n) return false;
  if (!mult.null() && (mult.ext || mult.n != n || ninst != 1)) return false;
  *n_out = n;
  return true;
}
inline std::vector<size_t> logup_full_blocks(size_t n, int cpi, int ninst, bool is_table) {
  std::vector<size_t> blocks;
  blocks.push_back((size_t)ninst * 8);
  unsigned nvars = dp_ceil_log2(n);
  for (unsigned lv = 1; lv + 1 <= nvars; lv++) {
    const bool lookup_final = lv == nvars - 1 && !is_table;
    blocks.push_back(((size_t)lv * 5 + 1 + (size_t)ninst * (lookup_final ? 2 : 4)) * 2);
  }
  blocks.push_back(((size_t)ninst * cpi + (is_table ? 1 : 0)) * 2);
  blocks.push_back(14);
  return blocks;
}
inline void logup_full_fill(LogupTailDesc* d, const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, const Challenger& ch, Dev& dev) {
  const size_t n = cols[0].n, half_max = n / 2;
  memset((void*)d, 0, sizeof(LogupTailDesc));
  d->full = 1; d->cpi = cpi; d->n = n; d->ninst = ninst; d->is_table = mult.null() ? 0 : 1;
  d->nlayers = (int)dp_ceil_log2(n); d->total_layers = d->nlayers - 1;
  for (int i = 0; i < ninst; i++) {
    for (int j = 0; j < cpi; j++) d->col[i][j] = (const u64*)cols[(size_t)i * cpi + j].p;
    d->den_all[i] = (Ext*)dev.alloc(2 * n, true).p;
    d->num_all[i] = (Ext*)dev.alloc(n, true).p;
  }
  d->mult = (const u64*)mult.p; d->c = c; d->chi = chi;
  d->eq = (Ext*)dev.alloc(half_max, true).p;
  d->eqn = (Ext*)dev.alloc(n, true).p;
  for (int t = 0; t < 1 + 4 * ninst; t++) {
    d->bufA[t] = (Ext*)dev.alloc(std::max<size_t>(half_max / 2, 1), true).p;
    d->bufB[t] = (Ext*)dev.alloc(std::max<size_t>(half_max / 4, 1), true).p;
  }
  for (int i = 0; i < 8; i++) d->state[i] = ch.state[i];
  for (int i = 0; i < 4; i++) d->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
  d->in_len = ch.in_len; d->out_len = ch.out_len;
  logup_tail_label("Internal round", d->lab_round); logup_tail_label("logup_batching", d->lab_batching);
  logup_tail_label("logup_alpha", d->lab_alpha); logup_tail_label("logup_lambda", d->lab_lambda);
  logup_tail_label("initial_batching", d->lab_ibatching); logup_tail_label("initial_alpha", d->lab_ialpha); logup_tail_label("initial_lambda", d->lab_ilambda);
  d->lds_ext = (unsigned)(logup_tail_lds_bytes(n, ninst) / 16);
}
inline void logup_full_parse(const u64* w, size_t n, int cpi, int ninst, bool is_table, const std::vector<size_t>& blocks, Challenger& ch, Dev::LogupFullOut& out) {
  for (int i = 0; i < 4 * ninst; i++) out.outputs.push_back(ex(w[2 * i], w[2 * i + 1]));
  const unsigned nvars = dp_ceil_log2(n);
  // the layer blocks and the sponge are parsed by the tail parser over a view that skips block 0 and the column block
  std::vector<LogupCircuitDev> none((size_t)ninst);
  Dev::LogupTailArgs ta{&none, !is_table, is_table, nvars - 1, ex_zero(), ex_zero(), ex_zero(), ex_zero()};
  std::vector<size_t> lb(blocks.begin() + 1, blocks.end() - 2);
  lb.push_back(14);
  size_t o = blocks[0], layer_words = 0;
  for (size_t i = 0; i + 1 < lb.size(); i++) layer_words += lb[i];
  // layers
  {
    std::vector<u64> view(w + o, w + o + layer_words);
    const size_t colw = blocks[blocks.size() - 2];
    view.insert(view.end(), w + o + layer_words + colw, w + o + layer_words + colw + 14);
    logup_tail_parse(view.data(), ta, lb, ch, out.layer_msgs, out.layer_points, out.round_evals, out.point);
  }
  o += layer_words;
  const size_t ncol = (size_t)ninst * cpi + (is_table ? 1 : 0);
  for (size_t i = 0; i < ncol; i++) out.col_evals.push_back(ex(w[o + 2 * i], w[o + 2 * i + 1]));
}

}  // namespace dp
