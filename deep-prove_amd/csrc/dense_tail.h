// Host <-> device interface of k_dense_tail (hip_dev.hip): the device part of a Dense layer's proof in one launch
// (Dev::dense_tail). Shared with the kernel-emulation test.
#pragma once
#include "dev.h"
#include <cstring>

namespace dp {

constexpr size_t DENSE_TAIL_MAX_R = 4096, DENSE_TAIL_MAX_C = 16384;

struct DenseTailDesc {
  const u64* bias; const u64* W; const Ext* in;   // bias[R], W[R x C] row-major (base field), in[C] (extension)
  Ext* eqr; Ext* mat; Ext* inA; Ext* bufA[2]; Ext* bufB[2];  // scratch: eq(point, .) over the rows, W(point, .), folds of (mat, in)
  unsigned R, C, lgR, lgC;
  Ext pt[16];                                     // the output point (log2 R coordinates)
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h): mapped request / reply areas of this proof and the last sequence number served; null: the sponge runs on the device from `state`
  u64 lab_round[2];                               // "Internal round"
};

inline bool dense_tail_accepts(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in) {
  if (R < 2 || C < 2 || R > DENSE_TAIL_MAX_R || C > DENSE_TAIL_MAX_C || (R & (R - 1)) || (C & (C - 1))) return false;
  if (bias.null() || bias.ext || bias.n != R || W.null() || W.ext || W.n != R * C || in.null() || !in.ext || in.n != C) return false;
  return dp_ceil_log2(R) <= 16;
}
// message, in three blocks (the tag sums (i + 1) * word_i with i relative to its block): [bias_eval] [3 evaluations per round,
// one challenge per round, 2 final evaluations] [the sponge: 8 state, 4 input buffer, in_len, out_len]
inline std::vector<size_t> dense_tail_blocks(size_t C) { const size_t lg = dp_ceil_log2(C); return {2, (lg * 3 + lg + 2) * 2, 14}; }

inline void dense_tail_fill(DenseTailDesc* d, const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, const Challenger& ch, Dev& dev) {
  memset((void*)d, 0, sizeof(DenseTailDesc));
  d->bias = (const u64*)bias.p; d->W = (const u64*)W.p; d->in = (const Ext*)in.p;
  d->R = (unsigned)R; d->C = (unsigned)C; d->lgR = dp_ceil_log2(R); d->lgC = dp_ceil_log2(C);
  d->eqr = (Ext*)dev.alloc(R, true).p; d->mat = (Ext*)dev.alloc(C, true).p;
  for (int t = 0; t < 2; t++) { d->bufA[t] = (Ext*)dev.alloc(C / 2, true).p; d->bufB[t] = (Ext*)dev.alloc(std::max<size_t>(C / 4, 1), true).p; }
  for (unsigned i = 0; i < d->lgR; i++) d->pt[i] = pt[i];
  for (int i = 0; i < 8; i++) d->state[i] = ch.state[i];
  for (int i = 0; i < 4; i++) d->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
  d->in_len = ch.in_len; d->out_len = ch.out_len;
  const char* lab = "Internal round";
  for (size_t i = 0, q = 0; i < strlen(lab) && q < 2; i += 8, q++) {
    u64 v = 0;
    size_t m = strlen(lab) - i < 8 ? strlen(lab) - i : 8;
    for (size_t b = 0; b < m; b++) v |= (u64)(uint8_t)lab[i + b] << (8 * b);
    d->lab_round[q] = gl_from_u64(v);
  }
}
inline void dense_tail_parse(const u64* w, size_t C, Challenger& ch, Dev::DenseTailOut& out) {
  const size_t lg = dp_ceil_log2(C);
  out.bias_eval = ex(w[0], w[1]);
  for (size_t q = 0; q < lg; q++) {
    std::vector<Ext> m(3);
    for (size_t j = 0; j < 3; j++) { size_t x = 2 + (q * 3 + j) * 2; m[j] = ex(w[x], w[x + 1]); }
    out.msgs.push_back(std::move(m));
  }
  for (size_t q = 0; q < lg; q++) { size_t x = 2 + (lg * 3 + q) * 2; out.point.push_back(ex(w[x], w[x + 1])); }
  const size_t xf = 2 + lg * 8;
  out.finals[0] = ex(w[xf], w[xf + 1]); out.finals[1] = ex(w[xf + 2], w[xf + 3]);
  const size_t o = (1 + lg * 4 + 2) * 2;
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

}  // namespace dp
