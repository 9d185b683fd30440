// Host <-> device interface of k_classic_tail (hip_dev.hip): the remaining rounds of the batch-opening sumcheck of
// pcs_batch_open with the transcript on the device (Dev::classic_tail). Shared with the kernel-emulation test.
#pragma once
#include "dev.h"
#include <cstring>

namespace dp {

constexpr int CT_MAXP = 128;                    // (f, eq) pairs of one batch opening
constexpr size_t CLASSIC_TAIL_MAX_N = 8192;     // the tail takes over once every table is at most this long

struct ClassicTailDesc {
  const void* f[CT_MAXP]; const Ext* eq[CT_MAXP];
  Ext* fA[CT_MAXP]; Ext* fB[CT_MAXP]; Ext* eA[CT_MAXP]; Ext* eB[CT_MAXP];  // ping-pong scratch of the folds (len / 2, len / 4)
  Ext eq_xt[CT_MAXP];
  unsigned len[CT_MAXP]; unsigned f_ext[CT_MAXP];
  int np, has_r; unsigned num_vars, round;
  Ext r, sum;
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h): mapped request / reply areas of this proof and the last sequence number served; null: the sponge runs on the device from `state`
  u64 lab[2];  // "sumcheck round"
};

inline bool classic_tail_accepts(const Dev::ClassicTailArgs& a) {
  if (a.np < 1 || a.np > CT_MAXP || a.round >= a.num_vars) return false;
  for (int i = 0; i < a.np; i++) {
    if (a.los && a.los[i].n) return false;  // factored eq tables (Dev::classic_round) are materialised before the tail's length
    if (a.fs[i].n > CLASSIC_TAIL_MAX_N || a.fs[i].n != a.eqs[i].n || !a.eqs[i].ext || a.fs[i].null() || a.eqs[i].null()) return false;
    if (a.fs[i].n == 0 || (a.fs[i].n & (a.fs[i].n - 1))) return false;
  }
  return true;
}
// message: [3 coefficients per remaining round][one challenge per round] then the sponge [8 state, 4 input buffer, in_len, out_len]
inline std::vector<size_t> classic_tail_blocks(const Dev::ClassicTailArgs& a) { return {(size_t)(a.num_vars - a.round) * 8, 14}; }

inline void classic_tail_fill(ClassicTailDesc* d, const Dev::ClassicTailArgs& a, const Challenger& ch, Dev& dev) {
  memset((void*)d, 0, sizeof(ClassicTailDesc));
  d->np = a.np; d->has_r = a.r ? 1 : 0; d->num_vars = a.num_vars; d->round = a.round;
  d->r = a.r ? *a.r : ex_zero(); d->sum = a.sum;
  for (int i = 0; i < a.np; i++) {
    const size_t n = a.fs[i].n;
    d->f[i] = a.fs[i].p; d->eq[i] = (const Ext*)a.eqs[i].p; d->len[i] = (unsigned)n; d->f_ext[i] = a.fs[i].ext ? 1 : 0; d->eq_xt[i] = a.eq_xt[i];
    if (n > 1) {
      d->fA[i] = (Ext*)dev.alloc(n / 2, true).p; d->eA[i] = (Ext*)dev.alloc(n / 2, true).p;
      d->fB[i] = (Ext*)dev.alloc(std::max<size_t>(n / 4, 1), true).p; d->eB[i] = (Ext*)dev.alloc(std::max<size_t>(n / 4, 1), true).p;
    }
  }
  for (int i = 0; i < 8; i++) d->state[i] = ch.state[i];
  for (int i = 0; i < 4; i++) d->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
  d->in_len = ch.in_len; d->out_len = ch.out_len;
  const char* lab = "sumcheck round";
  for (size_t i = 0, q = 0; i < strlen(lab) && q < 2; i += 8, q++) {
    u64 v = 0;
    size_t m = strlen(lab) - i < 8 ? strlen(lab) - i : 8;
    for (size_t b = 0; b < m; b++) v |= (u64)(uint8_t)lab[i + b] << (8 * b);
    d->lab[q] = gl_from_u64(v);
  }
}
inline void classic_tail_parse(const u64* w, const Dev::ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) {
  const size_t R = a.num_vars - a.round;
  for (size_t q = 0; q < R; q++) {
    std::vector<Ext> m(3);
    for (size_t j = 0; j < 3; j++) { size_t x = (q * 3 + j) * 2; m[j] = ex(w[x], w[x + 1]); }
    msgs.push_back(std::move(m));
  }
  for (size_t q = 0; q < R; q++) { size_t x = (R * 3 + q) * 2; challenges.push_back(ex(w[x], w[x + 1])); }
  const size_t o = R * 8;
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

}  // namespace dp
