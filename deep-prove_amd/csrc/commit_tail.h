// Host <-> device interface of k_commit_tail (hip_dev.hip): the last rounds of the Basefold commit phase in one launch
// (Dev::commit_tail). Shared with the kernel-emulation test.
#pragma once
#include <algorithm>
#include "dev.h"
#include <cstdlib>
#include <cstring>

namespace dp {

constexpr int CMT_MAXR = 8;                     // rounds one launch runs
constexpr int CMT_MAXM = 64;                    // committed codewords merged into the oracle of one round
constexpr size_t COMMIT_TAIL_MAX_N = 4096;      // the tail takes over when the previous round's folded oracle is at most this long
// Every round the tail takes saves the ~11 launches and 2 host round trips of a round driven from the host (fold, pair fold, message,
// reduction, leaves, Merkle layers, Merkle tail) at the price of ONE workgroup doing the work: a single proof (1024 threads, the GPU
// otherwise idle) is fastest with 4096; with hundreds of proofs in flight the throughput is the same from 4096 to 65536
// (profiles/r02_ctail_inflight_sweep.jsonl), so throughput mode takes over at 16384 and spends 22 launches less per proof.
constexpr size_t COMMIT_TAIL_MAX_N_THROUGHPUT = 16384;
inline size_t commit_tail_max_n(bool throughput_mode) {
  return throughput_mode ? COMMIT_TAIL_MAX_N_THROUGHPUT : COMMIT_TAIL_MAX_N;
}

struct CommitTailDesc {
  Ext last[3];
  const Ext* folded; const Ext* eq; const Ext* f;   // previous folded oracle (n values), sumcheck pairs (m values each)
  unsigned n, m; int rounds;
  int nmerge[CMT_MAXR]; const void* mcw[CMT_MAXR][CMT_MAXM]; int mext[CMT_MAXR][CMT_MAXM]; Ext mcoeff[CMT_MAXR][CMT_MAXM];
  Ext* run[CMT_MAXR];                               // merged oracle of round j (n >> j values), when it has merges
  Ext* leaves[CMT_MAXR]; u64* nodes[CMT_MAXR];      // tree of round j: folded oracle (n >> (j + 1) values) and its digests
  u64 gamma[CMT_MAXR], ninv[CMT_MAXR]; unsigned level[CMT_MAXR];
  const u64* tw; unsigned L;                        // tw[i] = w_{2^(L+1)}^i, i < 2^L (the RS parameters of the context)
  Ext* eqA; Ext* eqB; Ext* fA; Ext* fB;             // ping-pong of the folded sumcheck pairs
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h): mapped request / reply areas of this proof and the last sequence number served; null: the sponge runs on the device from `state`
  u64 lab[2];                                       // "commit round"
};

inline bool commit_tail_accepts(const Dev::CommitTailArgs& a, size_t max_n = COMMIT_TAIL_MAX_N) {
  if (a.rounds_left < 1 || a.rounds_left > (unsigned)CMT_MAXR || a.merges->size() != a.rounds_left) return false;
  const size_t n = a.folded.n, m = a.sum_evals.n;
  if (a.folded.null() || !a.folded.ext || n > max_n || (n & (n - 1)) || (n >> a.rounds_left) < 2) return false;
  if (a.eq.null() || a.sum_evals.null() || !a.eq.ext || !a.sum_evals.ext || a.eq.n != m || (m & (m - 1)) || (m >> a.rounds_left) < 1) return false;
  for (unsigned j = 0; j < a.rounds_left; j++) {
    if ((*a.merges)[j].size() > (size_t)CMT_MAXM) return false;
    for (const Dev::AxpyJob& job : (*a.merges)[j]) if (job.rep != 1 || job.x.null() || job.x.n != (n >> j)) return false;
  }
  return true;
}
// message: [3 coefficients and 4 root words per non-final round][final message, natural index order] then the sponge
inline std::vector<size_t> commit_tail_blocks(const Dev::CommitTailArgs& a) {
  const size_t R = a.rounds_left;
  return {(R - 1) * 10 + (a.sum_evals.n >> R) * 2, 14};
}
// tw / L: the RS tables of the device (fri_fold: x0 = gamma * tw[bitrev(i) << (L - level)], gamma = 7^(2^(L - level)))
inline void commit_tail_fill(CommitTailDesc* d, const Dev::CommitTailArgs& a, const Challenger& ch, Dev& dev, const u64* tw, unsigned L, std::vector<DevTree>& trees) {
  memset((void*)d, 0, sizeof(CommitTailDesc));
  const size_t n = a.folded.n, m = a.sum_evals.n;
  for (int q = 0; q < 3; q++) d->last[q] = a.last[q];
  d->folded = (const Ext*)a.folded.p; d->eq = (const Ext*)a.eq.p; d->f = (const Ext*)a.sum_evals.p;
  d->n = (unsigned)n; d->m = (unsigned)m; d->rounds = (int)a.rounds_left; d->tw = tw; d->L = L;
  // trees first: they outlive the call (the query phase reads them); the caller marks the arena after this function
  trees.clear();
  for (unsigned j = 0; j + 1 < a.rounds_left; j++) {
    DevTree t; t.nleaves = n >> (j + 1);
    t.leaves = dev.alloc(t.nleaves, true); t.nodes = dev.alloc(4 * (t.nleaves - 1), false);
    d->leaves[j] = (Ext*)t.leaves.p; d->nodes[j] = (u64*)t.nodes.p;
    trees.push_back(t);
  }
  for (unsigned j = 0; j < a.rounds_left; j++) {
    const std::vector<Dev::AxpyJob>& mj = (*a.merges)[j];
    d->nmerge[j] = (int)mj.size();
    for (size_t k = 0; k < mj.size(); k++) { d->mcw[j][k] = mj[k].x.p; d->mext[j][k] = mj[k].x.ext ? 1 : 0; d->mcoeff[j][k] = mj[k].coeff; }
    if (!mj.empty()) d->run[j] = (Ext*)dev.alloc(n >> j, true).p;
    const unsigned level = dp_ceil_log2(n >> j) - 1;
    u64 gam = GL_GENERATOR;
    for (unsigned i = 0; i < L + 1 - level - 1; i++) gam = gl_sqr(gam);
    d->level[j] = level; d->gamma[j] = gam; d->ninv[j] = gl_neg(gl_inv(gl_dbl(gam)));
  }
  d->eqA = (Ext*)dev.alloc(std::max<size_t>(m / 2, 1), true).p; d->eqB = (Ext*)dev.alloc(std::max<size_t>(m / 4, 1), true).p;
  d->fA = (Ext*)dev.alloc(std::max<size_t>(m / 2, 1), true).p; d->fB = (Ext*)dev.alloc(std::max<size_t>(m / 4, 1), true).p;
  for (int i = 0; i < 8; i++) d->state[i] = ch.state[i];
  for (int i = 0; i < 4; i++) d->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
  d->in_len = ch.in_len; d->out_len = ch.out_len;
  const char* lab = "commit round";
  for (size_t i = 0, q = 0; i < strlen(lab) && q < 2; i += 8, q++) {
    u64 v = 0;
    size_t mm = strlen(lab) - i < 8 ? strlen(lab) - i : 8;
    for (size_t b = 0; b < mm; b++) v |= (u64)(uint8_t)lab[i + b] << (8 * b);
    d->lab[q] = gl_from_u64(v);
  }
}
inline void commit_tail_parse(const u64* w, const Dev::CommitTailArgs& a, Challenger& ch, std::vector<DevTree>& trees, Dev::CommitTailOut& out) {
  const size_t R = a.rounds_left, mf = a.sum_evals.n >> R;
  for (size_t j = 0; j + 1 < R; j++) {
    const u64* b = w + j * 10;
    out.msgs.push_back({ex(b[0], b[1]), ex(b[2], b[3]), ex(b[4], b[5])});
    for (int k = 0; k < 4; k++) trees[j].root.v[k] = b[6 + k];
  }
  out.trees = trees;
  const u64* fm = w + (R - 1) * 10;
  for (size_t r = 0; r < mf; r++) out.final_message.push_back(ex(fm[2 * r], fm[2 * r + 1]));
  const size_t o = (R - 1) * 10 + mf * 2;
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

}  // namespace dp
