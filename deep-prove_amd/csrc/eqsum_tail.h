// Host <-> device interface of k_eqsum_tail (hip_dev.hip): eq tables followed by a whole sumcheck over tables that include
// them, in one launch (Dev::eqsum_tail). Shared with the kernel-emulation test.
#pragma once
#include "dev.h"
#include <cstring>

namespace dp {

constexpr int ES_MAXJ = 8, ES_MAXT = 32, ES_MAXTERM = 48, ES_MAXV = 14;  // eq jobs, tables, terms, variables (tables of <= 2^14 values)

struct EqSumDesc {
  Ext* job_out[ES_MAXJ]; Ext job_pt[ES_MAXJ][ES_MAXV]; Ext job_scale[ES_MAXJ]; int job_acc[ES_MAXJ];
  const void* tab[ES_MAXT]; int tab_ext[ES_MAXT]; Ext* bufA[ES_MAXT]; Ext* bufB[ES_MAXT];
  int tk[ES_MAXTERM]; int tt[ES_MAXTERM][3]; Ext coeff[ES_MAXTERM];
  int njobs, ntabs, nterms; unsigned nv, md;
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h): mapped request / reply areas of this proof and the last sequence number served; null: the sponge runs on the device from `state`
  u64 lab_round[2];  // "Internal round"
};

inline bool eqsum_tail_accepts(const Dev::EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, int nterms, unsigned nv, unsigned md) {
  if (njobs < 0 || njobs > ES_MAXJ || ntabs < 1 || ntabs > ES_MAXT || nterms < 1 || nterms > ES_MAXTERM || nv < 1 || nv > (unsigned)ES_MAXV || md < 1 || md > 3) return false;
  const size_t n = size_t(1) << nv;
  for (int i = 0; i < ntabs; i++) if (tabs[i].null() || tabs[i].n != n) return false;
  for (int j = 0; j < njobs; j++) if (jobs[j].out.null() || !jobs[j].out.ext || jobs[j].out.n != n || jobs[j].k != nv) return false;
  for (int i = 0; i < nterms; i++) {
    if (terms[i].k < 1 || terms[i].k > 3 || (unsigned)terms[i].k > md) return false;
    for (int q = 0; q < terms[i].k; q++) if (terms[i].t[q] < 0 || terms[i].t[q] >= ntabs) return false;
  }
  return true;
}
// message: [max_degree + 1 evaluations per round][one challenge per round][final evaluation of every table] then the sponge
inline std::vector<size_t> eqsum_tail_blocks(int ntabs, unsigned nv, unsigned md) { return {((size_t)nv * (md + 1) + nv + (size_t)ntabs) * 2, 14}; }

inline void eqsum_tail_fill(EqSumDesc* d, const Dev::EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms,
                            unsigned nv, unsigned md, const Challenger& ch, Dev& dev) {
  memset((void*)d, 0, sizeof(EqSumDesc));
  const size_t n = size_t(1) << nv;
  d->njobs = njobs; d->ntabs = ntabs; d->nterms = nterms; d->nv = nv; d->md = md;
  for (int j = 0; j < njobs; j++) {
    d->job_out[j] = (Ext*)jobs[j].out.p; d->job_scale[j] = jobs[j].scale; d->job_acc[j] = jobs[j].accumulate ? 1 : 0;
    for (unsigned q = 0; q < nv; q++) d->job_pt[j][q] = jobs[j].pt[q];
  }
  for (int i = 0; i < ntabs; i++) {
    d->tab[i] = tabs[i].p; d->tab_ext[i] = tabs[i].ext ? 1 : 0;
    d->bufA[i] = (Ext*)dev.alloc(n / 2, true).p; d->bufB[i] = (Ext*)dev.alloc(std::max<size_t>(n / 4, 1), true).p;
  }
  for (int i = 0; i < nterms; i++) { d->tk[i] = terms[i].k; for (int q = 0; q < 3; q++) d->tt[i][q] = q < terms[i].k ? terms[i].t[q] : 0; d->coeff[i] = coeffs[i]; }
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
inline void eqsum_tail_parse(const u64* w, int ntabs, unsigned nv, unsigned md, Challenger& ch, Dev::EqSumOut& out) {
  for (unsigned q = 0; q < nv; q++) {
    std::vector<Ext> m(md + 1);
    for (unsigned j = 0; j <= md; j++) { size_t x = ((size_t)q * (md + 1) + j) * 2; m[j] = ex(w[x], w[x + 1]); }
    out.msgs.push_back(std::move(m));
  }
  for (unsigned q = 0; q < nv; q++) { size_t x = ((size_t)nv * (md + 1) + q) * 2; out.point.push_back(ex(w[x], w[x + 1])); }
  const size_t xf = (size_t)nv * (md + 2) * 2;
  for (int i = 0; i < ntabs; i++) out.finals.push_back(ex(w[xf + 2 * i], w[xf + 2 * i + 1]));
  const size_t o = xf + 2 * (size_t)ntabs;
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

}  // namespace dp
