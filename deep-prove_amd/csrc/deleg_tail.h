// Host <-> device interface of k_deleg_tail (kernels.inc): ALL delegation sumchecks of one batch FFT / iFFT of the zkCNN convolution
// protocol (Prover::delegate_matrix_evaluation, zkml/src/iop/prover.rs:164-211) in one launch of one workgroup — per sumcheck the phi
// table (powers of the root of unity scaled by the previous point's last coordinate), the beta table, the degree-3 rounds with the
// transcript on the device; the next sumcheck's point is this one's challenges (Dev::deleg_tail). Shared with the kernel-emulation test.
#pragma once
#include "dev.h"
#include <cstring>

namespace dp {

constexpr int DG_MAXV = 12;  // variables of the longest intermediate table (CNN-264k: 10)

struct DelegDesc {
  const Ext* fmid;    // f_middle[0 .. fm) back to back: table l has 2^(l+1) entries and starts at 2^(l+1) - 2
  const u64* omegas;  // 2^(fm+1) powers of the root of unity (phi_pow_init)
  Ext* phi; Ext* beta;           // 2^fm entries each
  Ext* bufA[3]; Ext* bufB[3];    // fold ping-pong of (beta, phi, f_middle): 2^(fm-1) and 2^(fm-2) entries
  Ext r1[DG_MAXV + 1]; Ext r2[DG_MAXV + 1];
  int fm, is_fft;
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;
  u64* sp_req; const u64* sp_rep; unsigned long long sp_seq;  // host sponge (sponge_host.h), null: the sponge runs on the device from `state`
  u64 lab_round[2];  // "Internal round"
};

inline bool deleg_tail_accepts(const Dev::DelegTailArgs& a) {
  if (!a.f_middle || !a.r1 || !a.r2 || !a.omegas) return false;
  const size_t fm = a.f_middle->size();
  if (fm < 1 || fm > (size_t)DG_MAXV || a.n1 != fm + 1 || a.nomegas != (size_t(1) << (fm + 1))) return false;
  for (size_t l = 0; l < fm; l++) if ((*a.f_middle)[l].size() != (size_t(2) << l)) return false;
  return true;
}
// message: per sumcheck (nv = fm .. 1) [4 evaluations per round][one challenge per round][3 final evaluations], then the sponge
inline std::vector<size_t> deleg_tail_blocks(size_t fm) {
  std::vector<size_t> b;
  for (size_t nv = fm; nv >= 1; nv--) b.push_back((nv * 4 + nv + 3) * 2);
  b.push_back(14);
  return b;
}
// `staged` = [omegas (2^(fm+1) words)][f_middle tables as extension words] already on the device
inline void deleg_tail_fill(DelegDesc* d, const Dev::DelegTailArgs& a, const DBuf& staged, const Challenger& ch, Dev& dev) {
  memset((void*)d, 0, sizeof(DelegDesc));
  const size_t fm = a.f_middle->size();
  d->fm = (int)fm; d->is_fft = a.is_fft ? 1 : 0;
  d->omegas = (const u64*)staged.p; d->fmid = (const Ext*)((const u64*)staged.p + a.nomegas);
  for (size_t i = 0; i <= fm; i++) { d->r1[i] = a.r1[i]; d->r2[i] = a.r2[i]; }
  const size_t n = size_t(1) << fm;
  d->phi = (Ext*)dev.alloc(n, true).p; d->beta = (Ext*)dev.alloc(n, true).p;
  for (int t = 0; t < 3; t++) { d->bufA[t] = (Ext*)dev.alloc(std::max<size_t>(n / 2, 1), true).p; d->bufB[t] = (Ext*)dev.alloc(std::max<size_t>(n / 4, 1), true).p; }
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
// the words the host stages for the kernel: omegas, then the tables
inline std::vector<u64> deleg_tail_stage(const Dev::DelegTailArgs& a) {
  std::vector<u64> w(a.omegas, a.omegas + a.nomegas);
  for (const std::vector<Ext>& t : *a.f_middle) for (const Ext& e : t) { w.push_back(e.c0); w.push_back(e.c1); }
  return w;
}
inline void deleg_tail_parse(const u64* w, size_t fm, Challenger& ch, Dev::DelegTailOut& out) {
  size_t o = 0;
  for (size_t nv = fm; nv >= 1; nv--) {
    std::vector<std::vector<Ext>> msgs;
    for (size_t q = 0; q < nv; q++) { std::vector<Ext> m(4); for (size_t j = 0; j < 4; j++) { size_t x = o + (q * 4 + j) * 2; m[j] = ex(w[x], w[x + 1]); } msgs.push_back(std::move(m)); }
    std::vector<Ext> pt, fin;
    for (size_t q = 0; q < nv; q++) { size_t x = o + (nv * 4 + q) * 2; pt.push_back(ex(w[x], w[x + 1])); }
    for (size_t e = 0; e < 3; e++) { size_t x = o + (nv * 5 + e) * 2; fin.push_back(ex(w[x], w[x + 1])); }
    out.msgs.push_back(std::move(msgs)); out.points.push_back(std::move(pt)); out.finals.push_back(std::move(fin));
    o += (nv * 5 + 3) * 2;
  }
  for (int i = 0; i < 8; i++) ch.state[i] = w[o + i];
  ch.in_len = (int)w[o + 12]; ch.out_len = (int)w[o + 13];
  for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[o + 8 + i]; ch.out_buf[i] = ch.state[i]; }
}

}  // namespace dp
