// TEST INFRASTRUCTURE: dp::EmulDev — the CPU test double with Dev::logup_tail / Dev::logup_full served by the DEVICE SOURCE of
// k_logup_tail (cut out of deep-prove_amd/csrc/hip_dev.hip by extract.py) running on the SIMT emulator of simt.hpp, driven by
// the product's own host code (csrc/logup_tail.h: descriptor, message layout, parser).
#pragma once
#include "../cpu_dev.hpp"
#include "../../../deep-prove_amd/csrc/poseidon2_fast.h"
#include "../../../deep-prove_amd/csrc/logup_tail.h"
#include "../../../deep-prove_amd/csrc/axpy_many.h"
#include "../../../deep-prove_amd/csrc/classic_tail.h"
#include "../../../deep-prove_amd/csrc/dense_tail.h"
#include "../../../deep-prove_amd/csrc/eqsum_tail.h"
#include "../../../deep-prove_amd/csrc/deleg_tail.h"
#include "../../../deep-prove_amd/csrc/commit_tail.h"
#include "../../../deep-prove_amd/csrc/sponge_host.h"
#include "simt.hpp"
// the reply poll of the device's WaveChallenger in host mode: on the emulator the "host" serves the request from inside the poll
#define WC_POLL_BEGIN
#define WC_POLL_PAUSE(spin) (dp::sponge_serve_all(), (spin) > 1000000u)
#include <cstdio>

namespace dp {
static u64 c_rc[DP_POSEIDON2_RC_WORDS];
static u64 c_extrap[(SC_MAXK + 1) * (SC_MAXK + 1) * (SC_MAXK + 1)];
#include "_build/device_extract.inc"
inline void emul_init_constants() {
  memcpy(c_rc, POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  for (unsigned k = 1; k < (unsigned)SC_MAXK; k++) for (unsigned at = k + 1; at <= (unsigned)SC_MAXK; at++) for (unsigned i = 0; i <= k; i++)
    c_extrap[((size_t)k * (SC_MAXK + 1) + at) * (SC_MAXK + 1) + i] = extrapolation_coeffs(k, at)[i];
}

// the test double with Dev::logup_tail served by the emulated kernel
// DP_EMUL_HOST_SPONGE=1: the emulated kernels run their WaveChallenger in HOST mode (sponge_host.h) — requests served from inside the
// kernel's reply poll (WC_POLL_PAUSE above) by the same service the product's waiting threads run
struct EmulSponge {
  std::vector<u64> area; SpongeSlot* slot = nullptr; Challenger* ch = nullptr; Challenger fin; bool armed = false;
  template <class D> void arm(D& d, Challenger& c) {
    if (!getenv("DP_EMUL_HOST_SPONGE")) return;
    if (!slot) { area.assign(WC_REQ_WORDS + WC_REP_WORDS, 0); slot = sponge_slot_new(); slot->req = area.data(); slot->rep = area.data() + WC_REQ_WORDS; }
    d.sp_req = area.data(); d.sp_rep = area.data() + WC_REQ_WORDS; d.sp_seq = slot->served;
    slot->ch = &c; ch = &c; slot->active.store(1); sponge_nactive().fetch_add(1); armed = true;
  }
  void done() { if (armed) { slot->active.store(0); sponge_nactive().fetch_sub(1); fin = *ch; } }
  void restore() { if (armed) { *ch = fin; armed = false; served_total += 1; } }
  size_t served_total = 0;
  size_t requests() const { return slot ? (size_t)slot->nserved.load() : 0; }
};
struct EmulDev : CpuDev {
  EmulSponge sponge;
  unsigned threads = 64;
  unsigned lds_ext = 0;  // != 0: the dynamic LDS (in extension values) the emulated k_logup_tail launch brings, instead of what logup_tail_lds_bytes asks for
  size_t taken = 0, declined = 0;
  bool full = false;  // serve Dev::logup_full (the kernel's full mode) instead of Dev::logup_tail
  unsigned long long run_kernel(const LogupTailDesc& d, std::vector<u64>& res, const std::vector<size_t>& blocks) {
    unsigned long long flag = 0;
    const unsigned long long seq = 77 + taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_logup_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: tag does not match the payload\n"); exit(3); }
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: the kernel wrote past its message\n"); exit(3); }
    return flag;
  }
  // Dev::batch_tree with the row hashes from the device source of k_batch_row_hash (blocks emulated one after the other, grid-stride loop
  // with fewer lanes than rows); the tree over them is the double's
  size_t batch_rows_hashed = 0;
  DevTree batch_tree(const DBuf* cws, int k, bool persistent) override {
    const size_t n = cws[0].n;
    DBuf rows = persistent ? alloc_persistent(2 * n, true) : alloc(2 * n, true);
    BatchRowPtrs a{};
    for (int q = 0; q < k; q++) a.cw[q] = (const u64*)cws[q].p;
    const unsigned nblk = n >= 256 ? 3 : 1, th = 64;  // 3 x 64 lanes: every lane takes several rows
    blockDim.x.v = th; gridDim.x.v = nblk;
    for (unsigned b = 0; b < nblk; b++) {
      blockIdx.x.v = b;
      if (cws[0].ext) simt::launch(th, [&] { k_batch_row_hash<true>(a, k, (u64*)rows.p, n); }); else simt::launch(th, [&] { k_batch_row_hash<false>(a, k, (u64*)rows.p, n); });
    }
    blockIdx.x.v = 0; gridDim.x.v = 1;
    batch_rows_hashed += n;
    return build_tree(rows, persistent);
  }
  bool commit = true;  // serve Dev::commit_tail with the emulated k_commit_tail
  size_t commit_taken = 0, commit_max_n = 512, commit_rounds_run = 0, commit_merged = 0;
  std::vector<u64> tw_;  // tw[i] = w_{2^(L+1)}^i, i < 2^L, L = the RS parameter size of this context (what HipDev::pcs_init builds on the device)
  bool commit_tail(const CommitTailArgs& a, Challenger& ch, CommitTailOut& out) override {
    if (!commit || !commit_tail_accepts(a, commit_max_n)) return false;  // (512 by default for emulation speed; the device takes oracles up to COMMIT_TAIL_MAX_N, 16384 in throughput mode)
    const unsigned L = full_log_;
    if (tw_.size() != (size_t(1) << L)) {
      u64 w = GL_G32;
      for (unsigned i = L + 1; i < 32; i++) w = gl_sqr(w);
      tw_.assign(size_t(1) << L, 1);
      for (size_t i = 1; i < tw_.size(); i++) tw_[i] = gl_mul(tw_[i - 1], w);
    }
    const std::vector<size_t> blocks = commit_tail_blocks(a);
    const size_t nwords = blocks[0] + blocks[1];
    CommitTailDesc d;
    std::vector<DevTree> trees;
    commit_tail_fill(&d, a, ch, *this, tw_.data(), L, trees);
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    unsigned long long flag = 0;
    const unsigned long long seq = 13000 + commit_taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_commit_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: commit tail: tag does not match the payload\n"); exit(3); }
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: commit tail wrote past its message\n"); exit(3); }
    sponge.done();
    commit_tail_parse(res.data(), a, ch, trees, out);
    sponge.restore();
    commit_taken++; commit_rounds_run += a.rounds_left;
    for (auto& mj : *a.merges) commit_merged += mj.size();
    return true;
  }
  bool eqsum = true;  // serve Dev::eqsum_tail with the emulated k_eqsum_tail
  size_t eqsum_taken = 0;
  bool eqsum_tail(const EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned nv, unsigned md,
                  Challenger& ch, EqSumOut& out) override {
    if (!eqsum || !eqsum_tail_accepts(jobs, njobs, tabs, ntabs, terms, nterms, nv, md)) return false;
    const std::vector<size_t> blocks = eqsum_tail_blocks(ntabs, nv, md);
    const size_t nwords = blocks[0] + blocks[1];
    const size_t mk = mark();
    EqSumDesc d;
    eqsum_tail_fill(&d, jobs, njobs, tabs, ntabs, terms, coeffs, nterms, nv, md, ch, *this);
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    unsigned long long flag = 0;
    const unsigned long long seq = 9000 + eqsum_taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_eqsum_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: eqsum tail: tag does not match the payload\n"); exit(3); }
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: eqsum tail wrote past its message\n"); exit(3); }
    sponge.done();
    eqsum_tail_parse(res.data(), ntabs, nv, md, ch, out);
    sponge.restore();
    release(mk);
    eqsum_taken++;
    return true;
  }
  bool deleg = true;  // serve Dev::deleg_tail with the emulated k_deleg_tail
  size_t deleg_taken = 0, deleg_sumchecks = 0;
  bool deleg_tail(const DelegTailArgs& a, Challenger& ch, DelegTailOut& out) override {
    if (!deleg || !deleg_tail_accepts(a)) return false;
    const size_t fm = a.f_middle->size();
    const std::vector<size_t> blocks = deleg_tail_blocks(fm);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    const size_t mk = mark();
    const std::vector<u64> words = deleg_tail_stage(a);
    DBuf staged = alloc(words.size(), false);
    upload(staged, words.data());
    DelegDesc d;
    deleg_tail_fill(&d, a, staged, ch, *this);
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    unsigned long long flag = 0;
    const unsigned long long seq = 13000 + deleg_taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_deleg_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: delegation tail: tag does not match the payload\n"); exit(3); }
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: delegation tail wrote past its message\n"); exit(3); }
    sponge.done();
    deleg_tail_parse(res.data(), fm, ch, out);
    sponge.restore();
    release(mk);
    deleg_taken++; deleg_sumchecks += fm;
    return true;
  }
  bool dense = true;  // serve Dev::dense_tail with the emulated k_dense_tail
  size_t dense_taken = 0;
  bool dense_tail(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, Challenger& ch, DenseTailOut& out) override {
    if (!dense || !dense_tail_accepts(bias, W, R, C, in)) return false;
    const std::vector<size_t> blocks = dense_tail_blocks(C);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    const size_t mk = mark();
    DenseTailDesc d;
    dense_tail_fill(&d, bias, W, R, C, in, pt, ch, *this);
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    unsigned long long flag = 0;
    const unsigned long long seq = 5000 + dense_taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_dense_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: dense tail: tag does not match the payload\n"); exit(3); }
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: dense tail wrote past its message\n"); exit(3); }
    sponge.done();
    dense_tail_parse(res.data(), C, ch, out);
    sponge.restore();
    release(mk);
    dense_taken++;
    return true;
  }
  // Dev::classic_round from the device source of k_classic_fused — plain and FACTORED eq tables (classic_fused_factored) —, several
  // 64-lane workgroups per pair emulated one after the other; the sum over the workgroups (k_classic_reduce on the device) is the host's
  size_t classic_rounds_emulated = 0, classic_factored_pairs = 0, eq_outer_emulated = 0;
  void classic_round(DBuf* fs, DBuf* eqs, DBuf* los, int np, const Ext* r, Ext* out) override {
    std::vector<ClassicDesc> cd((size_t)np);
    std::vector<unsigned> first((size_t)np + 1);
    unsigned nblk = 0;
    for (int i = 0; i < np; i++) {
      const size_t ln = los ? los[i].n : 0;
      DP_REQUIRE(fs[i].n == (ln ? ln : 1) * eqs[i].n, DP_ERR_SHAPE, "classic_round: f/eq shapes");
      ClassicDesc& c = cd[i];
      c.f = fs[i].p; c.eq = (const Ext*)eqs[i].p; c.fout = nullptr; c.eqout = nullptr; c.n = fs[i].n; c.fext = fs[i].ext; c.ln = (unsigned)ln; c.lo = ln ? (const Ext*)los[i].p : nullptr; c.loout = nullptr;
      if (ln) classic_factored_pairs++;
      if (r && fs[i].n > 1) {
        DBuf fo = alloc(fs[i].n / 2, true); c.fout = (Ext*)fo.p; fs[i] = fo;
        if (ln > 1) { DBuf l2 = alloc(ln / 2, true); c.loout = (Ext*)l2.p; los[i] = l2; }
        else { DBuf eo = alloc(eqs[i].n / 2, true); c.eqout = (Ext*)eo.p; eqs[i] = eo; }
      }
      first[i] = nblk; nblk += c.n >= 64 ? 3 : 1;  // three workgroups of 64 lanes: a grid-stride loop with several turns per lane
    }
    first[np] = nblk;
    std::vector<Ext> partial(2 * (size_t)nblk);
    blockDim.x.v = 64; gridDim.x.v = nblk;
    for (unsigned b = 0; b < nblk; b++) {
      blockIdx.x.v = b;
      simt::launch(64, [&] { k_classic_fused(first.data(), cd.data(), np, r ? *r : ex_zero(), r ? 1 : 0, partial.data()); });
    }
    blockIdx.x.v = 0; gridDim.x.v = 1;
    for (int i = 0; i < np; i++) {
      Ext c0 = ex_zero(), c2 = ex_zero();
      for (unsigned b = first[i]; b < first[i + 1]; b++) { c0 = ex_add(c0, partial[2 * b]); c2 = ex_add(c2, partial[2 * b + 1]); }
      out[2 * i] = c0; out[2 * i + 1] = c2;
    }
    classic_rounds_emulated++;
  }
  // Dev::axpy_many from the device source of k_axpy_classes / k_axpy_many, planned by the product's own host code (csrc/axpy_many.h): the short jobs summed
  // among their own length first (blockIdx.y = class), then the pass over the accumulator — three workgroups of 64 lanes each (the unrolled four-elements-
  // per-lane body AND the scalar remainder loop run); DP_EMUL_AXPY_CLASSES=0: the one-pass form
  size_t axpy_emulated = 0, axpy_grouped = 0, axpy_classes_emulated = 0;
  void axpy_many(const DBuf& acc, const DBuf* init, const AxpyJob* jobs, size_t n) override {
    DP_REQUIRE(acc.ext && (!init || (init->ext && init->n == acc.n)), DP_ERR_SHAPE, "axpy_many: accumulator shape");
    const char* e = getenv("DP_EMUL_AXPY_CLASSES");
    AxpyPlan p = axpy_plan(jobs, n, acc.n, !(e && !atoi(e)));
    const size_t mk = mark();
    for (size_t c = 0; c < p.classes.size(); c++) { DBuf sum = alloc(p.classes[c].n, true); p.classes[c].out = (Ext*)sum.p; p.final_pass[p.class_final[c]].x = sum.p; }
    blockDim.x.v = 64; gridDim.x.v = 3;
    if (p.grouped) {
      gridDim.y.v = (unsigned)p.classes.size();
      for (unsigned y = 0; y < p.classes.size(); y++) for (unsigned b = 0; b < 3; b++) { blockIdx.x.v = b; blockIdx.y.v = y; simt::launch(64, [&] { k_axpy_classes(p.classes.data(), p.members.data()); }); }
      blockIdx.y.v = 0; gridDim.y.v = 1;
      axpy_grouped++; axpy_classes_emulated += p.classes.size();
    }
    for (unsigned b = 0; b < 3; b++) { blockIdx.x.v = b; simt::launch(64, [&] { k_axpy_many((Ext*)acc.p, init ? (const Ext*)init->p : (const Ext*)nullptr, acc.n, p.final_pass.data(), (int)p.final_pass.size()); }); }
    blockIdx.x.v = 0; gridDim.x.v = 1;
    release(mk);
    axpy_emulated++;
  }
  void eq_outer_many(const EqOuterJob* jobs, size_t n) override {
    std::vector<EqOuterDesc> d(n);
    for (size_t i = 0; i < n; i++) { d[i].out = (Ext*)jobs[i].out.p; d[i].lo = (const Ext*)jobs[i].lo.p; d[i].hi = (const Ext*)jobs[i].hi.p; d[i].ln = (unsigned)jobs[i].lo.n; d[i].n = (unsigned)jobs[i].out.n; }
    blockDim.x.v = 64; gridDim.x.v = 2; gridDim.y.v = (int)n;
    for (unsigned y = 0; y < n; y++) for (unsigned b = 0; b < 2; b++) { blockIdx.x.v = b; blockIdx.y.v = y; simt::launch(64, [&] { k_eq_outer_many(d.data()); }); }
    blockIdx.x.v = 0; blockIdx.y.v = 0; gridDim.x.v = 1; gridDim.y.v = 1;
    eq_outer_emulated += n;
  }
  bool classic = true;  // serve Dev::classic_tail with the emulated k_classic_tail
  size_t classic_max_n = 256, classic_taken = 0;
  size_t classic_eq_materialise_n() override { return classic ? classic_max_n : CpuDev::classic_eq_materialise_n(); }  // like the device: plain tables from the tail's length on
  bool classic_tail(const ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) override {
    if (!classic || !classic_tail_accepts(a)) return false;
    for (int i = 0; i < a.np; i++) if (a.fs[i].n > classic_max_n) return false;  // (emulation speed; the device takes tables up to CLASSIC_TAIL_MAX_N)
    const std::vector<size_t> blocks = classic_tail_blocks(a);
    const size_t nwords = blocks[0] + blocks[1];
    const size_t mk = mark();
    ClassicTailDesc d;
    classic_tail_fill(&d, a, ch, *this);
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    unsigned long long flag = 0;
    const unsigned long long seq = 1000 + classic_taken;
    blockDim.x.v = threads;
    simt::launch(threads, [&] { k_classic_tail(&d, res.data(), &flag, seq); });
    if (flag != pub_mix(seq) + logup_tail_checksum(res.data(), blocks)) { fprintf(stderr, "emul: classic tail: tag does not match the payload\n"); exit(3); }
    for (size_t i = nwords; i < res.size(); i++) if (res[i] != 0xDEADBEEFDEADBEEFull) { fprintf(stderr, "emul: classic tail wrote past its message\n"); exit(3); }
    sponge.done();
    classic_tail_parse(res.data(), a, ch, msgs, challenges);
    sponge.restore();
    release(mk);
    classic_taken++;
    return true;
  }
  bool logup_full(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, Challenger& ch, LogupFullOut& out) override {
    size_t n = 0;
    if (!full) return false;
    if (!logup_full_accepts(cols, cpi, ninst, mult, &n)) { declined++; return false; }
    const std::vector<size_t> blocks = logup_full_blocks(n, cpi, ninst, !mult.null());
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    const size_t mk = mark();
    LogupTailDesc d;
    logup_full_fill(&d, cols, cpi, ninst, mult, c, chi, ch, *this);
    if (lds_ext) d.lds_ext = lds_ext;
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    run_kernel(d, res, blocks);
    sponge.done();
    logup_full_parse(res.data(), n, cpi, ninst, !mult.null(), blocks, ch, out);
    sponge.restore();
    release(mk);
    taken++;
    return true;
  }
  bool logup_tail(const LogupTailArgs& a, Challenger& ch, std::vector<std::vector<std::vector<Ext>>>& layer_msgs,
                  std::vector<std::vector<Ext>>& layer_points, std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) override {
    if (full) return false;
    if (!logup_tail_accepts(a)) { declined++; return false; }
    const std::vector<size_t> blocks = logup_tail_blocks(a);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    const size_t mk = mark();
    LogupTailDesc d;
    logup_tail_fill(&d, a, ch, *this);
    if (lds_ext) d.lds_ext = lds_ext;
    sponge.arm(d, ch);
    std::vector<u64> res(nwords + 8, 0xDEADBEEFDEADBEEFull);
    run_kernel(d, res, blocks);
    sponge.done();
    logup_tail_parse(res.data(), a, blocks, ch, layer_msgs, layer_points, round_evals, point);
    sponge.restore();
    release(mk);
    taken++;
    return true;
  }
};
}  // namespace dp

