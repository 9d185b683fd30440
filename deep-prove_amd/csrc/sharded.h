// One sumcheck sharded over several devices, the round loop in the library (BASELINE config 5).
// The partition is the reference's thread-sharded prover IOPProverState::prove_batch_polys (sumcheck/src/prover.rs:37-321,
// merge step sumcheck/src/util.rs:215-243): worker g of W = 2^k owns the contiguous slice [g N/W, (g+1) N/W) of every table
// (the top k variables select the worker). Variables are bound low to high, so the first nv - k rounds are local: every worker
// computes the round sums of its slice, the shares are all-gathered — (degree + 1) extension elements per term, a few hundred
// bytes: pure latency — and added mod p (a mod-p sum is not an RCCL reduction: canonical words would not reduce), every worker
// runs the same Fiat-Shamir transcript on the total and folds its slice with the same challenge. After nv - k rounds each
// worker holds one value per table; those are all-gathered into tables of W entries and the last k rounds run (identically)
// on every worker. The transcript sequence is that of the unsharded prover: the proof is bit-identical to sumcheck_prove.
#pragma once
#include "sumcheck.h"
#include <condition_variable>
#include <mutex>

namespace dp {

struct Exchange {
  virtual ~Exchange() {}
  virtual int world() const = 0;
  virtual int rank() const = 0;
  // every rank contributes `nwords` u64 words; `out` receives world * nwords words in rank order
  virtual void all_gather(const u64* send, size_t nwords, u64* out) = 0;
  // the same exchange on device buffers (Dev::ShareExchange), for devices that keep the round's shares in HBM; null: host exchange only
  virtual Dev::ShareExchange* device() { return nullptr; }
};
struct SoloExchange : Exchange {  // world of one
  int world() const override { return 1; }
  int rank() const override { return 0; }
  void all_gather(const u64* send, size_t nwords, u64* out) override { for (size_t i = 0; i < nwords; i++) out[i] = send[i]; }
};
// W ranks as W threads of one process (tests, several contexts on one GPU): a reusable barrier around a shared buffer
struct ThreadExchangeHub {
  int world; std::mutex mu; std::condition_variable cv; std::vector<u64> buf; int arrived = 0, left = 0; unsigned long long gen = 0; size_t nwords = 0;
  bool aborted = false;  // a rank failed: every rank waiting in (or arriving at) an exchange throws instead of waiting for it forever
  explicit ThreadExchangeHub(int w) : world(w) {}
  void abort() { { std::lock_guard<std::mutex> lk(mu); aborted = true; } cv.notify_all(); }
};
struct ThreadExchange : Exchange {
  ThreadExchangeHub& h; int r;
  Dev::ShareExchange* dev_x = nullptr;  // (set by the caller that has a device-side implementation: capi.cpp)
  Dev::ShareExchange* device() override { return dev_x; }
  ThreadExchange(ThreadExchangeHub& hub, int rank_) : h(hub), r(rank_) {}
  int world() const override { return h.world; }
  int rank() const override { return r; }
  void all_gather(const u64* send, size_t nwords, u64* out) override {
    std::unique_lock<std::mutex> lk(h.mu);
    auto dead = [&] { if (h.aborted) throw DpError(DP_ERR_HIP, "sharded sumcheck: another rank failed, the exchange is abandoned"); };
    h.cv.wait(lk, [&] { return h.left == 0 || h.aborted; });  // the previous exchange has been read by everyone
    dead();
    if (h.arrived == 0) { h.nwords = nwords; h.buf.assign((size_t)h.world * nwords, 0); }
    if (nwords != h.nwords) { h.aborted = true; h.cv.notify_all(); throw DpError(DP_ERR_SHAPE, "sharded sumcheck: ranks disagree on the message size"); }
    for (size_t i = 0; i < nwords; i++) h.buf[(size_t)r * nwords + i] = send[i];
    const unsigned long long my = h.gen;
    if (++h.arrived == h.world) { h.arrived = 0; h.left = h.world; h.gen++; h.cv.notify_all(); }
    else { h.cv.wait(lk, [&] { return h.gen != my || h.aborted; }); if (h.gen == my) dead(); }
    for (size_t i = 0; i < (size_t)h.world * nwords; i++) out[i] = h.buf[i];
    if (--h.left == 0) h.cv.notify_all();
  }
};

// raw per-term sums (summed over the ranks) -> the round message, exactly as sumcheck_prove builds it
inline std::vector<Ext> sharded_message(const DevVP& vp, const std::vector<Ext>& raw) {
  const unsigned md = vp.max_degree;
  std::vector<Ext> msg(md + 1, ex_zero());
  size_t off = 0;
  for (size_t ti = 0; ti < vp.terms.size(); ti++) {
    unsigned k = vp.terms[ti].k;
    std::vector<Ext> s(k + 1);
    for (unsigned j = 0; j <= k; j++) s[j] = ex_mul(raw[off + j], vp.coeffs[ti]);
    off += k + 1;
    for (unsigned j = 0; j <= md; j++) msg[j] = ex_add(msg[j], j <= k ? s[j] : extrapolate_small(s.data(), k, j));
  }
  return msg;
}

// This rank's side of the sharded prover. `vp`: the virtual polynomial over THIS rank's slices (every table 2^(nv - k) entries,
// k = log2 world); `nv`: the number of variables of the whole polynomial. Returns the proof and final evaluations of the whole
// sumcheck (identical on every rank).
inline SumcheckOut sumcheck_prove_sharded(Dev& dev, Exchange& xch, unsigned nv, DevVP& vp, Transcript& t) {
  const int W = xch.world();
  unsigned k = 0; while ((1 << k) < W) k++;
  DP_REQUIRE((1 << k) == W && nv > k && vp.nv == nv - k, DP_ERR_SHAPE, "sharded sumcheck: world must be a power of two below 2^num_vars, local tables 2^(num_vars - log2 world) long");
  for (const DBuf& b : vp.tabs) DP_REQUIRE(b.n == (size_t(1) << vp.nv), DP_ERR_SHAPE, "sharded sumcheck: every local table has 2^(num_vars - log2 world) entries");
  const unsigned md = vp.max_degree, nv_local = vp.nv;
  const size_t nt = vp.tabs.size();
  size_t nraw = 0; for (auto& tm : vp.terms) nraw += tm.k + 1;
  SumcheckOut out;
  const size_t mk = dev.mark();
  t.append_usize(nv);
  t.append_usize(md);
  std::vector<DBuf> tabs = vp.tabs;
  std::vector<Ext> raw(nraw), total(nraw);
  std::vector<u64> send, recv;
  auto gather_sum = [&]() {  // shares of the round sums -> their sum mod p
    send.resize(2 * nraw); recv.resize((size_t)W * 2 * nraw);
    for (size_t i = 0; i < nraw; i++) { send[2 * i] = raw[i].c0; send[2 * i + 1] = raw[i].c1; }
    xch.all_gather(send.data(), 2 * nraw, recv.data());
    for (size_t i = 0; i < nraw; i++) {
      Ext acc = ex_zero();
      for (int g = 0; g < W; g++) acc = ex_add(acc, ex(recv[((size_t)g * nraw + i) * 2], recv[((size_t)g * nraw + i) * 2 + 1]));
      total[i] = acc;
    }
  };
  Ext ch = ex_zero();
  // shares on the device (ncclAllGather on HBM buffers, the mod-p sum in a kernel, ONE host wait per round) when both the exchange
  // and the device can; else the shares travel through the host (the CPU double, hosts that bring their own exchange)
  Dev::ShareExchange* sx = xch.device();
  bool dev_shares = sx && dev.sc_set_share_exchange(sx);
  struct Unset { Dev& d; bool& on; ~Unset() { if (on) d.sc_set_share_exchange(nullptr); } } unset{dev, dev_shares};
  // the ranks must AGREE on where the shares travel (device exchange = barrier-less ncclAllGather inside sc_round, host exchange = all_gather
  // here): a rank that decides differently (DP_SHARDED_HOST_EXCHANGE set on one rank only, a device that refuses) would deadlock the others
  if (W > 1) {
    u64 mine = dev_shares ? 1 : 0; std::vector<u64> all((size_t)W, 0);
    xch.all_gather(&mine, 1, all.data());
    bool every = true; for (u64 v : all) every = every && v == 1;
    if (!every && dev_shares) { dev.sc_set_share_exchange(nullptr); dev_shares = false; }
  }
  auto one_round = [&](bool first, bool local) {
    dev.sc_round(tabs.data(), (int)tabs.size(), first ? nullptr : &ch, vp.terms.data(), (int)vp.terms.size(), raw.data());
    if (local && !dev_shares) gather_sum(); else total = raw;  // (device shares: sc_round already returned the sum over the ranks)
    std::vector<Ext> msg = sharded_message(vp, total);
    for (const Ext& e : msg) t.append_ext(e);
    out.proof.proofs.push_back(msg);
    ch = t.get_and_append_challenge("Internal round");
    out.proof.point.push_back(ch);
  };
  for (unsigned round = 0; round < nv_local; round++) one_round(round == 0, true);
  if (dev_shares) { dev.sc_set_share_exchange(nullptr); dev_shares = false; }  // stage 2 runs on every rank alike: nothing to exchange
  std::vector<Ext> fin(nt);
  dev.sc_finish(tabs.data(), (int)tabs.size(), ch, fin.data());
  out.finals.resize(nt);
  if (k == 0) { out.finals = fin; dev.release(mk); return out; }
  // stage 2 (merge_sumcheck_polys, util.rs:215-243): table j has one entry per worker, in worker order; k more rounds, the
  // same on every rank
  send.resize(2 * nt); recv.resize((size_t)W * 2 * nt);
  for (size_t j = 0; j < nt; j++) { send[2 * j] = fin[j].c0; send[2 * j + 1] = fin[j].c1; }
  xch.all_gather(send.data(), 2 * nt, recv.data());
  dev.release(mk);
  const size_t mk2 = dev.mark();
  std::vector<u64> words(2 * (size_t)W);
  for (size_t j = 0; j < nt; j++) {
    for (int g = 0; g < W; g++) { words[2 * g] = recv[((size_t)g * nt + j) * 2]; words[2 * g + 1] = recv[((size_t)g * nt + j) * 2 + 1]; }
    DBuf b = dev.alloc((size_t)W, true);
    dev.upload(b, words.data());
    tabs[j] = b;
  }
  for (unsigned round = 0; round < k; round++) one_round(round == 0, false);
  dev.sc_finish(tabs.data(), (int)tabs.size(), ch, out.finals.data());
  dev.release(mk2);
  return out;
}

}  // namespace dp
