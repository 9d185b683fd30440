// Host-side sponge for the fused protocol kernels (DP_HOST_SPONGE=1, hip_dev.hip): the WaveChallenger of a kernel stages the words it
// observes in a mapped request area and, when it needs a challenge, posts the request and polls a reply area; any host thread that is
// waiting for the device serves pending requests of ANY proof in flight (a cohort's members ask together: served one after the other
// by one thread they cost members x 4.5 us per Fiat-Shamir round, profiles/r02_mailbox_under_load.txt). Shared with the kernel code:
// the layout of both areas and the tags.
//   request area (u64 words):  [0] tag  [1] n = observed words  [2] samples consumed since the last reply  [3] want (1: prepare a sample)
//                              [4 ..] the n words;   tag = req_mix(seq) + sum_i (i + 1) * word_i + 3 n + 5 consumed + 7 want
//   reply area:                [0] tag  [1] out_len  [2..5] the sponge's output buffer;   tag = rep_mix(seq) + out_len + sum_i (i + 1) * out_i
// A sample on the device pops out[--out_len] like DuplexChallenger::sample (poseidon2.h Challenger).
#pragma once
#include "dev.h"
#include "poseidon2.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>

namespace dp {

constexpr unsigned WC_REQ_CAP = 1000;                 // observed words per request (the area holds 4 + WC_REQ_CAP words)
constexpr unsigned WC_REQ_WORDS = 4 + WC_REQ_CAP, WC_REP_WORDS = 8;
DP_HD unsigned long long wc_req_mix(unsigned long long s) { return s * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull; }
DP_HD unsigned long long wc_rep_mix(unsigned long long s) { return s * 0xA0761D6478BD642Full + 0xE7037ED1A0B428DBull; }

// one proof's side of the service: its request / reply areas (host views) and the transcript sponge the requests act on
struct alignas(128) SpongeSlot {
  std::atomic<int> active{0}, busy{0};
  volatile u64* req = nullptr; volatile u64* rep = nullptr;
  Challenger* ch = nullptr;
  unsigned long long served = 0;      // sequence number of the last request served
  unsigned long long last_tag = 0;    // its tag (a changed tag word is the cheap "something new" test)
  std::atomic<unsigned long long> nserved{0};
};
constexpr int SPONGE_MAX_SLOTS = 2048;
inline SpongeSlot* sponge_slots() { static SpongeSlot s[SPONGE_MAX_SLOTS]; return s; }
inline std::atomic<int>& sponge_nslots() { static std::atomic<int> n{0}; return n; }
inline std::atomic<int>& sponge_nactive() { static std::atomic<int> n{0}; return n; }
// slots of destroyed contexts are handed out again (a long-lived process creates and destroys thousands of contexts)
inline std::mutex& sponge_free_mu() { static std::mutex m; return m; }
inline std::vector<SpongeSlot*>& sponge_free_list() { static std::vector<SpongeSlot*> v; return v; }
inline SpongeSlot* sponge_slot_new() {
  { std::lock_guard<std::mutex> g(sponge_free_mu()); auto& fl = sponge_free_list(); if (!fl.empty()) { SpongeSlot* s = fl.back(); fl.pop_back(); return s; } }
  int i = sponge_nslots().fetch_add(1);
  if (i >= SPONGE_MAX_SLOTS) { sponge_nslots().fetch_sub(1); throw DpError(DP_ERR_OOM, "too many live device contexts for the host sponge service"); }
  return sponge_slots() + i;
}
// the context is gone: the slot is inactive (sponge_disarm_), nobody serves it; its sequence state stays, the next owner continues it
inline void sponge_slot_free(SpongeSlot* s) { if (!s) return; s->req = s->rep = nullptr; s->ch = nullptr; std::lock_guard<std::mutex> g(sponge_free_mu()); sponge_free_list().push_back(s); }

// DuplexChallenger as the device drives it: drop the samples the device has popped, absorb, and (want) make a sample available
inline int challenger_serve(Challenger& c, const u64* words, unsigned n, unsigned consumed, bool want, u64 out[4]) {
  c.out_len = consumed >= (unsigned)c.out_len ? 0 : c.out_len - (int)consumed;
  for (unsigned i = 0; i < n; i++) c.observe(words[i]);
  if (want && (c.in_len != 0 || c.out_len == 0)) c.duplexing();
  for (int i = 0; i < 4; i++) out[i] = c.out_buf[i];
  return c.out_len;
}
// serve the pending request of one slot, if there is a complete one and nobody else is at it. Returns true when it served.
inline bool sponge_serve_slot(SpongeSlot& s) {
  if (!s.active.load(std::memory_order_acquire)) return false;
  const unsigned long long tag = s.req[0];
  if (tag == s.last_tag) return false;
  int expect = 0;
  if (!s.busy.compare_exchange_strong(expect, 1, std::memory_order_acquire)) return false;
  bool done = false;
  if (s.active.load(std::memory_order_acquire) && s.req[0] == tag && tag != s.last_tag) {
    const u64 n = s.req[1], consumed = s.req[2], want = s.req[3];
    if (n <= WC_REQ_CAP && want <= 1 && consumed <= 4) {
      u64 words[WC_REQ_CAP];
      unsigned long long cs = 0;
      for (u64 i = 0; i < n; i++) { words[i] = s.req[4 + i]; cs += (unsigned long long)(i + 1) * words[i]; }
      if (tag == wc_req_mix(s.served + 1) + cs + 3ull * n + 5ull * consumed + 7ull * want) {  // complete and the next in sequence
        u64 out[4];
        const int ol = challenger_serve(*s.ch, words, (unsigned)n, (unsigned)consumed, want != 0, out);
        unsigned long long rs = (unsigned long long)ol;
        for (int i = 0; i < 4; i++) { s.rep[2 + i] = out[i]; rs += (unsigned long long)(i + 1) * out[i]; }
        s.rep[1] = (u64)ol;
        std::atomic_thread_fence(std::memory_order_release);
        s.rep[0] = wc_rep_mix(s.served + 1) + rs;
        s.served++; s.last_tag = tag; s.nserved.fetch_add(1, std::memory_order_relaxed);
        done = true;
      }
    }
  }
  s.busy.store(0, std::memory_order_release);
  return done;
}
// The service of the product: S server threads (DP_SPONGE_THREADS, default 6), server k owns the slots i = k mod S — no slot is looked
// at by two threads (every thread scanning every mailbox costs ~20 us per round trip in cache-line traffic alone,
// profiles/r02_mailbox_under_load.txt), and the members of a cohort, whose requests arrive together, sit in slots of different
// servers. The threads start with the first armed kernel, sleep while nothing is armed, and live until the process ends.
inline void sponge_server_loop(int id, int S) {
  SpongeSlot* s = sponge_slots();
  unsigned idle = 0;
  for (;;) {
    if (sponge_nactive().load(std::memory_order_relaxed) == 0) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }
    const int n = sponge_nslots().load(std::memory_order_acquire);
    bool any = false;
    for (int i = id; i < n; i += S) any |= sponge_serve_slot(s[i]);
    if (any) idle = 0; else if (++idle > 64) __builtin_ia32_pause();
  }
}
inline int sponge_server_count() { static const int S = [] { const char* e = getenv("DP_SPONGE_THREADS"); int v = e ? atoi(e) : 6; return v < 1 ? 1 : v > 64 ? 64 : v; }(); return S; }
inline void sponge_servers_start() {
  static std::once_flag once;
  std::call_once(once, [] { const int S = sponge_server_count(); for (int k = 0; k < S; k++) std::thread(sponge_server_loop, k, S).detach(); });
}
// one pass over every active slot by the calling thread (the kernel emulator of tests/ serves requests from inside the kernel's poll)
inline void sponge_serve_all() {
  if (sponge_nactive().load(std::memory_order_relaxed) == 0) return;
  const int n = sponge_nslots().load(std::memory_order_acquire);
  SpongeSlot* s = sponge_slots();
  for (int i = 0; i < n; i++) sponge_serve_slot(s[i]);
}

}  // namespace dp
