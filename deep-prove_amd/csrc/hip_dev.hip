// HipDev: the MI355X (gfx950 / CDNA4) implementation of dp::Dev — hand-written HIP kernels for every O(n) loop of
// the deep-prove sumcheck / logup-GKR / Basefold hot path. All arithmetic is 64-bit modular
// integer work over Goldilocks: no MFMA anywhere; the kernels are HBM-streaming (K1-K7, K9-K13) or VALU-integer bound
// (K8 Poseidon2). Wave = 64 lanes, blocks of 256 threads, grids sized to >= a few waves per SIMD on 256 CUs.
//
// Layout in HBM: a table of n field elements is a dense array in natural (little-endian index) order; base elements
// are canonical u64, extension elements are 16-byte {c0,c1} pairs so one `global_load_dwordx4` per lane fetches one
// element. A sumcheck pair (2b, 2b+1) is therefore 32 contiguous bytes per lane.
#include "dev.h"
#include "poseidon2_fast.h"
#include "gl64_lazy.h"
#include "sumcheck.h"
#include "fiber.h"
#include "logup_tail.h"
#include "axpy_many.h"
#include "classic_tail.h"
#include "dense_tail.h"
#include "eqsum_tail.h"
#include "deleg_tail.h"
#include "commit_tail.h"
#include "sponge_host.h"
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <deque>
#include <type_traits>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>
#include <string>

namespace dp {

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { if (e_ == hipErrorOutOfMemory) (void)hipGetLastError(); throw DpError(e_ == hipErrorOutOfMemory ? DP_ERR_OOM : DP_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

#include "kernels.inc"


// ================================================================================================ HipDev
// Grid size for grid-stride kernels: `cap` bounds a launch of ONE proof (a merged cohort launch has gridDim.z = members times as many workgroups).
// A lower global cap was swept at 448 proofs in flight (8 / 24 / 64 / 256 workgroups per proof and launch against up to 4096): 470 / 513 / 566 / 491
// against 531-534 proofs/s, i.e. nothing outside the +-5 % between runs (profiles/r04_grid_cap_sweep.txt); there is no knob for it any more.
static inline int grid_for_uncapped(size_t n, int cap = 2048) {
  size_t b = (n + TPB - 1) / TPB;
  if (b < 1) b = 1;
  return (int)std::min<size_t>(b, cap);
}

struct ProfRec { const char* name; double bytes; hipEvent_t a, b; };
// every launch of this file goes through HipDev::launch_: kg<Body> on the context's own stream, or — when the context is a
// member of a cohort — an argument pack handed to the cohort, which launches kc<Body> once for all its members
#define DPL_B(kern, maxt, flags, grid, block, lds, ...) do { prof_begin(#kern); LaunchTimer lt_(this); launch_<kern, maxt, flags>(KArgs<decltype(&kern)>(), #kern, grid, block, lds, __VA_ARGS__); lt_.stop(); prof_end(); } while (0)
#define DPL(kern, grid, block, ...) DPL_B(kern, 1024, KF_NONE, grid, block, 0, __VA_ARGS__)
#define DPL_LDS(kern, grid, block, lds, ...) DPL_B(kern, 1024, KF_NONE, grid, block, lds, __VA_ARGS__)
// the bulk hash layers (kernels.inc DP_HASH_VGPRS: a build that makes their waves own a tail-sized share of the register file)
#if DP_HASH_VGPRS
#define DPL_HASH(kern, grid, block, ...) DPL_B(kern, 256, KF_HASH, grid, block, 0, __VA_ARGS__)
#else
#define DPL_HASH(kern, grid, block, ...) DPL(kern, grid, block, __VA_ARGS__)
#endif
// The throughput-mode form of a one-workgroup body runs at most SHARED_MAXT threads and is COMPILED for that: __launch_bounds__(256) leaves the
// register allocator 512 VGPRs per lane instead of the 128 of a 1024-thread workgroup — under __launch_bounds__(1024) k_logup_tail spilled 54 VGPRs and
// k_sc_persist 75 into scratch, on the dependent chain that bounds every tail (round-3 review; profiles/r04_kernel_resources_gfx950.csv).
constexpr int SHARED_MAXT = 256;
// The latency-mode (whole-CU) form of the same bodies: LAT_MAXT threads, and COMPILED for that many. Round 4 launched and compiled it for 1024 threads — 128 VGPRs
// per lane: k_sc_persist_lds<false> spilled 59 VGPRs, the claim forms of the tails 23 .. 104 (profiles/r04_kernel_resources_gfx950.csv) — and a Fiat-Shamir round
// of a single proof spent 14 of its 30 us in the fold and the round sums of kilobyte tables (DP_TIMING=2, tools/r05/call12.sh): scratch reloads on the dependent chain.
#ifndef DP_LAT_MAXT
#define DP_LAT_MAXT 512
#endif
constexpr int LAT_MAXT = DP_LAT_MAXT;
// one-workgroup-per-proof kernels (persistent sumchecks, fused protocol tails, Merkle tails): `threads` and the CU reservation
// apply in latency mode; in throughput mode the workgroup shrinks to shared_threads_ and reserves nothing (KF_PRIO above).
// `lds` = dynamic LDS the body really needs.
#define DPL_ONE(kern, grid, threads, lds, ...) do { if (shared_now()) { DPL_B(kern, SHARED_MAXT, KF_PRIO, grid, dim3(std::min<unsigned>((unsigned)(threads), (unsigned)shared_threads_)), (size_t)(lds), __VA_ARGS__); } \
                                                     else { DPL_B(kern, LAT_MAXT, KF_CLAIM, grid, dim3(std::min<unsigned>((unsigned)(threads), (unsigned)LAT_MAXT)), std::max<size_t>((size_t)(lds), excl_now()), __VA_ARGS__); } } while (0)
// ... and a WIDE throughput-mode form for the few one-workgroup launches whose table passes dominate (a lookup over 2^13 .. 2^16 rows: 3.8 ms of table passes
// against 0.3 ms for a 2^10-row column at 256 threads, profiles/r05_wgphases_448_summary.txt): WIDE_MAXT threads, still no reservation, the sponge wave unchanged
constexpr int WIDE_MAXT = 512;
#define DPL_ONE_W(kern, wide, grid, threads, lds, ...) do { if ((wide) && shared_now()) { DPL_B(kern, WIDE_MAXT, KF_PRIO, grid, dim3(std::min<unsigned>((unsigned)(threads), (unsigned)WIDE_MAXT)), (size_t)(lds), __VA_ARGS__); } \
                                                           else { DPL_ONE(kern, grid, threads, lds, __VA_ARGS__); } } while (0)
#define DPL_ONE_HI(kern, hi, grid, threads, lds, ...) do { if (hi) { DPL_ONE((kern<true>), grid, threads, lds, __VA_ARGS__); } else { DPL_ONE((kern<false>), grid, threads, lds, __VA_ARGS__); } } while (0)
#define DPL_HI(kern, hi, grid, block, ...) do { if (hi) { DPL((kern<true>), grid, block, __VA_ARGS__); } else { DPL((kern<false>), grid, block, __VA_ARGS__); } } while (0)
#define DPL_LDS_HI(kern, hi, grid, block, lds, ...) do { if (hi) { DPL_LDS((kern<true>), grid, block, lds, __VA_ARGS__); } else { DPL_LDS((kern<false>), grid, block, lds, __VA_ARGS__); } } while (0)
#define DP_SET_LDS(kern, maxt, bytes) set_lds_<kern, maxt, KF_NONE>(KArgs<decltype(&kern)>(), (int)(bytes))
#define DP_SET_LDS_ONE(kern, maxt, bytes) do { set_lds_<kern, LAT_MAXT, KF_CLAIM>(KArgs<decltype(&kern)>(), (int)(bytes)); set_lds_<kern, SHARED_MAXT, KF_PRIO>(KArgs<decltype(&kern)>(), (int)(bytes)); } while (0)

// Host memory that the HOST writes before a launch and kernels only READ — the cohorts' argument-pack rings, the descriptor ring — is mapped NON-coherent
// (coarse-grained): the GPU may keep its lines in L2 for the length of a kernel and drops them at the next kernel's system-scope acquire, so a descriptor costs one
// PCIe read per kernel and XCD instead of one per workgroup that looks at it (k_classic_fused / k_axpy_many / k_eq_table_many walk descriptor arrays in every
// workgroup). Memory a kernel WRITES for the host to poll (results, flags, mailboxes, the download staging) stays coherent. DP_HOST_NC=0: everything coherent, as before.
static const bool g_host_nc = !(getenv("DP_HOST_NC") && !atoi(getenv("DP_HOST_NC")));
static inline unsigned host_ro_flags() { return hipHostMallocMapped | (g_host_nc ? hipHostMallocNonCoherent : hipHostMallocCoherent); }
static const int g_timing_level = getenv("DP_TIMING") ? atoi(getenv("DP_TIMING")) : 0;  // 1: host / cohort accounting on stderr; 2: also launches by kernel, long host stretches, device cycle counters of the persistent sumcheck
static const bool g_host_stats = g_timing_level > 0;
// DP_WAIT_YIELD=1: a host thread that waits for the device outside a fiber gives its CPU away (sched_yield) instead of spinning — for
// seam-level hosts that run more proving threads than they have cores (tests/support/seam_bench.c)
static const bool g_wait_yield = getenv("DP_WAIT_YIELD") && atoi(getenv("DP_WAIT_YIELD"));
static inline void dp_spin_pause() { if (g_wait_yield) sched_yield(); else __builtin_ia32_pause(); }

// ------------------------------------------------------------------------------------------------ cohorts
// A cohort is a set of proofs of the SAME model proved in lock step on one stream by one host thread (each proof a fiber,
// fiber.h). Proofs of one model issue the same sequence of launches with the same shapes — only pointers and challenges
// differ — so launch number i of every member is merged into ONE kc<Body> launch with gridDim.z = members: the per-launch
// costs of the command processor (dispatch, barrier, end-of-kernel cache maintenance — what bounds a GPU that serves two
// dozen independent streams of tiny kernels) are paid once per cohort step instead of once per proof step.
// A member never blocks at a launch: it drops its argument pack and goes on to its next wait (where it yields to the next
// member); whoever completes a launch's set of packs fires it. Everything is driven from the cohort's one host thread.
struct Cohort {
  struct Pending {
    void (*fire)(const Pending&, hipStream_t);  // also the identity of the kernel (one instantiation per Body)
    const char* name;
    dim3 g, b; size_t lds, pack_bytes;
    char* packs; const char* packs_dev;
    int count, expected;
    size_t ring_begin, ring_end;
  };
  hipStream_t s = nullptr;
  int members = 0;  // proofs currently in the cohort
  char* ring = nullptr; const char* ring_dev = nullptr;
  size_t ring_cap = 0, ring_off = 0;
  std::deque<Pending> q; size_t q_base = 0;            // q[i] = launch number q_base + i of the common sequence, not yet fired
  std::deque<std::pair<size_t, size_t>> inflight;      // (launch number, ring_begin) of fired launches not known to have run
  size_t executed = 0;                                 // every launch number < executed has run to completion
  size_t nfired = 0, npacks = 0;
  // DP_TIMING: where a cohort's wall time goes — "device phase" = from a fire to the first member that sees a result afterwards,
  // "host phase" = from that wake-up to the next fire (the members digest the result one after the other on the cohort's
  // thread; nothing of this cohort is queued on the GPU meanwhile)
  std::chrono::steady_clock::time_point t_fire_{}, t_wake_{}; bool awake_ = false, have_fire_ = false;
  double dev_phase_us = 0, host_phase_us = 0; size_t nwakes = 0;
  void note_wake() {
    if (awake_ || !have_fire_) return;
    awake_ = true; t_wake_ = std::chrono::steady_clock::now(); nwakes++;
    const double d = std::chrono::duration<double, std::micro>(t_wake_ - t_fire_).count();
    dev_phase_us += d;
    if (g_timing_level > 2 && last_fired_) { auto& e = dev_by_[last_fired_]; e.first += d; e.second++; }
  }
  // (DP_TIMING=3) host and device phases by the launch that ENDS a host phase / is waited for: where in the proof a cohort's queue stands empty
  std::map<const char*, std::pair<double, size_t>> host_by_, dev_by_; const char* last_fired_ = nullptr;
  void note_fire(const char* name = nullptr) {
    auto t = std::chrono::steady_clock::now();
    if (awake_) { const double h = std::chrono::duration<double, std::micro>(t - t_wake_).count(); host_phase_us += h; awake_ = false; if (g_timing_level > 2 && name) { auto& e = host_by_[name]; e.first += h; e.second++; } }
    t_fire_ = t; have_fire_ = true; last_fired_ = name;
  }

  // `share`: run on the stream of another cohort (which must outlive this one: dp_model_prove_batch keeps the cohorts of a model together). Two cohorts on one
  // stream take turns on ONE hardware queue: while the members of one digest a result on the host — 21 members x ~65 us on one thread, the queue idle — the
  // other's launch runs (DP_STREAM_SHARE)
  bool owns_stream = true;
  explicit Cohort(size_t ring_bytes = size_t(32) << 20, const Cohort* share = nullptr) : ring_cap(ring_bytes) {
    if (share) { s = share->s; owns_stream = false; }
    else HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_CHECK(hipHostMalloc((void**)&ring, ring_cap, host_ro_flags()));
    HIP_CHECK(hipHostGetDevicePointer((void**)&ring_dev, ring, 0));
  }
  ~Cohort() { if (s) { hipStreamSynchronize(s); if (owns_stream) hipStreamDestroy(s); } if (ring) hipHostFree(ring); }
  Cohort(const Cohort&) = delete;
  Cohort& operator=(const Cohort&) = delete;

  // ring space for `bytes` (virtual offsets grow monotonically; physical = virtual mod capacity, regions never straddle
  // the end), never overlapping a region a launch may still read: those of unfired launches and of fired launches not yet
  // known to have run
  size_t ring_take(size_t bytes) {
    bytes = (bytes + 63) & ~size_t(63);
    while (!inflight.empty() && inflight.front().first < executed) inflight.pop_front();
    size_t head = ring_off;
    if (head % ring_cap + bytes > ring_cap) head += ring_cap - head % ring_cap;
    size_t tail = !inflight.empty() ? inflight.front().second : !q.empty() ? q.front().ring_begin : head;
    if (bytes > ring_cap || head + bytes - tail > ring_cap) throw DpError(DP_ERR_OOM, "cohort argument ring exhausted (DP_COHORT_RING_BYTES)");
    ring_off = head + bytes;
    return head;
  }
  // member `li`-th launch of its sequence: add its pack to launch number `li`, fire every launch whose set is complete
  void submit(size_t li, void (*fire)(const Pending&, hipStream_t), const char* name, dim3 g, dim3 b, size_t lds, const void* pack, size_t pack_bytes) {
    if (li < q_base) throw DpError(DP_ERR_SHAPE, std::string("cohort out of step: a member reached launch ") + name + " after it was fired (the proofs of a cohort must issue identical launch sequences)");
    size_t k = li - q_base;
    if (k > q.size()) throw DpError(DP_ERR_SHAPE, "cohort out of step: launch sequence gap");
    if (k == q.size()) {
      Pending p; p.fire = fire; p.name = name; p.g = g; p.b = b; p.lds = lds; p.pack_bytes = pack_bytes; p.count = 0; p.expected = members;
      p.ring_begin = ring_take(pack_bytes * (size_t)members); p.ring_end = ring_off;
      p.packs = ring + p.ring_begin % ring_cap; p.packs_dev = ring_dev + p.ring_begin % ring_cap;
      q.push_back(p);
    }
    Pending& p = q[k];
    if (p.fire != fire || p.g.x != g.x || p.g.y != g.y || p.b.x != b.x || p.lds != lds || p.pack_bytes != pack_bytes)
      throw DpError(DP_ERR_SHAPE, std::string("cohort out of step: launch ") + name + " of one member meets " + p.name + " of another (the proofs of a cohort must issue identical launch sequences)");
    if (p.count >= p.expected) throw DpError(DP_ERR_SHAPE, "cohort out of step: too many packs for one launch");
    memcpy(p.packs + (size_t)p.count * pack_bytes, pack, pack_bytes);
    p.count++; npacks++;
    flush();
  }
  void flush() {
    while (!q.empty() && q.front().count >= q.front().expected) {
      Pending& p = q.front();
      if (p.count > 0) {
        std::atomic_thread_fence(std::memory_order_release);
        if (g_host_stats) note_fire(p.name);
        p.fire(p, s);
        inflight.push_back({q_base, p.ring_begin});
        nfired++;
      }
      q.pop_front(); q_base++;
    }
  }
  // a member has seen the result of its launch number `li` (or of a later round of it): everything before it has run
  void note_executed(size_t li) { if (li > executed) executed = li; }
  int nominal = 0;  // members when the batch started (every member joins before the first launch): what launch shapes may depend on — `members` shrinks while a batch drains
  void join() { if (!q.empty()) throw DpError(DP_ERR_SHAPE, "a proof cannot join a cohort in the middle of a step"); members++; nominal = members; }
  // a member leaves (its proofs are done, or it failed) having issued `li` launches: later launches no longer wait for it
  void leave(size_t li) {
    members--;
    for (size_t k = li > q_base ? li - q_base : 0; k < q.size(); k++) q[k].expected--;
    flush();
  }
  void drain() { HIP_CHECK(hipStreamSynchronize(s)); inflight.clear(); executed = q_base; }
};


class HipDev : public Dev {
  // host-side cost accounting (DP_TIMING=1): time inside hipLaunchKernel and number of launches / device waits
  double launch_us_ = 0; size_t nlaunch_ = 0, nwait_ = 0, nyield_ = 0;
  // host time between two device waits (the proof's own host work: no yield happens there) and time from the first poll
  // of a wait to its success (device latency + the other fibers of this thread)
  struct Chunk { size_t wait; double us; const char* first; const char* last; };
  std::vector<Chunk> chunks_; const char* first_launch_ = nullptr; const char* last_launch_ = nullptr;
  std::map<std::pair<const char*, const char*>, std::pair<double, size_t>> chunk_by_;  // DP_TIMING=3: host work before a wait by (first, last) launch issued in it
  double work_us_ = 0, waitlat_us_ = 0; std::chrono::steady_clock::time_point last_exit_{}; bool have_exit_ = false;
  std::chrono::steady_clock::time_point wait_enter_() {
    auto t = std::chrono::steady_clock::now();
    if (g_host_stats && have_exit_) {
      double c = std::chrono::duration<double, std::micro>(t - last_exit_).count();
      work_us_ += c;
      if (c > 150.0 && chunks_.size() < 400) chunks_.push_back({nwait_, c, first_launch_, last_launch_});
      if (g_timing_level > 2) { auto& e = chunk_by_[{first_launch_, last_launch_}]; e.first += c; e.second++; }
    }
    first_launch_ = nullptr;
    return t;
  }
  void wait_exit_(std::chrono::steady_clock::time_point t0) {
    fiber_note_progress();
    if (!g_host_stats) return;
    last_exit_ = std::chrono::steady_clock::now(); have_exit_ = true;
    waitlat_us_ += std::chrono::duration<double, std::micro>(last_exit_ - t0).count();
    if (co_) co_->note_wake();
  }
  std::map<const char*, size_t> by_name_;  // (keys: the string literals of the launch macros)
  struct LaunchTimer {
    HipDev* d; std::chrono::steady_clock::time_point t0;
    explicit LaunchTimer(HipDev* d_) : d(d_) { if (g_host_stats) t0 = std::chrono::steady_clock::now(); }
    void stop() { if (g_host_stats) { d->launch_us_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); d->nlaunch_++; } }
  };
  int device_;
  bool prof_ = false;
  double nb_ = 0;  // algorithmic bytes of the next launch (SURVEY.md 8d ledger), consumed by prof_begin
  std::vector<ProfRec> recs_;
  std::vector<hipEvent_t> prof_pool_;  // events of earlier profiled runs, reused: creating two events per launch between the launches is what a cold profiled run paid for
  hipEvent_t prof_event_() { hipEvent_t e; if (!prof_pool_.empty()) { e = prof_pool_.back(); prof_pool_.pop_back(); return e; } hipEventCreate(&e); return e; }
  void prof_begin(const char* name) {
    if (!prof_) { nb_ = 0; return; }
    ProfRec r; r.name = name; r.bytes = nb_; nb_ = 0;
    r.a = prof_event_(); r.b = prof_event_();
    hipEventRecord(r.a, s_);
    recs_.push_back(r);
  }
  void prof_end() { if (prof_) hipEventRecord(recs_.back().b, s_); }

  hipStream_t s_ = nullptr;
  Cohort* co_ = nullptr;  // non-null while this context proves as a member of a cohort: launches go to the cohort's stream
  bool queued_() const { return co_ != nullptr; }  // launches are packs handed to someone else: data moves with kernels, never with stream commands
  size_t co_li_ = 0;      // number of launches this member has issued into the cohort's common sequence
  template <auto Body, int MAXT, int FLAGS, class... A>
  static void fire_(const Cohort::Pending& p, hipStream_t s) {
    hipLaunchKernelGGL((kc<Body, MAXT, FLAGS, std::decay_t<A>...>), dim3(p.g.x, p.g.y, (unsigned)p.count), p.b, p.lds, s, (const ArgPack<std::decay_t<A>...>*)p.packs_dev);
  }
  template <auto Body, int MAXT, int FLAGS, class... A, class... P>
  void launch_(KArgs<void (*)(A...)>, const char* name, dim3 g, dim3 b, size_t lds, P... args) {
    static_assert(sizeof...(A) == sizeof...(P), "kernel argument count");
    if (g_host_stats) { if (!first_launch_) first_launch_ = name; last_launch_ = name; by_name_[name]++; }
    if (!co_) { hipLaunchKernelGGL((kg<Body, MAXT, FLAGS, std::decay_t<A>...>), g, b, lds, s_, static_cast<std::decay_t<A>>(args)...); return; }
    DP_REQUIRE(g.z == 1, DP_ERR_SHAPE, "cohort launches use blockIdx.z for the proof");
    using Pack = ArgPack<std::decay_t<A>...>;
    static_assert(std::is_trivially_copyable<Pack>::value && std::is_trivially_destructible<Pack>::value, "argument packs travel as bytes");
    Pack pk(static_cast<std::decay_t<A>>(args)...);
    co_->submit(co_li_++, &fire_<Body, MAXT, FLAGS, A...>, name, g, b, lds, &pk, sizeof(Pack));
  }
  template <auto Body, int MAXT, int FLAGS, class... A>
  static void set_lds_(KArgs<void (*)(A...)>, int bytes) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kg<Body, MAXT, FLAGS, std::decay_t<A>...>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)kc<Body, MAXT, FLAGS, std::decay_t<A>...>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  }
  char* arena_ = nullptr;
  size_t arena_cap_ = 0, arena_off_ = 0, arena_peak_ = 0;
  u64* hres_ = nullptr;   // pinned, device-mapped host memory for small results (zero-copy readback)
  u64* hres_dev_ = nullptr;  // device view of hres_
  unsigned long long* hflag_ = nullptr;      // host view of the publish sequence number
  unsigned long long* hflag_dev_ = nullptr;  // device view
  unsigned long long seq_ = 0;
  unsigned long long last_tag_ = 0;
  bool zerocopy_ = true;  // DP_NO_ZEROCOPY=1 falls back to hipMemcpyAsync + hipStreamSynchronize
  bool persist_ = true;   // DP_NO_PERSIST=1 disables the persistent sumcheck kernel
  size_t excl_ = 0;       // dynamic LDS requested by one-workgroup kernels to keep a CU to themselves (latency mode)
  size_t excl_now() const { return excl_; }
  // throughput mode (several proofs in flight): one-workgroup kernels reserve nothing and run as 256-thread workgroups with
  // raised wave priority (KF_PRIO); round 1's whole-CU workgroups in throughput mode lost by 2x (tools/hol.hip) and are gone.
  static constexpr bool shared_tails_ = true;
  static constexpr int shared_threads_ = SHARED_MAXT;
  bool throughput_mode_ = false;
  bool shared_now() const { return throughput_mode_ && shared_tails_; }
  int persist_threads(size_t work) const {
    return excl_now() ? 1024 : work >= 2048 ? 1024 : work >= 512 ? 512 : 256;  // exclusive CU: always the full 16 waves
  }
  unsigned long long* scdbg_ = nullptr;  // DP_TIMING=2: device cycle counters of the persistent sumcheck kernel
  unsigned long long* hmail_ = nullptr;      // host view of the challenge mailbox [seq, c0, c1]
  unsigned long long* hmail_dev_ = nullptr;  // device view
  unsigned long long* vmail_ = nullptr;      // the mailbox in DEVICE memory the CPU writes through the PCIe BAR (mailbox_dev below)
  bool vmail_tried_ = false;
  struct ScSession { bool active = false; int ntabs = 0; size_t n = 0; unsigned long long seq = 0; std::vector<Ext*> a, b; bool nextA = true;
                     bool multi = false; int G = 0, rounds_a = 0, folds = 0, shift = 0; size_t slot_words = 0, n0 = 0; } sess_;
  static constexpr int MULTI_MAX_WG = 32;
  static constexpr size_t MULTI_MIN_N = 4096, MULTI_MAX_N = size_t(1) << 18, MULTI_TARGET_N = 1024;
  unsigned long long* hmflag_ = nullptr;      // host view of the per-workgroup flags
  unsigned long long* hmflag_dev_ = nullptr;  // device view
  unsigned long long last_tag_multi_[MULTI_MAX_WG];
  bool multi_ = true;  // the multi-workgroup phase of large sumchecks (latency mode only)
  static bool persist_flag_env(const char* name) { const char* e = getenv(name); return !(e && atoi(e)); }
  // all G workgroups have published round `seq`: every slot's tag matches its payload (same protocol as wait_flag)
  void wait_flags_multi(unsigned long long seq, size_t nwords, int G, size_t slot_words) {
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    int done = 0;
    while (done < G) {
      volatile unsigned long long* f = hmflag_ + done;
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent sumcheck (no challenge received)");
      if (tag != last_tag_multi_[done]) {
        std::atomic_thread_fence(std::memory_order_acquire);
        volatile u64* w = hres_ + (size_t)done * slot_words;
        unsigned long long cs = 0;
        for (size_t i = 0; i < nwords; i++) cs += (unsigned long long)(i + 1) * w[i];
        if (base + cs == tag) { last_tag_multi_[done] = tag; done++; continue; }
      }
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else dp_spin_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, std::string("timeout waiting for the device"));
    }
    desc_off_ = 0; stage_off_ = 0;
    if (co_ && co_li_) co_->note_executed(co_li_ - 1);
    wait_exit_(t0);
  }
  u64* dres_ = nullptr;   // device result buffer
  unsigned* fused_ticket_ = nullptr;  // "last workgroup" ticket of k_sc_fused (device, zero between launches)
  void* hstage_ = nullptr;  // pinned + device-mapped staging of bulk copies (from DESC_BYTES on; coherent: k_download's chunk tags are polled by the host)
  char* hstage_dev_ = nullptr;
  void* hdesc_ = nullptr;   // descriptor ring: written by the host before a launch, read by its kernels (host_ro_flags: non-coherent, cacheable on the GPU)
  char* hdesc_dev_ = nullptr;
  size_t desc_off_ = 0;
  // Asynchronous uploads (throughput mode; DP_ASYNC_UPLOAD=0 turns them off, =1 forces them for single proofs too): small
  // host-to-device copies take successive slots of the bulk staging area and are not waited for — like descriptors, the slots
  // are recycled when the host observes a later publication of this stream (everything launched before it has run). Otherwise
  // every upload costs a copy launch, a publish launch and a device wait (~29 per Dense-4M proof). The slot arithmetic uses
  // ASYNC_STAGE, not the context's own staging size: the members of a cohort (the model's context has a larger staging buffer
  // than the batch workers) must take the same decisions, or the cohort falls out of step.
  size_t stage_off_ = 0;
  static constexpr int async_upload_env_ = -1;  // (on in throughput mode, off for a single proof)
  bool async_upload_now() const { return async_upload_env_ < 0 ? throughput_mode_ : async_upload_env_ != 0; }
  static constexpr size_t ASYNC_STAGE = size_t(12) << 20, ASYNC_MAX = size_t(2) << 20;
  static constexpr size_t RES_WORDS = 1 << 16;
  size_t STAGE_BYTES = size_t(64) << 20;  // bulk staging of this context (workers of a batch get less: stage_bytes of the constructor)
  static constexpr size_t DESC_BYTES = 4 << 20;
  unsigned L_ = 0;  // full_message_size_log of the current PCS parameters
  u64* tw_ = nullptr;    // tw[i]   = w_{2^(L+1)}^i, i < 2^L   (all FFT root tables of rs.rs:31-68 in one array)
  u64* pow7_ = nullptr;  // pow7[i] = 7^i,          i < 2^L   (coset shifts)
  // The two tables are read-only after pcs_init: the workers of a model borrow the owning context's pair (pcs_share) — 2 x 8 B x 2^L
  // per worker otherwise (268 MB at L = 24), and one copy stays hot in L2 for every proof in flight instead of one copy per proof.
  struct PcsTables { u64* tw = nullptr; u64* pow7 = nullptr; int device = 0; ~PcsTables() { hipSetDevice(device); if (tw) hipFree(tw); if (pow7) hipFree(pow7); } };
  std::shared_ptr<PcsTables> pcs_tabs_;
  std::string name_;

  void* arena_alloc(size_t bytes) {
    size_t off = (arena_off_ + 255) & ~size_t(255);
    if (off + bytes > arena_cap_) throw DpError(DP_ERR_OOM, "device arena exhausted (raise DP_ARENA_BYTES / DP_WORKER_ARENA_BYTES)");
    arena_off_ = off + bytes;
    if (arena_off_ > arena_peak_) arena_peak_ = arena_off_;
    return arena_ + off;
  }
  static unsigned long long pub_mix(unsigned long long seq) { return seq * 0x9E3779B97F4A7C15ull + 0x7F4A7C159E3779B9ull; }
  // spin (bounded) until the message with sequence number `seq` and `nwords` payload words has fully landed in host
  // memory: the tag word must equal mix(seq) + sum (i+1)*word_i recomputed from what we read
  void wait_flag(unsigned long long seq, size_t nwords) {
    volatile unsigned long long* f = hflag_;
    volatile u64* w = hres_;
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    for (;;) {
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent sumcheck (no challenge received)");
      if (tag != last_tag_) {  // something new was written: check it against the payload
        std::atomic_thread_fence(std::memory_order_acquire);
        unsigned long long cs = 0;
        for (size_t i = 0; i < nwords; i++) cs += (unsigned long long)(i + 1) * w[i];
        if (base + cs == tag) { last_tag_ = tag; desc_off_ = 0; stage_off_ = 0; if (co_ && co_li_) co_->note_executed(co_li_ - 1); wait_exit_(t0); return; }
      }
      // inside a fiber the wait hands the host thread to the next proof in flight (fiber.h); otherwise spin
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else dp_spin_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, std::string("timeout waiting for the device"));
    }
  }
  // wait_flag for a message made of blocks whose checksum runs over block-relative word indices (k_logup_tail)
  void wait_flag_blocks(unsigned long long seq, const std::vector<size_t>& block_words) {
    volatile unsigned long long* f = hflag_;
    volatile u64* w = hres_;
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    for (;;) {
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent kernel");
      if (tag != last_tag_) {
        std::atomic_thread_fence(std::memory_order_acquire);
        const unsigned long long cs = logup_tail_checksum(w, block_words);
        if (base + cs == tag) { last_tag_ = tag; desc_off_ = 0; stage_off_ = 0; if (co_ && co_li_) co_->note_executed(co_li_ - 1); wait_exit_(t0); return; }
      }
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else dp_spin_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, std::string("timeout waiting for the device"));
    }
  }
  // descriptors for batched kernels: written by the host into the mapped ring and read by the kernel directly (no
  // H2D copy launch). The ring is recycled whenever the host has observed a publication, i.e. the stream is drained.
  template <class T> T* desc_alloc(size_t count, const T** dev_view) {
    size_t bytes = (count * sizeof(T) + 63) & ~size_t(63);
    DP_REQUIRE(bytes <= DESC_BYTES, DP_ERR_SHAPE, "descriptor batch too large");
    if (desc_off_ + bytes > DESC_BYTES) { stream_wait(); }
    T* h = (T*)((char*)hdesc_ + desc_off_);
    *dev_view = (const T*)(hdesc_dev_ + desc_off_);
    desc_off_ += bytes;
    return h;
  }
  char* bulk_stage() { return (char*)hstage_ + DESC_BYTES; }
  // wait until everything queued on the stream so far has executed, without entering hipStreamSynchronize (which
  // serialises against other host threads driving other proofs on the same GPU): a one-wave kernel posts a tag
  void stream_wait() {
    if (!zerocopy_) { HIP_CHECK(hipStreamSynchronize(s_)); desc_off_ = 0; stage_off_ = 0; return; }
    unsigned long long seq = ++seq_;
    nb_ = 0; DPL(k_publish, dim3(1), dim3(64), (const u64*)dres_, hres_dev_, (size_t)0, hflag_dev_, seq);
    wait_flag(seq, 0);
  }
  // second stage of a block-partial reduction, published straight to hres_ (see k_reduce_publish); nout <= 1024
  // ---- sharded sumcheck with the shares on the device (Dev::ShareExchange)
  ShareExchange* share_x_ = nullptr; u64* dshare_ = nullptr; u64* dgather_ = nullptr; size_t dgather_words_ = 0;
  bool sc_set_share_exchange(ShareExchange* x) override {
    if (x && (queued_() || !zerocopy_)) return false;
    if (x) {
      if (!dshare_) HIP_CHECK(hipMalloc((void**)&dshare_, (RES_WORDS + 8) * 8));
      const size_t need = RES_WORDS * (size_t)x->world();
      if (dgather_words_ < need) { if (dgather_) { HIP_CHECK(hipStreamSynchronize(s_)); HIP_CHECK(hipFree(dgather_)); } HIP_CHECK(hipMalloc((void**)&dgather_, need * 8)); dgather_words_ = need; }
    }
    share_x_ = x;
    return true;
  }
  // this rank's `nwords` raw words sit at dshare_: gather the ranks' words, add them mod p, bring the total to hres_ (one wait)
  void share_exchange_(size_t nwords) {
    DP_REQUIRE(nwords % 2 == 0 && nwords <= RES_WORDS && nwords / 2 <= 1024, DP_ERR_SHAPE, "sharded sumcheck: round message too large");
    share_x_->all_gather_device(dshare_, nwords, dgather_, (void*)s_);
    unsigned long long seq = ++seq_;
    nb_ = 8.0 * (double)nwords * share_x_->world(); DPL(k_shares_sum_publish, dim3(1), dim3(256), (const u64*)dgather_, share_x_->world(), (int)(nwords / 2), (Ext*)hres_dev_, hflag_dev_, seq);
    wait_flag(seq, nwords);
  }
  void reduce_publish(const Ext* partial, size_t nblocks, size_t inner, int nout) {
    DP_REQUIRE(nout >= 1 && nout <= 1024 && (size_t)nout * 2 <= RES_WORDS, DP_ERR_SHAPE, "reduce_publish: too many outputs");
    if (share_x_) {  // the reduction lands in device memory (the tag word too: nobody reads it), the exchange brings the ranks' total to hres_
      int threads = nout >= 8 ? 1024 : nout >= 4 ? 256 : 64 * nout;
      DPL(k_reduce_publish, dim3(1), dim3(threads), partial, nblocks, inner, nout, (Ext*)dshare_, (unsigned long long*)(dshare_ + RES_WORDS), 0ull);
      share_exchange_((size_t)nout * 2);
      return;
    }
    if (!zerocopy_) {
      if (inner == 4 && nout % 4 == 0 && nout > 4) DPL(k_reduce_terms, dim3(nout), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else if (inner == 2) DPL(k_reduce_pairs, dim3(nout), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else if (inner == 4 && nout <= 4) DPL(k_reduce_terms, dim3(4), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else DPL(k_reduce_partials, dim3(nout), dim3(TPB), partial, nblocks, inner, (Ext*)dres_);
      fetch((size_t)nout * 2);
      return;
    }
    unsigned long long seq = ++seq_;
    int threads = nout >= 8 ? 1024 : nout >= 4 ? 256 : 64 * nout;
    DPL(k_reduce_publish, dim3(1), dim3(threads), partial, nblocks, inner, nout, (Ext*)hres_dev_, hflag_dev_, seq);
    wait_flag(seq, (size_t)nout * 2);
  }
  // bring `nwords` of dres_ to hres_
  void fetch(size_t nwords) {
    DP_REQUIRE(nwords <= RES_WORDS, DP_ERR_ARG, "result too large");
    if (zerocopy_) {
      unsigned long long seq = ++seq_;
      nb_ = 0; DPL(k_publish, dim3(1), dim3(64), (const u64*)dres_, hres_dev_, nwords, hflag_dev_, seq);
      wait_flag(seq, nwords);
    } else {
      HIP_CHECK(hipMemcpyAsync(hres_, dres_, nwords * 8, hipMemcpyDeviceToHost, s_));
      HIP_CHECK(hipStreamSynchronize(s_));
    }
  }
  static PointArg make_point(const Ext* pt, unsigned k) {
    DP_REQUIRE(k <= MAX_PT, DP_ERR_SHAPE, "point too long");
    PointArg p;
    for (unsigned i = 0; i < k; i++) p.p[i] = pt[i];
    for (unsigned i = k; i < MAX_PT; i++) p.p[i] = ex_zero();
    return p;
  }

 public:
  // (the constructor allocates a stream, the arena — hundreds of megabytes for a batch worker — and pinned buffers one after the other: when a later one fails,
  // typically with the device out of memory, what the earlier ones took goes back before the error travels on; the destructor of a half-built object never runs)
  explicit HipDev(int device, size_t arena_bytes = 0, size_t stage_bytes = 0) : device_(device) {
    CPU_ZERO(&numa_cpus_); CPU_ZERO(&saved_affinity_);
    try { init_(device, arena_bytes, stage_bytes); } catch (...) { destroy_(); throw; }
  }
  void init_(int device, size_t arena_bytes, size_t stage_bytes) {
    if (stage_bytes) STAGE_BYTES = stage_bytes;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) throw DpError(DP_ERR_NODEVICE, "no HIP device available: the MI355X path is mandatory, there is no CPU fallback");
    DP_REQUIRE(device >= 0 && device < cnt, DP_ERR_ARG, "bad device id");
    HIP_CHECK(hipSetDevice(device));
    numa_mask_init_(device);
    if (!arena_bytes) numa_pin_creator_();  // (arena_bytes != 0: a worker of a batch or of the engine — the calling thread is not its to move)
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    name_ = std::string("hip:") + prop.name + ":" + prop.gcnArchName;
    HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
    const char* env = getenv("DP_ARENA_BYTES");
    arena_cap_ = arena_bytes ? arena_bytes : env ? strtoull(env, nullptr, 10) : (size_t(12) << 30);
    HIP_CHECK(hipMalloc((void**)&arena_, arena_cap_));
    HIP_CHECK(hipHostMalloc((void**)&hres_, RES_WORDS * 8 + 1024, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostGetDevicePointer((void**)&hres_dev_, hres_, 0));
    if (host_sponge_) {  // request + reply areas of the host-side sponge service (sponge_host.h)
      HIP_CHECK(hipHostMalloc((void**)&hsp_, (WC_REQ_WORDS + WC_REP_WORDS) * 8, hipHostMallocMapped | hipHostMallocCoherent));
      memset(hsp_, 0, (WC_REQ_WORDS + WC_REP_WORDS) * 8);
      HIP_CHECK(hipHostGetDevicePointer((void**)&hsp_dev_, hsp_, 0));
      sp_slot_ = sponge_slot_new();
      sp_slot_->req = hsp_; sp_slot_->rep = hsp_ + WC_REQ_WORDS;
    }
    hflag_ = (unsigned long long*)(hres_ + RES_WORDS);
    hflag_dev_ = (unsigned long long*)(hres_dev_ + RES_WORDS);
    *hflag_ = 0;
    hmail_ = hflag_ + 8; hmail_dev_ = hflag_dev_ + 8;
    hmail_[0] = hmail_[1] = hmail_[2] = hmail_[3] = 0;
    hmflag_ = hflag_ + 32; hmflag_dev_ = hflag_dev_ + 32;  // one flag word per workgroup of a multi-workgroup sumcheck phase
    for (int i = 0; i < MULTI_MAX_WG; i++) { hmflag_[i] = 0; last_tag_multi_[i] = 0; }
    zerocopy_ = !(getenv("DP_NO_ZEROCOPY") && atoi(getenv("DP_NO_ZEROCOPY")));
    persist_ = zerocopy_ && !(getenv("DP_NO_PERSIST") && atoi(getenv("DP_NO_PERSIST")));
    if (g_timing_level > 1) { HIP_CHECK(hipMalloc((void**)&scdbg_, 64)); HIP_CHECK(hipMemset(scdbg_, 0, 64)); }
    HIP_CHECK(hipMalloc((void**)&dres_, RES_WORDS * 8));
    HIP_CHECK(hipMalloc((void**)&fused_ticket_, 64)); HIP_CHECK(hipMemset(fused_ticket_, 0, 64));
    HIP_CHECK(hipHostMalloc(&hstage_, STAGE_BYTES + DESC_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostGetDevicePointer((void**)&hstage_dev_, hstage_, 0));
    HIP_CHECK(hipHostMalloc(&hdesc_, DESC_BYTES, host_ro_flags()));
    HIP_CHECK(hipHostGetDevicePointer((void**)&hdesc_dev_, hdesc_, 0));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST)));
    { std::vector<u64> ex((SC_MAXK + 1) * (SC_MAXK + 1) * (SC_MAXK + 1), 0);  // extrapolation_coeffs(k, at)[i] of sumcheck.h
      for (unsigned k = 1; k < (unsigned)SC_MAXK; k++) for (unsigned at = k + 1; at <= (unsigned)SC_MAXK; at++) for (unsigned i = 0; i <= k; i++)
        ex[((size_t)k * (SC_MAXK + 1) + at) * (SC_MAXK + 1) + i] = extrapolation_coeffs(k, at)[i];
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_extrap), ex.data(), ex.size() * 8)); }
    { double ts = getenv("DP_POLL_TIMEOUT_S") ? std::max(0.001, atof(getenv("DP_POLL_TIMEOUT_S"))) : 20.0; unsigned long long tk = (unsigned long long)(ts * 1e8); HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_poll_timeout_ticks), &tk, sizeof(tk))); }
#ifdef DP_DIAG_SKIP_HASH
#endif
    DP_SET_LDS_ONE((k_sc_persist_lds<false>), 1024, (int)SC_LDS_MAX);
    DP_SET_LDS_ONE((k_sc_persist_lds<true>), 1024, (int)SC_LDS_MAX);
    excl_ = EXCL_LDS;
    DP_SET_LDS_ONE((k_sc_persist<false>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_persist<true>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_small<false>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_small<true>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE(k_merkle_tail, 1024, (int)EXCL_LDS);
    if (devlogup_ || devlogup_full_) { DP_SET_LDS_ONE(k_logup_tail, 1024, 128 * 1024); set_lds_<k_logup_tail, WIDE_MAXT, KF_PRIO>(KArgs<decltype(&k_logup_tail)>(), 128 * 1024); }  // (logup_tail.h: logup_tail_lds_bytes)
    if (devclassic_) DP_SET_LDS_ONE(k_classic_tail, 1024, (int)EXCL_LDS);
    if (devdense_) DP_SET_LDS_ONE(k_dense_tail, 1024, (int)EXCL_LDS);
    if (deveqsum_) DP_SET_LDS_ONE(k_eqsum_tail, 1024, (int)EXCL_LDS);
    if (devdeleg_) DP_SET_LDS_ONE(k_deleg_tail, 1024, (int)EXCL_LDS);
    if (devcommit_) DP_SET_LDS_ONE(k_commit_tail, 1024, (int)EXCL_LDS);
    DP_SET_LDS((k_butterfly_pass<false, false>), 1024, 64 * 1024); DP_SET_LDS((k_butterfly_pass<false, true>), 1024, 64 * 1024);
    DP_SET_LDS((k_butterfly_pass<true, false>), 1024, 64 * 1024); DP_SET_LDS((k_butterfly_pass<true, true>), 1024, 64 * 1024);
    DP_SET_LDS(k_med_prepare, 1024, 128 * 1024);
    DP_SET_LDS(k_med_ntt_local, 1024, 64 * 1024);
  }
  ~HipDev() override { destroy_(); }
  void destroy_() noexcept {
    numa_unpin_creator_();
    hipSetDevice(device_);
    if (s_) hipStreamSynchronize(s_);
    pcs_tabs_.reset();
    if (arena_) hipFree(arena_);
    if (dres_) hipFree(dres_);
    for (auto& kv : plimbo_) hipFree(kv.second);  // (blocks in the device's pool stay for the next context)
    if (dshare_) hipFree(dshare_);
    if (dgather_) hipFree(dgather_);
    if (fused_ticket_) hipFree(fused_ticket_);
    if (sp_slot_) { sponge_disarm_(); sponge_slot_free(sp_slot_); sp_slot_ = nullptr; }
    if (hsp_) hipHostFree(hsp_);
    if (hres_) hipHostFree(hres_);
    if (vmail_) hipFree(vmail_);
    if (hstage_) hipHostFree(hstage_);
    if (hdesc_) hipHostFree(hdesc_);
    if (s_) hipStreamDestroy(s_);
  }
  const char* name() const override { return name_.c_str(); }
  size_t arena_peak() const { return arena_peak_; }
  void arena_peak_reset() { arena_peak_ = arena_off_; }
  // Dev::merkle_paths_check on the GPU: the job arrays go up in one staging pass, one launch, two words come back
  bool merkle_paths_check(const u64* leaf, const u64* root, const u64* x, const u64* path_off, const u64* depth, size_t n, const u64* pool, size_t pool_digests, size_t* first_bad) override {
    if (!n) return true;
    DP_REQUIRE(!queued_(), DP_ERR_ARG, "merkle_paths_check: not from inside a cohort / the resident executor");
    std::vector<u64> meta(3 * n);
    for (size_t j = 0; j < n; j++) { DP_REQUIRE(path_off[j] + depth[j] <= pool_digests, DP_ERR_ARG, "merkle_paths_check: path outside the pool"); meta[3 * j] = x[j]; meta[3 * j + 1] = path_off[j]; meta[3 * j + 2] = depth[j]; }
    struct ArenaMark { HipDev* d; size_t mk; ~ArenaMark() { d->release(mk); } } guard_{this, mark()};  // released on every exit: an adversarial proof must not leak the arena
    DBuf dl = alloc(4 * n, false), dr = alloc(4 * n, false), dm = alloc(3 * n, false), dp = alloc(std::max<size_t>(4 * pool_digests, 4), false), db = alloc(2, false);
    upload(dl, leaf); upload(dr, root); upload(dm, meta.data());
    if (pool_digests) upload(dp, pool);
    const u64 init[2] = {0, ~0ull};
    upload(db, init);
    nb_ = 96.0 * (double)pool_digests; DPL(k_merkle_paths, dim3(grid_for(n, 4096)), dim3(TPB), (const u64*)dl.p, (const u64*)dr.p, (const u64*)dm.p, (const u64*)dp.p, n, (unsigned long long*)db.p);
    u64 res[2];
    download(db, res);
    if (res[0] && first_bad) *first_bad = (size_t)res[1];
    return res[0] == 0;
  }
  // Poseidon2 compress() per second of the one-node-per-lane Merkle kernel on `nodes` nodes (chip-filling when nodes >> 458 752
  // = 256 CUs x 28 waves x 64 lanes): the VALU-integer peak bench.py prices the whole job's hashing against, measured with
  // HIP events on this context's stream in the same run. The input is whatever the arena holds: the arithmetic is branch-free.
  double probe_compress_rate(size_t nodes, int reps) {
    DP_REQUIRE(!queued_() && nodes >= 1024 && reps >= 1, DP_ERR_ARG, "probe: bad arguments");
    const size_t mk = mark();
    DBuf in = alloc(8 * nodes, false), out = alloc(4 * nodes, false);
    nb_ = 0; DPL(k_zero_words, dim3(grid_for(8 * nodes)), dim3(TPB), (u64*)in.p, 8 * nodes);
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    nb_ = 96.0 * nodes; DPL_HASH(k_merkle_layer, dim3(grid_for(nodes, 4096)), dim3(TPB), (const u64*)in.p, (u64*)out.p, nodes);
    HIP_CHECK(hipEventRecord(a, s_));
    for (int r = 0; r < reps; r++) { nb_ = 96.0 * nodes; DPL_HASH(k_merkle_layer, dim3(grid_for(nodes, 4096)), dim3(TPB), (const u64*)in.p, (u64*)out.p, nodes); }
    HIP_CHECK(hipEventRecord(b, s_)); HIP_CHECK(hipEventSynchronize(b));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    release(mk);
    return ms > 0 ? (double)nodes * reps / (ms * 1e-3) : 0.0;
  }
  void set_latency_mode(bool on) { multi_ = on; devfs_ = devfs_env_ < 0 ? !on : devfs_env_ != 0; throughput_mode_ = !on; }
  void dump_host_stats() {
    if (!g_host_stats) return;
    fprintf(stderr, "[dp timing] device context: %zu launches, %.1f us of host time per launch (%.1f ms total), %zu device waits, %zu fiber yields; host work between waits %.1f ms, inside waits %.1f ms\n", nlaunch_, nlaunch_ ? launch_us_ / nlaunch_ : 0.0, launch_us_ / 1000.0, nwait_, nyield_, work_us_ / 1000.0, waitlat_us_ / 1000.0);
    if (sp_slot_) fprintf(stderr, "[dp timing] host sponge: %llu requests served for this context so far\n", (unsigned long long)sp_slot_->nserved.load());
    if (g_timing_level > 1) {  // launches by kernel since the last dump (DP_TIMING=2)
      std::vector<std::pair<size_t, const char*>> v; for (auto& kv : by_name_) v.push_back({kv.second, kv.first});
      std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
      for (auto& e : v) fprintf(stderr, "[dp launches] %6zu  %s\n", e.first, e.second);
    }
    by_name_.clear();
    if (g_timing_level > 1) for (auto& c : chunks_) fprintf(stderr, "[dp chunk] before wait %zu: %.0f us of host work, launches %s .. %s\n", c.wait, c.us, c.first ? c.first : "-", c.last ? c.last : "-");
    chunks_.clear();
    if (g_timing_level > 2) {
      std::vector<std::pair<double, std::string>> v;
      for (auto& kv : chunk_by_) { char b[256]; snprintf(b, sizeof b, "%8.1f us total, %5zu times, %7.1f us each: %s .. %s", kv.second.first, kv.second.second, kv.second.first / kv.second.second, kv.first.first ? kv.first.first : "-", kv.first.second ? kv.first.second : "-"); v.push_back({kv.second.first, b}); }
      std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
      for (auto& e : v) fprintf(stderr, "[dp host-work] %s\n", e.second.c_str());
    }
    chunk_by_.clear();
    launch_us_ = 0; nlaunch_ = nwait_ = nyield_ = 0; work_us_ = waitlat_us_ = 0; have_exit_ = false;
  }
  void dump_sc_debug() {
    if (!scdbg_) return;
    unsigned long long h[8]; hipStreamSynchronize(s_); hipMemcpy(h, scdbg_, 64, hipMemcpyDeviceToHost); hipMemset(scdbg_, 0, 64);
    fprintf(stderr, "[dp sc-debug] rounds %llu: cycles/round fold %.0f sums %.0f publish %.0f wait-for-challenge %.0f; speculated rounds %llu, cycles per speculation (wave 1) %.0f; wave 0 at the barrier behind its wait: %.0f cycles per round\n", h[4], (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4],
            h[6], h[6] ? (double)h[5] / h[6] : 0.0, (double)h[7] / h[4]);
  }
  void bind_thread() override { HIP_CHECK(hipSetDevice(device_)); }
  hipStream_t stream() const { return s_; }
  // per-kernel HIP-event timing on the launch stream (bench.py roofline). report: name -> (launches, total ms, total bytes)
  void profile_enable(bool on) {
    hipStreamSynchronize(s_);
    for (auto& r : recs_) { prof_pool_.push_back(r.a); prof_pool_.push_back(r.b); }
    recs_.clear();
    if (!on) { for (auto e : prof_pool_) hipEventDestroy(e); prof_pool_.clear(); }
    prof_ = on;
  }
  std::string profile_report() {
    hipStreamSynchronize(s_);
    struct Agg { size_t n = 0; double ms = 0, bytes = 0; };
    std::vector<std::pair<std::string, Agg>> agg;
    for (auto& r : recs_) {
      float ms = 0; hipEventElapsedTime(&ms, r.a, r.b);
      std::string nm = r.name;  // "(k_x<3, true>)" -> "k_x<3, true>"
      if (nm.size() > 2 && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
      size_t k = 0; for (; k < agg.size(); k++) if (agg[k].first == nm) break;
      if (k == agg.size()) agg.push_back({nm, Agg()});
      agg[k].second.n++; agg[k].second.ms += ms; agg[k].second.bytes += r.bytes;
    }
    std::string out = "[";
    for (size_t k = 0; k < agg.size(); k++) {
      char buf[512];
      snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"launches\": %zu, \"total_ms\": %.6f, \"alg_bytes\": %.0f}", k ? ", " : "", agg[k].first.c_str(), agg[k].second.n, agg[k].second.ms, agg[k].second.bytes);
      out += buf;
    }
    return out + "]";
  }

  // ---- memory
  DBuf alloc(size_t n, bool ext) override { DBuf b; b.n = n; b.ext = ext; b.p = arena_alloc(std::max<size_t>(n, 1) * (ext ? 16 : 8)); return b; }
  size_t mark() override { return arena_off_; }
  void release(size_t m) override { arena_off_ = m; }
  // Persistent buffers (tables a caller uploads, commitments' trees, fixed matrices) come from hipMalloc; a seam-level host uploads and frees dozens
  // of small columns per proof, hipMalloc costs ~100 us and hipFree waits for the device. Freed blocks of up to 64 MB are therefore kept in ONE pool per
  // device (g_persist_pool: a block allocated by an engine worker is usually freed through the caller's context; at most 2 GB, DP_PERSIST_POOL_BYTES) and handed
  // out again for requests of the same rounded size. A freed block may still be read by launches its context has queued: it waits in the context's
  // `plimbo_` and enters the pool after the next stream_wait the context performs for that purpose (one per PLIMBO_MAX frees, or when a request finds
  // the pool empty) — a free is O(1).
  struct PersistPool { std::mutex mu; std::multimap<size_t, void*> blocks; size_t bytes = 0; size_t cap = [] { const char* e = getenv("DP_PERSIST_POOL_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t(2) << 30); }(); };
  static PersistPool& persist_pool_(int device) { static PersistPool pools[16]; return pools[device & 15]; }
  std::vector<std::pair<size_t, void*>> plimbo_;
  static constexpr size_t PLIMBO_MAX = 64;
  static size_t persist_round_(size_t n, bool ext) { return (std::max<size_t>(n, 1) * (ext ? 16 : 8) + 4095) & ~size_t(4095); }
  void plimbo_flush_() {
    if (plimbo_.empty()) return;
    stream_wait();
    PersistPool& pp = persist_pool_(device_);
    std::lock_guard<std::mutex> g(pp.mu);
    for (auto& kv : plimbo_) { if (pp.bytes + kv.first <= pp.cap) { pp.blocks.emplace(kv.first, kv.second); pp.bytes += kv.first; } else hipFree(kv.second); }
    plimbo_.clear();
  }
  DBuf alloc_persistent(size_t n, bool ext) override {
    DBuf b; b.n = n; b.ext = ext;
    const size_t bytes = persist_round_(n, ext);
    PersistPool& pp = persist_pool_(device_);
    for (int attempt = 0; attempt < 2; attempt++) {
      {
        std::lock_guard<std::mutex> g(pp.mu);
        auto it = pp.blocks.find(bytes);
        if (it != pp.blocks.end()) { b.p = it->second; pp.blocks.erase(it); pp.bytes -= bytes; return b; }
      }
      if (attempt == 0 && !plimbo_.empty()) plimbo_flush_(); else break;
    }
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipMalloc(&b.p, bytes));
    return b;
  }
  void free_persistent(DBuf& b) override {
    if (!b.p) return;
    const size_t bytes = persist_round_(b.n, b.ext);
    if (bytes <= (size_t(64) << 20)) {
      plimbo_.emplace_back(bytes, b.p); b.p = nullptr;
      if (plimbo_.size() >= PLIMBO_MAX) plimbo_flush_();
      return;
    }
    hipStreamSynchronize(s_); hipFree(b.p); b.p = nullptr;
  }
  // host <-> device copies go through the pinned staging buffer: hipMemcpyAsync on pageable memory pins the user pages
  // on the fly, which costs tens of milliseconds per MB on this stack
  // A cohort member moves data with kernels (k_copy_words through the mapped staging buffer, k_zero_words): a memcpy
  // command queued by one member would overtake the merged launches its cohort has not fired yet.
  void h2d(void* dst, const void* src, size_t bytes) {
    // an upload of up to the size of the asynchronous staging ring goes through it in slots of at most ASYNC_MAX (no device wait unless the ring wraps): until
    // round 5 anything above 2 MB took the synchronous path below — the witness columns of a transformer-layer proof (12 MB), member after member of a cohort
    if (async_upload_now() && zerocopy_ && bytes > ASYNC_MAX && bytes % 8 == 0 && bytes <= ASYNC_STAGE && STAGE_BYTES >= ASYNC_STAGE) {
      for (size_t off = 0; off < bytes; off += ASYNC_MAX) h2d((char*)dst + off, (const char*)src + off, std::min(ASYNC_MAX, bytes - off));
      return;
    }
    if (async_upload_now() && zerocopy_ && bytes > 0 && bytes % 8 == 0 && bytes <= ASYNC_MAX && STAGE_BYTES >= ASYNC_STAGE) {
      const size_t need = (bytes + 255) & ~size_t(255);
      if (stage_off_ + need > ASYNC_STAGE) { stream_wait(); stage_off_ = 0; }
      char* slot = bulk_stage() + stage_off_;
      memcpy(slot, src, bytes);
      if (g_host_stats) by_name_["  (k_copy_words as upload)"]++;
      if (queued_()) { nb_ = 0; DPL(k_copy_words, dim3(grid_for(bytes / 8, 256)), dim3(TPB), (u64*)dst, (const u64*)(hstage_dev_ + DESC_BYTES + stage_off_), bytes / 8); }
      else { nb_ = 0; prof_begin("memcpy_h2d"); HIP_CHECK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, s_)); prof_end(); }
      stage_off_ += need;
      return;
    }
    if (stage_off_) { stream_wait(); stage_off_ = 0; }  // pending slots of earlier asynchronous uploads must have been read before slot 0 is reused
    for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
      size_t m = std::min(STAGE_BYTES, bytes - off);
      memcpy(bulk_stage(), (const char*)src + off, m);
      if (queued_()) {
        DP_REQUIRE(m % 8 == 0, DP_ERR_ARG, "copies are whole words");
        nb_ = 0; DPL(k_copy_words, dim3(grid_for(m / 8, 256)), dim3(TPB), (u64*)((char*)dst + off), (const u64*)(hstage_dev_ + DESC_BYTES), m / 8);
      } else { nb_ = 0; prof_begin("memcpy_h2d"); HIP_CHECK(hipMemcpyAsync((char*)dst + off, bulk_stage(), m, hipMemcpyHostToDevice, s_)); prof_end(); }
      stream_wait();
    }
  }
  static unsigned long long dl_mix_host(unsigned long long seq, unsigned long long chunk) { return (seq * 0x9E3779B97F4A7C15ull + chunk) * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull; }
  void d2h(void* dst, const void* src, size_t bytes) {
    // (pending upload slots are read by copy kernels that precede this download's copy in the stream: no drain needed)
    if (queued_()) {
      // cohort member / executor slot: k_download — the copy and its per-chunk tags in one step, every chunk verified here (see the
      // kernel: a separate "done" publication does not order the copy's PCIe writes before it outside a kernel boundary)
      DP_REQUIRE(bytes % 8 == 0, DP_ERR_ARG, "copies are whole words");
      const size_t cap_words = ((STAGE_BYTES - 4096) / (8 * (DL_CHUNK + 1))) * DL_CHUNK;  // payload + one tag word per chunk fit the staging area
      for (size_t off = 0; off < bytes; off += cap_words * 8) {
        const size_t nwords = std::min(cap_words, (bytes - off) / 8), nch = (nwords + DL_CHUNK - 1) / DL_CHUNK;
        const size_t tag_off = (nwords * 8 + 255) & ~size_t(255);
        if (g_host_stats) by_name_["  (k_download)"]++;
        const unsigned long long seq = ++seq_;
        nb_ = 0; DPL(k_download, dim3((unsigned)std::min<size_t>(nch, (size_t)grid_for(nch * TPB, 256))), dim3(TPB), (const u64*)((const char*)src + off), (u64*)(hstage_dev_ + DESC_BYTES), nwords, (u64*)(hstage_dev_ + DESC_BYTES + tag_off), seq);
        auto t0 = wait_enter_();
        nwait_++;
        volatile u64* pay = (volatile u64*)bulk_stage();
        volatile u64* tags = (volatile u64*)(bulk_stage() + tag_off);
        u64* out = (u64*)((char*)dst + off);
        unsigned spins = 0;
        for (size_t ch = 0; ch < nch; ch++) {
          const size_t lo = ch * DL_CHUNK, hi = std::min(lo + DL_CHUNK, nwords);
          for (;;) {
            const unsigned long long tag = tags[ch];
            std::atomic_thread_fence(std::memory_order_acquire);
            unsigned long long cs = 0;
            if (auto f = dl_copy_sum_fast()) cs = f((const u64*)pay + lo, out + lo, hi - lo);  // (AVX-512: a quarter of the scalar loop's time; the acquire fence above orders the loads behind the tag's)
            else for (size_t i = lo; i < hi; i++) { const u64 v = pay[i]; out[i] = v; cs += (unsigned long long)(i - lo + 1) * v; }
            if (dl_mix_host(seq, ch) + cs == tag) break;
            const bool fib = fiber_active();
            if (fib) { nyield_++; fiber_yield(); } else dp_spin_pause();
            if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
              throw DpError(DP_ERR_HIP, std::string("timeout waiting for a download"));
          }
        }
        // every chunk has landed: the copy — and everything queued before it — has run
        desc_off_ = 0; stage_off_ = 0; if (co_ && co_li_) co_->note_executed(co_li_ - 1);
        wait_exit_(t0);
      }
      return;
    }
    for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
      size_t m = std::min(STAGE_BYTES, bytes - off);
      nb_ = 0; prof_begin("memcpy_d2h"); HIP_CHECK(hipMemcpyAsync(bulk_stage(), (const char*)src + off, m, hipMemcpyDeviceToHost, s_)); prof_end();
      stream_wait();
      memcpy((char*)dst + off, bulk_stage(), m);
    }
  }
  void upload(const DBuf& d, const u64* src) override { h2d(d.p, src, d.bytes()); }
  void upload_i64(const DBuf& d, const int64_t* src) override {
    DP_REQUIRE(!d.ext, DP_ERR_ARG, "upload_i64 needs a base buffer");
    size_t mk = mark();
    int64_t* tmp = (int64_t*)arena_alloc(d.n * 8);
    h2d(tmp, src, d.n * 8);
    DPL(k_fieldize, dim3(grid_for(d.n)), dim3(TPB), tmp, (u64*)d.p, d.n);
    stream_wait();
    release(mk);
  }
  void download(const DBuf& src, u64* dst) override { d2h(dst, src.p, src.bytes()); }
  void copy(const DBuf& d, const DBuf& s) override {
    nb_ = 2.0 * s.bytes();
    if (queued_()) { if (s.bytes()) DPL(k_copy_words, dim3(grid_for(s.bytes() / 8)), dim3(TPB), (u64*)d.p, (const u64*)s.p, s.bytes() / 8); return; }
    prof_begin("memcpy_d2d"); HIP_CHECK(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, s_)); prof_end();
  }
  void zero(const DBuf& d) override {
    nb_ = (double)d.bytes();
    if (queued_()) { if (d.bytes()) DPL(k_zero_words, dim3(grid_for(d.bytes() / 8)), dim3(TPB), (u64*)d.p, d.bytes() / 8); return; }
    prof_begin("memset"); HIP_CHECK(hipMemsetAsync(d.p, 0, d.bytes(), s_)); prof_end();
  }
  // ---- cohort membership (dp_model_prove_batch): while attached every launch of this context is one pack of a merged launch
  void cohort_attach(Cohort* co) {
    DP_REQUIRE(!co_ && !prof_ && zerocopy_, DP_ERR_ARG, "cohort members need the zero-copy publish path and no per-kernel profiling");
    HIP_CHECK(hipStreamSynchronize(s_));
    co->join(); co_ = co; co_li_ = co->q_base;
  }
  void cohort_detach() { if (co_) { Cohort* c = co_; co_ = nullptr; c->leave(co_li_); } }
  bool in_cohort() const { return co_ != nullptr; }
  void sync() override { stream_wait(); }
  void flush_uploads() override { if (stage_off_) { stream_wait(); stage_off_ = 0; } }
  void abort_call() override { sess_ = ScSession(); }

  // ---- MLE
  void eq_table_many(const EqJob* jobs, size_t n) override {
    if (!n) return;
    if (n * (sizeof(EqDesc) + 4) + 256 > DESC_BYTES) { Dev::eq_table_many(jobs, n); return; }
    if (desc_off_ + n * (sizeof(EqDesc) + 4) + 192 > DESC_BYTES) stream_wait();
    const EqDesc* dd = nullptr; const unsigned* fd = nullptr;
    unsigned* first = desc_alloc<unsigned>(n + 1, &fd);
    EqDesc* hd = desc_alloc<EqDesc>(n, &dd);
    unsigned nblk = 0; double bytes = 0;
    for (size_t i = 0; i < n; i++) {
      const EqJob& j = jobs[i];
      DP_REQUIRE(j.out.ext && j.out.n == (size_t(1) << j.k) && j.k <= (unsigned)MAX_PT, DP_ERR_SHAPE, "eq_table_many: output shape");
      hd[i].out = (Ext*)j.out.p; hd[i].k = j.k; hd[i].pad = 0;
      for (unsigned t = 0; t < j.k; t++) hd[i].pt[t] = j.pt[t];
      first[i] = nblk; nblk += (unsigned)((j.out.n + EQ_CHUNK - 1) / EQ_CHUNK);  // one workgroup per EQ_CHUNK entries
      bytes += 16.0 * j.out.n;
    }
    first[n] = nblk;
    nb_ = bytes; DPL(k_eq_table_many, dim3(nblk), dim3(TPB), fd, dd, (int)n);
  }
  void eq_table(const DBuf& out, const Ext* pt, unsigned k, Ext scale, bool acc) override {
    DP_REQUIRE(out.ext && out.n == (size_t(1) << k), DP_ERR_SHAPE, "eq_table: output shape");
    // a plain long table (the commit phase builds eq over 2^20 entries per proof): the chunked low x high form of k_eq_table_many, one
    // multiplication per entry instead of k (k_eq_table was 1.5 % of the kernel time of a Dense-4M batch, profiles/r03_bench448_kernel_stats.csv)
    if (!acc && k >= 12 && k <= (unsigned)MAX_PT && scale.c0 == 1 && scale.c1 == 0) { EqJob j{out, pt, k}; eq_table_many(&j, 1); return; }
    nb_ = 16.0 * out.n * (acc ? 2 : 1); DPL(k_eq_table, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, make_point(pt, k), k, scale, acc ? 1 : 0, out.n);
  }
  // ---- lazy eq: remembered here, built inside the persistent sumcheck kernel that consumes it (or materialised by a
  // plain launch if the next sumcheck takes another path)
  struct PendingEq { void* p = nullptr; unsigned k = 0; Ext pt[MAX_PT]; } pend_eq_;
  void eq_table_lazy(const DBuf& out, const Ext* pt, unsigned k) override {
    DP_REQUIRE(out.ext && out.n == (size_t(1) << k) && k <= (unsigned)MAX_PT, DP_ERR_SHAPE, "eq_table: output shape");
    flush_pending_eq();
    if (!persist_) { eq_table(out, pt, k, ex_one(), false); return; }
    pend_eq_.p = out.p; pend_eq_.k = k;
    for (unsigned i = 0; i < k; i++) pend_eq_.pt[i] = pt[i];
  }
  void flush_pending_eq() {
    if (!pend_eq_.p) return;
    DBuf b; b.p = pend_eq_.p; b.n = size_t(1) << pend_eq_.k; b.ext = true;
    pend_eq_.p = nullptr;
    eq_table(b, pend_eq_.pt, pend_eq_.k, ex_one(), false);
  }
  void eq_table_tiled(const DBuf& out, const Ext* pt, unsigned k) override {
    DP_REQUIRE(out.ext && out.n % (size_t(1) << k) == 0, DP_ERR_SHAPE, "eq_table_tiled: output shape");
    nb_ = 16.0 * out.n; DPL(k_eq_table, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, make_point(pt, k), k, ex_one(), 0, out.n);
  }
  void mle_eval_batch(const DBuf* fs, int nf, const Ext* pt, unsigned k, Ext* out) override {
    flush_pending_eq();
    size_t n = size_t(1) << k;
    PointArg p = make_point(pt, k);
    for (int s = 0; s < nf; s += 8) {
      EvalArgs a; a.nf = std::min(8, nf - s);
      for (int f = 0; f < 8; f++) { a.f[f] = nullptr; a.ext[f] = 0; }
      for (int f = 0; f < a.nf; f++) { DP_REQUIRE(fs[s + f].n == n, DP_ERR_SHAPE, "mle_eval: table size != 2^|point|"); a.f[f] = fs[s + f].p; a.ext[f] = fs[s + f].ext; }
      size_t mk = mark();
      int g = grid_for(n, 1024);
      Ext* partial = (Ext*)arena_alloc((size_t)g * 8 * 16);
      nb_ = [&] { double b = 0; for (int f = 0; f < a.nf; f++) b += fs[s + f].bytes(); return b; }(); DPL(k_mle_eval_partial, dim3(g), dim3(TPB), a, p, k, partial);
      reduce_publish(partial, (size_t)g, 8, a.nf);
      for (int f = 0; f < a.nf; f++) out[s + f] = ex(hres_[2 * f], hres_[2 * f + 1]);
      release(mk);
    }
  }
  void fix_high(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) override {
    DP_REQUIRE(!W.ext && W.n == R * C && out.ext && out.n == C, DP_ERR_SHAPE, "fix_high: shapes");
    unsigned k = dp_ceil_log2(R);
    size_t mk = mark();
    DBuf eq = alloc(R, true);
    eq_table(eq, pt, k, ex_one(), false);
    size_t nsplit = std::min<size_t>(R, 64);
    size_t rps = (R + nsplit - 1) / nsplit;
    Ext* partial = (Ext*)arena_alloc(nsplit * C * 16);
    dim3 g((unsigned)((C + TPB - 1) / TPB), (unsigned)nsplit);
    nb_ = 8.0 * R * C + 16.0 * R; DPL(k_fix_high_partial, g, dim3(TPB), (const u64*)W.p, (const Ext*)eq.p, R, C, rps, partial);
    DPL(k_colsum, dim3((unsigned)((C + TPB - 1) / TPB)), dim3(TPB), partial, nsplit, C, (Ext*)out.p);
    release(mk);  // safe: stream ordered, later allocations are only written by later kernels
  }

  void fix_low(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) override {
    DP_REQUIRE(!W.ext && W.n == R * C && out.ext && out.n == R && C >= 2 && (C & (C - 1)) == 0, DP_ERR_SHAPE, "fix_low: shapes");
    size_t mk = mark();
    DBuf eq = alloc(C, true);
    eq_table(eq, pt, dp_ceil_log2(C), ex_one(), false);
    nb_ = 8.0 * R * C + 16.0 * C + 16.0 * R; DPL(k_fix_low, dim3(grid_for(R * 64)), dim3(TPB), (const u64*)W.p, (const Ext*)eq.p, (Ext*)out.p, R, C);
    release(mk);  // safe: stream ordered, later allocations are only written by later kernels
  }

  // ---- sumcheck
  void fold_tables(DBuf* tabs, int nt, Ext r) {
    for (int s = 0; s < nt; s += MAX_TABS) {
      int m = std::min(MAX_TABS, nt - s);
      FoldArgs a; size_t maxh = 1;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.ext[i] = 0; a.half[i] = 0; }
      for (int i = 0; i < m; i++) {
        DBuf& t = tabs[s + i];
        DBuf o = alloc(t.n / 2, true);
        a.in[i] = t.p; a.out[i] = (Ext*)o.p; a.ext[i] = t.ext; a.half[i] = t.n / 2;
        maxh = std::max(maxh, t.n / 2);
        t = o;
      }
      nb_ = [&] { double b = 0; for (int i = 0; i < m; i++) b += a.half[i] * (a.ext[i] ? 32.0 : 16.0) + a.half[i] * 16.0; return b; }(); DPL(k_fold, dim3(grid_for(maxh), m), dim3(TPB), a, r);
    }
  }
  static constexpr size_t SC_LDS_MAX = 128 * 1024;  // dynamic LDS the LDS-resident sumcheck kernel may use
  // (a resident worker of class BIG has 64 KB for the kernel's frame and its tables: larger sumchecks take the global-memory kernel)
  size_t sc_lds_max() const { return SC_LDS_MAX; }
  // Latency-critical one-workgroup kernels ask for more than half of a CU's 160 KB of LDS even when they need none: two
  // such workgroups can then never share a CU. The workgroup dispatcher otherwise packs the small persistent kernels of
  // all proofs in flight onto the same first CUs (4 kernels of 256 threads fit on one), where they time-share the SIMDs.
  static constexpr size_t EXCL_LDS = 84 * 1024;
  static constexpr size_t SC_PERSIST_MAX = 16384;  // sumchecks whose tables are at most this long run in the persistent kernel
  // The challenge mailbox of the host-driven persistent sumchecks. In host memory every poll of the waiting kernel is a PCIe READ
  // (a round trip of microseconds, and the challenge is seen half a poll period later on average). With the whole of device
  // memory visible to the CPU (large BAR) the mailbox lives in fine-grained DEVICE memory instead: the CPU's store is a posted
  // PCIe write, the kernel polls its own HBM. Set up at the first host-driven session of a context (throughput-mode workers
  // never get here); checked, not assumed: the page must be CPU-writable (probed through a pipe, no fault) and a kernel must read
  // back two successive CPU writes. DP_MAILBOX_VRAM=0 keeps the mailbox in host memory.
  // Host threads that drive this GPU stay on the CPUs of the GPU's own NUMA node. Every Fiat-Shamir round of a single proof crosses PCIe twice; from the other
  // socket of a two-socket host each crossing also crosses the socket interconnect: one Dense-4M proof 34.7 ms from node 0 against 30.7 ms from the GPU's node 1,
  // four processes each (profiles/r05_numa_latency.txt). Who is pinned (round 6, after the advisor's finding that round 5 narrowed the APPLICATION's thread for
  // good and let the first GPU's node win for every later context):
  //  * the mask is per DEVICE: the node's CPUs (/sys/bus/pci/devices/<bdf>/local_cpulist) intersected with the affinity the PROCESS had when the library first
  //    looked (not with whatever an earlier context narrowed the thread to) — a context for a GPU on the other socket gets that socket's CPUs;
  //  * library-owned threads (cohort threads, engine threads, helpers) pin themselves for their whole life: pin_thread();
  //  * the thread that CREATES an owning context (dp_ctx_create) is narrowed too — it is the thread that proves single proofs — but its previous affinity is
  //    remembered and put back when the context is destroyed (or when the same thread creates a context on another device: the newest wins, the oldest
  //    saved mask is the one restored); workers of a batch (make_hip_worker) never touch the calling thread;
  //  * nothing happens when the intersection is empty, when the thread is already inside the node, or with DP_NUMA_PIN=0.
  static const cpu_set_t& process_affinity_() {
    static const cpu_set_t orig = [] { cpu_set_t c; CPU_ZERO(&c); if (sched_getaffinity(0, sizeof(c), &c) != 0) CPU_ZERO(&c); return c; }();
    return orig;
  }
  cpu_set_t numa_cpus_; bool have_numa_ = false;        // this device's mask (see above)
  cpu_set_t saved_affinity_; pid_t pinned_tid_ = 0;     // the creating thread's affinity before this context narrowed it
  void numa_mask_init_(int device) {
    CPU_ZERO(&numa_cpus_); CPU_ZERO(&saved_affinity_);
    if (getenv("DP_NUMA_PIN") && !atoi(getenv("DP_NUMA_PIN"))) return;
    const cpu_set_t& orig = process_affinity_();
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char* c = bdf; *c; c++) *c = (char)tolower(*c);
    std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t local; CPU_ZERO(&local);
    for (char* q = line; *q && *q != '\n';) {  // "64-127,192-255"
      char* e = nullptr;
      long a = strtol(q, &e, 10); if (e == q) break;
      long b = a;
      if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); if (e == q) break; }
      for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) CPU_SET((int)c, &local);
      q = *e == ',' ? e + 1 : e;
      if (*e != ',') break;
    }
    CPU_AND(&numa_cpus_, &orig, &local);
    const int nb = CPU_COUNT(&numa_cpus_), nc = CPU_COUNT(&orig);
    have_numa_ = nb > 0 && nb < nc;
    if (have_numa_ && g_timing_level) fprintf(stderr, "[dp] device %d: host threads of this context stay on the %d CPUs of the GPU's NUMA node (%s: %s)\n", device, nb, bdf, strtok(line, "\n"));
  }
  // the creating thread of an owning context: narrowed now, restored by the destructor
  void numa_pin_creator_() {
    if (!have_numa_) return;
    cpu_set_t cur; CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0 || CPU_EQUAL(&cur, &numa_cpus_)) return;
    if (sched_setaffinity(0, sizeof(numa_cpus_), &numa_cpus_) == 0) { saved_affinity_ = cur; pinned_tid_ = (pid_t)syscall(SYS_gettid); }
  }
  void numa_unpin_creator_() {
    if (!pinned_tid_) return;
    (void)sched_setaffinity(pinned_tid_, sizeof(saved_affinity_), &saved_affinity_);  // (ESRCH when the thread is gone: nothing to restore)
    pinned_tid_ = 0;
  }
  // a thread the LIBRARY owns (it ends with the call that spawned it) keeps to this device's node
  void pin_thread() override { if (have_numa_) (void)sched_setaffinity(0, sizeof(numa_cpus_), &numa_cpus_); }
  static bool cpu_can_write_(void* p) {
    int fd[2];
    if (pipe(fd) != 0) return false;
    unsigned long long v = 0; bool ok = false;
    if (write(fd[1], &v, 8) == 8) ok = read(fd[0], p, 8) == 8;  // copy_to_user: EFAULT instead of a fault when the page is not mapped writable
    close(fd[0]); close(fd[1]);
    return ok;
  }
  const unsigned long long* mailbox_dev() {
    if (vmail_tried_ || !zerocopy_ || throughput_mode_ || sess_.active) return hmail_dev_;  // (a worker of a cohort keeps the sponge on the device: no host-driven sessions worth the page)
    vmail_tried_ = true;
    static std::atomic<int> works{0};  // process-wide: -1 = a context found it not to work
    if (works.load() < 0 || (getenv("DP_MAILBOX_VRAM") && !atoi(getenv("DP_MAILBOX_VRAM")))) return hmail_dev_;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) != hipSuccess || !p) { (void)hipGetLastError(); works = -1; return hmail_dev_; }
    bool ok = cpu_can_write_((char*)p + 2048);
    for (unsigned long long probe = 0x5EED0001ull; ok && probe <= 0x5EED0002ull; probe++) {  // the second value: no cached copy of the first is served
      ((volatile unsigned long long*)p)[16] = probe;
      std::atomic_thread_fence(std::memory_order_seq_cst);
      unsigned long long seq = ++seq_;
      nb_ = 0; DPL(k_publish, dim3(1), dim3(64), (const u64*)p + 16, hres_dev_, (size_t)1, hflag_dev_, seq);
      wait_flag(seq, 1);
      ok = hres_[0] == probe;
    }
    if (!ok) { (void)hipGetLastError(); hipFree(p); works = -1; if (g_timing_level) fprintf(stderr, "[dp] challenge mailbox stays in host memory (device memory is not CPU-writable here)\n"); return hmail_dev_; }
    works = 1;
    vmail_ = (unsigned long long*)p;
    vmail_[0] = vmail_[1] = vmail_[2] = vmail_[3] = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    hmail_ = hmail_dev_ = vmail_;
    if (g_timing_level) fprintf(stderr, "[dp] challenge mailbox in device memory (CPU writes through the BAR)\n");
    return hmail_dev_;
  }
  void post_challenge(Ext r) {
    hmail_[1] = r.c0; hmail_[2] = r.c1; hmail_[3] = pub_mix(sess_.seq) + r.c0 + 2 * r.c1;  // tag: the device re-polls a torn payload
    std::atomic_thread_fence(std::memory_order_release);
    *(volatile unsigned long long*)hmail_ = sess_.seq;
    std::atomic_thread_fence(std::memory_order_seq_cst);
  }
  static constexpr size_t SC_SMALL_MAX = 8192;  // tables up to this length (after the fold) take the one-launch path
  const Ext* claim_hint_ = nullptr;  // set for the duration of sc_round_claim: the fused streaming path may skip t = 1
  // "grid2" (round 6; kernels.inc k_sc_terms2 / k_sc_fused2): the first two rounds of one product of <= 3 large BASE tables from ONE pass over them, both folds in one more.
  // stage 1: the sixteen grid sums are here and round 1 has been answered, the tables are still whole; stage 2: round 2 has been answered from the grid at r1, both folds are due.
  // One proof on the GPU only (latency mode, in-kernel ticket); DP_SC_GRID2=0 keeps the round-by-round form. 2^24 sumcheck 0.985 -> see profiles/r06_sumcheck_grid2_ab.txt.
  struct Grid2 { int stage = 0; int nt = 0; size_t n = 0; u64 A[16]; Ext r1; } grid2_;  // A[4 i + j]: the sums at the i-th point of x1 and the j-th of x2, points (0, 1, oo, -1)
  // coefficients of the polynomial of degree K <= 3 with the values v[0], v[1] at 0 and 1, the leading coefficient v[2] ("oo") and — needed for K = 3 only — the value v[3] at -1
  static void grid2_coeffs(int K, const Ext v[4], Ext c[4]) {
    c[0] = v[0]; c[1] = c[2] = c[3] = ex_zero();
    if (K == 1) c[1] = ex_sub(v[1], v[0]);
    else if (K == 2) { c[2] = v[2]; c[1] = ex_sub(ex_sub(v[1], v[0]), v[2]); }
    else {
      static const u64 half = gl_inv(2);
      const Ext s = ex_mul_base(ex_add(v[1], v[3]), half), d = ex_mul_base(ex_sub(v[1], v[3]), half);  // c0 + c2, c1 + c3
      c[3] = v[2]; c[2] = ex_sub(s, v[0]); c[1] = ex_sub(d, v[2]);
    }
  }
  static Ext grid2_eval(const Ext c[4], Ext x) { return ex_add(ex_mul(ex_add(ex_mul(ex_add(ex_mul(c[3], x), c[2]), x), c[1]), x), c[0]); }
  static constexpr size_t GRID2_MIN_N = size_t(1) << 20;
  void sc_round_claim(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, int nterms, const Ext* claim, Ext* out) override {
    claim_hint_ = nterms == 1 ? claim : nullptr;
    try { sc_round(tabs, nt, r, terms, nterms, out); } catch (...) { claim_hint_ = nullptr; throw; }
    claim_hint_ = nullptr;
  }
  // Every remaining round of a sumcheck in ONE launch with the Fiat-Shamir transcript on the device (ScFsArgs above): used
  // when several proofs are in flight (DP_DEVICE_FS: 0 never, 1 always, default: throughput mode only — alone, the host
  // sponge on a 5 GHz core plus two PCIe hops is faster per round than the lane-parallel permutation of one wave).
  int devfs_env_ = [] { const char* e = getenv("DP_DEVICE_FS"); return e ? atoi(e) : -1; }();
  bool devfs_ = devfs_env_ > 0;
  size_t nfs_ = 0;
  bool sc_tail(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned md, Challenger& ch,
               std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& point, Ext* finals) override {
    // latency mode (one proof on the GPU) keeps the host sponge: a round trip to the host costs 23-35 us per round, the 8-lane
    // wave sponge ~50-70 us (3-4 permutations of ~16 us: one wave on a dependent chain) — measured on the 2^24 sumcheck, whose
    // 14-round hand-over phase took 0.49 ms with the host sponge and 0.99 ms with the device sponge of round 2 (profiles/r02_sumcheck24_handover.txt)
    if (!devfs_ || !persist_ || !zerocopy_ || sess_.active) return false;
    if (nt > MAX_TABS || nterms > MAX_TERMS || nt <= 0 || nterms <= 0 || md < 1 || md > (unsigned)SC_MAXK) return false;
    size_t n_in = tabs[0].n;
    for (int i = 0; i < nt; i++) if (tabs[i].n != n_in) return false;
    size_t n_after = r ? n_in / 2 : n_in;
    if (n_after > SC_PERSIST_MAX || n_after < 4 || (n_after & (n_after - 1))) return false;
    bool hi = false;
    for (int i = 0; i < nterms; i++) { if (terms[i].k < 1 || terms[i].k > SC_MAXK || (unsigned)terms[i].k > md) return false; hi = hi || terms[i].k > 3; }
    unsigned rounds = 0; for (size_t m = n_after; m > 1; m >>= 1) rounds++;
    const size_t nwords = (size_t)rounds * (md + 2) * 2 + (size_t)nt * 2 + 14;
    if (nwords > RES_WORDS) return false;
    ScPersistArgs a;
    a.eq_tab = -1; a.eq_k = 0;
    if (pend_eq_.p && r) flush_pending_eq();
    if (pend_eq_.p) {
      for (int i = 0; i < nt; i++) if (tabs[i].p == pend_eq_.p && tabs[i].n == (size_t(1) << pend_eq_.k)) a.eq_tab = i;
      if (a.eq_tab >= 0) { a.eq_k = (int)pend_eq_.k; for (unsigned i = 0; i < pend_eq_.k; i++) a.eq_pt[i] = pend_eq_.pt[i]; pend_eq_.p = nullptr; }
      else flush_pending_eq();
    }
    for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
    for (int i = 0; i < MAX_TERMS; i++) { a.k[i] = 1; a.off[i] = 0; for (int j = 0; j < SC_MAXK; j++) a.t[i][j] = 0; }
    { int o = 0; for (int i = 0; i < nterms; i++) { a.k[i] = terms[i].k; for (int j = 0; j < SC_MAXK; j++) a.t[i][j] = j < terms[i].k ? terms[i].t[j] : 0; a.off[i] = o; o += terms[i].k + 1; } }
    const size_t lds = (size_t)nt * (n_in / 2) * 16;
    const size_t msg_lds = (nwords * 8 + 15) & ~size_t(15);  // the message is assembled in LDS (kernels.inc: MSG_PUT / msg_flush)
    const bool in_lds = lds + msg_lds <= sc_lds_max();
    size_t first = r ? n_after : n_after / 2;
    double tab_bytes = 0;
    for (int i = 0; i < nt; i++) {
      a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
      if (!in_lds) { a.bufA[i] = (Ext*)alloc(first, true).p; a.bufB[i] = (Ext*)alloc(std::max<size_t>(first / 2, 1), true).p; }
      tab_bytes += (double)n_in * (tabs[i].ext ? 16.0 : 8.0);
    }
    a.nwg = 1; a.rounds_a = 0; a.slot_ext = 0;
    a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = scdbg_;
    const ScFsArgs* fsd = nullptr;
    ScFsArgs* f = desc_alloc<ScFsArgs>(1, &fsd);
    for (int i = 0; i < 8; i++) f->state[i] = ch.state[i];
    for (int i = 0; i < 4; i++) f->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
    f->in_len = ch.in_len; f->out_len = ch.out_len; f->md = (int)md; f->rounds = (int)rounds;
    { static const char lab[] = "Internal round"; size_t n = sizeof(lab) - 1; f->nlabel = 0; f->pad = 0; f->label[0] = f->label[1] = 0;
      for (size_t i = 0; i < n; i += 8) { u64 v = 0; size_t m = n - i < 8 ? n - i : 8; for (size_t q = 0; q < m; q++) v |= (u64)(uint8_t)lab[i + q] << (8 * q); f->label[f->nlabel++] = gl_from_u64(v); } }
    for (int i = 0; i < MAX_TERMS; i++) f->coeff[i] = i < nterms ? coeffs[i] : ex_zero();
    unsigned long long seq = ++seq_;
    size_t work = (size_t)nterms * (n_after / 2) + (size_t)nt * n_after / 4;
    int threads = persist_threads(work);
    if (in_lds) { nb_ = tab_bytes; DPL_ONE_HI(k_sc_persist_lds, hi, dim3(1), threads, lds + msg_lds, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, seq - 1, fsd); }
    else { nb_ = tab_bytes + 24.0 * (double)n_in * nt; DPL_ONE_HI(k_sc_persist, hi, dim3(1), threads, msg_lds, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, seq - 1, fsd); }
    wait_flag(seq, nwords);
    const u64* w = hres_;
    for (unsigned q = 0; q < rounds; q++) {
      std::vector<Ext> m(md + 1);
      for (unsigned j = 0; j <= md; j++) { size_t o = ((size_t)q * (md + 1) + j) * 2; m[j] = ex(w[o], w[o + 1]); }
      msgs.push_back(std::move(m));
    }
    for (unsigned q = 0; q < rounds; q++) { size_t o = ((size_t)rounds * (md + 1) + q) * 2; point.push_back(ex(w[o], w[o + 1])); }
    size_t wf = (size_t)rounds * (md + 2) * 2;
    for (int i = 0; i < nt; i++) finals[i] = ex(w[wf + 2 * i], w[wf + 2 * i + 1]);
    size_t ws = wf + 2 * (size_t)nt;
    for (int i = 0; i < 8; i++) ch.state[i] = w[ws + i];
    ch.in_len = (int)w[ws + 12]; ch.out_len = (int)w[ws + 13];
    for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[ws + 8 + i]; ch.out_buf[i] = ch.state[i]; }
    nfs_++;
    return true;
  }
  // ---- Dev::logup_tail: DP_DEVICE_LOGUP=1, see k_logup_tail. Declines (returns false) whenever the shape is
  // outside what the kernel was written for; the caller then runs the layers one by one (logup_layers).
  // fused protocol kernels: on by default whenever the device-side transcript is (throughput mode); DP_FUSED_OFF=<x,y> / DP_DEVICE_LOGUP=0|1 turn them off.
  // Validated on MI355X in round 2 (tests/test_gpu_fused.py: every knob alone and all together, Dense-4M and CNN-264k batches
  // against the sequential proofs; profiles/r02_fused_knob_sweep.jsonl).
  static int knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
  // dynamic LDS of a fused protocol kernel = its message (assembled in LDS, sent in one burst: kernels.inc MSG_PUT / msg_flush); 0: too long for that, the caller declines
  static constexpr size_t MSG_LDS_MAX = 48 * 1024;
  static size_t msg_lds(const std::vector<size_t>& blocks) { size_t n = 0; for (size_t b : blocks) n += b; const size_t bytes = (n * 8 + 15) & ~size_t(15); return bytes <= MSG_LDS_MAX ? bytes : 0; }
  // DP_FUSED_OFF=<comma-separated subset of classic,commit,deleg,dense,eqsum>: those fused protocol kernels decline and their stretch runs launch by launch
  // (tests/test_gpu_fused.py turns each off alone and all together); the logup kernel has its own three-way switch DP_DEVICE_LOGUP
  static bool fused_on(const char* what) { const char* e = getenv("DP_FUSED_OFF"); if (!e) return true; const std::string s = std::string(",") + e + ","; return s.find(std::string(",") + what + ",") == std::string::npos; }
  // ---- DP_HOST_SPONGE=1 (sponge_host.h): the fused protocol kernels keep the transcript's sponge on the HOST — a kernel posts the
  // words it absorbs and asks for challenges through a mapped mailbox, DP_SPONGE_THREADS server threads answer (the members of a
  // cohort ask together and sit with different servers). Off by default: new at the end of round 2 (the wave sponge costs ~12 us per permutation, the mailbox 2.9 us per round trip).
  bool host_sponge_ = knob("DP_HOST_SPONGE", 0) != 0;
  SpongeSlot* sp_slot_ = nullptr; u64* hsp_ = nullptr; u64* hsp_dev_ = nullptr; bool sp_active_ = false;
  void sponge_disarm_() {
    if (!sp_active_) return;
    while (sp_slot_->busy.load(std::memory_order_acquire)) __builtin_ia32_pause();
    sp_slot_->active.store(0, std::memory_order_release);
    while (sp_slot_->busy.load(std::memory_order_acquire)) __builtin_ia32_pause();  // a server that had passed the first check
    sponge_nactive().fetch_sub(1);
    sp_slot_->ch = nullptr; sp_active_ = false;
  }
  // between the descriptor fill and the launch: point the kernel at this context's mailbox and the service at the transcript; after the
  // wait the transcript is final (done()); the parse functions overwrite it with the kernel's unused sponge words (restore())
  struct SpongeArm {
    HipDev* dev; Challenger* ch; bool armed = false; Challenger fin;
    template <class D> SpongeArm(HipDev* dv, D* d, Challenger& c) : dev(dv), ch(&c) {
      if (!dv->host_sponge_ || !dv->sp_slot_) return;
      d->sp_req = dv->hsp_dev_; d->sp_rep = dv->hsp_dev_ + WC_REQ_WORDS; d->sp_seq = dv->sp_slot_->served;
      sponge_servers_start();
      dv->sp_slot_->ch = &c; dv->sp_slot_->active.store(1, std::memory_order_release); sponge_nactive().fetch_add(1); dv->sp_active_ = true;
      armed = true;
    }
    void done() { if (armed) { dev->sponge_disarm_(); fin = *ch; } }
    void restore() { if (armed) { *ch = fin; armed = false; } }
    ~SpongeArm() { if (armed) dev->sponge_disarm_(); }
  };
  bool devlogup_ = knob("DP_DEVICE_LOGUP", 2) == 1;
  size_t nlogup_tail_ = 0;
  bool logup_tail(const LogupTailArgs& a, Challenger& ch, std::vector<std::vector<std::vector<Ext>>>& layer_msgs,
                  std::vector<std::vector<Ext>>& layer_points, std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) override {
    if (!devlogup_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!logup_tail_accepts(a)) return false;
    const std::vector<size_t> blocks = logup_tail_blocks(a);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    if (nwords > RES_WORDS || nwords * 8 > MSG_LDS_MAX) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const LogupTailDesc* dd = nullptr;
    LogupTailDesc* d = desc_alloc<LogupTailDesc>(1, &dd);
    logup_tail_fill(d, a, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (const LogupCircuitDev& c : *a.circuits) for (const DBuf& l : c.den) nb_ += 2.0 * 16.0 * (double)l.n;
    DPL_ONE(k_logup_tail, dim3(1), 1024, (size_t)d->lds_ext * 16 + ((nwords * 8 + 15) & ~size_t(15)), dd, (u64*)hres_dev_, hflag_dev_, seq);  // (table slots + the message, assembled in LDS)
    wait_flag_blocks(seq, blocks);
    sponge.done();
    logup_tail_parse(hres_, a, blocks, ch, layer_msgs, layer_points, round_evals, point);
    sponge.restore();
    release(mk);
    nlogup_tail_++;
    return true;
  }
  // ---- Dev::commit_tail: DP_DEVICE_COMMIT (default on): k_commit_tail, the last rounds of the Basefold commit phase
  bool devcommit_ = fused_on("commit");
  bool commit_tail(const CommitTailArgs& a, Challenger& ch, CommitTailOut& out) override {
    if (!devcommit_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_ || !tw_) return false;
    if (!commit_tail_accepts(a, commit_tail_max_n(throughput_mode_)) || dp_ceil_log2(a.folded.n) - 1 > L_) return false;
    const std::vector<size_t> blocks = commit_tail_blocks(a);
    if (blocks[0] + blocks[1] > RES_WORDS || !msg_lds(blocks)) return false;
    const CommitTailDesc* dd = nullptr;
    CommitTailDesc* d = desc_alloc<CommitTailDesc>(1, &dd);
    std::vector<DevTree> trees;
    CommitTailDesc fill;
    commit_tail_fill(&fill, a, ch, *this, (const u64*)tw_, L_, trees);  // (the trees stay allocated: the query phase reads them)
    memcpy((void*)d, &fill, sizeof(CommitTailDesc));
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 16.0 * (double)a.folded.n * 2.0 + 32.0 * (double)a.sum_evals.n;
    DPL_ONE(k_commit_tail, dim3(1), 1024, msg_lds(blocks), dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    commit_tail_parse(hres_, a, ch, trees, out);
    sponge.restore();
    return true;
  }
  // ---- Dev::eqsum_tail: DP_DEVICE_EQSUM (default on): k_eqsum_tail, eq tables + accumulation sumcheck in one launch
  bool deveqsum_ = fused_on("eqsum");
  bool eqsum_tail(const EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned nv, unsigned md,
                  Challenger& ch, EqSumOut& out) override {
    if (!deveqsum_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!eqsum_tail_accepts(jobs, njobs, tabs, ntabs, terms, nterms, nv, md)) return false;
    const std::vector<size_t> blocks = eqsum_tail_blocks(ntabs, nv, md);
    if (blocks[0] + blocks[1] > RES_WORDS || !msg_lds(blocks)) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const EqSumDesc* dd = nullptr;
    EqSumDesc* d = desc_alloc<EqSumDesc>(1, &dd);
    eqsum_tail_fill(d, jobs, njobs, tabs, ntabs, terms, coeffs, nterms, nv, md, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (int i = 0; i < ntabs; i++) nb_ += (double)tabs[i].bytes();
    DPL_ONE(k_eqsum_tail, dim3(1), 1024, msg_lds(blocks), dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    eqsum_tail_parse(hres_, ntabs, nv, md, ch, out);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::deleg_tail: DP_DEVICE_DELEG (default on): k_deleg_tail, all delegation sumchecks of one batch FFT / iFFT in one launch
  bool devdeleg_ = fused_on("deleg");
  bool deleg_tail(const DelegTailArgs& a, Challenger& ch, DelegTailOut& out) override {
    if (!devdeleg_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!deleg_tail_accepts(a)) return false;
    const size_t fm = a.f_middle->size();
    const std::vector<size_t> blocks = deleg_tail_blocks(fm);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    if (nwords > RES_WORDS || !msg_lds(blocks)) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const std::vector<u64> words = deleg_tail_stage(a);
    DBuf staged = alloc(words.size(), false);
    upload(staged, words.data());
    const DelegDesc* dd = nullptr;
    DelegDesc* d = desc_alloc<DelegDesc>(1, &dd);
    deleg_tail_fill(d, a, staged, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 8.0 * (double)words.size();
    DPL_ONE(k_deleg_tail, dim3(1), 1024, msg_lds(blocks), dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    deleg_tail_parse(hres_, fm, ch, out);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::dense_tail: DP_DEVICE_DENSE (default on): k_dense_tail, a Dense layer's device work in one launch
  bool devdense_ = fused_on("dense");
  bool dense_tail(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, Challenger& ch, DenseTailOut& out) override {
    if (!devdense_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!dense_tail_accepts(bias, W, R, C, in)) return false;
    const std::vector<size_t> blocks = dense_tail_blocks(C);
    flush_pending_eq();
    const size_t mk = mark();
    const DenseTailDesc* dd = nullptr;
    DenseTailDesc* d = desc_alloc<DenseTailDesc>(1, &dd);
    dense_tail_fill(d, bias, W, R, C, in, pt, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 8.0 * (double)R * (double)C + 16.0 * (double)R + 16.0 * (double)C * 4.0;
    DPL_ONE(k_dense_tail, dim3(1), 1024, msg_lds(blocks), dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    dense_tail_parse(hres_, C, ch, out);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::classic_tail: DP_DEVICE_CLASSIC (default on): k_classic_tail, the last rounds of the batch-opening sumcheck
  bool devclassic_ = fused_on("classic");
  bool classic_tail(const ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) override {
    if (!devclassic_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!classic_tail_accepts(a)) return false;
    const std::vector<size_t> blocks = classic_tail_blocks(a);
    if (blocks[0] + blocks[1] > RES_WORDS || !msg_lds(blocks)) return false;
    const size_t mk = mark();
    const ClassicTailDesc* dd = nullptr;
    ClassicTailDesc* d = desc_alloc<ClassicTailDesc>(1, &dd);
    classic_tail_fill(d, a, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (int i = 0; i < a.np; i++) nb_ += a.fs[i].bytes() + a.eqs[i].bytes();
    DPL_ONE(k_classic_tail, dim3(1), 1024, msg_lds(blocks), dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    classic_tail_parse(hres_, a, ch, msgs, challenges);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::logup_full: DP_DEVICE_LOGUP=2 (the default): k_logup_tail in full mode — one launch and one device wait per
  // logup-GKR batch proof
  bool devlogup_full_ = knob("DP_DEVICE_LOGUP", 2) == 2;
  size_t logup_wide_n_ = (size_t)knob("DP_LOGUP_WIDE_N", 2048);  // lookups of at least this many rows take the 512-thread form in throughput mode (0: never)
  bool logup_full(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, Challenger& ch, LogupFullOut& out) override {
    if (!devlogup_full_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    size_t n = 0;
    if (!logup_full_accepts(cols, cpi, ninst, mult, &n)) return false;
    const std::vector<size_t> blocks = logup_full_blocks(n, cpi, ninst, !mult.null());
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    if (nwords > RES_WORDS || nwords * 8 > MSG_LDS_MAX) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const LogupTailDesc* dd = nullptr;
    LogupTailDesc* d = desc_alloc<LogupTailDesc>(1, &dd);
    logup_full_fill(d, cols, cpi, ninst, mult, c, chi, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = (double)ninst * (8.0 * cpi * n + 16.0 * 3 * n) * 2.0;
    DPL_ONE_W(k_logup_tail, logup_wide_n_ && n >= logup_wide_n_, dim3(1), 1024, (size_t)d->lds_ext * 16 + ((nwords * 8 + 15) & ~size_t(15)), dd, (u64*)hres_dev_, hflag_dev_, seq);  // (table slots + the message, assembled in LDS)
    wait_flag_blocks(seq, blocks);
    sponge.done();
    logup_full_parse(hres_, n, cpi, ninst, !mult.null(), blocks, ch, out);
    sponge.restore();
    release(mk);
    nlogup_tail_++;
    return true;
  }
  void sc_round(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, int nterms, Ext* out) override {
    DP_REQUIRE(nt <= MAX_TABS && nterms <= MAX_TERMS && nt > 0 && nterms > 0, DP_ERR_SHAPE, "sumcheck: too many tables/terms for one launch");
    size_t n_in = tabs[0].n;
    for (int i = 0; i < nt; i++) DP_REQUIRE(tabs[i].n == n_in, DP_ERR_SHAPE, "sumcheck: tables must have equal length");
    size_t n_after = r ? n_in / 2 : n_in;
    DP_REQUIRE(n_after >= 2, DP_ERR_SHAPE, "sumcheck: tables must keep length >= 2");
    // (n_in and r are rewritten below when a multi-workgroup phase hands its folded tables to a single-workgroup session)
    size_t nraw = 0;  // a degree-k term contributes k + 1 values; single-workgroup kernels publish them packed
    for (int i = 0; i < nterms; i++) { DP_REQUIRE(terms[i].k >= 1 && terms[i].k <= SC_MAXK, DP_ERR_SHAPE, "sumcheck: term degree must be 1..5"); nraw += terms[i].k + 1; }
    bool hi = false; for (int i = 0; i < nterms; i++) hi = hi || terms[i].k > 3;
    auto read_terms = [&]() { for (size_t o = 0; o < nraw; o++) out[o] = ex(hres_[2 * o], hres_[2 * o + 1]); };
    auto fill_terms = [&](int (*tk), int (*tt)[SC_MAXK], int* toff) {
      for (int i = 0; i < MAX_TERMS; i++) { tk[i] = 1; for (int j = 0; j < SC_MAXK; j++) tt[i][j] = 0; if (toff) toff[i] = 0; }
      int o = 0;
      for (int i = 0; i < nterms; i++) { tk[i] = terms[i].k; for (int j = 0; j < SC_MAXK; j++) tt[i][j] = j < terms[i].k ? terms[i].t[j] : 0; if (toff) toff[i] = o; o += terms[i].k + 1; }
    };
    auto read_shares = [&](int G, size_t slot_words) {  // the round sums are the sums of the workgroups' shares
      for (size_t o = 0; o < nraw; o++) {
        Ext acc = ex_zero();
        for (int g = 0; g < G; g++) { const u64* w = hres_ + (size_t)g * slot_words; acc = ex_add(acc, ex(w[2 * o], w[2 * o + 1])); }
        out[o] = acc;
      }
    };
    if (!r) grid2_.stage = 0;  // (a sumcheck abandoned between its first rounds leaves nothing behind)
    if (grid2_.stage) {
      DP_REQUIRE(r && nt == grid2_.nt && n_in == grid2_.n && nterms == 1 && terms[0].k == nt && !sess_.active, DP_ERR_ARG, "sumcheck out of sync (two-round grid)");
      const int K = nt;
      if (grid2_.stage == 1) {  // round 2 from the grid: g(X2) = Q(r1, X2) from its values at 0, 1, -1 and its leading coefficient, each Q(., p2) evaluated at r1 — no launch, nothing folded yet
        Ext gv[4], c[4];
        for (int p2 = 0; p2 < 4; p2++) {
          const Ext col[4] = {ex_base(grid2_.A[0 + p2]), ex_base(grid2_.A[4 + p2]), ex_base(grid2_.A[8 + p2]), ex_base(grid2_.A[12 + p2])};
          grid2_coeffs(K, col, c);
          gv[p2] = grid2_eval(c, *r);
        }
        grid2_coeffs(K, gv, c);
        for (int t = 0; t <= K; t++) out[t] = grid2_eval(c, ex_from_u64((u64)t));
        grid2_.r1 = *r; grid2_.stage = 2;
        return;
      }
      // stage 2: both folds and the sums of round 3 in one pass over the base tables
      grid2_.stage = 0;
      const void* in[3] = {nullptr, nullptr, nullptr}; Ext* outp[3] = {nullptr, nullptr, nullptr};
      const size_t n_out = n_in / 4;
      double bytes = 0;
      for (int j = 0; j < nt; j++) {
        int ti = terms[0].t[j];
        DP_REQUIRE(!tabs[ti].ext, DP_ERR_ARG, "sumcheck out of sync (two-round grid: folded tables)");
        DBuf o = alloc(n_out, true);
        in[j] = tabs[ti].p; outp[j] = (Ext*)o.p;
        bytes += tabs[ti].bytes() + o.bytes();
        tabs[ti] = o;
      }
      const size_t nocts = n_in / 8;
      size_t mk = mark();
      int g = grid_for(nocts, 1024);  // (4096 -> 1024: 154 -> 136 us at 3 x 2^24 entries; 512: 152 — tools/r06/call47.sh)
      Ext* partial = (Ext*)arena_alloc((size_t)g * 4 * 16);
      nb_ = bytes;
      const bool skip1 = claim_hint_ != nullptr;
      const unsigned long long fseq = ++seq_;
      #define LAUNCH_F2(KK) do { if (skip1) DPL_B((k_sc_fused2<KK, true>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nocts, grid2_.r1, *r, partial, fused_ticket_, (Ext*)hres_dev_, hflag_dev_, fseq); \
                                 else DPL_B((k_sc_fused2<KK, false>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nocts, grid2_.r1, *r, partial, fused_ticket_, (Ext*)hres_dev_, hflag_dev_, fseq); } while (0)
      if (nt == 1) LAUNCH_F2(1); else if (nt == 2) LAUNCH_F2(2); else LAUNCH_F2(3);
      #undef LAUNCH_F2
      wait_flag(fseq, 8);
      for (int t = 0; t <= K; t++) out[t] = ex(hres_[2 * t], hres_[2 * t + 1]);
      if (skip1) out[1] = ex_sub(*claim_hint_, out[0]);  // s(0) + s(1) = claim, exactly
      release(mk);
      return;
    }
    {
      static const bool grid2_env = !(getenv("DP_SC_GRID2") && !atoi(getenv("DP_SC_GRID2")));
      bool take = grid2_env && !r && !throughput_mode_ && !share_x_ && zerocopy_ && !queued_() && fused_ticket_ != nullptr && !sess_.active && !pend_eq_.p
                  && nterms == 1 && terms[0].k == nt && nt <= 3 && n_in >= GRID2_MIN_N;
      for (int i = 0; take && i < nt; i++) { take = !tabs[i].ext; for (int j = 0; j < i; j++) take = take && terms[0].t[i] != terms[0].t[j]; }
      if (take) {
        const void* in[3] = {nullptr, nullptr, nullptr};
        double bytes = 0;
        for (int j = 0; j < nt; j++) { in[j] = tabs[terms[0].t[j]].p; bytes += tabs[terms[0].t[j]].bytes(); }
        const size_t nquads = n_in / 4;
        size_t mk = mark();
        int g = grid_for(nquads, 768);  // (chip-sized: kernels.inc, k_sc_terms2)
        Ext* partial = (Ext*)arena_alloc((size_t)g * 8 * 16);
        nb_ = bytes;
        const unsigned long long fseq = ++seq_;
        if (nt == 1) DPL_B((k_sc_terms2<1>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], nquads, partial, fused_ticket_, (Ext*)hres_dev_, hflag_dev_, fseq);
        else if (nt == 2) DPL_B((k_sc_terms2<2>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], nquads, partial, fused_ticket_, (Ext*)hres_dev_, hflag_dev_, fseq);
        else DPL_B((k_sc_terms2<3>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], nquads, partial, fused_ticket_, (Ext*)hres_dev_, hflag_dev_, fseq);
        wait_flag(fseq, 16);
        for (int i = 0; i < 16; i++) grid2_.A[i] = hres_[i];
        release(mk);
        {  // s1(t) = Q(t, 0) + Q(t, 1): the two polynomials in X1 from their values at 0, 1, -1 and their leading coefficients
          Ext c0[4], c1[4];
          const Ext col0[4] = {ex_base(grid2_.A[0]), ex_base(grid2_.A[4]), ex_base(grid2_.A[8]), ex_base(grid2_.A[12])}, col1[4] = {ex_base(grid2_.A[1]), ex_base(grid2_.A[5]), ex_base(grid2_.A[9]), ex_base(grid2_.A[13])};
          grid2_coeffs(nt, col0, c0); grid2_coeffs(nt, col1, c1);
          for (int t = 0; t <= nt; t++) out[t] = ex_add(grid2_eval(c0, ex_from_u64((u64)t)), grid2_eval(c1, ex_from_u64((u64)t)));
        }
        grid2_.stage = 1; grid2_.nt = nt; grid2_.n = n_in;
        return;
      }
    }
    if (sess_.active && sess_.multi) {  // multi-workgroup phase: every workgroup folds its slice with this challenge
      DP_REQUIRE(r && nt == sess_.ntabs && n_in == sess_.n, DP_ERR_ARG, "sumcheck session out of sync");
      post_challenge(*r);
      sess_.folds++;
      const size_t lvl_off = sess_.n0 - (sess_.n0 >> (sess_.folds - 1 + sess_.shift));  // level j starts at n0 (1 - 2^-(j-1)); one level later when the phase began with a fold
      for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i] + lvl_off; tabs[i].n = n_after; tabs[i].ext = true; }
      sess_.n = n_after;
      if (sess_.folds < sess_.rounds_a) {
        wait_flags_multi(++sess_.seq, 2 * nraw, sess_.G, sess_.slot_words);
        read_shares(sess_.G, sess_.slot_words);
        return;
      }
      // the kernel leaves after this fold: the compact folded tables feed a single-workgroup session (stream ordered)
      sess_.active = false; sess_.multi = false;
      r = nullptr; n_in = n_after;
    }
    if (sess_.active) {  // the persistent kernel is waiting for this challenge
      DP_REQUIRE(r && nt == sess_.ntabs && n_in == sess_.n, DP_ERR_ARG, "sumcheck session out of sync");
      post_challenge(*r);
      wait_flag(++sess_.seq, 2 * nraw);
      sess_.n = n_after;
      for (int i = 0; i < nt; i++) { tabs[i].p = sess_.nextA ? sess_.a[i] : sess_.b[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      sess_.nextA = !sess_.nextA;
      read_terms();
      return;
    }
    static const bool multi_mid = getenv("DP_MULTI_MID") && atoi(getenv("DP_MULTI_MID"));  // (experiment, round 6: enter the phase in the middle of a streaming sumcheck — the mailbox is device memory now)
    if (!share_x_ && (!r || (multi_mid && !throughput_mode_)) && multi_ && persist_ && n_after >= MULTI_MIN_N && n_in <= MULTI_MAX_N && nraw * 2 * MULTI_MAX_WG <= RES_WORDS) {
      // ---- multi-workgroup phase: G workgroups own contiguous slices, fold until the tables are MULTI_TARGET_N long. It may
      // begin in the middle of a sumcheck (r given: the streaming rounds of a large sumcheck hand over as soon as the tables
      // fit): every workgroup then first folds its slice with the pending challenge. Measured on the 2^24 sumcheck that costs
      // 75 us per round (32 workgroups x PCIe mailbox) against 23 us for one workgroup in LDS, so a sumcheck only ENTERS this phase at its first round (round 2 measured the mid-sumcheck entry and rejected it):
      // the hand-over from the streaming rounds goes to the device-side transcript instead (sc_tail).
      flush_pending_eq();
      mailbox_dev();
      int G = (int)std::min<size_t>(MULTI_MAX_WG, n_after / 512);
      int rounds_a = (int)(dp_ceil_log2(n_after) - dp_ceil_log2(MULTI_TARGET_N));
      ScPersistArgs a;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
      fill_terms(a.k, a.t, a.off);
      sess_.a.assign(nt, nullptr); sess_.b.assign(nt, nullptr);
      double bytes = 0;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        sess_.a[i] = (Ext*)alloc(n_in, true).p; sess_.b[i] = nullptr;  // every fold level has its own region of this buffer
        a.bufA[i] = sess_.a[i]; a.bufB[i] = nullptr;
        bytes += tabs[i].bytes() + 3.0 * 16.0 * (n_in / 2);  // read once + the halving folded tables written and re-read
      }
      a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = nullptr; a.eq_tab = -1; a.eq_k = 0;
      a.nwg = G; a.rounds_a = rounds_a; a.slot_ext = (int)nraw;
      sess_.active = true; sess_.multi = true; sess_.G = G; sess_.rounds_a = rounds_a; sess_.folds = 0; sess_.shift = r ? 1 : 0; sess_.slot_words = 2 * nraw;
      sess_.ntabs = nt; sess_.n = n_after; sess_.n0 = n_in; sess_.seq = seq_;
      seq_ += (unsigned)rounds_a;  // one publication per round of the phase
      nb_ = bytes; if (hi) { DPL_B((k_sc_persist<true>), LAT_MAXT, KF_CLAIM, dim3(G), dim3(LAT_MAXT), 0, a, (Ext*)hres_dev_, hmflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); } else { DPL_B((k_sc_persist<false>), LAT_MAXT, KF_CLAIM, dim3(G), dim3(LAT_MAXT), 0, a, (Ext*)hres_dev_, hmflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }  // (latency mode only: G whole-CU workgroups)
      wait_flags_multi(++sess_.seq, 2 * nraw, G, sess_.slot_words);
      if (r) for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      read_shares(G, sess_.slot_words);
      return;
    }
    // A sumcheck that arrives here in the middle (r pending: the streaming rounds of a large one are handing over) enters the
    // persistent kernel only once its tables fit in LDS: the global-memory variant costs 35 us per round on 2^15..2^13-entry
    // tables against ~20 us for another streaming / one-launch round (round 2 measured the early hand-over and rejected it).
    const bool lds_fits = (size_t)nt * (n_in / 2) * 16 <= sc_lds_max();
    const bool persist_here = !share_x_ && persist_ && n_after <= SC_PERSIST_MAX && n_after >= 4 && 2 * nraw <= RES_WORDS && (!r || lds_fits || throughput_mode_);
    const bool take_persistent = !sess_.active && persist_here;
    if (pend_eq_.p && !(take_persistent && !r)) flush_pending_eq();
    if (persist_here) {
      if (take_persistent) mailbox_dev();
      ScPersistArgs a;
      a.eq_tab = -1; a.eq_k = 0;
      if (pend_eq_.p) {
        for (int i = 0; i < nt; i++) if (tabs[i].p == pend_eq_.p && tabs[i].n == (size_t(1) << pend_eq_.k)) a.eq_tab = i;
        if (a.eq_tab >= 0) { a.eq_k = (int)pend_eq_.k; for (unsigned i = 0; i < pend_eq_.k; i++) a.eq_pt[i] = pend_eq_.pt[i]; pend_eq_.p = nullptr; }
        else flush_pending_eq();
      }
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
      fill_terms(a.k, a.t, a.off);
      sess_.a.assign(nt, nullptr); sess_.b.assign(nt, nullptr);
      // ping-pong buffers: A takes the first fold output, B the second, A the third, ...
      size_t first = r ? n_after : n_after / 2;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        sess_.a[i] = (Ext*)alloc(first, true).p; sess_.b[i] = (Ext*)alloc(std::max<size_t>(first / 2, 1), true).p;
        a.bufA[i] = sess_.a[i]; a.bufB[i] = sess_.b[i];
      }
      a.nwg = 1; a.rounds_a = 0; a.slot_ext = 0;
      a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = scdbg_;
      sess_.active = true; sess_.ntabs = nt; sess_.n = n_after; sess_.seq = seq_; sess_.nextA = r ? false : true;
      // reserve the sequence numbers of all rounds + the final message
      unsigned rounds = 0; for (size_t m = n_after; m > 1; m >>= 1) rounds++;
      seq_ += rounds + 1;
      size_t work = (size_t)nterms * (n_after / 2) + (size_t)nt * n_after / 4;
      int threads = persist_threads(work);
      size_t lds = (size_t)nt * (n_in / 2) * 16;
      double tab_bytes = 0; for (int i = 0; i < nt; i++) tab_bytes += (double)n_in * (tabs[i].ext && !r ? 16.0 : tabs[i].ext ? 16.0 : 8.0);
      // algorithmic HBM bytes of the launch: every table is read once (the LDS variant never touches HBM again; the
      // global variant also writes and re-reads the halving ping-pong buffers: + 3 x 16 B x n/2 per table in total)
      if (lds <= sc_lds_max()) { nb_ = tab_bytes; DPL_ONE_HI(k_sc_persist_lds, hi, dim3(1), threads, lds, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }
      else { nb_ = tab_bytes + 24.0 * (double)n_in * nt; DPL_ONE_HI(k_sc_persist, hi, dim3(1), threads, 0, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }
      wait_flag(++sess_.seq, 2 * nraw);
      if (r) for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      read_terms();
      return;
    }
    // (a single product still 4096+ entries long after the fold, one proof on the GPU: the streaming kernel below does the
    // round on many workgroups in ~15 us; the one-workgroup kernel here needs ~55 us for it)
    const bool stream_instead = !throughput_mode_ && r && nterms == 1 && terms[0].k == nt && nt <= 3 && n_after >= 4096;
    if (zerocopy_ && n_after <= SC_SMALL_MAX && 2 * nraw <= RES_WORDS && !stream_instead) {
      ScSmallArgs a;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.in_ext[i] = 0; }
      fill_terms(a.k, a.t, a.off);
      double bytes = 0;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        if (r) { DBuf o = alloc(n_after, true); a.out[i] = (Ext*)o.p; bytes += tabs[i].bytes() + o.bytes(); tabs[i] = o; }
      }
      for (int i = 0; i < nterms; i++) bytes += 16.0 * n_after * terms[i].k;
      a.ntabs = nt; a.nterms = nterms; a.has_r = r ? 1 : 0; a.n_after = n_after; a.r = r ? *r : ex_zero();
      unsigned long long seq = ++seq_;
      size_t work = (size_t)nterms * (n_after / 2) + (r ? (size_t)nt * n_after / 4 : 0);
      int threads = persist_threads(work);
      if (share_x_) { nb_ = bytes; DPL_ONE_HI(k_sc_small, hi, dim3(1), threads, 0, a, (Ext*)dshare_, (unsigned long long*)(dshare_ + RES_WORDS), seq); share_exchange_(2 * nraw); read_terms(); return; }
      nb_ = bytes; DPL_ONE_HI(k_sc_small, hi, dim3(1), threads, 0, a, (Ext*)hres_dev_, hflag_dev_, seq);
      wait_flag(seq, 2 * nraw);
      read_terms();
      return;
    }
    if (r && nterms == 1 && terms[0].k == nt && nt <= 3 && n_in >= 8) {
      bool uniform = true, distinct = true;
      for (int i = 0; i < nt; i++) { uniform &= tabs[i].ext == tabs[0].ext; for (int j = 0; j < i; j++) distinct &= terms[0].t[i] != terms[0].t[j]; }
      if (uniform && distinct) {
        const void* in[3] = {nullptr, nullptr, nullptr}; Ext* outp[3] = {nullptr, nullptr, nullptr};
        bool base = !tabs[0].ext;
        double bytes = 0;
        for (int j = 0; j < nt; j++) {
          int ti = terms[0].t[j];
          DBuf o = alloc(n_after, true);
          in[j] = tabs[ti].p; outp[j] = (Ext*)o.p;
          bytes += tabs[ti].bytes() + o.bytes();
          tabs[ti] = o;
        }
        size_t nquads = n_in / 4;
        size_t mk = mark();
        // grid: at most 512 workgroups since round 6 (4096 before: the ten rounds of a 2^24 sumcheck between 2^22 and 2^13 entries 0.302 -> 0.251 ms, 1024: 0.269; every workgroup
        // ends with four block reductions, a write-through of its sums and a ticket, and the last one reads all of them past its L2 — tools/r06/call47.sh, profiles/r06_sumcheck_grid2_ab.txt)
        int g = grid_for(nquads, 512);
        Ext* partial = (Ext*)arena_alloc((size_t)g * 4 * 16);
        nb_ = bytes;
        const bool skip1 = claim_hint_ != nullptr;
        // one proof on the GPU: the kernel's last workgroup reduces and publishes (no second launch); cohort members keep the
        // separate reduction (their workgroups of one launch belong to different proofs)
        // (round 2 had this OFF: its agent-scope release fence per workgroup wrote the XCD's whole L2 back 4096 times per launch, 162 / 40 us -> 1122 / 230 us.
        // Round 4 publishes the 64 bytes of block sums write-through instead — kernels.inc, k_sc_fused — and it is the default; DP_FUSED_TICKET=0 restores the
        // separate 14 us reduction launch per round.)
        static const bool ticket_env = !(getenv("DP_FUSED_TICKET") && !atoi(getenv("DP_FUSED_TICKET")));
        const bool inkernel = ticket_env && zerocopy_ && !queued_() && !share_x_ && fused_ticket_ != nullptr;
        unsigned* tick = inkernel ? fused_ticket_ : nullptr;
        const unsigned long long fseq = inkernel ? ++seq_ : 0;
        #define LAUNCH_FUSED2(KK, BB) do { if (skip1) DPL_B((k_sc_fused<KK, BB, true>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nquads, *r, partial, tick, (Ext*)hres_dev_, hflag_dev_, fseq); \
                                           else DPL_B((k_sc_fused<KK, BB, false>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nquads, *r, partial, tick, (Ext*)hres_dev_, hflag_dev_, fseq); } while (0)
        #define LAUNCH_FUSED(KK) do { if (base) LAUNCH_FUSED2(KK, true); else LAUNCH_FUSED2(KK, false); } while (0)
        if (nt == 1) LAUNCH_FUSED(1); else if (nt == 2) LAUNCH_FUSED(2); else LAUNCH_FUSED(3);
        #undef LAUNCH_FUSED
        #undef LAUNCH_FUSED2
        if (inkernel) wait_flag(fseq, 8); else reduce_publish(partial, (size_t)g, 4, 4);
        for (int t = 0; t <= terms[0].k; t++) out[t] = ex(hres_[2 * t], hres_[2 * t + 1]);
        if (skip1) out[1] = ex_sub(*claim_hint_, out[0]);  // s(0) + s(1) = claim, exactly
        release(mk);
        return;
      }
    }
    if (r) fold_tables(tabs, nt, *r);
    size_t n = tabs[0].n;
    TermArgs a;
    for (int i = 0; i < MAX_TABS; i++) { a.tab[i] = nullptr; a.ext[i] = 0; }
    for (int i = 0; i < nt; i++) { a.tab[i] = tabs[i].p; a.ext[i] = tabs[i].ext; }
    fill_terms(a.k, a.t, nullptr);
    a.npairs = n / 2;
    size_t mk = mark();
    int g = grid_for(a.npairs, 2048);
    DP_REQUIRE(nterms * SC_SLOTS <= 1024, DP_ERR_SHAPE, "sumcheck: too many terms for one reduction");
    Ext* partial = (Ext*)arena_alloc((size_t)nterms * g * SC_SLOTS * 16);
    nb_ = [&] { double b = 0; for (int i = 0; i < nterms; i++) for (int j = 0; j < terms[i].k; j++) b += tabs[terms[i].t[j]].bytes(); return b; }(); DPL_HI(k_sc_terms, hi, dim3(g, nterms), dim3(TPB), a, partial);
    reduce_publish(partial, (size_t)g, SC_SLOTS, nterms * SC_SLOTS);
    size_t o = 0;
    for (int i = 0; i < nterms; i++)
      for (int t = 0; t <= terms[i].k; t++) out[o++] = ex(hres_[(i * SC_SLOTS + t) * 2], hres_[(i * SC_SLOTS + t) * 2 + 1]);
    release(mk);
  }
  void sc_finish(DBuf* tabs, int nt, Ext r, Ext* finals) override {
    DP_REQUIRE(nt <= MAX_TABS, DP_ERR_SHAPE, "sumcheck: too many tables");
    flush_pending_eq();
    if (sess_.active) {
      DP_REQUIRE(nt == sess_.ntabs && sess_.n == 2, DP_ERR_ARG, "sumcheck session out of sync at finish");
      post_challenge(r);
      wait_flag(++sess_.seq, (size_t)nt * 2);
      for (int i = 0; i < nt; i++) finals[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
      sess_.active = false;
      return;
    }
    FoldArgs a;
    for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.ext[i] = 0; a.half[i] = 0; }
    for (int i = 0; i < nt; i++) { DP_REQUIRE(tabs[i].n == 2, DP_ERR_SHAPE, "sc_finish: tables must have 2 entries"); a.in[i] = tabs[i].p; a.ext[i] = tabs[i].ext; }
    if (zerocopy_) { unsigned long long seq = ++seq_; DPL(k_finish_publish, dim3(1), dim3(64), a, r, nt, (Ext*)hres_dev_, hflag_dev_, seq); wait_flag(seq, 2 * (size_t)nt); }
    else { DPL(k_finish, dim3(1), dim3(64), a, r, nt, (Ext*)dres_); fetch(2 * (size_t)nt); }
    for (int i = 0; i < nt; i++) finals[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
  }

  // ---- logup
  void logup_den(const DBuf& out, const DBuf* cols, int nc, Ext c, Ext chi) override {
    DP_REQUIRE(nc >= 1 && nc <= 16 && out.ext, DP_ERR_SHAPE, "logup_den: 1..16 columns");
    ColsArg a; a.n = nc;
    for (int i = 0; i < 16; i++) a.col[i] = nullptr;
    for (int i = 0; i < nc; i++) { DP_REQUIRE(!cols[i].ext && cols[i].n == out.n, DP_ERR_SHAPE, "logup_den: column shape"); a.col[i] = (const u64*)cols[i].p; }
    nb_ = 8.0 * nc * out.n + 16.0 * out.n; DPL(k_logup_den, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, a, c, chi, out.n);
  }
  void logup_layer(const DBuf& ni, const DBuf& di, const DBuf& no, const DBuf& dout) override {
    size_t h = di.n / 2;
    DP_REQUIRE(di.ext && no.n == h && dout.n == h && (ni.null() || ni.n == di.n), DP_ERR_SHAPE, "logup_layer: shapes");
    int mode = ni.null() ? 0 : (ni.ext ? 2 : 1);
    nb_ = (mode == 0 ? 32.0 : mode == 1 ? 48.0 : 64.0) * h + 32.0 * h; DPL(k_logup_layer, dim3(grid_for(h)), dim3(TPB), (const void*)ni.p, mode, (const Ext*)di.p, (Ext*)no.p, (Ext*)dout.p, h);
  }

  void logup_build(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi,
                   std::vector<LogupCircuitDev>& circuits, std::vector<Ext>& outputs) override {
    size_t n = cols[0].n;
    if (n > 16384 || n < 4 || cpi > 8 || (size_t)ninst * sizeof(LogupTreeDesc) > DESC_BYTES || (size_t)ninst * 8 > RES_WORDS) {
      Dev::logup_build(cols, cpi, ninst, mult, c, chi, circuits, outputs);
      return;
    }
    circuits.clear(); outputs.clear();
    const LogupTreeDesc* dd = nullptr;
    LogupTreeDesc* hd = desc_alloc<LogupTreeDesc>((size_t)ninst, &dd);
    for (int s = 0; s < ninst; s++) {
      LogupCircuitDev cd;
      DBuf den_all = alloc(2 * n, true), num_all = alloc(n, true);
      LogupTreeDesc& t = hd[s];
      for (int j = 0; j < 8; j++) t.col[j] = nullptr;
      for (int j = 0; j < cpi; j++) { const DBuf& col = cols[(size_t)s * cpi + j]; DP_REQUIRE(!col.ext && col.n == n, DP_ERR_SHAPE, "logup: column shape"); t.col[j] = (const u64*)col.p; }
      t.ncols = cpi; t.num0 = mult.p; t.num_mode = mult.null() ? 0 : (mult.ext ? 2 : 1);
      t.den_all = (Ext*)den_all.p; t.num_all = (Ext*)num_all.p;
      size_t doff = 0, noff = 0;
      cd.num.push_back(mult);
      for (size_t len = n; len >= 2; len >>= 1) {
        cd.den.push_back(den_all.slice(doff, len));
        if (len < n) { cd.num.push_back(num_all.slice(noff, len)); noff += len; }
        doff += len;
      }
      circuits.push_back(cd);
    }
    size_t mk = mark();
    int threads = n >= 2048 ? 1024 : n >= 512 ? 512 : 256;
    nb_ = (double)ninst * (8.0 * cpi * n + 16.0 * 3 * n); DPL(k_logup_tree, dim3(ninst), dim3(threads), dd, n, c, chi, (Ext*)dres_);
    fetch((size_t)ninst * 8);
    for (int i = 0; i < 4 * ninst; i++) outputs.push_back(ex(hres_[2 * i], hres_[2 * i + 1]));
    release(mk);
  }

  // ---- PCS
  void pcs_init(unsigned L) override {
    if (L == L_ && tw_) return;
    DP_REQUIRE(L >= 1 && L <= 28, DP_ERR_ARG, "pcs_init: unsupported parameter size");
    HIP_CHECK(hipStreamSynchronize(s_));
    tw_ = pow7_ = nullptr;
    pcs_tabs_ = std::make_shared<PcsTables>();
    pcs_tabs_->device = device_;
    L_ = L;
    size_t n = size_t(1) << L;
    HIP_CHECK(hipMalloc((void**)&pcs_tabs_->tw, n * 8));
    HIP_CHECK(hipMalloc((void**)&pcs_tabs_->pow7, n * 8));
    tw_ = pcs_tabs_->tw; pow7_ = pcs_tabs_->pow7;
    u64 w = GL_G32;
    for (unsigned i = L + 1; i < 32; i++) w = gl_sqr(w);
    DPL(k_pow_table, dim3(grid_for(n)), dim3(TPB), tw_, w, n);
    DPL(k_pow_table, dim3(grid_for(n)), dim3(TPB), pow7_, GL_GENERATOR, n);
    HIP_CHECK(hipStreamSynchronize(s_));
  }
  void pcs_share(HipDev& owner) {
    DP_REQUIRE(owner.device_ == device_ && owner.pcs_tabs_, DP_ERR_ARG, "pcs_share: the owner has no tables on this device");
    HIP_CHECK(hipStreamSynchronize(s_));
    pcs_tabs_ = owner.pcs_tabs_; tw_ = pcs_tabs_->tw; pow7_ = pcs_tabs_->pow7; L_ = owner.L_;
  }
  void bitrev_copy(const DBuf& d, const DBuf& s) override {
    unsigned lg = dp_ceil_log2(s.n);
    if (s.ext) { nb_ = 32.0 * s.n; DPL(k_bitrev<true>, dim3(grid_for(s.n)), dim3(TPB), d.p, (const void*)s.p, lg); }
    else { nb_ = 16.0 * s.n; DPL(k_bitrev<false>, dim3(grid_for(s.n)), dim3(TPB), d.p, (const void*)s.p, lg); }
  }
  // Run k_merkle_tail over `nd` descriptors of the mapped ring and bring the roots to hres_[4*i..]. A single tree
  // publishes its root itself; several trees land in dres_ and are published together.
  void tails_to_host(const TailDesc* dd, size_t nd) {
    if (nd == 1 && zerocopy_) {
      unsigned long long seq = ++seq_;
      nb_ = 0; DPL_ONE(k_merkle_tail, dim3(1), 1024, 0, dd, dres_, hres_dev_, hflag_dev_, seq);
      wait_flag(seq, 4);
      return;
    }
    nb_ = 0; DPL_ONE(k_merkle_tail, dim3((unsigned)nd), 1024, 0, dd, dres_, (u64*)nullptr, (unsigned long long*)nullptr, 0ull);
    fetch(4 * nd);
  }
  // layers of at most this many digests are finished by k_merkle_tail (one workgroup, no relaunch between layers); wider
  // layers get their own multi-workgroup launch: a 512-parent layer is 4 sequential passes inside the tail workgroup but one
  // pass spread over 16 CUs as a launch (2048 measured equal in round 2)
  static constexpr size_t TAIL_MAX = 256;
  // Layers with at most lp_max_ parent nodes use the 8-lanes-per-node kernel (lowest latency per layer, but ~2.2x the
  // VALU work of one node per lane and a grid 8x as large: with many proofs in flight those grids fill the chip and
  // every other stream queues behind them), wider layers hash one node per lane.
  size_t lp_max_ = [] { const char* e = getenv("DP_LP_MAX"); return e ? (size_t)strtoull(e, nullptr, 10) : size_t(1) << 16; }();  // (latency mode: one proof has the chip to itself, a 2^16-parent layer is one pass of 8-lane groups at the latency of two permutations instead of a one-lane compress; 2^12 -> 2^16: CNN-264k 57.6 -> 55.0 ms, Dense-4M ~ -1 ms, tools/r05/call28.sh)
  // ... in THROUGHPUT mode the launch of a layer is merged over the ~20 members of a cohort: 20 x 1024 parents fill the chip with one node per lane, and
  // the 8-lane form's 2.2x VALU work is paid for nothing (k_merkle_layer_lp was 35 % of the Merkle kernel time of the cohort regime for ~10 % of the
  // nodes: profiles/r05_bench448_kernel_stats_lds_msgs.csv). DP_LP_MAX_TP (default 512): the widest layer that still takes the 8-lane kernel there.
  size_t lp_max_tp_ = [] { const char* e = getenv("DP_LP_MAX_TP"); return e ? (size_t)strtoull(e, nullptr, 10) : size_t(512); }();
  size_t lp_max_now() const { return throughput_mode_ ? lp_max_tp_ : lp_max_; }
  // DP_MERKLE_WG_CAP (throughput mode; 0 = off): workgroups of ONE merged k_merkle_layer launch, all members together. A hash workgroup lives ~200 us (18 700 VALU
  // instructions per lane, 7 waves per SIMD) and an uncapped layer of a cohort (21 x 2048 workgroups) takes every wave slot of the chip for milliseconds: the
  // workgroups of every other queue — the streaming kernels of the batch opening, the one-workgroup tails, k_publish — then wait for a slot to drain
  // (k_axpy_many: 12.8 ms per launch for 0.2 ms of work, profiles/r05_bench448_kernel_stats.csv). The VALU is saturated by 2-3 hash waves per SIMD.
  size_t merkle_wg_cap_ = [] { const char* e = getenv("DP_MERKLE_WG_CAP"); return e ? (size_t)strtoull(e, nullptr, 10) : size_t(0); }();
  int merkle_grid(size_t nodes) const {
    if (!throughput_mode_ || !merkle_wg_cap_) return grid_for(nodes, 4096);
    const size_t m = co_ && co_->nominal > 0 ? (size_t)co_->nominal : 1;
    return grid_for_uncapped(nodes, (int)std::max<size_t>(1, merkle_wg_cap_ / m));
  }
  // ---- every grid-stride launch of a cohort member is PLACEABLE AT ONCE (DP_WIDE_WG_CAP, throughput mode). Measured with tools/r06/qprobe.hip
  // (profiles/r06_qprobe.txt): a chain of dependent one-wave kernels on one queue costs 2.8 us per link on an idle chip, 45 us when 22 other queues run
  // long LIGHT kernels, 55 us when they run VALU-saturating persistent kernels of 64 workgroups each (22 x 64 < the ~2 000 workgroup slots of the chip) —
  // and 1.8 ms per link when those queues launch grids that do NOT fit (chip-filling, or 512 workgroups each): a grid that cannot be placed keeps its queue's
  // pipe of the command processor until its last workgroup has found a slot, and every other queue served by that pipe (22 queues on 4 pipes) waits behind it,
  // whatever it wants to launch. That is the "small kernels stretch with the number of active kernels" of rounds 2-5 (k_publish 1.7 ms with 20 kernels
  // active) and why capping ONE kernel family never moved the rate: the sum over all queues has to fit. The cap is per merged launch, all members together.
  // Default 256 (round 6, tools/r06/call20.sh, call21.sh: 12 waves of 448 Dense-4M proofs, alternating on one box): 983 / 1 006 / 1 009 / 995 proofs/s with the cap against
  // 937 / 944 / 928 without, 1 034 against 959 at 660 in flight; 192 the same, 128 and 384 less. The hash layers take the same cap (merkle_grid goes through grid_for)
  // unless DP_MERKLE_WG_CAP gives them their own: 512 / 1 024 / 2 048 for them measured 1 037 / 1 023 / 987 against 1 047-1 058 (tools/r06/call24.sh).
  size_t wide_wg_cap_ = [] { const char* e = getenv("DP_WIDE_WG_CAP"); return e ? (size_t)strtoull(e, nullptr, 10) : size_t(256); }();
  int grid_for(size_t n, int cap = 2048) const {
    if (throughput_mode_ && wide_wg_cap_ && co_) {
      const size_t m = co_->nominal > 0 ? (size_t)co_->nominal : 1;
      cap = std::min<int>(cap, (int)std::max<size_t>(1, wide_wg_cap_ / m));
    }
    return grid_for_uncapped(n, cap);
  }
  // `nodes` must hold 4*(n-1) words; synchronises (root is copied to the host)
  DevTree build_tree_into(const DBuf& leaves, const DBuf& nodes) {
    DevTree t; t.leaves = leaves; t.nleaves = leaves.n; t.nodes = nodes;
    size_t n = leaves.n;
    u64* nd = (u64*)t.nodes.p;
    if (leaves.ext) { nb_ = 16.0 * n + 16.0 * n; DPL(k_merkle_leaves<true>, dim3(grid_for(n / 2)), dim3(TPB), (const void*)leaves.p, nd, n / 2); }
    else { nb_ = 8.0 * n + 16.0 * n; DPL(k_merkle_leaves<false>, dim3(grid_for(n / 2)), dim3(TPB), (const void*)leaves.p, nd, n / 2); }
    size_t off = 0, cnt = n / 2;
    while (cnt > TAIL_MAX) {
      size_t next = cnt / 2;
      if (next <= lp_max_now()) { nb_ = 96.0 * next; DPL(k_merkle_layer_lp, dim3((unsigned)grid_for(next * 8, 8192)), dim3(256), (const u64*)(nd + 4 * off), nd + 4 * (off + cnt), next); }
      else { nb_ = 96.0 * next; DPL_HASH(k_merkle_layer, dim3(merkle_grid(next)), dim3(TPB), (const u64*)(nd + 4 * off), nd + 4 * (off + cnt), next); }
      off += cnt; cnt /= 2;
    }
    const TailDesc* dd = nullptr;
    TailDesc* hd = desc_alloc<TailDesc>(1, &dd);
    hd[0].nodes = nd; hd[0].off = off; hd[0].cnt = cnt;
    tails_to_host(dd, 1);
    for (int k = 0; k < 4; k++) t.root.v[k] = hres_[k];
    return t;
  }
  DevTree build_tree(const DBuf& leaves, bool persistent) {
    size_t n = leaves.n;
    DP_REQUIRE(n >= 2 && (n & (n - 1)) == 0, DP_ERR_SHAPE, "merkle: leaf count must be a power of two >= 2");
    DBuf nodes = persistent ? alloc_persistent(4 * (n - 1), false) : alloc(4 * (n - 1), false);
    return build_tree_into(leaves, nodes);
  }
  // stages [s0, s1) of the Moebius transform / DIT NTT of `buf` as LDS-tiled passes (k_butterfly_pass): tiles of 64 KB (2^13 base
  // or 2^12 extension elements); the first pass works on contiguous blocks, later ones on strided tiles of 128-byte segments
  void butterfly_passes(const DBuf& buf, unsigned s0, unsigned s1, bool ntt) {
    const unsigned lgtile = buf.ext ? 12 : 13, lgseg = buf.ext ? 3 : 4;
    const unsigned lgn = dp_ceil_log2(buf.n);
    const size_t lds = (size_t(1) << lgtile) * (buf.ext ? 16 : 8);
    while (s0 < s1) {
      const unsigned lgc = std::min(s0, lgseg), span = std::min(s1 - s0, std::min(lgtile, lgn) - lgc), s_hi = s0 + span;
      const size_t tiles = buf.n >> (span + lgc);
      const size_t bytes = (size_t(1) << (span + lgc)) * (buf.ext ? 16 : 8);
      nb_ = 2.0 * (double)buf.bytes();
      if (buf.ext) { if (ntt) { DPL_LDS((k_butterfly_pass<true, true>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } else { DPL_LDS((k_butterfly_pass<true, false>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } }
      else { if (ntt) { DPL_LDS((k_butterfly_pass<false, true>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } else { DPL_LDS((k_butterfly_pass<false, false>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } }
      (void)lds;
      s0 = s_hi;
    }
  }
  DevCommit commit(const DBuf& evals, bool persistent) override {
    DevCommit c; c.nv = dp_ceil_log2(evals.n); c.is_base = !evals.ext; c.evals = evals;
    DP_REQUIRE((size_t(1) << c.nv) == evals.n && evals.n >= 2, DP_ERR_SHAPE, "commit: polynomial length must be a power of two >= 2");
    if (c.nv <= 7) { c.bh_evals = evals; c.tree = build_tree(evals, persistent); return c; }
    DP_REQUIRE(tw_ && c.nv <= L_, DP_ERR_SHAPE, "commit: polynomial larger than the PCS parameters (PolynomialTooLarge)");
    size_t n = evals.n;
    auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
    DBuf cw = A(2 * n, evals.ext);
    c.bh_evals = A(n, evals.ext);
    DBuf nodes = A(4 * (2 * n - 1), false);
    size_t mk = mark();
    DBuf co = alloc(n, evals.ext);
    DBuf tmp = alloc(2 * n, evals.ext);
    copy(co, evals);
    bool E = evals.ext;
    butterfly_passes(co, 0, c.nv, false);  // K5 in LDS tiles: 2 passes for 2^20
    if (E) { nb_ = 48.0 * n; DPL(k_rs_prepare<true>, dim3(grid_for(n)), dim3(TPB), (const void*)co.p, tmp.p, (const u64*)pow7_, c.nv, L_); }
    else { nb_ = 24.0 * n; DPL(k_rs_prepare<false>, dim3(grid_for(n)), dim3(TPB), (const void*)co.p, tmp.p, (const u64*)pow7_, c.nv, L_); }
    butterfly_passes(tmp, 1, c.nv + 1, true);  // K7 stages 1..nv of the 2n-point DIT
    bitrev_copy(cw, tmp);            // K6
    bitrev_copy(c.bh_evals, evals);  // K6
    c.tree = build_tree_into(cw, nodes);  // K8 (synchronises: root to host)
    release(mk);
    return c;
  }
  // Commit many polynomials at once (witness columns of one inference): equal-size groups share batched launches —
  // one LDS-resident Moebius+NTT workgroup per polynomial, one layer-0 launch, one fused Merkle-tail workgroup per tree.
  // Medium base-field polynomials (2^12..2^14: the witness columns of a convolution) of equal size: Moebius in LDS, the
  // NTT in LDS-resident blocks of 2^13 points plus at most two global stages, batched Merkle layers — a dozen launches for
  // the whole group instead of ~45 per polynomial.
  void commit_medium_group(const std::vector<DBuf>& evals, size_t first, std::vector<DevCommit>& out, std::vector<bool>& done, bool persistent) {
    const DBuf& e0 = evals[first];
    unsigned nv = dp_ceil_log2(e0.n);
    size_t n = e0.n, N = 2 * n;
    std::vector<size_t> grp;
    for (size_t j = first; j < evals.size(); j++) if (!done[j] && evals[j].n == n && !evals[j].ext) grp.push_back(j);
    size_t g = grp.size();
    DP_REQUIRE(g * (sizeof(SmallCommitDesc) + sizeof(TailDesc)) + 256 <= DESC_BYTES && 4 * g <= RES_WORDS && g <= 65535, DP_ERR_SHAPE, "commit_many: group too large");
    if (desc_off_ + g * (sizeof(SmallCommitDesc) + sizeof(TailDesc)) + 256 > DESC_BYTES) stream_wait();
    auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
    const SmallCommitDesc* dd = nullptr; const TailDesc* tdd = nullptr;
    SmallCommitDesc* hd = desc_alloc<SmallCommitDesc>(g, &dd);
    TailDesc* td = desc_alloc<TailDesc>(g, &tdd);
    std::vector<DBuf> cws(g), bhs(g), nodes(g);
    for (size_t q = 0; q < g; q++) { cws[q] = A(N, false); bhs[q] = A(n, false); nodes[q] = A(4 * (N - 1), false); }
    size_t mk = mark();
    // layers above TAIL_MAX digests are hashed by batched launches; the tail kernel finishes each tree
    size_t off = 0, cnt = N / 2;
    std::vector<std::pair<size_t, size_t>> layers;
    while (cnt > TAIL_MAX) { layers.push_back({off, cnt}); off += cnt; cnt /= 2; }
    for (size_t q = 0; q < g; q++) {
      DevCommit& c = out[grp[q]];
      const DBuf& ev = evals[grp[q]];
      c.nv = nv; c.is_base = true; c.evals = ev; c.bh_evals = bhs[q];
      c.tree.leaves = cws[q]; c.tree.nleaves = N; c.tree.nodes = nodes[q];
      DBuf tmp = alloc(N, false);
      hd[q].evals = ev.p; hd[q].cw = cws[q].p; hd[q].bh = bhs[q].p; hd[q].nodes = (u64*)nodes[q].p; hd[q].tmp = tmp.p;
      td[q].nodes = (u64*)nodes[q].p; td[q].off = off; td[q].cnt = cnt;
    }
    nb_ = g * 32.0 * n; DPL_LDS(k_med_prepare, dim3((unsigned)g), dim3(1024), n * 8, dd, nv, L_, (const u64*)pow7_);
    unsigned lgblk = std::min<unsigned>(nv + 1, MED_NTT_LG), smax = std::min<unsigned>(nv, lgblk - 1);
    nb_ = g * 32.0 * n; DPL_LDS(k_med_ntt_local, dim3((unsigned)(N >> lgblk), (unsigned)g), dim3(1024), (size_t(1) << lgblk) * 8, dd, lgblk, smax, (const u64*)tw_, L_);
    for (unsigned st = smax + 1; st <= nv; st++) { nb_ = g * 32.0 * n; DPL(k_ntt_stage_many, dim3(grid_for(n, 64), (unsigned)g), dim3(TPB), dd, N, st, (const u64*)tw_, L_); }
    nb_ = g * 32.0 * n; DPL(k_bitrev_many, dim3(grid_for(N, 64), (unsigned)g), dim3(TPB), dd, nv + 1);
    nb_ = g * 24.0 * N; DPL(k_merkle_leaves_many<false>, dim3(grid_for(N / 2, 64), (unsigned)g), dim3(TPB), dd, N / 2);
    for (auto& l : layers) { nb_ = g * 96.0 * (l.second / 2); DPL(k_merkle_layer_many, dim3(grid_for(l.second / 2, 64), (unsigned)g), dim3(TPB), tdd, l.first, l.second); }
    tails_to_host(tdd, g);
    for (size_t q = 0; q < g; q++) { for (int k = 0; k < 4; k++) out[grp[q]].tree.root.v[k] = hres_[4 * q + k]; done[grp[q]] = true; }
    release(mk);
  }
  std::vector<DevCommit> commit_many(const std::vector<DBuf>& evals, bool persistent) override {
    std::vector<DevCommit> out(evals.size());
    std::vector<bool> done(evals.size(), false);
    for (size_t i = 0; i < evals.size(); i++) {
      if (done[i]) continue;
      const DBuf& e0 = evals[i];
      unsigned nv = dp_ceil_log2(e0.n);
      bool small = (size_t(1) << nv) == e0.n && nv >= 1 && (nv <= 7 || (tw_ && nv <= L_ && nv <= (e0.ext ? 10u : 11u)));
      // (the medium path's kernels keep a whole polynomial in up to 128 KB of LDS: not for a resident worker — behind the executor these
      // sizes take the LDS-tiled passes of the large path, whose extra launches cost nothing there)
      bool medium = !small && !e0.ext && (size_t(1) << nv) == e0.n && tw_ && nv <= L_ && nv >= 12 && nv <= 14;
      if (medium) { commit_medium_group(evals, i, out, done, persistent); continue; }
      if (!small) { out[i] = commit(e0, persistent); done[i] = true; continue; }
      std::vector<size_t> grp;
      for (size_t j = i; j < evals.size(); j++) if (!done[j] && evals[j].n == e0.n && evals[j].ext == e0.ext) grp.push_back(j);
      size_t g = grp.size(), n = e0.n;
      bool trivial = nv <= 7;
      size_t nleaves = trivial ? n : 2 * n;
      DP_REQUIRE(g * sizeof(SmallCommitDesc) + g * sizeof(TailDesc) + 128 <= DESC_BYTES && 4 * g <= RES_WORDS, DP_ERR_SHAPE, "commit_many: group too large");
      auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
      if (desc_off_ + g * sizeof(SmallCommitDesc) + g * sizeof(TailDesc) + 128 > DESC_BYTES) stream_wait();
      const SmallCommitDesc* dd = nullptr; const TailDesc* tdd = nullptr;
      SmallCommitDesc* hd = desc_alloc<SmallCommitDesc>(g, &dd);
      TailDesc* td = desc_alloc<TailDesc>(g, &tdd);
      for (size_t q = 0; q < g; q++) {
        DevCommit& c = out[grp[q]];
        const DBuf& ev = evals[grp[q]];
        c.nv = nv; c.is_base = !ev.ext; c.evals = ev;
        DBuf cw = trivial ? ev : A(2 * n, ev.ext);
        c.bh_evals = trivial ? ev : A(n, ev.ext);
        DBuf nodes = A(4 * (nleaves - 1), false);
        c.tree.leaves = cw; c.tree.nleaves = nleaves; c.tree.nodes = nodes;
        hd[q].evals = ev.p; hd[q].cw = cw.p; hd[q].bh = c.bh_evals.p; hd[q].nodes = (u64*)nodes.p; hd[q].tmp = nullptr;
        td[q].nodes = (u64*)nodes.p; td[q].off = 0; td[q].cnt = nleaves / 2;
      }
      size_t mk = mark();
      if (!trivial) {
        size_t lds = 3 * n * (e0.ext ? 16 : 8);
        if (e0.ext) { nb_ = g * 64.0 * n; DPL_B((k_commit_small<true>), 256, KF_NONE, dim3((unsigned)g), dim3(256), lds, dd, nv, L_, (const u64*)tw_, (const u64*)pow7_); }
        else { nb_ = g * 32.0 * n; DPL_B((k_commit_small<false>), 256, KF_NONE, dim3((unsigned)g), dim3(256), lds, dd, nv, L_, (const u64*)tw_, (const u64*)pow7_); }
      }
      if (e0.ext) { nb_ = g * 32.0 * nleaves; DPL(k_merkle_leaves_many<true>, dim3(grid_for(nleaves / 2, 64), (unsigned)g), dim3(TPB), dd, nleaves / 2); }
      else { nb_ = g * 24.0 * nleaves; DPL(k_merkle_leaves_many<false>, dim3(grid_for(nleaves / 2, 64), (unsigned)g), dim3(TPB), dd, nleaves / 2); }
      tails_to_host(tdd, g);
      for (size_t q = 0; q < g; q++) { for (int k = 0; k < 4; k++) out[grp[q]].tree.root.v[k] = hres_[4 * q + k]; done[grp[q]] = true; }
      release(mk);
    }
    return out;
  }
  void free_commit(DevCommit& c) override {
    if (c.bh_evals.p && c.bh_evals.p != c.evals.p) free_persistent(c.bh_evals);
    if (c.tree.leaves.p && c.tree.leaves.p != c.evals.p) free_persistent(c.tree.leaves);
    free_persistent(c.tree.nodes);
    free_persistent(c.evals);
  }
  DevTree merkle_ext(const DBuf& leaves) override { return build_tree(leaves, false); }
  DevTree batch_tree(const DBuf* cws, int k, bool persistent) override {
    DP_REQUIRE(k >= 2 && k <= BATCH_ROW_MAX, DP_ERR_SHAPE, "batch_tree: 2..32 polynomials per tree");
    const size_t n = cws[0].n; const bool E = cws[0].ext;
    BatchRowPtrs a{};
    for (int q = 0; q < k; q++) { DP_REQUIRE(cws[q].n == n && cws[q].ext == E, DP_ERR_SHAPE, "batch_tree: equal sizes and one field expected"); a.cw[q] = (const u64*)cws[q].p; }
    DBuf rows = persistent ? alloc_persistent(2 * n, true) : alloc(2 * n, true);
    nb_ = (E ? 16.0 : 8.0) * n * k + 32.0 * n;
    if (E) { DPL(k_batch_row_hash<true>, dim3(grid_for(n)), dim3(TPB), a, k, (u64*)rows.p, n); }
    else { DPL(k_batch_row_hash<false>, dim3(grid_for(n)), dim3(TPB), a, k, (u64*)rows.p, n); }
    return build_tree(rows, persistent);
  }

  // factored eq tables of the batch-opening sumcheck (Dev::classic_round): polynomials of 2^15 entries and more keep eq(x, z) as
  // (low half, high half); round 3 measured the materialised form: 320 MB more traffic per proof, same rate
  static constexpr bool eq_split_ = true;
  unsigned classic_eq_split(unsigned nv) override { return eq_split_ && zerocopy_ && nv >= 15 ? nv / 2 : 0; }
  size_t classic_eq_materialise_n() override { return CLASSIC_TAIL_MAX_N; }
  void eq_outer_many(const EqOuterJob* jobs, size_t n) override {
    if (!n) return;
    DP_REQUIRE(n * sizeof(EqOuterDesc) + 64 <= DESC_BYTES, DP_ERR_SHAPE, "eq_outer_many: too many tables");
    const EqOuterDesc* dd = nullptr;
    EqOuterDesc* d = desc_alloc<EqOuterDesc>(n, &dd);
    size_t maxn = 1; double bytes = 0;
    for (size_t i = 0; i < n; i++) {
      const EqOuterJob& j = jobs[i];
      DP_REQUIRE(j.out.ext && j.lo.ext && j.hi.ext && j.lo.n && !(j.lo.n & (j.lo.n - 1)) && j.out.n == j.lo.n * j.hi.n && j.out.n <= 0xffffffffu, DP_ERR_SHAPE, "eq_outer_many: shapes");
      d[i].out = (Ext*)j.out.p; d[i].lo = (const Ext*)j.lo.p; d[i].hi = (const Ext*)j.hi.p; d[i].ln = (unsigned)j.lo.n; d[i].n = (unsigned)j.out.n;
      maxn = std::max(maxn, j.out.n); bytes += 16.0 * j.out.n;
    }
    nb_ = bytes; DPL(k_eq_outer_many, dim3(grid_for(maxn, 64), (unsigned)n), dim3(TPB), dd);
  }
  void classic_round(DBuf* fs, DBuf* eqs, DBuf* los, int np, const Ext* r, Ext* out) override {
    DP_REQUIRE((size_t)np * (sizeof(PolyDesc) * 2 + sizeof(ClassicDesc) + 4) + 320 <= DESC_BYTES && (size_t)np * 4 <= RES_WORDS && np * 2 <= 1024, DP_ERR_SHAPE, "classic_round: too many polynomials");
    if (desc_off_ + (size_t)np * (sizeof(PolyDesc) * 2 + sizeof(ClassicDesc) + 4) + 320 > DESC_BYTES) stream_wait();
    const PolyDesc* dd = nullptr;
    PolyDesc* hd = desc_alloc<PolyDesc>((size_t)np, &dd);
    // the factored pairs (los[i].n > 0): lo / loout / ln of the fused launch's descriptor
    struct Fac { const Ext* lo = nullptr; Ext* loout = nullptr; unsigned ln = 0; };
    std::vector<Fac> fac((size_t)np);
    size_t maxn = 1;
    for (int i = 0; i < np; i++) {
      const size_t ln = los ? los[i].n : 0;
      DP_REQUIRE(fs[i].n == (ln ? ln : 1) * eqs[i].n && eqs[i].ext && (!ln || (los[i].ext && !(ln & (ln - 1)) && zerocopy_)), DP_ERR_SHAPE, "classic_round: f/eq shapes");
      hd[i].f = fs[i].p; hd[i].eq = (const Ext*)eqs[i].p; hd[i].n = fs[i].n; hd[i].fext = fs[i].ext; hd[i].pad = 0; hd[i].fout = nullptr; hd[i].eqout = nullptr;
      if (ln) { fac[i].lo = (const Ext*)los[i].p; fac[i].ln = (unsigned)ln; }
      if (r && fs[i].n > 1) {
        DBuf fo = alloc(fs[i].n / 2, true);
        hd[i].fout = (Ext*)fo.p; fs[i] = fo;
        if (ln > 1) { DBuf lo2 = alloc(ln / 2, true); fac[i].loout = (Ext*)lo2.p; los[i] = lo2; }
        else { DBuf eo = alloc(eqs[i].n / 2, true); hd[i].eqout = (Ext*)eo.p; eqs[i] = eo; }
      }
      maxn = std::max(maxn, hd[i].n);
    }
    size_t mk = mark();
    if (zerocopy_) {
      // one fused launch (fold + sums of the folded tables) on a work-proportional grid, one reduction that publishes
      const unsigned* fd = nullptr; const ClassicDesc* cdd = nullptr;
      unsigned* first = desc_alloc<unsigned>((size_t)np + 1, &fd);
      ClassicDesc* cd = desc_alloc<ClassicDesc>((size_t)np, &cdd);
      unsigned nblk = 0; double bytes = 0;
      for (int i = 0; i < np; i++) {
        cd[i].f = hd[i].f; cd[i].eq = hd[i].eq; cd[i].fout = hd[i].fout; cd[i].eqout = hd[i].eqout; cd[i].n = hd[i].n; cd[i].fext = hd[i].fext;
        cd[i].ln = fac[i].ln; cd[i].lo = fac[i].lo; cd[i].loout = fac[i].loout;
        const bool folds = r && hd[i].n > 1;
        size_t items = folds ? hd[i].n / 4 : hd[i].n / 2;  // loop iterations of the pair: 4 (2) entries of each table per iteration
        // (behind the resident executor a tile costs ~20 us of queue protocol whatever it does: 32 iterations per thread instead of 4)
        const size_t per_blk = (size_t)TPB * 4;
        size_t nb = std::min<size_t>(std::max<size_t>((items + per_blk - 1) / per_blk, 1), (size_t)grid_for(items, 1024));  // (per polynomial; every polynomial keeps one workgroup)
        first[i] = nblk; nblk += (unsigned)nb;
        bytes += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + (folds ? hd[i].n * 8.0 : 0.0) + (fac[i].ln ? 0.0 : hd[i].n * 16.0 + (folds ? hd[i].n * 8.0 : 0.0));
      }
      first[np] = nblk;
      Ext* partial = (Ext*)arena_alloc((size_t)nblk * 2 * 16);
      unsigned long long seq = ++seq_;
      nb_ = bytes; DPL(k_classic_fused, dim3(nblk), dim3(TPB), fd, cdd, np, r ? *r : ex_zero(), r ? 1 : 0, partial);
      DP_REQUIRE(2 * np <= 1024, DP_ERR_SHAPE, "classic round: too many polynomials for one reduction");
      nb_ = 0; DPL(k_classic_reduce, dim3(1), dim3(np >= 8 && !throughput_mode_ ? 1024 : 256), fd, np, (const Ext*)partial, (Ext*)hres_dev_, hflag_dev_, seq);
      wait_flag(seq, (size_t)np * 4);
      for (int i = 0; i < 2 * np; i++) out[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
      release(mk);
      return;
    }
    if (r) {
      nb_ = [&] { double b = 0; for (int i = 0; i < np; i++) if (hd[i].fout) b += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + hd[i].n * 16.0 + hd[i].n * 16.0; return b; }(); DPL(k_classic_fold, dim3(grid_for(maxn / 2, 1024), np), dim3(TPB), dd, *r);
      // descriptors for the sums: the folded tables (a second ring slot — the fold may still be reading the first)
      hd = desc_alloc<PolyDesc>((size_t)np, &dd);
      maxn = 1;
      for (int i = 0; i < np; i++) { hd[i].f = fs[i].p; hd[i].eq = (const Ext*)eqs[i].p; hd[i].n = fs[i].n; hd[i].fext = fs[i].ext; hd[i].pad = 0; hd[i].fout = nullptr; hd[i].eqout = nullptr; maxn = std::max(maxn, hd[i].n); }
    }
    int g = grid_for(std::max<size_t>(maxn / 2, 1), 256);
    Ext* partial = (Ext*)arena_alloc((size_t)np * g * 2 * 16);
    nb_ = [&] { double b = 0; for (int i = 0; i < np; i++) b += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + hd[i].n * 16.0; return b; }(); DPL(k_classic_sums, dim3(g, np), dim3(TPB), dd, partial);
    reduce_publish(partial, (size_t)g, 2, np * 2);
    for (int i = 0; i < 2 * np; i++) out[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
    release(mk);
  }
  // DP_AXPY_CLASSES=0: every descriptor straight into the one pass over the accumulator (nd x n_acc multiplications), as before round 4's last session
  bool axpy_classes_ = knob("DP_AXPY_CLASSES", 1) != 0;
  void axpy_many(const DBuf& acc, const DBuf* init, const AxpyJob* jobs, size_t n) override {
    DP_REQUIRE(acc.ext && (!init || (init->ext && init->n == acc.n)), DP_ERR_SHAPE, "axpy_many: accumulator shape");
    AxpyPlan p = axpy_plan(jobs, n, acc.n, axpy_classes_);  // axpy_many.h: the jobs shorter than the accumulator are summed among their own length first
    const size_t nfinal = std::max<size_t>(p.final_pass.size(), 1), nclass_desc = (p.classes.size() * sizeof(AxpyClass) + sizeof(AxpyDesc) - 1) / sizeof(AxpyDesc);
    if ((p.members.size() + nfinal + nclass_desc) * sizeof(AxpyDesc) + 256 > DESC_BYTES) { Dev::axpy_many(acc, init, jobs, n); return; }
    const size_t mk = mark();
    // one descriptor block for both launches (a second desc_alloc may wrap the staging ring): [class members][final pass][classes]
    const AxpyDesc* cdd = nullptr;
    AxpyDesc* chd = desc_alloc<AxpyDesc>(p.members.size() + nfinal + nclass_desc, &cdd);
    AxpyDesc* fhd = chd + p.members.size(); const AxpyDesc* fdd = cdd + p.members.size();
    AxpyClass* cc = (AxpyClass*)(fhd + nfinal); const AxpyClass* ccd = (const AxpyClass*)(fdd + nfinal);
    for (size_t c = 0; c < p.classes.size(); c++) { DBuf sum = alloc(p.classes[c].n, true); p.classes[c].out = (Ext*)sum.p; p.final_pass[p.class_final[c]].x = sum.p; }
    std::copy(p.members.begin(), p.members.end(), chd); std::copy(p.final_pass.begin(), p.final_pass.end(), fhd); std::copy(p.classes.begin(), p.classes.end(), cc);
    if (p.grouped) { nb_ = p.member_bytes; DPL(k_axpy_classes, dim3(grid_for(p.max_class_n, 256), (unsigned)p.classes.size()), dim3(TPB), ccd, cdd); }
    nb_ = 16.0 * acc.n * (init ? 2 : 1) + p.final_bytes;
    DPL(k_axpy_many, dim3(grid_for(acc.n)), dim3(TPB), (Ext*)acc.p, init ? (const Ext*)init->p : (const Ext*)nullptr, acc.n, fdd, (int)p.final_pass.size());
    release(mk);
  }
  void axpy_rep(const DBuf& acc, const DBuf& x, Ext coeff, size_t rep) override {
    DP_REQUIRE(acc.ext && acc.n == x.n * rep && (rep & (rep - 1)) == 0, DP_ERR_SHAPE, "axpy_rep: shapes");
    nb_ = x.bytes() + 32.0 * acc.n; DPL(k_axpy_rep, dim3(grid_for(acc.n)), dim3(TPB), (Ext*)acc.p, (const void*)x.p, (int)x.ext, coeff, acc.n, dp_ceil_log2(rep));
  }
  void bf_round(DBuf& eq, DBuf& f, const Ext* ch, Ext* msg) override {
    DP_REQUIRE(eq.ext && f.ext && eq.n == f.n, DP_ERR_SHAPE, "bf_round: shapes");
    if (ch) { DBuf t[2] = {eq, f}; fold_tables(t, 2, *ch); eq = t[0]; f = t[1]; }
    if (!msg) return;
    if (f.n == 1) { u64 w[2]; download(f, w); msg[0] = msg[1] = msg[2] = ex(w[0], w[1]); return; }
    size_t mk = mark();
    int g = grid_for(f.n / 2, 512);
    Ext* partial = (Ext*)arena_alloc((size_t)g * 4 * 16);
    nb_ = 32.0 * f.n; DPL(k_bf_msg, dim3(g), dim3(TPB), (const Ext*)f.p, (const Ext*)eq.p, f.n / 2, partial);
    reduce_publish(partial, (size_t)g, 4, 3);
    for (int i = 0; i < 3; i++) msg[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
    release(mk);
  }
  DBuf fri_fold(const DBuf& o, unsigned level, Ext ch) override {
    DP_REQUIRE(o.ext && o.n == (size_t(2) << level) && level <= L_, DP_ERR_SHAPE, "fri_fold: shapes");
    DBuf out = alloc(o.n / 2, true);
    u64 gam = GL_GENERATOR;
    for (unsigned i = 0; i < L_ + 1 - level - 1; i++) gam = gl_sqr(gam);
    u64 ninv = gl_neg(gl_inv(gl_dbl(gam)));  // -1/(2 gamma)
    nb_ = 16.0 * o.n + 8.0 * o.n; DPL(k_fri_fold, dim3(grid_for(out.n)), dim3(TPB), (const Ext*)o.p, (Ext*)out.p, out.n, level, (const u64*)tw_, L_, gam, ninv, ch);
    return out;
  }
  void query_gather_flat(const QueryDesc* d, size_t nd, std::vector<u64>& flat, std::vector<size_t>& off) override {
    off.assign(nd + 1, 0);
    flat.clear();
    if (!nd) return;
    const GatherDesc* dd = nullptr;
    GatherDesc* hd = desc_alloc<GatherDesc>(nd, &dd);
    size_t total = 0;
    for (size_t i = 0; i < nd; i++) {
      const DevTree& t = *d[i].tree;
      hd[i].leaves = t.leaves.p; hd[i].nodes = (const u64*)t.nodes.p; hd[i].nleaves = t.nleaves; hd[i].p0 = d[i].p0;
      hd[i].ext = t.leaves.ext; hd[i].height = (int)t.height(); hd[i].out_off = total; hd[i].path_off = total + (t.leaves.ext ? 4 : 2);
      total += (t.leaves.ext ? 4 : 2) + 4 * (size_t)(t.height() - 1);
      off[i + 1] = total;
    }
    size_t mk = mark();
    u64* dout = (u64*)arena_alloc(total * 8);
    DPL(k_query_gather, dim3((unsigned)grid_for(nd * 64, 1 << 20)), dim3(TPB), dd, nd, dout);  // (one wave per descriptor, grid-stride)
    flat.resize(total);
    d2h(flat.data(), dout, total * 8);  // (the copy follows the gather on the stream: no wait of its own in between, as there was until round 5)
    release(mk);
  }
  void query_gather_into(const QueryDesc* d, size_t nd, const size_t* pair_off, const size_t* path_off, u64* dst, size_t total) override {
    if (!nd) return;
    const GatherDesc* dd = nullptr;
    GatherDesc* hd = desc_alloc<GatherDesc>(nd, &dd);
    for (size_t i = 0; i < nd; i++) {
      const DevTree& t = *d[i].tree;
      hd[i].leaves = t.leaves.p; hd[i].nodes = (const u64*)t.nodes.p; hd[i].nleaves = t.nleaves; hd[i].p0 = d[i].p0;
      hd[i].ext = t.leaves.ext; hd[i].height = (int)t.height(); hd[i].out_off = pair_off[i]; hd[i].path_off = path_off[i];
      DP_REQUIRE(pair_off[i] + (t.leaves.ext ? 4 : 2) <= total && path_off[i] + 4 * (size_t)(t.height() - 1) <= total, DP_ERR_SHAPE, "query_gather_into: layout outside the buffer");
    }
    size_t mk = mark();
    u64* dout = (u64*)arena_alloc(total * 8);
    DPL(k_query_gather, dim3((unsigned)grid_for(nd * 64, 1 << 20)), dim3(TPB), dd, nd, dout);
    d2h(dst, dout, total * 8);
    release(mk);
  }
  // the image written by the device from the query indices and the tree list (kernels.inc k_query_section): 4 KB of host-written words per opening instead of
  // 11 200 expanded descriptors (627 KB read by the kernel across PCIe) and a host pass over the image for its headers — 1.27 ms of the proving thread per
  // Dense-4M proof, spent with the cohort's queue empty (profiles/r06_cohort_phases_704.txt: 40.7 ms per cohort and pass before k_query_gather). DP_QUERY_SECTION_HOST=1: the host form.
  void query_section(const size_t* qidx, size_t nq, const QueryTree* trees, size_t noracle, size_t ncomm, u64* dst, size_t total) override {
    static const bool host_form = getenv("DP_QUERY_SECTION_HOST") && atoi(getenv("DP_QUERY_SECTION_HOST"));
    const size_t nt = noracle + ncomm;
    if (host_form || !nq || !nt) { Dev::query_section(qidx, nq, trees, noracle, ncomm, dst, total); return; }
    std::vector<size_t> rel; size_t cpos;
    const size_t stride = query_section_layout(trees, noracle, ncomm, rel, cpos);
    DP_REQUIRE(total == 1 + nq * stride, DP_ERR_SHAPE, "query_section: layout");
    const u64* dd = nullptr;
    u64* hd = desc_alloc<u64>(8 + nq + 5 * nt, &dd);
    hd[0] = nq; hd[1] = nt; hd[2] = noracle; hd[3] = ncomm; hd[4] = stride; hd[5] = cpos; hd[6] = hd[7] = 0;
    for (size_t i = 0; i < nq; i++) hd[8 + i] = qidx[i];
    for (size_t k = 0; k < nt; k++) {
      const DevTree& t = *trees[k].tree;
      DP_REQUIRE(trees[k].shift < 64 && t.height() >= 1 && ((size_t(1) << t.height()) == t.nleaves), DP_ERR_SHAPE, "query_section: tree");
      u64* w = hd + 8 + nq + 5 * k;
      w[0] = (u64)t.leaves.p; w[1] = (u64)t.nodes.p; w[2] = t.nleaves; w[3] = (u64)trees[k].shift | ((u64)(t.leaves.ext ? 1 : 0) << 8) | ((u64)t.height() << 16); w[4] = rel[k];
    }
    for (size_t i = 0; i < nq; i++) for (size_t k = 0; k < nt; k++) DP_REQUIRE((((qidx[i] >> trees[k].shift) | 1) < trees[k].tree->nleaves), DP_ERR_SHAPE, "query_section: query index outside a tree");
    size_t mk = mark();
    u64* dout = (u64*)arena_alloc(total * 8);
    DPL(k_query_section, dim3((unsigned)grid_for(nq * nt * 64, 1 << 20)), dim3(TPB), dd, dout);
    d2h(dst, dout, total * 8);
    release(mk);
  }
  void query_gather(const QueryDesc* d, size_t nd, std::vector<std::vector<u64>>& out) override {
    std::vector<u64> flat; std::vector<size_t> off;
    query_gather_flat(d, nd, flat, off);
    out.resize(nd);
    for (size_t i = 0; i < nd; i++) out[i].assign(flat.begin() + off[i], flat.begin() + off[i + 1]);
  }
};

Dev* make_hip_dev(int device) { return new HipDev(device); }
Dev* make_hip_worker(int device, size_t arena_bytes) { return new HipDev(device, arena_bytes, size_t(16) << 20); }
// cohorts (lock-step batches of proofs, see struct Cohort): created and driven by dp_model_prove_batch
Cohort* hip_cohort_new() { const char* e = getenv("DP_COHORT_RING_BYTES"); return e ? new Cohort(strtoull(e, nullptr, 10)) : new Cohort(); }
Cohort* hip_cohort_new_sharing(Cohort* with) { const char* e = getenv("DP_COHORT_RING_BYTES"); return new Cohort(e ? strtoull(e, nullptr, 10) : size_t(32) << 20, with); }
void hip_cohort_free(Cohort* c) { delete c; }
void hip_cohort_drain(Cohort* c) { c->drain(); }
void hip_cohort_stats(Cohort* c, size_t* fired, size_t* packs) {
  *fired = c->nfired; *packs = c->npacks; c->nfired = c->npacks = 0;
  if (g_host_stats && c->nwakes) fprintf(stderr, "[dp timing] cohort: %zu wake-ups; device phases (fire -> first member sees a result) %.1f ms, host phases (wake-up -> next fire) %.1f ms = %.1f us each\n",
                                        c->nwakes, c->dev_phase_us / 1000.0, c->host_phase_us / 1000.0, c->host_phase_us / c->nwakes);
  if (g_timing_level > 2) {
    for (int which = 0; which < 2; which++) {
      std::vector<std::pair<double, std::string>> v;
      for (auto& kv : which ? c->dev_by_ : c->host_by_) { char b[200]; snprintf(b, sizeof b, "%9.1f ms in %5zu phases, %8.1f us each: %s", kv.second.first / 1000.0, kv.second.second, kv.second.first / kv.second.second, kv.first); v.push_back({kv.second.first, b}); }
      std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
      for (size_t i = 0; i < v.size() && i < 14; i++) fprintf(stderr, "[dp cohort %s] %s\n", which ? "device phase ended by the result of" : "host phase before the fire of", v[i].second.c_str());
    }
  }
  c->host_by_.clear(); c->dev_by_.clear();
  c->dev_phase_us = c->host_phase_us = 0; c->nwakes = 0; c->awake_ = false; c->have_fire_ = false;
}
void hip_dev_cohort_attach(Dev* d, Cohort* c) { static_cast<HipDev*>(d)->cohort_attach(c); }
void hip_dev_cohort_detach(Dev* d) { static_cast<HipDev*>(d)->cohort_detach(); }
void hip_dev_dump_sc_debug(Dev* d) { static_cast<HipDev*>(d)->dump_sc_debug(); }
size_t hip_dev_arena_peak(Dev* d) { return static_cast<HipDev*>(d)->arena_peak(); }
double hip_dev_probe_compress_rate(Dev* d, size_t nodes, int reps) { return static_cast<HipDev*>(d)->probe_compress_rate(nodes, reps); }
void hip_dev_arena_peak_reset(Dev* d) { static_cast<HipDev*>(d)->arena_peak_reset(); }
// free / total bytes of the device's HBM: dp_model_prove_batch sizes the number of proofs in flight against it
void hip_mem_info(int device, size_t* free_bytes, size_t* total_bytes) { HIP_CHECK(hipSetDevice(device)); HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes)); }
void hip_dev_dump_host_stats(Dev* d) { static_cast<HipDev*>(d)->dump_host_stats(); }
// latency mode (one proof on the GPU): large sumcheck rounds spread over several workgroups; throughput mode (many proofs
// in flight): one workgroup per sumcheck — spreading costs more CUs and host polls than it saves when the GPU is shared
// DIAGNOSTIC BUILD ONLY (DP_WG_TIMES): per merged launch of k_logup_tail, the spread between its members
void hip_dump_wg_times() {
#ifdef DP_WG_TIMES
  {
    unsigned long long sp = 0, np = 0, mt = 0, zero = 0;
    hipMemcpyFromSymbol(&sp, HIP_SYMBOL(g_sponge_ticks), 8); hipMemcpyFromSymbol(&np, HIP_SYMBOL(g_sponge_perms), 8); hipMemcpyFromSymbol(&mt, HIP_SYMBOL(g_member_ticks), 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_sponge_ticks), &zero, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_sponge_perms), &zero, 8); hipMemcpyToSymbol(HIP_SYMBOL(g_member_ticks), &zero, 8);
    {
      unsigned long long pha[16], z16[16];
      for (int k = 0; k < 16; k++) { pha[k] = 0; z16[k] = 0; }
      hipMemcpyFromSymbol(pha, HIP_SYMBOL(g_phase_ticks), sizeof(pha)); hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z16, sizeof(z16));
      for (int big = 0; big < 2; big++) {
        const unsigned long long* ph = pha + 8 * big;
        if (ph[7]) fprintf(stderr, "[dp wg-times] k_logup_tail (%s), sponge-free time of the sponge wave per member (us): tree + first transcript %.0f, layer set-up %.0f, passes %.0f, message + challenge bookkeeping %.0f, "
                                   "layer end (final fold, three challenges, next claim) %.0f, column claims %.0f, tail %.0f; %llu members\n", big ? "columns of more than 1024 rows" : "columns of at most 1024 rows",
                           (double)ph[0] / 100.0 / (double)ph[7], (double)ph[1] / 100.0 / (double)ph[7], (double)ph[2] / 100.0 / (double)ph[7], (double)ph[3] / 100.0 / (double)ph[7],
                           (double)ph[4] / 100.0 / (double)ph[7], (double)ph[5] / 100.0 / (double)ph[7], (double)ph[6] / 100.0 / (double)ph[7], ph[7]);
      }
    }
    if (np) fprintf(stderr, "[dp wg-times] k_logup_tail members: %.1f %% of entry -> exit inside the sponge permutation (%llu permutations, %.2f us each); the rest (table work, barriers, round arithmetic) %.0f us per 100 permutations\n",
                    100.0 * (double)sp / (double)mt, np, (double)sp / 100.0 / (double)np, (double)(mt - sp) / 100.0 / (double)np * 100.0);
  }
  unsigned n = 0; if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgt_n), sizeof(n)) != hipSuccess) return;
  n = std::min(n, 65536u);
  std::vector<unsigned long long> w(4 * (size_t)n);
  if (n && hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_wgt), w.size() * 8) != hipSuccess) return;
  unsigned zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_wgt_n), &zero, sizeof(zero));
  // s_memrealtime ticks at 100 MHz. Packs addresses recur (a ring): a group = records with one address whose entries lie within 50 ms
  std::map<unsigned long long, std::vector<size_t>> by;
  for (size_t i = 0; i < n; i++) by[w[4 * i + 2]].push_back(i);
  double sum_kernel = 0, sum_med = 0, sum_spread_end = 0, sum_skew = 0, sum_minmax = 0; size_t groups = 0, members = 0, samecu = 0;
  for (auto& kv : by) {
    auto idx = kv.second;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return w[4 * a] < w[4 * b]; });
    size_t s = 0;
    while (s < idx.size()) {
      size_t e = s + 1;
      while (e < idx.size() && w[4 * idx[e]] - w[4 * idx[s]] < 5000000ull) e++;
      if (e - s >= 2) {
        unsigned long long t0 = ~0ull, t0max = 0, t1min = ~0ull, t1 = 0; std::vector<double> dur; std::map<unsigned long long, int> cu;
        for (size_t k = s; k < e; k++) {
          size_t i = idx[k];
          t0 = std::min(t0, w[4 * i]); t0max = std::max(t0max, w[4 * i]); t1min = std::min(t1min, w[4 * i + 1]); t1 = std::max(t1, w[4 * i + 1]);
          dur.push_back((double)(w[4 * i + 1] - w[4 * i]) / 100.0);
          unsigned long long id = w[4 * i + 3]; unsigned hw = (unsigned)id; unsigned xcc = (unsigned)(id >> 32) & 0xF;
          cu[((unsigned long long)xcc << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 0x7) << 12)]++;  // (xcc, se/sh, cu)
        }
        std::sort(dur.begin(), dur.end());
        sum_kernel += (double)(t1 - t0) / 100.0; sum_med += dur[dur.size() / 2]; sum_spread_end += (double)(t1 - t1min) / 100.0; sum_skew += (double)(t0max - t0) / 100.0;
        sum_minmax += dur.back() - dur.front();
        for (auto& c : cu) if (c.second > 1) samecu += c.second;
        groups++; members += e - s;
      }
      s = e;
    }
  }
  if (groups) fprintf(stderr, "[dp wg-times] k_logup_tail: %zu merged launches, %.1f members each: first entry -> last exit %.0f us; median member %.0f us; slowest - fastest member %.0f us; "
                              "start skew (last entry - first entry) %.0f us; last exit - first exit %.0f us; %.1f %% of the members shared a CU with another member of their launch\n",
                      groups, (double)members / groups, sum_kernel / groups, sum_med / groups, sum_minmax / groups, sum_skew / groups, sum_spread_end / groups, 100.0 * samecu / members);
#endif
}
void hip_dev_pcs_share(Dev* worker, Dev* owner) { static_cast<HipDev*>(worker)->pcs_share(*static_cast<HipDev*>(owner)); }
void hip_dev_set_latency_mode(Dev* d, bool on) { static_cast<HipDev*>(d)->set_latency_mode(on); }
void hip_dev_profile_enable(Dev* d, bool on) { static_cast<HipDev*>(d)->profile_enable(on); }
std::string hip_dev_profile_report(Dev* d) { return static_cast<HipDev*>(d)->profile_report(); }

}  // namespace dp
