// RESIDENT EXECUTOR (see rx.h for the design): the worker kernels — the bodies of kernels.inc behind a work-queue loop instead
// of behind one launch each — and the host side that feeds them.
#include "dev.h"
#include "poseidon2_fast.h"
#include "gl64_lazy.h"
#include "sumcheck.h"
#include "fiber.h"
#include "logup_tail.h"
#include "classic_tail.h"
#include "dense_tail.h"
#include "eqsum_tail.h"
#include "deleg_tail.h"
#include "commit_tail.h"
#include "sponge_host.h"
#include "rx.h"
#include "rx_bodies.h"
#include <hip/hip_runtime.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#undef DP_WG_TIMES  // (a diagnostic of the cohort path)
#define DP_RX 1

namespace dp {
namespace rxk {

// ---- what the bodies see instead of the hardware's block coordinates and of separate LDS objects
struct RxVec3 { unsigned x, y, z; };
__shared__ RxVec3 rx_s_block, rx_s_grid;  // virtual blockIdx / gridDim of the tile this worker is running
__device__ __forceinline__ RxVec3 rx_block() {
  RxVec3 r; r.x = (unsigned)__builtin_amdgcn_readfirstlane((int)rx_s_block.x); r.y = (unsigned)__builtin_amdgcn_readfirstlane((int)rx_s_block.y); r.z = 0; return r;
}
__device__ __forceinline__ RxVec3 rx_grid() {
  RxVec3 r; r.x = (unsigned)__builtin_amdgcn_readfirstlane((int)rx_s_grid.x); r.y = (unsigned)__builtin_amdgcn_readfirstlane((int)rx_s_grid.y); r.z = 1; return r;
}
extern __shared__ __align__(16) unsigned char rx_lds[];  // the worker's LDS arena: [frame of the body][its dynamic part]
__device__ __forceinline__ unsigned char* rx_lds_arena() { return rx_lds; }
#define blockIdx (::dp::rxk::rx_block())
#define gridDim (::dp::rxk::rx_grid())

#include "kernels.inc"

#undef blockIdx
#undef gridDim

// ---- device-side atomics on queue words. Device memory: agent scope (performed in the XCD's L2); host-mapped memory: system scope.
__device__ __forceinline__ unsigned long long ld_dev(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_dev32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// every memory operation this wave has issued has been performed (stores: written to L2, which is all a consumer on the same XCD needs)
__device__ __forceinline__ void rx_drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// consumer side: nothing this CU cached before may be used for data another CU of the XCD has produced since
__device__ __forceinline__ void rx_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: the vector L1
  __builtin_amdgcn_s_dcache_inv();                    // the scalar cache (uniform loads of tables and descriptors)
}
__device__ __forceinline__ unsigned long long rx_mix_dev(unsigned long long step) { return step * 0x9E3779B97F4A7C15ull + 0x51A7C0DEB16B00B5ull; }

constexpr unsigned long long CELL_MASK = RX_CELLS - 1;
__device__ __forceinline__ unsigned long long cell_make(unsigned long long pos, unsigned slot, unsigned first, unsigned count) {
  return ((((pos / RX_CELLS) + 1) & 0xFFFFFFull) << 40) | ((unsigned long long)slot << 28) | ((unsigned long long)first << 8) | count;
}
// one cell of this ring, or false when none is published at the head
__device__ __forceinline__ bool rx_pop(RxRing* r, unsigned long long* cell) {
  for (int tries = 0; tries < 8; tries++) {
    unsigned long long h = ld_dev(&r->head);
    unsigned long long c = ld_dev(&r->cells[h & CELL_MASK]);
    if ((c >> 40) != (((h / RX_CELLS) + 1) & 0xFFFFFFull)) return false;
    unsigned long long expect = h;
    if (__hip_atomic_compare_exchange_strong(&r->head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { *cell = c; return true; }
  }
  return false;
}
// The next step of `slot`, if its predecessor is done and its descriptor has arrived: claims it (issued++), copies the
// descriptor into the slot and publishes its tiles as cells of the step's class. Called (by ONE thread) after a completion and
// for every doorbell; whichever of the two comes second finds the other's work done — no wake-up is lost:
//   completion:  done = s            (performed)  then  reads descriptor s
//   doorbell:    host wrote descriptor s, then the doorbell; the handler reads the doorbell, then `done`
__device__ __forceinline__ void rx_try_issue(const RxArgs& a, RxSlot* slot, unsigned slot_id) {
  for (int guard = 0; guard < 4; guard++) {
    const unsigned long long s = ld_dev(&slot->issued);
    if (ld_dev(&slot->done) < s) return;  // step s - 1 is still running: its completion comes here again
    const unsigned long long* d = (const unsigned long long*)(slot->ring + (s & (RX_DESC_RING - 1)));
    unsigned long long w[8];
    bool ok = false;
    for (int rd = 0; rd < 3 && !ok; rd++) {
#pragma unroll
      for (int i = 0; i < 8; i++) w[i] = ld_sys(d + i);
      if (w[0] != rx_mix_dev(a.session + s + 1)) return;  // not pushed yet: its doorbell will come
      unsigned long long cs = w[0];
#pragma unroll
      for (int i = 1; i < 7; i++) cs += (unsigned long long)(i + 1) * w[i];
      ok = cs == w[7];  // (the host writes the fields and the checksum before the tag: a mismatch is a torn read, read again)
    }
    if (!ok) return;
    unsigned long long expect = s;
    if (!__hip_atomic_compare_exchange_strong(&slot->issued, &expect, s + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;  // someone else took it
    const unsigned gx = (unsigned)(w[2] & 0xFFFFFFFFull), gy = (unsigned)(w[2] >> 32);
    const unsigned ntiles = gx * gy, per = (unsigned)(w[4] & 0xFFFFFFFFull), cls = (unsigned)(w[4] >> 32) & 1u;
    const unsigned which = cls == RX_BIG ? RX_RING_BIG : ntiles <= RX_URGENT_TILES ? RX_RING_URGENT : RX_RING_WIDE;
    if (slot_id == a.trace_slot && s < RX_TRACE_STEPS) { st_sys(a.trace + 4 * s, __builtin_amdgcn_s_memrealtime()); st_sys(a.trace + 4 * s + 3, (w[1] & 0xFFFFFFFFull) | ((unsigned long long)ntiles << 32)); }
    st_dev(&slot->cur_body_flags, w[1]); st_dev(&slot->cur_grid, w[2]); st_dev(&slot->cur_pack, w[3]); st_dev(&slot->cur_pack_words, w[5]);
    st_dev32(&slot->tiles_left, ntiles);
    rx_drain();
    RxRing* r = &a.xcd[slot->xcd].ring[which];
    const unsigned ncells = (ntiles + per - 1) / per;
    const unsigned long long pos = __hip_atomic_fetch_add(&r->tail, (unsigned long long)ncells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (pos + ncells - ld_dev(&r->head) > RX_CELLS) __builtin_amdgcn_s_sleep(8);  // (ring full: the other workers are draining it)
    for (unsigned i = 0; i < ncells; i++) {
      const unsigned first = i * per, cnt = ntiles - first < per ? ntiles - first : per;
      st_dev(&r->cells[(pos + i) & CELL_MASK], cell_make(pos + i, slot_id, first, cnt));
    }
    return;
  }
}
// doorbells of this XCD (one poller at a time): slot ids whose rings have received a descriptor
__device__ __forceinline__ void rx_poll_doorbells(const RxArgs& a, RxXcd* x, unsigned xcd) {
  unsigned long long expect = 0;
  if (!__hip_atomic_compare_exchange_strong(&x->db_lock, &expect, 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const unsigned long long* db = a.doorbells + (size_t)xcd * RX_DOORBELLS;
  for (int n = 0; n < 32; n++) {
    const unsigned long long h = ld_dev(&x->db_head);
    const unsigned long long v = ld_sys(db + (h & (RX_DOORBELLS - 1)));
    if ((v >> 32) != ((h + 1) & 0xFFFFFFFFull)) break;
    st_dev(&x->db_head, h + 1);
    const unsigned slot_id = (unsigned)(v & 0xFFFFFFFFull);
    if (slot_id < RX_MAX_SLOTS) rx_try_issue(a, a.slots + slot_id, slot_id);
  }
  rx_drain();
  st_dev(&x->db_lock, 0ull);
}

template <auto Body, class... A> __device__ __forceinline__ void rx_invoke_(KArgs<void (*)(A...)>, const void* pk) {
  reinterpret_cast<const ArgPack<std::decay_t<A>...>*>(pk)->call(Body);
}
template <auto Body> __device__ __forceinline__ void rx_invoke(const void* pk) { rx_invoke_<Body>(KArgs<decltype(Body)>(), pk); }
enum { RX_ID_BASE = __COUNTER__ + 1 };
template <int CLS> __device__ __forceinline__ void rx_run_body(int body, const void* pack) {
  switch (body) {
#define X(cls, ...) case (__COUNTER__ - RX_ID_BASE): if constexpr ((cls) == CLS) rx_invoke<&__VA_ARGS__>(pack); break;
    RX_BODY_LIST(X)
#undef X
    default: break;
  }
}
enum { RX_NBODIES = __COUNTER__ - RX_ID_BASE };

// which XCDs does this device expose? (a partitioned GPU shows fewer than 8: slots are dealt to the XCDs that answer)
__global__ void k_rx_probe(unsigned long long* seen) {
  if (threadIdx.x == 0) { const unsigned xcd = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u; st_sys(seen + xcd, 1ull); }
}
// The worker. One workgroup of 256 threads; thread 0 runs the queue protocol, everyone runs the bodies.
// Register budget: BIG 128 VGPRs (4 waves per SIMD — what the one-workgroup bodies were compiled for behind kg / kc), STREAM 96 (5 waves
// per SIMD): one BIG and four STREAM waves share a SIMD's 512 registers. Without the bound the compiler takes 300 registers
// for the union of the bodies and one worker fills a SIMD.
template <int CLS> __global__ void __launch_bounds__(RX_WORKER_THREADS) __attribute__((amdgpu_waves_per_eu(CLS == RX_BIG ? 4 : 5, CLS == RX_BIG ? 4 : 5))) k_rx_worker(RxArgs a) {
  __shared__ unsigned long long s_cell, s_bf, s_grid, s_pack, s_pack_words;
  __shared__ __align__(16) unsigned long long s_args[CLS == RX_BIG ? RX_PACK_WORDS_BIG : RX_PACK_WORDS_STREAM];  // the step's argument pack, staged once per cell (the bodies read their arguments many times)
  __shared__ int s_state;  // 0: nothing to do, 1: run s_cell, 2: leave
  const unsigned xcd = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u;  // HW_REG_XCC_ID
  RxXcd* x = a.xcd + xcd;
  RxRing* ring = &x->ring[CLS == RX_BIG ? RX_RING_BIG : RX_RING_WIDE];
  RxRing* urgent = &x->ring[RX_RING_URGENT];
  if (CLS == RX_BIG) __builtin_amdgcn_s_setprio(3);  // the protocol bodies are one-wave dependent chains: issue them first
  if (threadIdx.x == 0 && __hip_atomic_fetch_add(&x->alive[CLS], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) st_sys(a.heartbeat + 32 + xcd * RX_NCLASS + CLS, 1ull);
  unsigned idle = 0, ran = 0;
  for (;;) {
    if (threadIdx.x == 0) {
      unsigned long long c = 0;
      int st = 0;
      if ((CLS == RX_STREAM && rx_pop(urgent, &c)) || rx_pop(ring, &c)) st = 1;
      else {
        rx_poll_doorbells(a, x, xcd);
        if ((CLS == RX_STREAM && rx_pop(urgent, &c)) || rx_pop(ring, &c)) st = 1;
        else if ((idle & 15u) == 15u && ld_sys(a.control) != 0) st = 2;
      }
      if (st == 1) {
        RxSlot* slot = a.slots + ((c >> 28) & 0xFFFu);
        s_cell = c; s_bf = ld_dev(&slot->cur_body_flags); s_grid = ld_dev(&slot->cur_grid); s_pack = ld_dev(&slot->cur_pack); s_pack_words = ld_dev(&slot->cur_pack_words);
      }
      s_state = st;
    }
    __syncthreads();
    const int st = s_state;
    if (st == 2) break;
    if (st == 0) {  // nothing published for this class on this XCD: back off (bounded: a doorbell must not wait long)
      idle++;
      if (threadIdx.x == 0 && (idle & 63u) == 0) __hip_atomic_fetch_add(&x->idle_iters[CLS], 64ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (idle < 8) __builtin_amdgcn_s_sleep(8); else if (idle < 64) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(96);
      __syncthreads();
      continue;
    }
    idle = 0;
    rx_acquire();  // (every wave: what this CU cached before the step's predecessor finished must not be used)
    const unsigned long long c = s_cell;
    const unsigned first = (unsigned)((c >> 8) & 0xFFFFFu), cnt = (unsigned)(c & 0xFFu);
    const unsigned gx = (unsigned)(s_grid & 0xFFFFFFFFull), gy = (unsigned)(s_grid >> 32);
    const int body = (int)(s_bf & 0xFFFFFFFFull);
    { const unsigned long long* src = (const unsigned long long*)s_pack; const unsigned nw = (unsigned)s_pack_words;
      for (unsigned i = threadIdx.x; i < nw; i += RX_WORKER_THREADS) s_args[i] = ld_sys(src + i); }
    const void* pack = (const void*)s_args;
    __syncthreads();
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && first == 0 && ((c >> 28) & 0xFFFu) == a.trace_slot) { const unsigned long long sn = ld_dev(&a.slots[a.trace_slot].issued) - 1; if (sn < RX_TRACE_STEPS) st_sys(a.trace + 4 * sn + 1, t_in); }
    for (unsigned t = first; t < first + cnt; t++) {
      if (threadIdx.x == 0) { rx_s_block.x = t % gx; rx_s_block.y = t / gx; rx_s_block.z = 0; rx_s_grid.x = gx; rx_s_grid.y = gy; rx_s_grid.z = 1; }
      __syncthreads();
      rx_run_body<CLS>(body, pack);
      __syncthreads();  // the next tile (or step) reuses the LDS arena and the block coordinates
    }
    rx_drain();  // this wave's stores are in L2 ...
    __syncthreads();  // ... and so are everyone's
    if (threadIdx.x == 0) {
      RxSlot* slot = a.slots + ((c >> 28) & 0xFFFu);
      const unsigned left = __hip_atomic_fetch_sub(&slot->tiles_left, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left == cnt) {  // the last tiles of the step: the step is done, the proof's next step may start
        const unsigned long long s1 = ld_dev(&slot->issued);
        st_dev(&slot->done, s1);
        st_sys(slot->host_done, s1);
        if (((c >> 28) & 0xFFFu) == a.trace_slot && s1 - 1 < RX_TRACE_STEPS) st_sys(a.trace + 4 * (s1 - 1) + 2, __builtin_amdgcn_s_memrealtime());
        rx_drain();
        rx_try_issue(a, slot, (unsigned)((c >> 28) & 0xFFFu));
      }
      if ((++ran & 255u) == 0) st_sys(a.heartbeat + xcd * RX_NCLASS + CLS, (unsigned long long)ran);
      const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_in;
      const int b = body & (RX_STAT_BODIES - 1);
      __hip_atomic_fetch_add(&x->body_ticks[b], dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&x->body_cells[b], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&x->body_tiles[b], (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&x->busy_ticks[CLS], dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

}  // namespace rxk

// ================================================================================================ host side
#define RX_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw DpError(DP_ERR_HIP, std::string("resident executor: " #x ": ") + hipGetErrorString(e_)); } while (0)

namespace {
constexpr int RX_NBODIES_HOST = rxk::RX_NBODIES;
struct BodyInfo { int cls; size_t frame; const char* name; };
template <class T> constexpr size_t frame_of() { return sizeof(T); }
// LDS frame (bytes at the head of the arena) of the bodies that have one
const BodyInfo* body_table() {
  static BodyInfo tab[RX_NBODIES_HOST];
  static bool init = false;
  if (!init) {
    int i = 0;
#define X(klass, ...) tab[i].cls = (klass); tab[i].frame = 0; tab[i].name = #__VA_ARGS__; i++;
    RX_BODY_LIST(X)
#undef X
    auto set = [&](const char* prefix, size_t bytes) { for (int k = 0; k < RX_NBODIES_HOST; k++) if (!strncmp(tab[k].name, prefix, strlen(prefix)) && (tab[k].name[strlen(prefix)] == 0 || tab[k].name[strlen(prefix)] == '<')) tab[k].frame = (bytes + 15) & ~size_t(15); };
    set("k_sc_small", sizeof(rxk::Lds_k_sc_small)); set("k_sc_persist", sizeof(rxk::Lds_k_sc_persist)); set("k_sc_persist_lds", sizeof(rxk::Lds_k_sc_persist_lds));
    set("k_logup_tail", sizeof(rxk::Lds_k_logup_tail)); set("k_classic_tail", sizeof(rxk::Lds_k_classic_tail)); set("k_dense_tail", sizeof(rxk::Lds_k_dense_tail));
    set("k_eqsum_tail", sizeof(rxk::Lds_k_eqsum_tail)); set("k_deleg_tail", sizeof(rxk::Lds_k_deleg_tail)); set("k_commit_tail", sizeof(rxk::Lds_k_commit_tail));
    set("k_reduce_publish", sizeof(rxk::Lds_k_reduce_publish)); set("k_classic_reduce", sizeof(rxk::Lds_k_classic_reduce)); set("k_eq_table_many", sizeof(rxk::Lds_k_eq_table_many));
    init = true;
  }
  return tab;
}
}  // namespace

struct RxEngine {
  int device = 0;
  hipStream_t stream[RX_NCLASS] = {nullptr, nullptr};
  RxXcd* d_xcd = nullptr; RxSlot* d_slots = nullptr;
  // host-mapped: control, heartbeat, host_done, doorbells (one block); descriptor + pack rings (one block per slot, grown on demand)
  char* hm = nullptr; char* hm_dev = nullptr;
  unsigned long long *control = nullptr, *heartbeat = nullptr, *host_done = nullptr, *doorbells = nullptr, *trace = nullptr;
  unsigned trace_slot = RX_MAX_SLOTS;
  struct SlotHost {
    char* mem = nullptr; char* mem_dev = nullptr;         // [RX_DESC_RING descriptors][RX_PACK_RING bytes of packs]
    unsigned long long pushed = 0;                         // steps pushed in this session
    unsigned long long pack_head = 0, pack_tail = 0;       // virtual byte offsets into the pack ring
    unsigned long long pack_end[RX_DESC_RING];             // pack_head after step s (indexed s % RX_DESC_RING): what step s's completion frees
    unsigned long long freed_upto = 0;                     // steps whose pack space has been returned
    const char* names[RX_DESC_RING];                       // launch names of the ring's steps (diagnostics)
  };
  std::vector<SlotHost> slots;
  std::vector<unsigned> xcds;  // the XCC ids this device answers with (k_rx_probe)
  unsigned xcd_of(unsigned slot) const { return xcds[slot % xcds.size()]; }
  std::atomic<unsigned long long> db_tail[RX_XCDS];
  unsigned long long session = 0;
  bool running = false; unsigned nslots = 0;
  std::chrono::steady_clock::time_point t_start{}; double last_session_ms = 0;
  int nworkers[RX_NCLASS] = {0, 0};
  std::mutex mu;
};

size_t rx_lds_budget(int cls) { return cls == RX_BIG ? RX_LDS_BIG : RX_LDS_STREAM; }

RxEngine* rx_engine_new(int device) {
  std::unique_ptr<RxEngine> e(new RxEngine());
  e->device = device;
  RX_HIP(hipSetDevice(device));
  for (int c = 0; c < RX_NCLASS; c++) RX_HIP(hipStreamCreateWithFlags(&e->stream[c], hipStreamNonBlocking));
  RX_HIP(hipMalloc((void**)&e->d_xcd, sizeof(RxXcd) * RX_XCDS));
  RX_HIP(hipMalloc((void**)&e->d_slots, sizeof(RxSlot) * RX_MAX_SLOTS));
  const size_t words = 64 + 64 + RX_MAX_SLOTS * 16 + (size_t)RX_XCDS * RX_DOORBELLS + (size_t)RX_TRACE_STEPS * 4;  // host_done: one word per 128-byte line (the device writes them)
  RX_HIP(hipHostMalloc((void**)&e->hm, words * 8, hipHostMallocMapped | hipHostMallocCoherent));
  RX_HIP(hipHostGetDevicePointer((void**)&e->hm_dev, e->hm, 0));
  memset(e->hm, 0, words * 8);
  e->control = (unsigned long long*)e->hm; e->heartbeat = e->control + 64; e->host_done = e->heartbeat + 64; e->doorbells = e->host_done + RX_MAX_SLOTS * 16; e->trace = e->doorbells + (size_t)RX_XCDS * RX_DOORBELLS;
  RX_HIP(hipFuncSetAttribute((const void*)rxk::k_rx_worker<RX_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RX_LDS_BIG_LAUNCH));
  RX_HIP(hipFuncSetAttribute((const void*)rxk::k_rx_worker<RX_STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RX_LDS_STREAM));
  // the Poseidon2 round constants and extrapolation weights of THIS translation unit's device code
  RX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rxk::c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST)));
  { std::vector<u64> ex((SC_MAXK + 1) * (SC_MAXK + 1) * (SC_MAXK + 1), 0);
    for (unsigned k = 1; k < (unsigned)SC_MAXK; k++) for (unsigned at = k + 1; at <= (unsigned)SC_MAXK; at++) for (unsigned i = 0; i <= k; i++)
      ex[((size_t)k * (SC_MAXK + 1) + at) * (SC_MAXK + 1) + i] = extrapolation_coeffs(k, at)[i];
    RX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rxk::c_extrap), ex.data(), ex.size() * 8)); }
  { double ts = getenv("DP_POLL_TIMEOUT_S") ? std::max(0.001, atof(getenv("DP_POLL_TIMEOUT_S"))) : 20.0; unsigned long long tk = (unsigned long long)(ts * 1e8); RX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rxk::c_poll_timeout_ticks), &tk, sizeof(tk))); }
  { int ps = getenv("DP_POLL_SLEEP") ? std::max(0, atoi(getenv("DP_POLL_SLEEP"))) : 1; RX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rxk::c_poll_sleep), &ps, sizeof(int))); }
  body_table();  // (built here, single-threaded)
  { for (int x = 0; x < 16; x++) e->heartbeat[48 + x] = 0;
    hipLaunchKernelGGL(rxk::k_rx_probe, dim3(4096), dim3(64), 0, e->stream[0], (unsigned long long*)(e->hm_dev + ((char*)(e->heartbeat + 48) - e->hm)));
    RX_HIP(hipStreamSynchronize(e->stream[0]));
    for (unsigned x = 0; x < RX_XCDS; x++) if (e->heartbeat[48 + x]) e->xcds.push_back(x);
    DP_REQUIRE(!e->xcds.empty(), DP_ERR_HIP, "resident executor: no XCD answered the probe"); }
  // workers that are resident together: the hardware's answer for each kernel alone, then BIG first (one per CU) and STREAM in what is left
  hipDeviceProp_t prop; RX_HIP(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount;
  int per_cu_big = 0, per_cu_stream = 0;
  RX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_big, (const void*)rxk::k_rx_worker<RX_BIG>, RX_WORKER_THREADS, RX_LDS_BIG_LAUNCH));
  RX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_stream, (const void*)rxk::k_rx_worker<RX_STREAM>, RX_WORKER_THREADS, RX_LDS_STREAM));
  const char* eb = getenv("DP_RX_BIG_PER_CU"); const char* es = getenv("DP_RX_STREAM_PER_CU");
  const int big = eb ? atoi(eb) : 1, str = es ? atoi(es) : 4;
  e->nworkers[RX_BIG] = cus * std::max(1, std::min(big, per_cu_big));
  e->nworkers[RX_STREAM] = cus * std::max(1, std::min(str, per_cu_stream));
  if (getenv("DP_TIMING") && atoi(getenv("DP_TIMING")))
    fprintf(stderr, "[dp timing] resident executor: %d CUs, occupancy limits %d BIG / %d STREAM workgroups per CU alone; launching %d BIG + %d STREAM workers\n", cus, per_cu_big, per_cu_stream, e->nworkers[RX_BIG], e->nworkers[RX_STREAM]);
  return e.release();
}
void rx_engine_free(RxEngine* e) {
  if (!e) return;
  if (e->running) { try { rx_engine_stop(e); } catch (...) {} }
  hipSetDevice(e->device);
  for (auto& s : e->slots) if (s.mem) hipHostFree(s.mem);
  if (e->hm) hipHostFree(e->hm);
  if (e->d_xcd) hipFree(e->d_xcd);
  if (e->d_slots) hipFree(e->d_slots);
  for (int c = 0; c < RX_NCLASS; c++) if (e->stream[c]) hipStreamDestroy(e->stream[c]);
  delete e;
}
bool rx_engine_running(const RxEngine* e) { return e && e->running; }

void rx_engine_start(RxEngine* e, unsigned nslots) {
  std::lock_guard<std::mutex> g(e->mu);
  DP_REQUIRE(!e->running, DP_ERR_ARG, "resident executor: already running");
  DP_REQUIRE(nslots >= 1 && nslots <= RX_MAX_SLOTS, DP_ERR_ARG, "resident executor: bad slot count");
  RX_HIP(hipSetDevice(e->device));
  if (e->slots.size() < nslots) e->slots.resize(nslots);
  const size_t per_slot = (size_t)RX_DESC_RING * sizeof(RxDesc) + RX_PACK_RING;
  e->session += (unsigned long long)1 << 40;  // descriptor tags of earlier sessions can never match
  std::vector<RxSlot> init(nslots);
  for (unsigned i = 0; i < nslots; i++) {
    RxEngine::SlotHost& s = e->slots[i];
    if (!s.mem) { RX_HIP(hipHostMalloc((void**)&s.mem, per_slot, hipHostMallocMapped | hipHostMallocCoherent)); RX_HIP(hipHostGetDevicePointer((void**)&s.mem_dev, s.mem, 0)); memset(s.mem, 0, (size_t)RX_DESC_RING * sizeof(RxDesc)); }
    s.pushed = 0; s.pack_head = s.pack_tail = 0; s.freed_upto = 0;
    memset(&init[i], 0, sizeof(RxSlot));
    init[i].xcd = e->xcd_of(i);
    init[i].ring = (const RxDesc*)s.mem_dev;
    init[i].host_done = (unsigned long long*)(e->hm_dev + ((char*)(e->host_done + (size_t)i * 16) - e->hm));
    e->host_done[(size_t)i * 16] = 0;
  }
  memset(e->doorbells, 0, (size_t)RX_XCDS * RX_DOORBELLS * 8);
  for (int x = 0; x < RX_XCDS; x++) e->db_tail[x].store(0);
  e->control[0] = 0;
  for (int i = 0; i < 48; i++) e->heartbeat[i] = 0;
  { const char* ts = getenv("DP_RX_TRACE"); e->trace_slot = ts ? (unsigned)atoi(ts) : RX_MAX_SLOTS; if (e->trace_slot < nslots) memset(e->trace, 0, (size_t)RX_TRACE_STEPS * 32); else e->trace_slot = RX_MAX_SLOTS; }
  RX_HIP(hipMemsetAsync(e->d_xcd, 0, sizeof(RxXcd) * RX_XCDS, e->stream[0]));
  RX_HIP(hipMemcpyAsync(e->d_slots, init.data(), sizeof(RxSlot) * nslots, hipMemcpyHostToDevice, e->stream[0]));
  RX_HIP(hipStreamSynchronize(e->stream[0]));
  std::atomic_thread_fence(std::memory_order_seq_cst);
  auto launch = [&](int c) {
    RxArgs a;
    a.xcd = e->d_xcd; a.slots = e->d_slots;
    a.doorbells = (const unsigned long long*)(e->hm_dev + ((char*)e->doorbells - e->hm));
    a.control = (const unsigned long long*)(e->hm_dev + ((char*)e->control - e->hm));
    a.heartbeat = (unsigned long long*)(e->hm_dev + ((char*)e->heartbeat - e->hm));
    a.session = e->session; a.cls = c;
    a.trace = (unsigned long long*)(e->hm_dev + ((char*)e->trace - e->hm)); a.trace_slot = e->trace_slot;
    if (c == RX_BIG) hipLaunchKernelGGL((rxk::k_rx_worker<RX_BIG>), dim3(e->nworkers[c]), dim3(RX_WORKER_THREADS), RX_LDS_BIG_LAUNCH, e->stream[c], a);
    else hipLaunchKernelGGL((rxk::k_rx_worker<RX_STREAM>), dim3(e->nworkers[c]), dim3(RX_WORKER_THREADS), RX_LDS_STREAM, e->stream[c], a);
    RX_HIP(hipGetLastError());
  };
  // BIG first, and the STREAM workers only once every XCD has reported a BIG worker (see RX_LDS_BIG_LAUNCH)
  launch(RX_BIG);
  { auto tb = std::chrono::steady_clock::now();
    for (;;) {
      bool all = true;
      for (unsigned x : e->xcds) if (!__atomic_load_n(e->heartbeat + 32 + x * RX_NCLASS + RX_BIG, __ATOMIC_ACQUIRE)) all = false;
      if (all || std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count() > 2.0) break;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(2)); }  // (the dispatcher places the rest of the BIG grid)
  launch(RX_STREAM);
  e->nslots = nslots; e->running = true; e->t_start = std::chrono::steady_clock::now();
  // every XCD must have workers of both classes resident before a proof is pinned to it
  auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true;
    for (unsigned x : e->xcds) for (int c = 0; c < RX_NCLASS; c++) if (!__atomic_load_n(e->heartbeat + 32 + x * RX_NCLASS + c, __ATOMIC_ACQUIRE)) all = false;
    if (all) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
      std::string miss; for (unsigned x : e->xcds) for (int c = 0; c < RX_NCLASS; c++) if (!e->heartbeat[32 + x * RX_NCLASS + c]) miss += " xcd" + std::to_string(x) + (c == RX_BIG ? ":big" : ":stream");
      __atomic_store_n(e->control, 1ull, __ATOMIC_SEQ_CST); hipStreamSynchronize(e->stream[0]); hipStreamSynchronize(e->stream[1]); __atomic_store_n(e->control, 0ull, __ATOMIC_SEQ_CST); e->running = false;
      throw DpError(DP_ERR_HIP, "resident executor: no worker became resident on" + miss);
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
void rx_engine_stop(RxEngine* e) {
  std::lock_guard<std::mutex> g(e->mu);
  if (!e->running) return;
  hipSetDevice(e->device);
  __atomic_store_n(e->control, 1ull, __ATOMIC_SEQ_CST);
  hipError_t r0 = hipStreamSynchronize(e->stream[0]), r1 = hipStreamSynchronize(e->stream[1]);
  __atomic_store_n(e->control, 0ull, __ATOMIC_SEQ_CST);
  e->running = false; e->last_session_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e->t_start).count();
  if (r0 != hipSuccess || r1 != hipSuccess) throw DpError(DP_ERR_HIP, std::string("resident executor: the workers ended with ") + hipGetErrorString(r0 != hipSuccess ? r0 : r1));
}

static inline unsigned long long slot_done(const RxEngine* e, unsigned slot) { return __atomic_load_n(e->host_done + (size_t)slot * 16, __ATOMIC_ACQUIRE); }
bool rx_slot_idle(RxEngine* e, unsigned slot) { return slot_done(e, slot) >= e->slots[slot].pushed; }
void rx_slot_confirm(RxEngine*, unsigned) {}  // (ring space is recycled from the device's own progress word)

void rx_submit(RxEngine* e, unsigned slot, int body, int cls, int flags, unsigned gx, unsigned gy, size_t lds, const void* pack, size_t pack_bytes, const char* name) {
  DP_REQUIRE(e && e->running && slot < e->nslots, DP_ERR_ARG, "resident executor: not running / bad slot");
  DP_REQUIRE(body >= 0 && body < RX_NBODIES_HOST, DP_ERR_SHAPE, std::string("kernel ") + name + " is not available in the resident executor (csrc/rx_bodies.h)");
  const BodyInfo& bi = body_table()[body];
  DP_REQUIRE(bi.cls == cls, DP_ERR_SHAPE, "resident executor: body class mismatch between the translation units");
  DP_REQUIRE(bi.frame + lds <= rx_lds_budget(cls), DP_ERR_SHAPE, std::string("kernel ") + name + " needs more LDS than a resident worker of its class has");
  const unsigned long long ntiles = (unsigned long long)gx * gy;
  DP_REQUIRE(gx >= 1 && gy >= 1 && ntiles < (1u << 20), DP_ERR_SHAPE, "resident executor: grid out of range");
  RxEngine::SlotHost& s = e->slots[slot];
  const size_t need = (pack_bytes + 63) & ~size_t(63);
  DP_REQUIRE(need <= RX_PACK_RING / 4 && (pack_bytes + 7) / 8 <= (cls == RX_BIG ? RX_PACK_WORDS_BIG : RX_PACK_WORDS_STREAM), DP_ERR_SHAPE, std::string("resident executor: argument pack of ") + name + " too large");
  // ring space: descriptors and packs are recycled as the device reports steps done
  auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const unsigned long long done = slot_done(e, slot);
    while (s.freed_upto < done) { s.pack_tail = s.pack_end[s.freed_upto & (RX_DESC_RING - 1)]; s.freed_upto++; }
    unsigned long long head = s.pack_head;
    if (head % RX_PACK_RING + need > RX_PACK_RING) head += RX_PACK_RING - head % RX_PACK_RING;
    if (s.pushed - done < RX_DESC_RING - 1 && head + need - s.pack_tail <= RX_PACK_RING) { s.pack_head = head; break; }
    if (fiber_active()) fiber_yield(); else { static const bool wy = getenv("DP_WAIT_YIELD") && atoi(getenv("DP_WAIT_YIELD")); if (wy) sched_yield(); else __builtin_ia32_pause(); }
    if ((++spins & 0x3FFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
      throw DpError(DP_ERR_HIP, "resident executor: no progress on a full ring\n" + rx_engine_dump(e, slot));
  }
  char* pk = s.mem + (size_t)RX_DESC_RING * sizeof(RxDesc) + s.pack_head % RX_PACK_RING;
  memcpy(pk, pack, pack_bytes);
  const unsigned long long pk_dev = (unsigned long long)(uintptr_t)(s.mem_dev + (size_t)RX_DESC_RING * sizeof(RxDesc) + s.pack_head % RX_PACK_RING);
  s.pack_head += need;
  s.pack_end[s.pushed & (RX_DESC_RING - 1)] = s.pack_head;
  const unsigned per = (unsigned)std::min<unsigned long long>(255, std::max<unsigned long long>(1, (ntiles + 255) / 256));
  volatile unsigned long long* d = (volatile unsigned long long*)(s.mem + (s.pushed & (RX_DESC_RING - 1)) * sizeof(RxDesc));
  unsigned long long w[8];
  w[0] = rx_mix(e->session + s.pushed + 1);
  w[1] = (unsigned long long)(unsigned)body | ((unsigned long long)(unsigned)flags << 32);
  w[2] = (unsigned long long)gx | ((unsigned long long)gy << 32);
  w[3] = pk_dev;
  w[4] = (unsigned long long)per | ((unsigned long long)(unsigned)cls << 32);
  w[5] = (pack_bytes + 7) / 8; w[6] = s.pushed;
  s.names[s.pushed & (RX_DESC_RING - 1)] = name;
  w[7] = w[0]; for (int i = 1; i < 7; i++) w[7] += (unsigned long long)(i + 1) * w[i];
  for (int i = 1; i < 8; i++) d[i] = w[i];
  std::atomic_thread_fence(std::memory_order_release);
  d[0] = w[0];  // the tag last: a reader that sees it sees the fields
  s.pushed++;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const unsigned xcd = e->xcd_of(slot);
  const unsigned long long t = e->db_tail[xcd].fetch_add(1);
  __atomic_store_n(e->doorbells + (size_t)xcd * RX_DOORBELLS + (t & (RX_DOORBELLS - 1)), ((t + 1) << 32) | slot, __ATOMIC_RELEASE);
}

std::string rx_engine_stats(RxEngine* e) {
  if (!e || e->running) return "{}";
  hipSetDevice(e->device);
  std::vector<RxXcd> x(RX_XCDS);
  // only the counters at the end of each RxXcd are needed, but the struct is 0.5 MB: copy the tails
  const size_t off = offsetof(RxXcd, alive);
  std::vector<char> tail((sizeof(RxXcd) - off) * RX_XCDS);
  for (int i = 0; i < RX_XCDS; i++) if (hipMemcpy(tail.data() + (size_t)i * (sizeof(RxXcd) - off), (const char*)(e->d_xcd + i) + off, sizeof(RxXcd) - off, hipMemcpyDeviceToHost) != hipSuccess) return "{}";
  auto at = [&](int i) { return (const RxXcd*)(tail.data() + (size_t)i * (sizeof(RxXcd) - off) - off); };  // (only members from `alive` on are valid)
  const BodyInfo* bt = body_table();
  double busy[RX_NCLASS] = {0, 0};
  for (int i = 0; i < RX_XCDS; i++) for (int c = 0; c < RX_NCLASS; c++) busy[c] += (double)at(i)->busy_ticks[c];
  std::string out = "{\"alive_per_xcd_stream_big\": [";
  char buf[512];
  for (int i = 0; i < RX_XCDS; i++) { snprintf(buf, sizeof buf, "%s[%llu, %llu]", i ? ", " : "", at(i)->alive[RX_STREAM], at(i)->alive[RX_BIG]); out += buf; }
  out += "], \"idle_iters_stream_big\": [";
  { unsigned long long is = 0, ib = 0; for (int i = 0; i < RX_XCDS; i++) { is += at(i)->idle_iters[RX_STREAM]; ib += at(i)->idle_iters[RX_BIG]; } snprintf(buf, sizeof buf, "%llu, %llu], ", is, ib); out += buf; }
  snprintf(buf, sizeof buf, "\"session_ms\": %.3f, \"workers\": [%d, %d], \"busy_frac\": [%.4f, %.4f], \"bodies\": [", e->last_session_ms, e->nworkers[RX_STREAM], e->nworkers[RX_BIG],
           e->last_session_ms > 0 ? busy[RX_STREAM] * 1e-5 / (e->last_session_ms * e->nworkers[RX_STREAM]) : 0.0, e->last_session_ms > 0 ? busy[RX_BIG] * 1e-5 / (e->last_session_ms * e->nworkers[RX_BIG]) : 0.0);
  out += buf;
  bool firstb = true;
  for (int b = 0; b < RX_NBODIES_HOST && b < RX_STAT_BODIES; b++) {
    unsigned long long ticks = 0, cells = 0, tiles = 0;
    for (int i = 0; i < RX_XCDS; i++) { ticks += at(i)->body_ticks[b]; cells += at(i)->body_cells[b]; tiles += at(i)->body_tiles[b]; }
    if (!cells) continue;
    snprintf(buf, sizeof buf, "%s{\"body\": \"%s\", \"class\": \"%s\", \"cells\": %llu, \"tiles\": %llu, \"total_ms\": %.3f, \"avg_us_per_tile\": %.3f}", firstb ? "" : ", ", bt[b].name,
             bt[b].cls == RX_BIG ? "big" : "stream", cells, tiles, ticks * 1e-5, tiles ? ticks * 1e-2 / (double)tiles : 0.0);
    out += buf; firstb = false;
  }
  return out + "]}";
}

std::string rx_engine_trace(RxEngine* e) {
  if (!e || e->running || e->trace_slot >= RX_MAX_SLOTS) return std::string();
  const BodyInfo* bt = body_table();
  const unsigned long long n = std::min<unsigned long long>(e->slots[e->trace_slot].pushed, RX_TRACE_STEPS);
  std::string out = "# step body tiles issue_us start_us done_us | gap_since_prev_done_us queue_us run_us\n";
  unsigned long long t0 = n ? e->trace[0] : 0, prev_done = t0;
  char buf[256];
  for (unsigned long long s = 0; s < n; s++) {
    const unsigned long long ti = e->trace[4 * s], ts = e->trace[4 * s + 1], td = e->trace[4 * s + 2], bw = e->trace[4 * s + 3];
    const unsigned body = (unsigned)(bw & 0xFFFFFFFFull), tiles = (unsigned)(bw >> 32);
    snprintf(buf, sizeof buf, "%llu %s %u %.2f %.2f %.2f | %.2f %.2f %.2f\n", s, body < (unsigned)RX_NBODIES_HOST ? bt[body].name : "?", tiles, (ti - t0) * 0.01, (ts - t0) * 0.01, (td - t0) * 0.01,
             ((double)ti - (double)prev_done) * 0.01, ((double)ts - (double)ti) * 0.01, ((double)td - (double)ts) * 0.01);
    out += buf; prev_done = td;
  }
  return out;
}

std::string rx_engine_dump(RxEngine* e, unsigned slot) {
  char buf[1024];
  const RxEngine::SlotHost& s = e->slots[slot];
  const unsigned long long done = slot_done(e, slot);
  const char* nm = "?";
  if (done < s.pushed) nm = s.names[done & (RX_DESC_RING - 1)];
  std::string hb;
  for (int x = 0; x < RX_XCDS; x++) { char t[64]; snprintf(t, sizeof t, " xcd%d:%llu/%llu", x, e->heartbeat[x * RX_NCLASS + RX_STREAM], e->heartbeat[x * RX_NCLASS + RX_BIG]); hb += t; }
  snprintf(buf, sizeof buf, "[resident executor] slot %u (XCD %u): %llu steps pushed, %llu done, waiting on step %llu (%s); doorbells rung on its XCD: %llu; cells run (stream/big, x256):%s",
           slot, e->xcd_of(slot), s.pushed, done, done, nm ? nm : "?", e->db_tail[e->xcd_of(slot)].load(), hb.c_str());
  return buf;
}

}  // namespace dp
