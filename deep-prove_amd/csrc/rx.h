// RESIDENT EXECUTOR — the throughput-mode launch path of dp_model_prove_batch.
//
// With hundreds of proofs in flight every proof is a chain of ~230 dependent, mostly microsecond-sized steps. Dispatched through the
// command processor — even merged 12 proofs to a launch (struct Cohort) — a step costs 100–500 us of queueing under 20+ active queues
// and the members of a cohort wait for each other at every launch. Here nothing is launched per step: two persistent kernels per GPU
// (class STREAM: 256-thread workers for the grid-stride bodies; class BIG: one worker per CU with 64 KB of LDS for the one-workgroup
// protocol bodies and the LDS-tiled passes) pull work from queues in device memory, and a step becomes runnable the moment ITS OWN
// predecessor has finished — per-proof sequence words, no cohort barrier, no command processor.
//
//   host (one fiber per proof)                       device
//   ------------------------------------------      ---------------------------------------------------------------------------------
//   launch_(Body, grid, args)                        worker (idle): polls its XCD's doorbell ring (host-mapped) -> try_issue(slot)
//     pack  -> the slot's pack ring (host-mapped)    try_issue(slot): previous step done && descriptor `issued` present ->
//     desc  -> the slot's descriptor ring              copy it into the slot, tiles_left = gx*gy, push cells {slot, first tile, count}
//     doorbell: slot id -> the XCD's doorbell ring     into the XCD's cell ring of the step's class
//   wait_flag(...): as before — the bodies           worker: pop a cell -> acquire (invalidate L1 / scalar cache) -> run the body for each
//     publish into host-mapped memory with             tile with blockIdx / gridDim = the step's virtual coordinates -> stores drained
//     checksummed tags                                 -> tiles_left -= count; the worker that reaches 0 marks the step done
//                                                      (slot.done, and a host-visible copy) and calls try_issue for the next one
//
// XCD awareness is what makes this correct AND cheap: a proof (slot) is pinned to ONE XCD — descriptors, cells and every tile of
// the proof run on workers of that XCD, so all of the proof's data lives under one L2, which is coherent for the CUs of its XCD. A
// step therefore needs no L2 write-back (what made in-kernel hand-over 7x slower in round 2, DP_FUSED_TICKET), only "my stores have
// reached L2" (s_waitcnt) on the producer and an L1 / scalar-cache invalidate on the consumer. Queue words live in device memory
// and are only ever touched by atomics of their own XCD; everything the host writes or reads is host-mapped (uncached on the GPU).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace dp {

constexpr int RX_XCDS = 8;                    // MI355X: 8 XCDs x 32 CUs, one L2 each
constexpr unsigned RX_MAX_SLOTS = 1024;       // proofs in flight (one slot per worker context)
constexpr unsigned RX_DESC_RING = 1024;       // steps a proof may have pushed and not yet seen completed (power of two)
constexpr size_t RX_PACK_RING = size_t(1) << 20;  // bytes of argument packs per slot (ring, recycled like the descriptor ring)
constexpr unsigned RX_CELLS = 1u << 15;       // cells per (XCD, class) ring (power of two)
constexpr unsigned RX_DOORBELLS = 1u << 16;   // doorbell ring entries per XCD (power of two)
constexpr unsigned RX_WORKER_THREADS = 256;
// dynamic LDS arena of a worker. A BIG worker ASKS for 84 KB (RX_LDS_BIG_LAUNCH) although its bodies use at most 64: two of them then
// cannot share a CU, and the 256 BIG workers — launched first — land one per CU; the STREAM workers (9 KB + 4.5 KB of statics)
// fill in four per CU behind them (registers: 128 + 4 x 96 of a SIMD's 512). Launched the other way round the STREAM workers pack
// five to a CU and leave the BIG workers a sixth of the chip.
constexpr size_t RX_LDS_BIG = 64 * 1024, RX_LDS_BIG_LAUNCH = 84 * 1024, RX_LDS_STREAM = 9 * 1024;
constexpr int RX_NCLASS = 2;
// cell rings per XCD: STREAM wide steps, BIG steps, and URGENT = STREAM steps of a few tiles (copies, publications, reductions: the
// links of every proof's chain) which the STREAM workers take first — behind a burst of 256 four-millisecond Merkle tiles a
// one-tile step would otherwise wait its turn like a small kernel waits in a command-processor queue
constexpr int RX_NRINGS = 3, RX_RING_WIDE = 0, RX_RING_BIG = 1, RX_RING_URGENT = 2;
constexpr unsigned RX_URGENT_TILES = 8;
constexpr unsigned RX_TRACE_STEPS = 1u << 14;  // steps of ONE slot a session can trace (DP_RX_TRACE)
constexpr unsigned RX_PACK_WORDS_STREAM = 256, RX_PACK_WORDS_BIG = 512;  // largest argument pack a worker stages in LDS (TermArgs 1.6 KB / ScPersistArgs ~3 KB)
constexpr int RX_STAT_BODIES = 128;           // per-body counters of the workers (>= number of bodies in rx_bodies.h)

// One step of one proof. 8 words in host-mapped memory; word 0 is a tag over the other seven (torn reads are retried):
// tag = rx_mix(step + 1) + sum_i (i + 1) * w[i]
struct alignas(64) RxDesc {
  unsigned long long tag;
  unsigned long long body_flags;   // body id | flags << 32
  unsigned long long grid;         // gx | gy << 32
  unsigned long long pack;         // device view of the argument pack
  unsigned long long cells;        // tiles per cell | class << 32
  unsigned long long pack_words;   // 8-byte words of the pack (the worker stages them in LDS: one PCIe burst per cell)
  unsigned long long w6, w7;       // step index (diagnostics), checksum
};
static_assert(sizeof(RxDesc) == 64, "one descriptor per 64-byte line");
inline unsigned long long rx_mix(unsigned long long step) { return step * 0x9E3779B97F4A7C15ull + 0x51A7C0DEB16B00B5ull; }

// device-resident state of one proof slot (touched only by atomics of the slot's XCD)
struct alignas(128) RxSlot {
  unsigned long long issued;       // steps handed to the cell rings so far
  unsigned long long done;         // steps completed
  unsigned tiles_left;             // tiles of the step in flight not yet finished
  unsigned xcd;
  const RxDesc* ring;              // device view of the slot's descriptor ring (host-mapped)
  unsigned long long* host_done;   // device view of the host-visible copy of `done`
  // the step in flight (copied from its descriptor by whoever issued it)
  unsigned long long cur_body_flags, cur_grid, cur_pack, cur_pack_words;
  unsigned long long pad[7];
};
static_assert(sizeof(RxSlot) == 128, "RxSlot layout");

struct alignas(128) RxRing {  // MPMC ring of cells: producers reserve with tail, consumers advance head one cell at a time
  unsigned long long head; unsigned long long pad0[15];
  unsigned long long tail; unsigned long long pad1[15];
  unsigned long long cells[RX_CELLS];  // lap + 1 (24 bits) | slot (12) | first tile (20) | tile count (8)
};
struct alignas(128) RxXcd {
  RxRing ring[RX_NRINGS];
  unsigned long long db_head; unsigned long long db_lock; unsigned long long db_last_poll; unsigned long long pad[13];
  unsigned long long alive[RX_NCLASS], pad2[14];  // workers of each class that have started on this XCD
  // what the workers of this XCD did (s_memrealtime ticks of 10 ns; read by the host after the session: rx_engine_stats)
  unsigned long long body_ticks[RX_STAT_BODIES], body_cells[RX_STAT_BODIES], body_tiles[RX_STAT_BODIES];
  unsigned long long busy_ticks[RX_NCLASS], idle_iters[RX_NCLASS], issue_ticks, doorbells;
};

// what the worker kernels get (by value)
struct RxArgs {
  RxXcd* xcd;                            // [RX_XCDS], device memory
  RxSlot* slots;                         // [RX_MAX_SLOTS], device memory
  const unsigned long long* doorbells;   // [RX_XCDS][RX_DOORBELLS], host-mapped: (ticket + 1) << 32 | slot
  const unsigned long long* control;     // host-mapped: [0] = stop flag
  unsigned long long* heartbeat;         // host-mapped: [xcd * 2 + class] = cells run (diagnostics); [32 + xcd * 2 + class] = "a worker of this class is resident on this XCD"
  unsigned long long session;            // salt of this session's descriptor tags
  unsigned long long* trace;             // host-mapped, [RX_TRACE_STEPS][4]: issue / first tile start / done ticks and body | tiles << 32 of every step of `trace_slot`
  unsigned trace_slot;                   // RX_MAX_SLOTS: nothing is traced
  int cls;
};

// ---- host interface (rx.hip), used by hip_dev.hip and capi.cpp
struct RxEngine;
RxEngine* rx_engine_new(int device);
void rx_engine_free(RxEngine* e);
void rx_engine_start(RxEngine* e, unsigned nslots);   // resets every slot and ring, launches the worker kernels
void rx_engine_stop(RxEngine* e);                     // raises the stop flag, waits for the workers to leave
bool rx_engine_running(const RxEngine* e);
size_t rx_lds_budget(int cls);                        // dynamic LDS a body of this class may use (frame included)
// queue one step of `slot`: Body id, launch flags (KF_*), virtual grid, argument pack (copied), name for diagnostics.
// Blocks (yielding the fiber) while the slot's rings are full. Throws DpError on a body the executor does not carry.
void rx_submit(RxEngine* e, unsigned slot, int body, int cls, int flags, unsigned gx, unsigned gy, size_t lds, const void* pack, size_t pack_bytes, const char* name);
// every step pushed so far has completed (host-visible progress word)
bool rx_slot_idle(RxEngine* e, unsigned slot);
// the host has observed a publication of the slot's LAST pushed step: everything before it has run, its ring space is free
void rx_slot_confirm(RxEngine* e, unsigned slot);
std::string rx_engine_dump(RxEngine* e, unsigned slot);  // state of a slot and of its XCD's queues, for error messages
// per-body accounting of the last session (after rx_engine_stop): JSON {"session_ms", "workers": [stream, big], "busy_frac": [..],
// "bodies": [{"body", "class", "cells", "tiles", "total_ms", "avg_us_per_tile"}...]} — the executor's stand-in for a kernel trace
// (rocprofv3 sees two launches per session)
std::string rx_engine_stats(RxEngine* e);
// DP_RX_TRACE=<slot>: the timeline of that slot's steps in the last session, one line per step (ticks of 10 ns relative to the first)
std::string rx_engine_trace(RxEngine* e);

}  // namespace dp
