// TEST INFRASTRUCTURE — a minimal SIMT emulator that runs DEVICE SOURCE of deep-prove_amd/csrc/hip_dev.hip on the CPU.
// One workgroup; every GPU thread is a cooperative fiber (csrc/fiber.h) of one OS thread, so execution is deterministic.
// __syncthreads is a rendezvous of all fibers, every cross-lane operation (__shfl, __shfl_down, DPP moves) one of the 64
// fibers of a wave — valid for code whose cross-lane operations sit in wave-uniform control flow, which is what the
// hardware requires too.
// It checks the LOGIC of a kernel (indices, layouts, transcript order, arithmetic), not memory-model or timing behaviour.
#pragma once
#include <type_traits>
#include "../../../deep-prove_amd/csrc/fiber.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <unordered_map>
#include <vector>

namespace simt {
struct State {
  dp::FiberSched sched;
  std::unordered_map<const dp::Fiber*, unsigned> lane_of;
  unsigned n = 0, arrived = 0, gen = 0;
  std::vector<unsigned> wave_arrived, wave_gen;  // cross-lane operations are rendezvous points of ONE wave (64 threads)
  std::vector<uint32_t> xch;
};
inline State*& st() { static State* s = nullptr; return s; }
inline unsigned tid() { return st()->lane_of.at(st()->sched.cur); }
inline void barrier() {
  State& s = *st();
  unsigned g = s.gen;
  if (++s.arrived == s.n) { s.arrived = 0; s.gen++; }
  else while (s.gen == g) dp::fiber_yield();
}
inline void wave_barrier(unsigned wave) {
  State& s = *st();
  unsigned g = s.wave_gen[wave];
  if (++s.wave_arrived[wave] == 64) { s.wave_arrived[wave] = 0; s.wave_gen[wave]++; }
  else while (s.wave_gen[wave] == g) dp::fiber_yield();
}
// value of `v` in thread `src` (a thread of the caller's wave); all 64 threads of the wave take part
inline uint32_t exchange(uint32_t v, unsigned src) {
  State& s = *st();
  const unsigned me = tid(), wave = me >> 6;
  if ((src >> 6) != wave) { fprintf(stderr, "simt: cross-lane read outside the wave\n"); abort(); }
  s.xch[me] = v;
  wave_barrier(wave);
  uint32_t r = s.xch[src];
  wave_barrier(wave);
  return r;
}
// run `body` as a workgroup of `nthreads` (a multiple of 64)
inline void launch(unsigned nthreads, std::function<void()> body) {
  State s;
  s.n = nthreads; s.xch.assign(nthreads, 0); s.wave_arrived.assign(nthreads / 64, 0); s.wave_gen.assign(nthreads / 64, 0);
  State* saved = st();
  st() = &s;
  for (unsigned i = 0; i < nthreads; i++) dp::fiber_spawn(s.sched, body, size_t(256) << 10);
  for (unsigned i = 0; i < nthreads; i++) s.lane_of[s.sched.fibers[i].get()] = i;
  dp::fiber_run_all(s.sched);
  st() = saved;
}
struct Idx { operator unsigned() const { return tid(); } };
struct Const { unsigned v; operator unsigned() const { return v; } };
}  // namespace simt

// ---- the HIP vocabulary the extracted device code uses
static struct { simt::Idx x; } threadIdx;
static struct { simt::Const x{64}, y{1}, z{1}; } blockDim;  // blockDim.x.v is set by the harness before a launch
static struct { simt::Const x{0}, y{0}, z{0}; } blockIdx;
static struct { simt::Const x{1}, y{1}, z{1}; } gridDim;
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__
#define KBODY inline void
#define DP_LDS_FRAME(T, f) static T f  // (one workgroup at a time: the frame of a body is a plain static object here)
#define DP_LDS_DYN(name, T) alignas(16) static unsigned char name[160 << 10]  // (the dynamic LDS of a launch: a CU's whole LDS)
#define MSG_PUT(p, v) (*(p) = (v))  // (kernels.inc: a word of the message a one-workgroup kernel assembles in LDS)
#define DP_WAVE_SYNC() simt::wave_barrier(simt::tid() >> 6)  // lanes of one wave hand data to each other through LDS: in lock step on the hardware, a rendezvous here
#define DP_CLAIM_ALL_VGPRS() ((void)0)
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(volatile const std::remove_pointer_t<decltype(p)>*)(p))
inline void __syncthreads() { simt::barrier(); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline unsigned long long __brevll(unsigned long long x) { unsigned long long r = 0; for (int i = 0; i < 64; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
inline int __shfl(int v, int src, int width = 64) {
  unsigned me = simt::tid();
  return (int)simt::exchange((uint32_t)v, (me & ~63u) | ((unsigned)src & (unsigned)(width - 1)));
}
inline int __shfl_down(int v, int d, int width = 64) {
  (void)width;
  unsigned me = simt::tid(), lane = me & 63u;
  unsigned src = lane + (unsigned)d < 64u ? me + (unsigned)d : me;
  return (int)simt::exchange((uint32_t)v, src);
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)simt::exchange((uint32_t)v, (simt::tid() & ~63u) | ((unsigned)lane & 63u)); }
// DPP with full row / bank masks: quad_perm (ctrl < 0x100: two selector bits per lane of a quad) and row_half_mirror (0x141)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)old; (void)bound_ctrl;
  if (row_mask != 0xF || bank_mask != 0xF) { fprintf(stderr, "simt: partial DPP masks are not emulated\n"); abort(); }
  unsigned me = simt::tid(), from;
  if (ctrl >= 0 && ctrl < 0x100) from = (me & ~3u) | (((unsigned)ctrl >> (2 * (me & 3u))) & 3u);
  else if (ctrl == 0x141) from = (me & ~7u) | (7u - (me & 7u));
  else { fprintf(stderr, "simt: DPP control 0x%x is not emulated\n", ctrl); abort(); }
  return (int)simt::exchange((uint32_t)src, from);
}
