// Cooperative fibers for the host orchestrator. A proof spends most of its wall time waiting for the device (one
// Fiat-Shamir round trip per sumcheck round, ~2000 waits per proof); a host thread that spins through those waits burns a
// whole core per proof in flight, and the MI355X boxes give a process ~16 cores per GPU (cgroup quota). With fibers one
// host thread drives several proofs: every device wait (HipDev::wait_flag) yields to the next proof of the thread instead
// of spinning. Fibers never migrate between threads (HIP's per-thread state and the thread_locals of this library stay
// valid). x86-64 System V only (the hosts of MI355X nodes are EPYC).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <time.h>
#include <unistd.h>
#include <immintrin.h>

#if !defined(__x86_64__)
#error "fiber.h: x86-64 hosts only"
#endif

// saves the callee-saved registers on the current stack, stores rsp to *save_sp, switches to load_sp and restores
extern "C" void dp_fiber_switch(void** save_sp, void* load_sp);
#define DP_FIBER_SWITCH_ASM                                                                                             \
  __asm__(".text\n.globl dp_fiber_switch\n.type dp_fiber_switch,@function\ndp_fiber_switch:\n"                           \
          "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"                         \
          "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"                                                                    \
          "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"                        \
          ".size dp_fiber_switch, .-dp_fiber_switch\n");

namespace dp {

struct FiberSched;
struct Fiber {
  void* sp = nullptr;
  void* stack = nullptr;
  size_t stack_bytes = 0;
  bool failed = false;  // the body let an exception escape
  std::function<void()> fn;
  bool done = false;
  FiberSched* sched = nullptr;
  ~Fiber() { if (stack) munmap(stack, stack_bytes); }
};
struct FiberSched {
  std::vector<std::unique_ptr<Fiber>> fibers;
  Fiber* cur = nullptr;
  void* main_sp = nullptr;
  bool progressed = false;  // a fiber of this scheduler has come out of a wait since the sweep began (fiber_note_progress)
  bool idle_sleep = false;  // the owner allows the thread to sleep when no fiber made progress (throughput mode only: a single proof's waits are microseconds long)
};
inline FiberSched*& fiber_current_sched() { static thread_local FiberSched* s = nullptr; return s; }
inline bool fiber_active() { FiberSched* s = fiber_current_sched(); return s && s->cur; }
// a wait of the calling fiber has succeeded: its thread has work (see fiber_run_all)
inline void fiber_note_progress() { FiberSched* s = fiber_current_sched(); if (s) s->progressed = true; }
// DP_IDLE_SLEEP_US (default 20; 0 = spin): when a whole round over the fibers of a thread found every one of them still waiting for the device, the thread sleeps
// this long instead of polling on. Measured at 448 Dense-4M proofs in flight (tools/r06/call6.sh, profiles/r06_idle_sleep_host_accounting.txt): the process goes
// from 13.2 busy cores (polling included) to 4.9 with the same 14 threads and to 6.3 with one thread per cohort (22), at the same or a slightly better rate
// (837 / 858 / 872 proofs/s): the device waits of a cohort are milliseconds long, 20 us of wake-up latency per step is 0.4 % of a pass. What it buys is the
// host's other half: a process that shares its cgroup quota with the application no longer starves it, and the thread count is no longer tied to the quota.
inline long fiber_idle_sleep_ns() { static const long v = [] { const char* e = getenv("DP_IDLE_SLEEP_US"); return e ? (long)(atof(e) * 1000.0) : 20000L; }(); return v; }
// give the thread to the next fiber (called from inside a fiber at a device wait)
inline void fiber_yield() {
  FiberSched* s = fiber_current_sched();
  Fiber* f = s->cur;
  dp_fiber_switch(&f->sp, s->main_sp);
}
inline void fiber_entry_trampoline() {
  FiberSched* s = fiber_current_sched();
  Fiber* f = s->cur;
  try { f->fn(); } catch (...) { f->failed = true; }  // the body catches its own exceptions; nothing may unwind past this frame (fake return address)
  f->done = true;
  dp_fiber_switch(&f->sp, s->main_sp);
  abort();  // a finished fiber is never resumed
}
// Whole proofs run on these stacks (std::function, exceptions, HIP runtime calls): 4 MB, committed lazily, and the lowest
// page is a PROT_NONE guard so that an overflow faults instead of corrupting the mapping below (another fiber's stack).
inline void fiber_spawn(FiberSched& s, std::function<void()> fn, size_t stack_bytes = size_t(4) << 20) {
  std::unique_ptr<Fiber> f(new Fiber());
  const size_t page = 4096;
  stack_bytes = ((stack_bytes + page - 1) & ~(page - 1)) + page;
  f->fn = std::move(fn); f->sched = &s; f->stack_bytes = stack_bytes;
  f->stack = mmap(nullptr, stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
  if (f->stack == MAP_FAILED) { f->stack = nullptr; throw std::bad_alloc(); }
  mprotect(f->stack, page, PROT_NONE);
  uintptr_t top = ((uintptr_t)f->stack + stack_bytes) & ~uintptr_t(15);
  void** sp = (void**)top;
  *(--sp) = nullptr;                             // fake return address of the trampoline (keeps rsp = 8 mod 16 at entry)
  *(--sp) = (void*)&fiber_entry_trampoline;      // `ret` of the first switch lands here
  for (int i = 0; i < 6; i++) *(--sp) = nullptr;  // rbp rbx r12 r13 r14 r15
  f->sp = (void*)sp;
  s.fibers.push_back(std::move(f));
}
// run every fiber of `s` on the calling thread until all are finished, round robin
inline void fiber_run_all(FiberSched& s) {
  FiberSched*& cur = fiber_current_sched();
  FiberSched* saved = cur;
  cur = &s;
  const long idle_ns = s.idle_sleep ? fiber_idle_sleep_ns() : 0;
  if (idle_ns > 0) prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);  // (the default slack of 50 us would round every sleep up)
  for (;;) {
    bool alive = false;
    s.progressed = false;
    for (auto& f : s.fibers) {
      if (f->done) continue;
      alive = true;
      s.cur = f.get();
      dp_fiber_switch(&s.main_sp, f->sp);
      s.cur = nullptr;
    }
    if (!alive) break;
    if (idle_ns > 0 && !s.progressed) { struct timespec ts = {0, idle_ns}; nanosleep(&ts, nullptr); }
    else _mm_pause();
  }
  cur = saved;
}

// CPUs this process may actually use: the cgroup quota when there is one (cpu.max of cgroup v2, cfs_quota of v1), else
// the number of hardware threads
inline double host_cpu_budget() {
  double hw = 0;
  { long n = sysconf(_SC_NPROCESSORS_ONLN); hw = n > 0 ? (double)n : 1.0; }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64]; double period = 0;
    if (fscanf(f, "%63s %lf", a, &period) == 2 && std::string(a) != "max" && period > 0) { double q = atof(a) / period; fclose(f); return q > 0 && q < hw ? q : hw; }
    fclose(f);
  }
  double quota = -1, period = -1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lf", &quota) != 1) quota = -1; fclose(f); }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lf", &period) != 1) period = -1; fclose(f); }
  if (quota > 0 && period > 0 && quota / period < hw) return quota / period;
  return hw;
}

}  // namespace dp
