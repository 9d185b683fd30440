// Host <-> device mailbox round trip UNDER LOAD: the question behind "sponge on the host for the fused protocol kernels".
// C cohorts x G workgroups (one stream per cohort, like the merged launches of dp_model_prove_batch) ping-pong with the host: each
// workgroup publishes a request word (system-scope store into mapped host memory) and polls its reply word; T host threads serve the
// mailboxes round robin (thread t owns the mailboxes of the cohorts c = t mod T, as the cohort threads do) and spend `work_us` per
// request (3 Poseidon2 permutations on the host are ~4.5 us). Optionally a wide VALU-bound kernel keeps every CU busy meanwhile.
// Prints the round trip per request as the device sees it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mailbox_load tools/mailbox_load.hip -lpthread && /tmp/mailbox_load
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void echo(unsigned long long* req, const unsigned long long* rep, int base, int iters, int sleep, unsigned long long* ticks) {
  const int m = base + blockIdx.x;
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 1; i <= iters; i++) {
    __hip_atomic_store(req + 8 * m, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned spin = 0; spin < (1u << 26); spin++) {
      if (__hip_atomic_load(rep + 8 * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (unsigned long long)i) break;
      if (sleep) __builtin_amdgcn_s_sleep(8);
    }
  }
  ticks[m] = __builtin_amdgcn_s_memrealtime() - t0;
}
// background: every CU busy with dependent 64-bit multiply-adds (what the wide Poseidon2 layers look like to the scheduler)
__global__ void busy(unsigned long long* out, int rounds) {
  unsigned long long a = threadIdx.x + 1, b = blockIdx.x + 3;
  for (int r = 0; r < rounds; r++) {
#pragma unroll 16
    for (int i = 0; i < 256; i++) a = a * b + (a >> 7);
  }
  if (a == 42) out[0] = a;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  CK(hipSetDevice(0));
  struct Cfg { int C, G, T; double work_us; int load; int shared; };  // shared: every thread scans every mailbox and claims a request with a CAS (csrc/sponge_host.h)
  const Cfg cfgs[] = {{1, 1, 1, 0.0, 0, 0}, {22, 12, 14, 0.0, 0, 0}, {22, 12, 14, 4.5, 0, 0}, {22, 12, 14, 4.5, 0, 1}, {22, 12, 14, 4.5, 1, 1}, {22, 12, 14, 0.0, 0, 1}, {22, 12, 7, 4.5, 0, 1}};
  for (const Cfg& cf : cfgs) {
    const int M = cf.C * cf.G;
    unsigned long long *req, *rep, *ticks, *dreq, *drep, *dticks, *sink;
    CK(hipHostMalloc((void**)&req, M * 64, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostMalloc((void**)&rep, M * 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&ticks, M * 8, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipMalloc((void**)&sink, 64));
    for (int i = 0; i < M * 8; i++) { req[i] = 0; rep[i] = 0; }
    CK(hipHostGetDevicePointer((void**)&dreq, req, 0)); CK(hipHostGetDevicePointer((void**)&drep, rep, 0)); CK(hipHostGetDevicePointer((void**)&dticks, ticks, 0));
    std::vector<hipStream_t> st(cf.C + 1);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::atomic<int> done(0);
    std::vector<std::thread> th;
    std::vector<unsigned long long> served(cf.T, 0);
    for (int t = 0; t < cf.T; t++) th.emplace_back([&, t] {
      if (cf.shared) {  // every thread scans every mailbox; a request is claimed with a CAS on the mailbox's lock word
        static std::vector<std::atomic<int>> lock(4096); static std::vector<unsigned long long> lastv(4096);
        if (t == 0) for (int m = 0; m < M; m++) { lock[m] = 0; lastv[m] = 0; }
        static std::atomic<int> ready(0), finished(0); if (t == 0) { finished = 0; ready = 1; } while (!ready.load()) {}
        while (finished.load(std::memory_order_relaxed) < M && !done.load(std::memory_order_relaxed)) {
          for (int m = 0; m < M; m++) {
            unsigned long long v = __atomic_load_n(req + 8 * m, __ATOMIC_ACQUIRE);
            if (v == __atomic_load_n(&lastv[m], __ATOMIC_RELAXED)) continue;
            int e = 0; if (!lock[m].compare_exchange_strong(e, 1, std::memory_order_acquire)) continue;
            v = __atomic_load_n(req + 8 * m, __ATOMIC_ACQUIRE);
            if (v != lastv[m]) {
              if (cf.work_us > 0) { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < cf.work_us) {} }
              __atomic_store_n(rep + 8 * m, v, __ATOMIC_RELEASE); __atomic_store_n(&lastv[m], v, __ATOMIC_RELAXED); served[t]++;
              if ((int)v == iters) finished.fetch_add(1);
            }
            lock[m].store(0, std::memory_order_release);
          }
        }
        if (t == 0) { while (finished.load() < M && !done.load()) {} ready = 0; }
        return;
      }
      std::vector<int> mine; for (int c = t; c < cf.C; c += cf.T) for (int g = 0; g < cf.G; g++) mine.push_back(c * cf.G + g);
      std::vector<unsigned long long> last(mine.size(), 0);
      size_t fin = 0;
      while (fin < mine.size() && !done.load(std::memory_order_relaxed)) {
        for (size_t k = 0; k < mine.size(); k++) {
          const int m = mine[k];
          unsigned long long v = __atomic_load_n(req + 8 * m, __ATOMIC_ACQUIRE);
          if (v != last[k]) {
            if (cf.work_us > 0) { auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < cf.work_us) {} }
            __atomic_store_n(rep + 8 * m, v, __ATOMIC_RELEASE);
            last[k] = v; served[t]++;
            if ((int)v == iters) fin++;
          }
        }
      }
    });
    auto t0 = std::chrono::steady_clock::now();
    if (cf.load) for (int r = 0; r < 4000; r++) hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, st[cf.C], sink, 64);
    for (int c = 0; c < cf.C; c++) hipLaunchKernelGGL(echo, dim3(cf.G), dim3(256), 0, st[c], dreq, (const unsigned long long*)drep, c * cf.G, iters, 1, dticks);
    for (int c = 0; c < cf.C; c++) CK(hipStreamSynchronize(st[c]));
    double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    done = 1;
    for (auto& t : th) t.join();
    CK(hipDeviceSynchronize());
    double avg = 0, mx = 0; for (int m = 0; m < M; m++) { double us = (double)ticks[m] / 100.0 / iters; avg += us; mx = std::max(mx, us); }
    printf("%2d cohorts x %2d workgroups, %2d host threads (%s), %.1f us of host work per request, wide kernel running: %s -> round trip %.2f us average, %.2f us slowest workgroup (wall %.0f ms, %.0f k requests/s)\n",
           cf.C, cf.G, cf.T, cf.shared ? "every thread serves every mailbox" : "a thread serves its cohorts", cf.work_us, cf.load ? "yes" : "no ", avg / M, mx, wall, (double)M * iters / wall);
    for (auto& s : st) hipStreamDestroy(s);
    hipHostFree(req); hipHostFree(rep); hipHostFree(ticks); hipFree(sink);
  }
  return 0;
}
