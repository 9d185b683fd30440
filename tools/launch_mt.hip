// What makes tiny kernels slow when many proofs are in flight? N host threads, each with its own stream, run
// {launch a one-wave kernel that posts a tag to host memory; spin on the tag} `iters` times, with optional background:
//   bg=1  every thread also keeps a persistent kernel spinning on a host mailbox (like the persistent sumcheck)
//   bg=2  one extra stream keeps streaming 256 MB through the L2s (like the Basefold oracle merges)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
#include <immintrin.h>
typedef unsigned long long ull;
__global__ void post(ull* flag, ull v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void spin(const ull* mailbox) {
  for (unsigned s = 0; s < (1u << 30); s++) { if (__hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 1) break; __builtin_amdgcn_s_sleep(4); }
}
__global__ void stream_rw(ull* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 3 + 1; }
// same traffic, but the stores are non-temporal (bg=3) / write-through at system scope (bg=4): no dirty lines stay in the L2s
__global__ void stream_rw_nt(ull* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(p + i) * 3 + 1, p + i); }
__global__ void stream_rw_sc(ull* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __hip_atomic_store(p + i, p[i] * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// read-only background (bg=5)
__global__ void stream_ro(ull* p, size_t n, ull* out) { ull a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i]; if (a == 12345) *out = a; }
int main() {
  int iters = 3000;
  for (int bg : {0, 2, 3, 4, 5}) for (int n : {1, 16}) {
    std::vector<double> res(n);
    std::vector<std::thread> th;
    std::atomic<int> ready(0); std::atomic<int> done(0);
    std::thread bgth;
    if (bg >= 2) bgth = std::thread([&] {
      hipSetDevice(0); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      ull* p; hipMalloc(&p, 256u << 20);
      size_t cnt = (size_t)(256u << 20) / 8;
      while (done.load() < n) {
        if (bg == 2) hipLaunchKernelGGL(stream_rw, dim3(4096), dim3(256), 0, s, p, cnt);
        else if (bg == 3) hipLaunchKernelGGL(stream_rw_nt, dim3(4096), dim3(256), 0, s, p, cnt);
        else if (bg == 4) hipLaunchKernelGGL(stream_rw_sc, dim3(4096), dim3(256), 0, s, p, cnt);
        else hipLaunchKernelGGL(stream_ro, dim3(4096), dim3(256), 0, s, p, cnt, p);
        hipStreamSynchronize(s);
      }
      hipFree(p); hipStreamDestroy(s);
    });
    for (int t = 0; t < n; t++) th.emplace_back([&, t] {
      hipSetDevice(0);
      hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      ull *h, *d; hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer((void**)&d, h, 0);
      h[0] = 0; h[64] = 0;
      if (bg == 1) hipLaunchKernelGGL(spin, dim3(1), dim3(1024), 0, s2, (const ull*)(d + 64));
      hipLaunchKernelGGL(post, dim3(1), dim3(64), 0, s, d, (ull)0);
      ready++; while (ready.load() < n) _mm_pause();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= iters; i++) { hipLaunchKernelGGL(post, dim3(1), dim3(64), 0, s, d, (ull)i); while (*(volatile ull*)h != (ull)i) _mm_pause(); }
      res[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      *(volatile ull*)(h + 64) = 1; _mm_sfence();
      hipStreamSynchronize(s2); hipStreamSynchronize(s); done++;
      hipHostFree(h); hipStreamDestroy(s); hipStreamDestroy(s2);
    });
    for (auto& x : th) x.join();
    if (bg >= 2) bgth.join();
    double avg = 0, mx = 0; for (double v : res) { avg += v / n; mx = v > mx ? v : mx; }
    printf("bg=%d threads=%2d : %.2f us per launch+post round trip (avg), %.2f max\n", bg, n, avg, mx); fflush(stdout);
  }
  return 0;
}
