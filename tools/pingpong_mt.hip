// N concurrent host<->device ping-pong pairs (one host thread + one stream + one echo kernel each): does the round trip
// of the persistent-sumcheck protocol degrade when several proofs are in flight?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>
#include <immintrin.h>
__global__ void echo(const unsigned long long* mailbox, unsigned long long* flag, int iters, int sleep) {
  for (int i = 1; i <= iters; i++) {
    unsigned long long got = 0;
    for (unsigned spin = 0; spin < (1u << 24); spin++) {
      got = __hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (got == (unsigned long long)i) break;
      if (sleep) __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(flag, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main(int argc, char** argv) {
  int iters = 20000;
  for (int n : {1, 2, 4, 8, 16}) for (int sleep : {0, 32}) {
    std::vector<double> res(n);
    std::vector<std::thread> th;
    std::atomic<int> ready(0);
    for (int t = 0; t < n; t++) th.emplace_back([&, t] {
      hipSetDevice(0);
      hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      unsigned long long *h, *d; hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer((void**)&d, h, 0);
      h[0] = 0; h[64] = 0;
      hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, s, (const unsigned long long*)d, d + 64, iters, sleep);
      ready++; while (ready.load() < n) _mm_pause();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= iters; i++) { *(volatile unsigned long long*)h = i; _mm_sfence(); while (*(volatile unsigned long long*)(h + 64) != (unsigned long long)i) _mm_pause(); }
      res[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      hipStreamSynchronize(s); hipHostFree(h); hipStreamDestroy(s);
    });
    for (auto& x : th) x.join();
    double avg = 0; for (double v : res) avg += v / n;
    printf("pairs=%2d s_sleep=%2d : %.2f us per round trip (avg)\n", n, sleep, avg);
  }
  return 0;
}
