// Does the single-stream launch round trip degrade once many streams (HSA queues) exist? Creates N streams, touches each
// with one kernel, then times {launch one-wave kernel; spin on its tag} on stream 0.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <immintrin.h>
typedef unsigned long long ull;
__global__ void post(ull* flag, ull v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
  ull *h, *d; hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer((void**)&d, h, 0);
  std::vector<hipStream_t> ss;
  for (int target : {1, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64}) {
    while ((int)ss.size() < target) { hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipLaunchKernelGGL(post, dim3(1), dim3(64), 0, s, d + 8, (ull)1); hipStreamSynchronize(s); ss.push_back(s); }
    int iters = 2000; h[0] = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; i++) { hipLaunchKernelGGL(post, dim3(1), dim3(64), 0, ss[0], d, (ull)i); while (*(volatile ull*)h != (ull)i) _mm_pause(); }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    // and the last created stream
    h[0] = 0; t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; i++) { hipLaunchKernelGGL(post, dim3(1), dim3(64), 0, ss.back(), d, (ull)i); while (*(volatile ull*)h != (ull)i) _mm_pause(); }
    double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    printf("streams=%2d : stream0 %.2f us, newest stream %.2f us per launch+post round trip\n", target, us, us2); fflush(stdout);
  }
  return 0;
}
