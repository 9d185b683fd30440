// Micro-benchmark of the host<->device round trip used by the persistent sumcheck kernel.
// Variants: mailbox (host->device word) in pinned host memory polled over PCIe, or in fine-grained device memory written
// by the CPU through the BAR; flag (device->host word) always in pinned host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <atomic>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void echo(const unsigned long long* mailbox, unsigned long long* flag, int iters, int sleep) {
  for (int i = 1; i <= iters; i++) {
    unsigned long long got = 0;
    for (unsigned spin = 0; spin < (1u << 24); spin++) {
      got = __hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (got == (unsigned long long)i) break;
      if (sleep) __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(flag, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double run(unsigned long long* mail_host_view, unsigned long long* mail_dev_view, unsigned long long* flag_h, unsigned long long* flag_d, int iters, int sleep, hipStream_t s) {
  *mail_host_view = 0; *flag_h = 0;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, s, (const unsigned long long*)mail_dev_view, flag_d, iters, sleep);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 1; i <= iters; i++) {
    *(volatile unsigned long long*)mail_host_view = i;
    _mm_sfence();
    while (*(volatile unsigned long long*)flag_h != (unsigned long long)i) _mm_pause();
  }
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *hbuf, *hbuf_d; CK(hipHostMalloc((void**)&hbuf, 4096, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&hbuf_d, hbuf, 0));
  int iters = 20000;
  printf("host-memory mailbox, no sleep : %.2f us / round trip\n", run(hbuf, hbuf_d, hbuf + 64, hbuf_d + 64, iters, 0, s));
  printf("host-memory mailbox, s_sleep 4: %.2f us / round trip\n", run(hbuf, hbuf_d, hbuf + 64, hbuf_d + 64, iters, 1, s));
  unsigned long long* dfine = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&dfine, 4096, hipDeviceMallocFinegrained);
  if (e == hipSuccess) {
    // is it CPU-accessible? try (large BAR): guarded by a pointer-attribute query
    hipPointerAttribute_t at; e = hipPointerGetAttributes(&at, dfine);
    printf("fine-grained device alloc ok (type %d); trying CPU stores through the BAR...\n", (int)at.type); fflush(stdout);
    printf("device-memory mailbox, no sleep: %.2f us / round trip\n", run(dfine, dfine, hbuf + 64, hbuf_d + 64, iters, 0, s));
  } else printf("hipExtMallocWithFlags(finegrained) failed: %s\n", hipGetErrorString(e));
  return 0;
}
