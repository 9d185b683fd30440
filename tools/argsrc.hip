// Where may a lock-step launch keep its per-proof argument packs? One kernel launch serves B proofs (blockIdx.z = proof)
// and every workgroup starts by fetching ITS arguments. Candidates for the table of packs:
//   0 kernarg   : one pack by value (the single-proof launch, reference point)
//   1 devmem    : table in hipMalloc memory, refreshed by a hipMemcpyAsync before the launch (one more stream command)
//   2 hostmap   : table in pinned host memory mapped into the device (every uniform load is a PCIe read unless cached)
//   3 bar       : table in fine-grained VRAM written directly by the host through the PCIe BAR (if the platform allows it)
// The kernel does `iters` dependent, dynamically indexed reads of the pack (what a persistent sumcheck round does with
// its table pointers), so the per-read latency of each source shows. Prints us per launch+wait for each source.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
typedef unsigned long long ull;
struct Pack { ull tab[96]; ull* flag; ull seq; int iters; int pad; };  // ~800 B, like ScPersistArgs
__device__ __forceinline__ void body(const Pack& p, ull* out) {
  ull idx = threadIdx.x & 1, acc = 0;
  for (int i = 0; i < p.iters; i++) { ull v = p.tab[(idx + acc) % 96]; acc += v; }
  if (threadIdx.x == 0) { if (acc == 0xdeadbeef) out[0] = acc; __hip_atomic_store(p.flag, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__global__ void k_byval(Pack p, ull* out) { body(p, out); }
__global__ void k_table(const Pack* t, ull* out) { body(t[blockIdx.z], out); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
int main() {
  hipSetDevice(0);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  ull *hflag, *dflag; CK(hipHostMalloc((void**)&hflag, 4096, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
  ull* out; CK(hipMalloc(&out, 64));
  Pack* dev; CK(hipMalloc(&dev, 64 * sizeof(Pack)));
  Pack *hm, *hm_dev; CK(hipHostMalloc((void**)&hm, 64 * sizeof(Pack), hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&hm_dev, hm, 0));
  Pack *hm_nc, *hm_nc_dev; CK(hipHostMalloc((void**)&hm_nc, 64 * sizeof(Pack), hipHostMallocMapped | hipHostMallocNonCoherent)); CK(hipHostGetDevicePointer((void**)&hm_nc_dev, hm_nc, 0));
  Pack* bar = nullptr;
  hipError_t be = hipExtMallocWithFlags((void**)&bar, 64 * sizeof(Pack), hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s ptr=%p\n", hipGetErrorString(be), (void*)bar);
  bool bar_ok = false;
  if (be == hipSuccess && bar) {
    hipPointerAttribute_t at; memset(&at, 0, sizeof at);
    hipError_t pe = hipPointerGetAttributes(&at, bar);
    printf("  attributes: %s type=%d hostPointer=%p devicePointer=%p\n", hipGetErrorString(pe), (int)at.type, at.hostPointer, at.devicePointer);
    const char* e = getenv("TRY_BAR");
    bar_ok = e && atoi(e);  // a CPU store to VRAM segfaults when the BAR does not cover it: only tried on request
  }
  Pack p; for (int i = 0; i < 96; i++) p.tab[i] = (i * 7 + 1) % 5; p.flag = dflag; p.pad = 0;
  const int B = 8, reps = 2000;
  for (int iters : {0, 16, 256}) {
    p.iters = iters;
    for (int src = 0; src < 5; src++) {
      if (src == 4 && !bar_ok) continue;
      ull seq = *(volatile ull*)hflag;
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; r++) {
        p.seq = ++seq;
        if (src == 0) hipLaunchKernelGGL(k_byval, dim3(1), dim3(64), 0, s, p, out);
        else {
          Pack* host = src == 1 || src == 2 ? hm : src == 3 ? hm_nc : bar;
          for (int b = 0; b < B; b++) { host[b] = p; }
          _mm_sfence();
          if (src == 1) CK(hipMemcpyAsync(dev, hm, B * sizeof(Pack), hipMemcpyHostToDevice, s));
          const Pack* t = src == 1 ? dev : src == 2 ? hm_dev : src == 3 ? hm_nc_dev : bar;
          hipLaunchKernelGGL(k_table, dim3(1, 1, B), dim3(64), 0, s, t, out);
        }
        while (*(volatile ull*)hflag != seq) _mm_pause();
        if (src != 0) hipStreamSynchronize(s);  // all B workgroups done before the table is rewritten
      }
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      const char* nm[] = {"kernarg(1 proof)", "devmem+memcpy(B=8)", "hostmap coherent(B=8)", "hostmap noncoherent(B=8)", "bar(B=8)"};
      printf("iters=%3d %-26s %.2f us per launch+wait\n", iters, nm[src], us);
    }
  }
  return 0;
}
