// Head-of-line probe (tools/, not product code): how long does a ONE-WAVE kernel on its own stream take from launch to
// completion while B other streams run "wide" work, as a function of HOW that work is shaped?
//   thin : 16384 workgroups x 256 threads, short workgroups   (round 1's Merkle-layer cohort launches)
//   fat  : 1024 workgroups x 256 threads, 16x the work each   (all workgroups resident at once: the launch phase is short)
//   excl : thin on B-2 streams + 2 streams of 8 x 1024-thread workgroups with 84 KB of LDS (whole-CU workgroups)
//   small: thin on B-2 streams + 2 streams of 8 x 256-thread workgroups, no LDS (the same serial work, not exclusive)
// Reports the probe latency (median / p90) and the background throughput (work units / ms) per shape.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
__global__ void __launch_bounds__(256) k_work(unsigned long long* out, int iters, int reps) {
  unsigned long long x = blockIdx.x * 256 + threadIdx.x + 1, acc = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    acc ^= x;
  }
  if (acc == 42) out[0] = acc;
}
__global__ void __launch_bounds__(1024) k_serial(unsigned long long* out, int iters) {
  extern __shared__ unsigned long long lds[];
  unsigned long long x = threadIdx.x + 1;
  if (threadIdx.x < 64) for (int i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;  // one wave works, like a sponge
  if (x == 42) { lds[0] = x; out[0] = lds[0]; }
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16;
  unsigned long long* d; CK(hipMalloc(&d, 1024)); unsigned* cnt; CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
  CK(hipFuncSetAttribute((const void*)k_serial, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  std::vector<hipStream_t> st(B); for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipStream_t probe; CK(hipStreamCreateWithFlags(&probe, hipStreamNonBlocking));
  const int ITERS = 2000;  // ~ a few us per workgroup pass
  const char* names[] = {"idle", "thin", "fat", "thin+excl", "thin+small", "fat+small"};
  for (int shape = 0; shape < 6; shape++) {
    CK(hipDeviceSynchronize());
    const int L = 60;  // launches per background stream
    auto t0 = std::chrono::steady_clock::now();
    double units = 0;
    if (shape > 0) for (int l = 0; l < L; l++) for (int b = 0; b < B; b++) {
      const bool special = b >= B - 2 && shape >= 3;
      if (special) {
        if (shape == 3) hipLaunchKernelGGL(k_serial, dim3(8), dim3(1024), 84 * 1024, st[b], d, 400000);
        else hipLaunchKernelGGL(k_serial, dim3(8), dim3(256), 0, st[b], d, 400000);
      } else {
        const bool fat = shape == 2 || shape == 5;
        if (fat) hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, st[b], d, ITERS, 16);
        else hipLaunchKernelGGL(k_work, dim3(16384), dim3(256), 0, st[b], d, ITERS, 1);
        units += 16384;
      }
    }
    std::vector<double> lat;
    bool bg_running = true;
    for (int i = 0; i < 400 && bg_running; i++) {
      auto a = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, probe, cnt);
      CK(hipStreamSynchronize(probe));
      lat.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
      if (shape > 0 && (i & 15) == 15) { bg_running = false; for (auto& s : st) if (hipStreamQuery(s) == hipErrorNotReady) { bg_running = true; break; } }
    }
    CK(hipDeviceSynchronize());
    double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::sort(lat.begin(), lat.end());
    printf("%-11s B=%2d  probe n=%3zu median %8.1f us  p90 %8.1f us  max %8.1f us | background %.1f ms, %.0f workgroup-units/ms\n", names[shape], B, lat.size(),
           lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back(), wall, units / wall);
  }
  return 0;
}
