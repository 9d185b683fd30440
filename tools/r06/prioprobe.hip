// Do hardware queue priorities (hipStreamCreateWithPriority) let a chain of small dependent kernels through while other queues fill the chip with VALU-bound grids?
// qprobe.hip measured 1.8 ms per link of such a chain beside 22 queues of chip-filling launches, all at the default priority. Here the probe stream is created at
// the HIGH priority and / or the load streams at the LOW one. Printed per (probe priority, load priority, busy queues): time per link of a chain of K one-wave kernels,
// time per link of a chain of K "cohort-like" kernels (21 workgroups x 256 threads, ~20 us of work each), and the load launches completed per ms meanwhile.
// usage: prioprobe [K=300] [T_us=2000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
typedef unsigned long long ull;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void probe(ull* t, int i) { if (threadIdx.x == 0) t[2 * i] = wall_clock64(); __syncthreads(); if (threadIdx.x == 0) t[2 * i + 1] = wall_clock64(); }
__global__ void probe21(ull* t, int i, ull ticks) {
  const ull t0 = wall_clock64(); ull a = threadIdx.x;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * i] = t0;
  while (wall_clock64() - t0 < ticks) { for (int k = 0; k < 32; k++) a = a * a + 3; }
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * i + 1] = wall_clock64() + (a == 0x1234567);
}
__global__ void valu(ull* out, int iters) {
  ull a = threadIdx.x + 1, b = blockIdx.x * 2654435761ull + 12345;
  for (size_t w = blockIdx.x; w < (size_t)iters; w += gridDim.x)
    for (int i = 0; i < 4096; i++) { a = a * b + (a >> 7); b = b * a + (b >> 9); }
  if (a == 0x1234567 && b == 17) out[0] = a;
}
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 300;
  const double T_us = argc > 2 ? atof(argv[2]) : 2000.0;
  int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
  const double tick_us = 1000.0 / (double)wc_khz;
  int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: least %d .. greatest %d\n", lo, hi);
  ull* ts; CK(hipMalloc((void**)&ts, sizeof(ull) * 2 * K)); ull* sink; CK(hipMalloc((void**)&sink, 64));
  std::vector<ull> h(2 * K);
  hipStream_t ps[2]; CK(hipStreamCreateWithPriority(&ps[0], hipStreamNonBlocking, 0)); CK(hipStreamCreateWithPriority(&ps[1], hipStreamNonBlocking, hi));
  std::vector<hipStream_t> bs[2]; bs[0].resize(22); bs[1].resize(22);
  for (auto& s : bs[0]) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
  for (auto& s : bs[1]) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
  int units = 4096;
  for (int it = 0; it < 4; it++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, bs[0][0]); hipLaunchKernelGGL(valu, dim3(std::min(units, 65536)), dim3(256), 0, bs[0][0], sink, units); hipEventRecord(b, bs[0][0]); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (it) units = std::max(256, (int)(units * (T_us / 1000.0) / std::max(ms, 0.01f)));
    hipEventDestroy(a); hipEventDestroy(b);
  }
  printf("valu load: %d work units per launch (~%.0f us alone, chip-filling grid)\n", units, T_us);
  for (int cap : {0, 256})
  for (int pp = 0; pp < 2; pp++) for (int lp = 0; lp < 2; lp++) for (int N : {8, 22}) for (int wide = 0; wide < 2; wide++) {
    std::vector<long> done(22, 0);
    volatile bool stop = false;
    std::vector<std::thread> th;
    for (int q = 0; q < N; q++) th.emplace_back([&, q] {
      hipSetDevice(0);
      while (!stop) {
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(valu, dim3(cap ? cap : std::min(units, 65536)), dim3(256), 0, bs[lp][q], sink, units);
        hipStreamSynchronize(bs[lp][q]); done[q] += 4;
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    auto t0 = std::chrono::steady_clock::now();
    long d0 = 0; for (int q = 0; q < N; q++) d0 += done[q];
    for (int i = 0; i < K; i++) { if (wide) hipLaunchKernelGGL(probe21, dim3(21), dim3(256), 0, ps[pp], ts, i, (ull)(20.0 / tick_us)); else hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, ps[pp], ts, i); }
    CK(hipStreamSynchronize(ps[pp]));
    const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    long d1 = 0; for (int q = 0; q < N; q++) d1 += done[q];
    stop = true; for (auto& t : th) t.join();
    for (int q = 0; q < N; q++) hipStreamSynchronize(bs[lp][q]);
    CK(hipMemcpy(h.data(), ts, sizeof(ull) * 2 * K, hipMemcpyDeviceToHost));
    std::vector<double> dur;
    for (int i = 0; i < K; i++) dur.push_back((double)(h[2 * i + 1] - h[2 * i]) * tick_us);
    std::sort(dur.begin(), dur.end());
    printf("load grid %-5s probe %-6s prio %-6s load prio %-6s busy queues %2d: chain %8.1f us per link; inside the probe kernel p50 %7.2f p90 %8.2f us; load launches completed %.2f per ms\n",
           cap ? "256" : "full", wide ? "21x256" : "1 wave", pp ? "HIGH" : "normal", lp ? "LOW" : "normal", N, wall_us / K, dur[dur.size() / 2], dur[dur.size() * 9 / 10], (double)(d1 - d0) / (wall_us / 1000.0));
    fflush(stdout);
  }
  return 0;
}
