// What sets the duration of a SMALL kernel in a chain of dependent launches when other hardware queues are busy — the number of busy QUEUES (command
// processor / pipe service) or the RESOURCES their kernels hold (wave slots, issue slots)? One probe stream runs a chain of K tiny kernels (one wave each,
// every kernel records the GPU clock at entry and exit); N busy streams run back-to-back "load" kernels of one of three kinds:
//   spin   : 21 workgroups x 256 threads that watch the clock for T us (a merged one-workgroup tail: long, holds almost nothing)
//   valu   : a chip-filling grid of dependent 64-bit multiply-adds, ~T us per launch (a merged Merkle layer: every wave slot, every issue slot)
//   valucap: the same work from 512 workgroups (grid-stride); cap128 / cap64 / cap32: from 128 / 64 / 32 workgroups — with N x G below the chip's ~2 000
//            workgroup slots every launch can be PLACED at once and leaves its queue's pipe at once
// The probe kernel raises its wave priority (s_setprio 3) in the kinds whose name ends in "+p".
//   hbm    : 64 workgroups per queue streaming 256 MB device -> device (the batch opening's streaming kernels, capped)
//   d2h    : 8 workgroups per queue copying 4 MB device -> pinned host memory per launch (k_download: the proofs leave the device)
//   mix    : a third of the queues each of cap64 / hbm / d2h
// Probe variants (second table): "args" = the probe reads one word from NON-COHERENT host-mapped memory first (a cohort's argument packs), "pub" = it stores one
// word to COHERENT host memory last (a publication), "both".
// Printed per (kind, N): the probe chain's time per link = (exit of the last - entry of the first) / K, and the median entry(i+1) - exit(i) gap.
// usage: qprobe [K=400] [T_us=2000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
typedef unsigned long long ull;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void probe(ull* t, int i, int prio) { if (prio) __builtin_amdgcn_s_setprio(3); if (threadIdx.x == 0) { t[2 * i] = wall_clock64(); } __syncthreads(); if (threadIdx.x == 0) t[2 * i + 1] = wall_clock64(); }
__global__ void probe_io(ull* t, int i, const ull* hargs, ull* hpub, int mode) {
  __builtin_amdgcn_s_setprio(3);
  ull a = 0;
  if (threadIdx.x == 0) { t[2 * i] = wall_clock64(); if (mode & 1) a = __builtin_nontemporal_load(hargs + (i & 63) * 8); }
  __syncthreads();
  if (threadIdx.x == 0) { if (mode & 2) __hip_atomic_store(hpub + (i & 63) * 8, a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); t[2 * i + 1] = wall_clock64() + (a >> 63); }
}
// a merged one-workgroup tail as the library launches it: 21 workgroups of 256 threads, `lds` bytes of dynamic LDS each, ~170 VGPRs, T us of dependent work
__global__ void __launch_bounds__(256) tail_like(ull ticks, ull* out) {
  extern __shared__ ull sm[];
  sm[threadIdx.x] = threadIdx.x;
  asm volatile("v_mov_b32 v170, 0" ::: "v170");  // (the register allocation of the real tails: 169-200 VGPRs, two waves per SIMD)
  const ull t0 = wall_clock64(); ull a = sm[(threadIdx.x + 1) & 255];
  while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 64; i++) a = a * a + 3; }
  if (a == 0x1234567) out[0] = a;
}
__global__ void stream_copy(const ulonglong2* src, ulonglong2* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void spin(ull ticks) { const ull t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void valu(ull* out, int iters) {
  ull a = threadIdx.x + 1, b = blockIdx.x * 2654435761ull + 12345;
  for (size_t w = blockIdx.x; w < (size_t)iters; w += gridDim.x)
    for (int i = 0; i < 4096; i++) { a = a * b + (a >> 7); b = b * a + (b >> 9); }
  if (a == 0x1234567 && b == 17) out[0] = a;
}
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 400;
  const double T_us = argc > 2 ? atof(argv[2]) : 2000.0;
  int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
  const double tick_us = 1000.0 / (double)wc_khz;
  ull* ts; CK(hipMalloc((void**)&ts, sizeof(ull) * 2 * K)); ull* sink; CK(hipMalloc((void**)&sink, 64));
  std::vector<ull> h(2 * K);
  hipStream_t ps; CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  std::vector<hipStream_t> bs(23);
  for (auto& s : bs) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // calibrate the valu load: work units so that one uncapped launch alone takes ~T_us
  int units = 4096;
  for (int it = 0; it < 4; it++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, bs[0]); hipLaunchKernelGGL(valu, dim3(std::min(units, 65536)), dim3(256), 0, bs[0], sink, units); hipEventRecord(b, bs[0]); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (it) units = std::max(256, (int)(units * (T_us / 1000.0) / std::max(ms, 0.01f)));
    hipEventDestroy(a); hipEventDestroy(b);
  }
  printf("wall clock %.1f MHz; valu load: %d work units of 4096 x 2 dependent 64-bit multiply-adds per lane per launch (~%.0f us alone)\n", wc_khz / 1000.0, units, T_us);
  const char* kinds[] = {"none", "spin", "valu", "valucap", "cap128", "cap64", "cap32", "cap64+p", "valu+p"};
  const int capof[] = {0, 0, 0, 512, 128, 64, 32, 64, 0};
  const bool skip_first = argc > 3 && atoi(argv[3]);
  for (int kind = 0; kind < 9 && !skip_first; kind++) {
    for (int N : {0, 4, 8, 12, 16, 20, 22}) {
      if ((kind == 0) != (N == 0)) continue;
      if (kind >= 4 && N != 8 && N != 16 && N != 22) continue;
      const int prio = kind >= 7;
      std::vector<long> done(23, 0);
      volatile bool stop = false;
      std::vector<std::thread> th;
      for (int q = 0; q < N; q++) th.emplace_back([&, q] {
        hipSetDevice(0);
        while (!stop) {
          for (int r = 0; r < 4; r++) {
            if (kind == 1) hipLaunchKernelGGL(spin, dim3(21), dim3(256), 0, bs[q], (ull)(T_us / tick_us));
            else if (!capof[kind]) hipLaunchKernelGGL(valu, dim3(std::min(units, 65536)), dim3(256), 0, bs[q], sink, units);
            else hipLaunchKernelGGL(valu, dim3(capof[kind]), dim3(256), 0, bs[q], sink, units);
          }
          hipStreamSynchronize(bs[q]); done[q] += 4;
        }
      });
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
      auto t0 = std::chrono::steady_clock::now();
      long d0 = 0; for (int q = 0; q < N; q++) d0 += done[q];
      for (int i = 0; i < K; i++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, ps, ts, i, prio);
      CK(hipStreamSynchronize(ps));
      const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      long d1 = 0; for (int q = 0; q < N; q++) d1 += done[q];
      stop = true; for (auto& t : th) t.join();
      for (int q = 0; q < N; q++) hipStreamSynchronize(bs[q]);
      CK(hipMemcpy(h.data(), ts, sizeof(ull) * 2 * K, hipMemcpyDeviceToHost));
      std::vector<double> gap, dur;
      for (int i = 0; i + 1 < K; i++) gap.push_back((double)(h[2 * (i + 1)] - h[2 * i + 1]) * tick_us);
      for (int i = 0; i < K; i++) dur.push_back((double)(h[2 * i + 1] - h[2 * i]) * tick_us);
      std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
      printf("%-8s busy queues %2d: chain %8.1f us per link (host wall %8.1f); gap exit->next entry p50 %8.1f p90 %8.1f max %8.1f; inside the probe kernel p50 %6.2f us; load launches completed meanwhile %.1f per ms\n", kinds[kind], N,
             (double)(h[2 * K - 1] - h[0]) * tick_us / K, wall_us / K, gap[gap.size() / 2], gap[gap.size() * 9 / 10], gap.back(), dur[dur.size() / 2], (double)(d1 - d0) / (wall_us / 1000.0));
      fflush(stdout);
    }
  }
  // ---- second table: what the probe touches over PCIe, under loads that can all be placed
  ull *hargs, *hargs_d, *hpub, *hpub_d;
  CK(hipHostMalloc((void**)&hargs, 4096, hipHostMallocMapped | hipHostMallocNonCoherent)); CK(hipHostGetDevicePointer((void**)&hargs_d, hargs, 0));
  CK(hipHostMalloc((void**)&hpub, 4096, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&hpub_d, hpub, 0));
  for (int i = 0; i < 512; i++) hargs[i] = i;
  const size_t HB = size_t(256) << 20, DB = size_t(4) << 20;
  std::vector<char*> dsrc(22), ddst(22), hdst(22), hdst_d(22);
  for (int q = 0; q < 22; q++) { CK(hipMalloc((void**)&dsrc[q], HB)); CK(hipMalloc((void**)&ddst[q], HB)); CK(hipHostMalloc((void**)&hdst[q], DB, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&hdst_d[q], hdst[q], 0)); }
  CK(hipFuncSetAttribute((const void*)tail_like, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const char* loads[] = {"idle", "cap64", "hbm", "d2h", "mix", "tail16k", "tail100k", "t100k+v", "t16k+v"};
  const char* pmodes[] = {"plain", "args", "pub", "both"};
  for (int load = 0; load < 9; load++) for (int pm = 0; pm < 4; pm++) {
    if (load >= 5 && pm != 0 && pm != 3) continue;
    const int N = load ? 22 : 0;
    volatile bool stop = false;
    std::vector<std::thread> th;
    for (int q = 0; q < N; q++) th.emplace_back([&, q] {
      hipSetDevice(0);
      const int kind = load == 4 ? 1 + q % 3 : load;
      while (!stop) {
        for (int r = 0; r < 4; r++) {
          if (kind >= 5) {  // tails of 2 ms (16 KB / 100 KB of LDS), in the "+v" loads every second launch is a capped VALU kernel instead
            const size_t lds = (kind == 5 || kind == 8) ? 16 * 1024 : 100 * 1024;
            if (kind >= 7 && (r & 1)) hipLaunchKernelGGL(valu, dim3(64), dim3(256), 0, bs[q], sink, units / 8);
            else hipLaunchKernelGGL(tail_like, dim3(21), dim3(256), lds, bs[q], (ull)(T_us / tick_us), sink);
          }
          else if (kind == 1) hipLaunchKernelGGL(valu, dim3(64), dim3(256), 0, bs[q], sink, units / 8);
          else if (kind == 2) hipLaunchKernelGGL(stream_copy, dim3(64), dim3(256), 0, bs[q], (const ulonglong2*)dsrc[q], (ulonglong2*)ddst[q], HB / 16);
          else hipLaunchKernelGGL(stream_copy, dim3(8), dim3(256), 0, bs[q], (const ulonglong2*)dsrc[q], (ulonglong2*)hdst_d[q], DB / 16);
        }
        hipStreamSynchronize(bs[q]);
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; i++) hipLaunchKernelGGL(probe_io, dim3(1), dim3(64), 0, ps, ts, i, (const ull*)hargs_d, hpub_d, pm);
    CK(hipStreamSynchronize(ps));
    const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    stop = true; for (auto& t : th) t.join();
    for (int q = 0; q < N; q++) hipStreamSynchronize(bs[q]);
    CK(hipMemcpy(h.data(), ts, sizeof(ull) * 2 * K, hipMemcpyDeviceToHost));
    std::vector<double> gap, dur;
    for (int i = 0; i + 1 < K; i++) gap.push_back((double)(h[2 * (i + 1)] - h[2 * i + 1]) * tick_us);
    for (int i = 0; i < K; i++) dur.push_back((double)(h[2 * i + 1] - h[2 * i]) * tick_us);
    std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
    printf("load %-6s probe %-6s: chain %8.1f us per link; gap p50 %8.1f p90 %8.1f; inside p50 %7.2f p90 %7.2f us\n", loads[load], pmodes[pm], wall_us / K, gap[gap.size() / 2], gap[gap.size() * 9 / 10], dur[dur.size() / 2], dur[dur.size() * 9 / 10]);
    fflush(stdout);
  }
  return 0;
}
