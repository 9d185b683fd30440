// micro-benchmark: Goldilocks multiplication on gfx950 — the compiler's 64x64->128 (mul_lo/mul_hi mix) against a
// hand-split 4 x v_mad_u64_u32 product; dependent-chain latency (one wave) and throughput (full device)
#include "../deep-prove_amd/csrc/gl64.h"
#include <hip/hip_runtime.h>
#include <cstdio>
using namespace dp;
__device__ __forceinline__ u64 mul_new(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00 = (u64)a0 * b0;
  u64 mid = (u64)a0 * b1 + (p00 >> 32);
  u64 mid2 = (u64)a1 * b0 + (u32)mid;
  u64 hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
  u64 lo = (mid2 << 32) | (u32)p00;
  u64 hh = hi >> 32;
  u64 t0 = lo - hh;
  if (lo < hh) t0 -= GL_EPS;
  u32 h32 = (u32)hi; u32 t1lo = 0u - h32; u32 t1hi = h32 - (h32 != 0 ? 1u : 0u);
  u64 t1 = ((u64)t1hi << 32) | t1lo;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}
template <int MODE> __global__ void k_chain(u64* io, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  u64 x = io[i], y = x ^ 0x9E3779B97F4A7C15ULL; if (y >= GL_P) y -= GL_P;
  for (int k = 0; k < iters; k++) { x = MODE == 0 ? gl_mul(x, y) : mul_new(x, y); y = gl_add(y, x); }
  io[i] = x;
}
template <int MODE> float run(int blocks, int threads, int iters, u64* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10);
  hipEventRecord(a); hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  size_t n = 4096 * 256; u64* d; hipMalloc(&d, n * 8);
  u64* h = new u64[n]; for (size_t i = 0; i < n; i++) h[i] = (i * 0x9E3779B97F4A7C15ULL + 12345) % GL_P;
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  // correctness of the split product against the library one
  hipLaunchKernelGGL(k_chain<0>, dim3(64), dim3(256), 0, 0, d, 1000); u64* r0 = new u64[n]; hipMemcpy(r0, d, 64 * 256 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_chain<1>, dim3(64), dim3(256), 0, 0, d, 1000); u64* r1 = new u64[n]; hipMemcpy(r1, d, 64 * 256 * 8, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < 64 * 256; i++) bad += r0[i] != r1[i];
  printf("mismatches %zu of %d\n", bad, 64 * 256);
  int it = 20000;
  float l0 = run<0>(1, 64, it, d), l1 = run<1>(1, 64, it, d);
  printf("latency  (1 wave, dependent chain): lib %.1f ns/mul+add, split %.1f ns/mul+add\n", 1e6 * l0 / it, 1e6 * l1 / it);
  float w0 = run<0>(1, 1024, it, d), w1 = run<1>(1, 1024, it, d);
  printf("one CU   (1024 threads):            lib %.1f ns/iter,   split %.1f ns/iter\n", 1e6 * w0 / it, 1e6 * w1 / it);
  it = 2000;
  float t0 = run<0>(4096, 256, it, d), t1 = run<1>(4096, 256, it, d);
  printf("throughput (4096x256 threads): lib %.2f Gmul/s, split %.2f Gmul/s\n", n * (double)it / (t0 * 1e6), n * (double)it / (t1 * 1e6));
  return 0;
}
