// r04: the hand-written gfx950 Goldilocks multiplication (csrc/gl64_gfx950.h) against the compiler's — correctness on edge + random inputs
// (host reference: 128-bit product mod p), dependent-chain latency on one wave, throughput on the full chip, and Poseidon2 compress() throughput
// (one node per lane, 2^21 nodes) with a checksum of the digests. Build twice: default (asm) and -DDP_NO_GFX950_ASM (compiler); the checksums must agree.
#include "../../deep-prove_amd/csrc/poseidon2.h"
#include "../../deep-prove_amd/csrc/poseidon2_fast.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using namespace dp;
__constant__ u64 c_rc[DP_POSEIDON2_RC_WORDS];
__global__ void k_check(const u64* a, const u64* b, const u64* d, u64* om, u64* of, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  om[i] = p2f::canon(p2f::mul(a[i], b[i]));
#ifdef DP_GFX950_ASM
  of[i] = p2f::canon(gx::fma(a[i], b[i], d[i]));
#else
  { unsigned __int128 x = (unsigned __int128)a[i] * b[i] + d[i]; of[i] = p2f::canon(p2f::red128((u64)x, (u64)(x >> 64))); }
#endif
}
__global__ void k_chain(u64* io, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  u64 x = io[i], y = x ^ 0x9E3779B97F4A7C15ULL;
  for (int k = 0; k < iters; k++) { x = p2f::mul(x, y); y = p2f::mul(y, x); }
  io[i] = p2f::canon(x) ^ p2f::canon(y);
}
__global__ void k_sbox_chain(u64* io, int iters) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  u64 x = io[i];
  for (int k = 0; k < iters; k++) x = p2f::sbox(x + 1);
  io[i] = p2f::canon(x);
}
__global__ void __launch_bounds__(256) k_layer(const u64* in, u64* out, size_t cnt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2* p = (const ulonglong2*)(in + 8 * i);
    ulonglong2 x01 = p[0], x23 = p[1], y01 = p[2], y23 = p[3];
    u64 x[4] = {x01.x, x01.y, x23.x, x23.y}, y[4] = {y01.x, y01.y, y23.x, y23.y}, o[4];
    p2f::compress(x, y, o, c_rc);
    ulonglong2* q = (ulonglong2*)(out + 4 * i);
    q[0] = make_ulonglong2(o[0], o[1]);
    q[1] = make_ulonglong2(o[2], o[3]);
  }
}
static u64 ref_mul(u64 a, u64 b, u64 d) { unsigned __int128 x = (unsigned __int128)a * b + d; return (u64)(x % GL_P); }
template <class F> float timed(F f) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); return ms; }
int main() {
#ifdef DP_GFX950_ASM
  printf("variant: gfx950 asm\n");
#else
  printf("variant: compiler\n");
#endif
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  // ---- correctness
  std::vector<u64> edge = {0, 1, 2, 3, 7, GL_EPS - 1, GL_EPS, GL_EPS + 1, GL_EPS + 2, 1ull << 33, 1ull << 48, (1ull << 48) + 1, 1ull << 63, (1ull << 63) + 1, GL_P - 2, GL_P - 1, GL_P, GL_P + 1,
                           ~0ull, ~0ull - 1, ~0ull - GL_EPS, ~0ull - GL_EPS - 1, 0xFFFFFFFE00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x00000001FFFFFFFFull, 0x0000000100000000ull, 0xFFFFFFFF00000000ull,
                           0x8000000000000000ull - 1, 0x0001000000000000ull, 0x0000FFFFFFFFFFFFull, 0x1000000010000001ull};
  std::vector<u64> a, b, d;
  for (u64 x : edge) for (u64 y : edge) for (u64 z : std::vector<u64>{0, 1, GL_P - 1, (u64)~0ull, GL_EPS, (u64)0xFFFFFFFF00000000ull}) { a.push_back(x); b.push_back(y); d.push_back(z); }
  u64 s = 88172645463325252ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < (1 << 22); i++) { u64 x = rnd(), y = rnd(), z = rnd(); if ((i & 15) == 1) x &= 0xFFFFFFFFull; if ((i & 15) == 2) y >>= 16; if ((i & 15) == 3) { x = (x & 0xFFFF) << 48; y = (y & 0xFFFF) << 48; } a.push_back(x); b.push_back(y); d.push_back(z); }
  size_t n = a.size();
  u64 *da, *db, *dd, *dm, *df;
  (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8); (void)hipMalloc(&dd, n * 8); (void)hipMalloc(&dm, n * 8); (void)hipMalloc(&df, n * 8);
  (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dd, dm, df, n);
  std::vector<u64> om(n), of(n);
  (void)hipMemcpy(om.data(), dm, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(of.data(), df, n * 8, hipMemcpyDeviceToHost);
  size_t badm = 0, badf = 0;
  for (size_t i = 0; i < n; i++) {
    if (om[i] != ref_mul(a[i], b[i], 0)) { if (badm < 5) printf("  mul mismatch a=%016llx b=%016llx got %016llx want %016llx\n", (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)om[i], (unsigned long long)ref_mul(a[i], b[i], 0)); badm++; }
    if (of[i] != ref_mul(a[i], b[i], d[i])) { if (badf < 5) printf("  fma mismatch a=%016llx b=%016llx d=%016llx got %016llx want %016llx\n", (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)d[i], (unsigned long long)of[i], (unsigned long long)ref_mul(a[i], b[i], d[i])); badf++; }
  }
  printf("correctness: %zu cases, mul mismatches %zu, fma mismatches %zu\n", n, badm, badf);
  // ---- latency / throughput of the multiplication
  size_t m = 4096 * 256; u64* dio; (void)hipMalloc(&dio, m * 8); (void)hipMemcpy(dio, a.data() + 6000, m * 8, hipMemcpyHostToDevice);
  int it = 20000;
  float l = timed([&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dio, it); });
  printf("one wave, dependent chain: %.1f ns per multiplication\n", 1e6 * l / (2.0 * it));
  float ls = timed([&] { hipLaunchKernelGGL(k_sbox_chain, dim3(1), dim3(64), 0, 0, dio, it / 4); });
  printf("one wave, dependent S-boxes: %.1f ns per x^7\n", 1e6 * ls / (it / 4));
  it = 2000;
  float t = timed([&] { hipLaunchKernelGGL(k_chain, dim3(4096), dim3(256), 0, 0, dio, it); });
  printf("full chip (4096 x 256 threads): %.3e multiplications/s\n", m * 2.0 * it / (t * 1e-3));
  // ---- Poseidon2 compress
  const size_t nn = size_t(1) << 21;
  std::vector<u64> h(8 * nn); for (auto& v : h) v = rnd() % GL_P;
  for (int i = 0; i < 64; i++) h[i] = (i & 1) ? GL_P - 1 - i : i;
  u64 *din, *dout; (void)hipMalloc(&din, 8 * nn * 8); (void)hipMalloc(&dout, 4 * nn * 8);
  (void)hipMemcpy(din, h.data(), 8 * nn * 8, hipMemcpyHostToDevice);
  for (int blocks : {8192, 2048, 1024}) {
    float tt = timed([&] { for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_layer, dim3(blocks), dim3(256), 0, 0, din, dout, nn); }) / 5;
    printf("compress, %5d blocks: %.3f ms = %.3e compress/s\n", blocks, tt, nn / (tt * 1e-3));
  }
  std::vector<u64> r(4 * nn); (void)hipMemcpy(r.data(), dout, 4 * nn * 8, hipMemcpyDeviceToHost);
  u64 cs = 0; for (size_t i = 0; i < 4 * nn; i++) cs = cs * 0x100000001B3ull + r[i];
  u64 ref[4]; poseidon2_compress(h.data(), h.data() + 4, ref, POSEIDON2_RC_HOST);
  size_t badh = 0; for (size_t i = 0; i < 4096; i++) { u64 o[4]; poseidon2_compress(h.data() + 8 * i, h.data() + 8 * i + 4, o, POSEIDON2_RC_HOST); for (int k = 0; k < 4; k++) badh += o[k] != r[4 * i + k]; }
  printf("digest checksum %016llx; first 4096 nodes vs host poseidon2_compress: %zu mismatching words\n", (unsigned long long)cs, badh);
  return (badm || badf || badh) ? 1 : 0;
}
