// Poseidon2 compress throughput on gfx950: the canonical formulation (poseidon2.h, what k_merkle_layer ran in round 1)
// against the wide-accumulation / any-representative one (poseidon2_fast.h). One compress per lane over 2^21 nodes,
// results compared word for word. tools/, not product code.   usage: p2bench
#include "../deep-prove_amd/csrc/poseidon2.h"
#include "../deep-prove_amd/csrc/poseidon2_fast.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using namespace dp;
__constant__ u64 c_rc[DP_POSEIDON2_RC_WORDS];
template <int V> __global__ void __launch_bounds__(256) k_layer(const u64* in, u64* out, size_t cnt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2* p = (const ulonglong2*)(in + 8 * i);
    ulonglong2 x01 = p[0], x23 = p[1], y01 = p[2], y23 = p[3];
    u64 x[4] = {x01.x, x01.y, x23.x, x23.y}, y[4] = {y01.x, y01.y, y23.x, y23.y}, o[4];
    if (V == 0) poseidon2_compress(x, y, o, c_rc); else p2f::compress(x, y, o, c_rc);
    ulonglong2* q = (ulonglong2*)(out + 4 * i);
    q[0] = make_ulonglong2(o[0], o[1]);
    q[1] = make_ulonglong2(o[2], o[3]);
  }
}
template <int V> double run(const u64* in, u64* out, size_t n, int blocks, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k_layer<V>, dim3(blocks), dim3(256), 0, 0, in, out, n);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_layer<V>, dim3(blocks), dim3(256), 0, 0, in, out, n);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}
int main() {
  const size_t n = size_t(1) << 21;
  std::vector<u64> h(8 * n);
  u64 x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % GL_P; }
  for (int i = 0; i < 64; i++) h[i] = (i & 1) ? GL_P - 1 - i : i;  // edge values
  u64 *din, *d0, *d1;
  (void)hipMalloc(&din, 8 * n * 8); (void)hipMalloc(&d0, 4 * n * 8); (void)hipMalloc(&d1, 4 * n * 8);
  (void)hipMemcpy(din, h.data(), 8 * n * 8, hipMemcpyHostToDevice);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  for (int blocks : {8192, 2048}) {
    double t0 = run<0>(din, d0, n, blocks, 5), t1 = run<1>(din, d1, n, blocks, 5);
    printf("blocks %5d: canonical %.3f ms = %.3e compress/s (%.3e mul/s) | fast %.3f ms = %.3e compress/s (%.3e mul/s) | x%.2f\n", blocks,
           t0, n / (t0 * 1e-3), 1040.0 * n / (t0 * 1e-3), t1, n / (t1 * 1e-3), 1040.0 * n / (t1 * 1e-3), t0 / t1);
  }
  std::vector<u64> r0(4 * n), r1(4 * n);
  (void)hipMemcpy(r0.data(), d0, 4 * n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(r1.data(), d1, 4 * n * 8, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < 4 * n; i++) bad += r0[i] != r1[i];
  u64 ref[4]; poseidon2_compress(h.data(), h.data() + 4, ref, POSEIDON2_RC_HOST);
  bool host_ok = ref[0] == r0[0] && ref[1] == r0[1] && ref[2] == r0[2] && ref[3] == r0[3];
  printf("fast vs canonical on device: %zu mismatching words of %zu; node 0 vs host: %s\n", bad, 4 * n, host_ok ? "ok" : "DIFFERENT");
  return bad != 0 || !host_ok;
}
