// r04: the one-wave sponge permutation of the protocol tails (p2l_permute, csrc/kernels.inc) alone on a wave: microseconds per permutation of a
// dependent chain (what a Fiat-Shamir round pays), checked against the host permutation.   usage: p2l_bench
#include "../../deep-prove_amd/csrc/dev.h"
#include "../../deep-prove_amd/csrc/poseidon2.h"
#include "../../deep-prove_amd/csrc/poseidon2_fast.h"
#include "../../deep-prove_amd/csrc/gl64_lazy.h"
#include "../../deep-prove_amd/csrc/sumcheck.h"
#include "../../deep-prove_amd/csrc/fiber.h"
#include "../../deep-prove_amd/csrc/logup_tail.h"
#include "../../deep-prove_amd/csrc/classic_tail.h"
#include "../../deep-prove_amd/csrc/dense_tail.h"
#include "../../deep-prove_amd/csrc/eqsum_tail.h"
#include "../../deep-prove_amd/csrc/deleg_tail.h"
#include "../../deep-prove_amd/csrc/commit_tail.h"
#include "../../deep-prove_amd/csrc/sponge_host.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
namespace dp {
#include "../../deep-prove_amd/csrc/kernels.inc"
__global__ void k_p2l_chain(u64* io, int iters) {
  const int lane = threadIdx.x & 63;
  u64 s = io[lane & 7];
  const P2lK pk = p2l_load(lane);
  for (int k = 0; k < iters; k++) s = p2l_permute(s, lane, pk);
  if (lane < 8) io[8 + lane] = s;
}
}  // namespace dp
using namespace dp;
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dp::c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  u64 h[16]; for (int i = 0; i < 8; i++) h[i] = 0x0123456789ABCDEFull * (i + 1) % GL_P;
  u64* d; (void)hipMalloc(&d, 128); (void)hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  const int it = 2000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(dp::k_p2l_chain, dim3(1), dim3(64), 0, 0, d, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); hipLaunchKernelGGL(dp::k_p2l_chain, dim3(1), dim3(64), 0, 0, d, it); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipMemcpy(h + 8, d + 8, 64, hipMemcpyDeviceToHost);
  u64 ref[8]; memcpy(ref, h, 64); for (int k = 0; k < it; k++) poseidon2_permute(ref, POSEIDON2_RC_HOST);
  int bad = 0; for (int i = 0; i < 8; i++) bad += ref[i] != h[8 + i];
  printf("p2l_permute, one wave, dependent chain of %d: %.2f us per permutation (%.0f cycles at 2.4 GHz); vs host permutation: %d differing words\n", it, 1e3 * ms / it, 2.4e6 * ms / it, bad);
  return bad != 0;
}
