// r04: why does a one-wave sponge run 3x slower next to hashing waves? One wave runs a dependent chain of p2l_permute while (a) the chip is idle,
// (b) every SIMD also hosts Poseidon2 hashing waves (k_bg fills the chip from another stream). Variables: s_setprio 0 / 3 on the sponge wave, and the
// size of the code the chain walks through (COPIES inlined copies of the permutation per loop iteration: 1 copy = ~4 KB, 32 copies = ~120 KB > the
// 64 KB instruction cache two CUs share).   usage: contend
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
#include <vector>
#include <chrono>
namespace dp {
#include "../../deep-prove_amd/csrc/kernels.inc"
template <int PRIO, int COPIES> __global__ void k_sponge(u64* io, int iters, unsigned long long* ticks) {
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  u64 s = io[lane & 7] + blockIdx.x;
  const P2lK pk = p2l_load(lane);
  const unsigned long long t0 = wall_clock64();
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int c = 0; c < COPIES; c++) s = p2l_permute(s, lane, pk);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane < 8) io[8 + lane] = s;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_bg(u64* io, int iters, volatile int* stop) {
  u64 x[4], y[4], o[4];
  for (int i = 0; i < 4; i++) { x[i] = io[(threadIdx.x + i) & 7] + blockIdx.x; y[i] = x[i] ^ 0x5555; }
  for (int k = 0; k < iters; k++) { p2f::compress(x, y, o, c_rc); for (int i = 0; i < 4; i++) { x[i] = o[i]; y[i] ^= o[3 - i]; } if ((k & 15) == 15 && *stop) break; }
  if (x[0] == 0x1234) io[20] = x[1];
}
}  // namespace dp
using namespace dp;
template <int PRIO, int COPIES> double run(u64* d, unsigned long long* dt, int nblocks, int perms, hipStream_t s) {
  hipLaunchKernelGGL((dp::k_sponge<PRIO, COPIES>), dim3(nblocks), dim3(64), 0, s, d, perms / COPIES, dt);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned long long> h(nblocks); (void)hipMemcpy(h.data(), dt, nblocks * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  return sum / nblocks * 10.0 / perms;  // ns per permutation (100 MHz wall clock)
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dp::c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  u64 h[8]; for (int i = 0; i < 8; i++) h[i] = 0x0123456789ABCDEFull * (i + 1) % GL_P;
  u64* d; (void)hipMalloc(&d, 4096); (void)hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  u64* dbg; (void)hipMalloc(&dbg, 4096); (void)hipMemcpy(dbg, h, 64, hipMemcpyHostToDevice);
  unsigned long long* dt; (void)hipMalloc(&dt, 8 * 4096);
  int* stop; (void)hipHostMalloc(&stop, 4, hipHostMallocMapped); *stop = 0;
  hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  const int perms = 256;
  for (int nb : {1, 64, 256}) {
    printf("---- %d sponge workgroups (one wave each), %d permutations\n", nb, perms);
    for (int load = 0; load < 3; load++) {  // 0: idle chip, 1: 4 hashing waves per SIMD, 2: 8 (as many as fit)
      if (load) { *stop = 0; hipLaunchKernelGGL(dp::k_bg, dim3(256 * (load == 1 ? 4 : 8)), dim3(256), 0, sb, dbg, 1 << 20, stop); std::this_thread::sleep_for(std::chrono::milliseconds(20)); }
      double a = run<0, 1>(d, dt, nb, perms, sa), b = run<1, 1>(d, dt, nb, perms, sa), c = run<0, 8>(d, dt, nb, perms, sa), e = run<1, 8>(d, dt, nb, perms, sa), f = run<0, 32>(d, dt, nb, perms, sa), g = run<1, 32>(d, dt, nb, perms, sa);
      if (load) { *stop = 1; (void)hipStreamSynchronize(sb); }
      printf("  load %d (%s): us per permutation  1 copy: prio0 %.2f prio3 %.2f | 8 copies: prio0 %.2f prio3 %.2f | 32 copies: prio0 %.2f prio3 %.2f\n", load, load == 0 ? "idle chip" : load == 1 ? "4 x 256-thread hashing workgroups per CU" : "8 per CU", a / 1e3, b / 1e3, c / 1e3, e / 1e3, f / 1e3, g / 1e3);
    }
  }
  return 0;
}
