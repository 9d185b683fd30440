// r04: where do the one-wave sponges of concurrently launched small kernels land? K streams each launch a 20-workgroup kernel of 256 threads whose
// wave 0 runs a chain of p2l_permute at s_setprio 3 (waves 1-3 wait at the barrier, like the table-work waves of a protocol tail); every sponge wave
// records its XCC / SE / CU / SIMD and its time per permutation. Question: do the sponge waves (always wave 0) pile up on one SIMD of a few CUs?
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
#include <map>
#include <algorithm>
namespace dp {
#include "../../deep-prove_amd/csrc/kernels.inc"
// ROT: which wave of the workgroup is the sponge wave: 0 = always wave 0, 1 = (blockIdx.x + salt) & 3
template <int ROT> __global__ void __launch_bounds__(256) k_tail(u64* io, int perms, unsigned long long* rec, int salt) {
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sw = ROT ? ((blockIdx.x + salt) & 3) : 0;
  if (wave == sw) {
    u64 s = io[lane & 7] + blockIdx.x;
    const P2lK pk = p2l_load(lane);
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < perms; k++) s = p2l_permute(s, lane, pk);
    const unsigned long long t1 = wall_clock64();
    if (lane < 8) io[64 + lane] = s;
    if (lane == 0) {
      unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
      rec[2 * (salt * 64 + blockIdx.x)] = t1 - t0; rec[2 * (salt * 64 + blockIdx.x) + 1] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
  }
  __syncthreads();
}
}  // namespace dp
using namespace dp;
template <int ROT> void run(int K, int wgs, int perms, u64* d, unsigned long long* drec, std::vector<hipStream_t>& st) {
  (void)hipMemset(drec, 0, 16 * 64 * 64);
  for (int k = 0; k < K; k++) hipLaunchKernelGGL((dp::k_tail<ROT>), dim3(wgs), dim3(256), 0, st[k], d, perms, drec, k);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * 64 * 64); (void)hipMemcpy(h.data(), drec, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, int> per_simd, per_cu; double sum = 0, mx = 0; int n = 0;
  std::vector<double> us;
  int simd_hist[4] = {0, 0, 0, 0};
  for (int k = 0; k < K; k++) for (int b = 0; b < wgs; b++) {
    unsigned long long t = h[2 * (k * 64 + b)], id = h[2 * (k * 64 + b) + 1];
    unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned long long cukey = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
    per_cu[cukey]++; per_simd[(cukey << 4) | simd]++; simd_hist[simd]++;
    double u = t * 10.0 / perms / 1e3; us.push_back(u); sum += u; mx = std::max(mx, u); n++;
  }
  std::sort(us.begin(), us.end());
  int worst = 0; for (auto& kv : per_simd) worst = std::max(worst, kv.second);
  printf("  %s: %3d kernels x %d workgroups: %zu CUs used, %zu SIMDs used, most sponge waves on one SIMD %d, by SIMD id [%d %d %d %d]; us per permutation median %.2f mean %.2f max %.2f\n", ROT ? "rotated sponge wave" : "sponge = wave 0     ", K, wgs, per_cu.size(), per_simd.size(), worst, simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3], us[us.size() / 2], sum / n, mx);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dp::c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST));
  u64 h[8]; for (int i = 0; i < 8; i++) h[i] = 0x0123456789ABCDEFull * (i + 1) % GL_P;
  u64* d; (void)hipMalloc(&d, 4096); (void)hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
  unsigned long long* drec; (void)hipMalloc(&drec, 16 * 64 * 64);
  std::vector<hipStream_t> st(64); for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int K : {1, 4, 8, 16, 22}) { run<0>(K, 20, 128, d, drec, st); run<1>(K, 20, 128, d, drec, st); }
  return 0;
}
