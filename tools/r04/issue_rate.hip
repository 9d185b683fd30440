// r04 micro-benchmark: what does ONE wave pay per instruction on gfx950? (dependent vs independent issue, SGPR-carry hazards, DPP moves, s_nop)
// The one-wave protocol tails (k_logup_tail & co) run a Poseidon2 sponge on a single wave: this is the price list their instruction
// sequences are designed against.   usage: issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
// every body is 256 copies of a snippet; the loop runs it `it` times
#define KERNEL(name, snippet, clob...)                                                        \
  __global__ void name(u64* out, int it) {                                                    \
    u64 t0 = clock64(), w0 = wall_clock64();                                                  \
    for (int k = 0; k < it; k++) { asm volatile(R256(snippet) ::: clob); }                    \
    u64 t1 = clock64(), w1 = wall_clock64();                                                  \
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; } \
  }
KERNEL(k_add_dep, "v_add_u32 v10, v10, v11\n", "v10")
KERNEL(k_add_ind4, "v_add_u32 v10, v10, v11\n v_add_u32 v12, v12, v11\n v_add_u32 v13, v13, v11\n v_add_u32 v14, v14, v11\n", "v10", "v12", "v13", "v14")
KERNEL(k_mad_dep, "v_mad_u64_u32 v[10:11], vcc, v12, v13, v[10:11]\n", "v10", "v11", "vcc")
KERNEL(k_mad_dep_lo, "v_mad_u64_u32 v[10:11], vcc, v10, v13, v[14:15]\n", "v10", "v11", "vcc")
KERNEL(k_mad_ind4, "v_mad_u64_u32 v[10:11], vcc, v12, v13, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v12, v13, v[14:15]\n v_mad_u64_u32 v[16:17], vcc, v12, v13, v[16:17]\n v_mad_u64_u32 v[18:19], vcc, v12, v13, v[18:19]\n", "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19", "vcc")
KERNEL(k_lshladd_dep, "v_lshl_add_u64 v[10:11], v[10:11], 0, v[12:13]\n", "v10", "v11")
KERNEL(k_lshladd_ind4, "v_lshl_add_u64 v[10:11], v[10:11], 0, v[12:13]\n v_lshl_add_u64 v[14:15], v[14:15], 0, v[12:13]\n v_lshl_add_u64 v[16:17], v[16:17], 0, v[12:13]\n v_lshl_add_u64 v[18:19], v[18:19], 0, v[12:13]\n", "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19")
KERNEL(k_addc_pair_nop, "v_add_co_u32 v10, vcc, v10, v12\n s_nop 1\n v_addc_co_u32 v11, vcc, v11, v13, vcc\n", "v10", "v11", "vcc")
KERNEL(k_addc_pair_fill, "v_add_co_u32 v10, vcc, v10, v12\n v_add_u32 v14, v14, v12\n v_add_u32 v15, v15, v12\n v_addc_co_u32 v11, vcc, v11, v13, vcc\n", "v10", "v11", "v14", "v15", "vcc")
KERNEL(k_cmp_cnd_nop, "v_cmp_lt_u32 vcc, v10, v12\n s_nop 1\n v_cndmask_b32 v10, v10, v13, vcc\n", "v10", "vcc")
KERNEL(k_mov_dep, "v_mov_b32 v10, v11\n v_mov_b32 v11, v10\n", "v10", "v11")
KERNEL(k_dpp_dep, "v_mov_b32_dpp v10, v11 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp v11, v10 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n s_nop 1\n", "v10", "v11")
KERNEL(k_dpp_add_dep, "v_add_u32_dpp v10, v10, v10 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n s_nop 1\n", "v10")
KERNEL(k_snop0, "s_nop 0\n", "v10")
KERNEL(k_snop1, "s_nop 1\n", "v10")
KERNEL(k_snop3, "s_nop 3\n", "v10")
KERNEL(k_mulhi_dep, "v_mul_hi_u32 v10, v10, v11\n", "v10")
KERNEL(k_mullo_dep, "v_mul_lo_u32 v10, v10, v11\n", "v10")
KERNEL(k_pkmov_dep, "v_pk_mov_b32 v[10:11], v[10:11], v[12:13] op_sel:[1,0]\n", "v10", "v11")
KERNEL(k_sub_subb_chain, "v_sub_co_u32 v10, vcc, v10, v12\n s_nop 1\n v_subbrev_co_u32 v11, vcc, 0, v11, vcc\n s_nop 1\n v_cndmask_b32 v14, 0, -1, vcc\n", "v10", "v11", "v14", "vcc")
KERNEL(k_readlane, "v_readlane_b32 s10, v10, 3\n s_nop 3\n v_add_u32 v10, s10, v10\n", "v10", "s10")
KERNEL(k_salu_dep, "s_add_u32 s10, s10, s11\n", "s10", "scc")
KERNEL(k_smul_dep, "s_mul_i32 s10, s10, s11\n", "s10")
KERNEL(k_smulhi_dep, "s_mul_hi_u32 s10, s10, s11\n", "s10")

template <class K> void run(const char* name, K k, int instr_per_snip, int blocks, int threads, u64* d) {
  u64 h[2];
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 4);
  (void)hipDeviceSynchronize();
  const int it = 64;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, it);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  double n = 256.0 * it * instr_per_snip;
  printf("%-18s waves/blk %2d  %7.2f clk64-ticks/instr  %7.3f ns/instr (wall 100 MHz)\n", name, threads / 64, h[0] / n, h[1] * 10.0 / n);
}
int main() {
  u64* d; (void)hipMalloc(&d, 1 << 16);
#define RUN(k, n) run(#k, k, n, 1, 64, d); run(#k, k, n, 1, 512, d);
  RUN(k_add_dep, 1) RUN(k_add_ind4, 4) RUN(k_mad_dep, 1) RUN(k_mad_dep_lo, 1) RUN(k_mad_ind4, 4) RUN(k_lshladd_dep, 1) RUN(k_lshladd_ind4, 4)
  RUN(k_addc_pair_nop, 2) RUN(k_addc_pair_fill, 4) RUN(k_cmp_cnd_nop, 2) RUN(k_mov_dep, 2) RUN(k_dpp_dep, 2) RUN(k_dpp_add_dep, 1)
  RUN(k_snop0, 1) RUN(k_snop1, 1) RUN(k_snop3, 1) RUN(k_mulhi_dep, 1) RUN(k_mullo_dep, 1) RUN(k_pkmov_dep, 1) RUN(k_sub_subb_chain, 3)
  RUN(k_readlane, 2) RUN(k_salu_dep, 1) RUN(k_smul_dep, 1) RUN(k_smulhi_dep, 1)
  return 0;
}
