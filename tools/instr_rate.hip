// gfx950 VALU issue-rate probe for the integer instructions Goldilocks arithmetic is made of (tools/, not product code).
// Every kernel runs ITERS x 16 independent instances of one instruction per lane on 4096 x 256 threads; the table gives
// wave-instructions per second and cycles per wave-instruction per SIMD relative to v_add_u32 (= 4 cycles for wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 512
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL32(NAME, ASM)                                                                                     \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                   \
    uint32_t a[16], b = seed + threadIdx.x, c = seed * 3 + 1;                                                   \
    for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;                                                 \
    for (int it = 0; it < ITERS; it++) {                                                                        \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));           \
    }                                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 16; i++) s ^= a[i];                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                             \
  }
#define KERNEL64(NAME, ASM)                                                                                     \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                   \
    uint64_t a[16]; uint32_t b = seed + threadIdx.x, c = seed * 3 + 1; uint64_t d = ((uint64_t)seed << 32) | threadIdx.x;  \
    for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;                                                 \
    for (int it = 0; it < ITERS; it++) {                                                                        \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(d));   \
    }                                                                                                           \
    uint64_t s = 0; for (int i = 0; i < 16; i++) s ^= a[i];                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);                             \
  }
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %2")
KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL32(k_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 22")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_add_u32_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_pair_add_mul24, "v_add_u32 %0, %0, %1\n v_mul_u32_u24 %0, %0, %2")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_u64_u32_s, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %3\n v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %3, %3")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %3")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %3")
template <class K> double run(K k, uint32_t* d, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 7u + r);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 3.0;
}
int main() {
  const int blocks = 4096; uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double simds = p.multiProcessorCount * 4.0, clk = p.clockRate * 1e3;
  const double winstr = (double)blocks * 4 * ITERS * 16;  // wave-instructions per launch
  printf("device %s, %d CUs, clock %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, clk / 1e6);
  double base = 0;
#define RUN(NAME, N) { double ms = run(NAME, d, blocks); double cyc = (ms * 1e-3) * clk * simds / (winstr * N); if (!base) base = cyc; printf("%-22s %8.3f ms  %6.2f cycles/wave-instr (x%.2f of v_add_u32)\n", #NAME, ms, cyc, cyc / base); }
  RUN(k_add_u32, 1) RUN(k_add_co_u32, 1) RUN(k_addc_co_u32, 1) RUN(k_cndmask, 1) RUN(k_mul_lo_u32, 1) RUN(k_mul_hi_u32, 1)
  RUN(k_mad_u64_u32, 1) RUN(k_mad_u64_u32_s, 1) RUN(k_lshl_add_u64, 1) RUN(k_cmp_u64, 2) RUN(k_lshlrev_b64, 1)
  // round 5 (review item 7: is there a multiplier cheaper than v_mad_u64_u32 to build the S-box on?): the narrow multipliers, the packed 16-bit forms, the
  // dot products, the shift / mask helpers of a limb representation, a DPP add, FP64 (exact for 26-bit limbs), and an add + 24-bit multiply pair
  RUN(k_mul_u32_u24, 1) RUN(k_mad_u32_u24, 1) RUN(k_mul_hi_u32_u24, 1) RUN(k_mad_u32_u16, 1) RUN(k_dot4_u32_u8, 1) RUN(k_dot2_u32_u16, 1)
  RUN(k_pk_mul_lo_u16, 1) RUN(k_pk_mad_u16, 1) RUN(k_pk_add_u16, 1) RUN(k_add3, 1) RUN(k_lshl_add_u32, 1) RUN(k_alignbit, 1) RUN(k_and_or, 1)
  RUN(k_add_u32_dpp, 1) RUN(k_cmp_u32, 2) RUN(k_pair_add_mul24, 2) RUN(k_fma_f64, 1) RUN(k_mul_f64, 1) RUN(k_add_f64, 1)
  return 0;
}
