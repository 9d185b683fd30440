// Goldilocks multiplication for gfx950, written instruction by instruction (device code only).
//
// Why: the compiler's 64 x 64 -> 128 product plus 128 -> 64 reduction is 23 VALU instructions and 4 s_nop on gfx950 (five v_mov_b32
// to build zero-extended operand pairs, two v_cmp_*_u64 + v_cndmask pairs to re-derive carries it already had in VCC). Every hot
// kernel of this library is bound by exactly that sequence: the Poseidon2 Merkle layers (1 040 multiplications per node), the fused
// sumcheck rounds (VALU issue 0.93) and the one-wave sponge of every protocol tail (a dependent chain that pays per
// instruction). The sequence below is 12 VALU instructions: four v_mad_u64_u32 for the product (the 64-bit accumulate of the multiplier
// carries the columns), one more for hl * (2^32 - 1) + lo, and carry-chain adds that consume the carries where they are produced.
//
// Arithmetic (same field element as gl64.h gl_mul / poseidon2_fast.h p2f::mul, any u64 representative on output, ANY u64 on input):
//   a = a1:a0, b = b1:b0.   P = a0 b0;  Q = a1 b0 + (P >> 32);  R + cR 2^64 = a0 b1 + Q;  H = a1 b1 + (R >> 32) + cR 2^32
//   a b = x0 + x1 2^32 + H 2^64 with x0 = P.lo, x1 = R.lo, H = hh:hl.      2^64 = eps = 2^32 - 1, 2^96 = -1 (mod p):
//   a b = (x0 + x1 2^32 + hl eps) - hh.   u + c 2^64 = x1:x0 + hl eps (one v_mad_u64_u32 with carry out), c 2^64 = c 2^32 - c:
//   r = (u.lo - hh - c) + (u.hi + c) 2^32. With c = 1 the wrapped u is <= 2^64 - 2^33, so neither limb can leave its range; with c = 0 the
//   high limb underflows only if u < hh (hl = 0, x1 = 0, x0 < hh: e.g. 2^48 * 2^48) — then r -= eps once, on a wave-uniform branch that random
//   data never takes (probability 2^-64 per lane) and that is exact when taken.
//
// Registers: the block works in ten fixed VGPRs (DP_GLT0 .. DP_GLT0+9, default v[48:57]) because a 64-bit operand of v_mad_u64_u32 must be an
// even-aligned pair and inline asm cannot name the halves of a compiler-allocated pair. SGPR hazards (VALU writes an SGPR / VCC, VALU reads
// it as carry: 2 wait states on gfx940+) are covered by instruction order and explicit s_nop.
//
// The host (and the CPU SIMT emulator of tests/support/kernel_emul) compile the portable forms in poseidon2_fast.h / gl64.h instead.
#pragma once
#include "gl64.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(DP_NO_GFX950_ASM)
#define DP_GFX950_ASM 1
namespace dp {
namespace gx {

// a * b, any representative
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u32 r0, r1; u64 sc;
  asm("v_mad_u64_u32 v[48:49], vcc, %3, %5, 0\n"          // P = a0 b0                      v48 = x0
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"             // {P.hi, 0}
      "v_mad_u64_u32 v[52:53], vcc, %4, %5, v[50:51]\n"    // Q = a1 b0 + P.hi
      "v_mad_u64_u32 v[54:55], %2, %3, %6, v[52:53]\n"     // R = a0 b1 + Q, carry cR        v54 = x1
      "v_lshrrev_b64 v[50:51], 32, v[54:55]\n"             // {R.hi, 0}
      "v_mad_u64_u32 v[56:57], vcc, %4, %6, v[50:51]\n"    // H = a1 b1 + R.hi
      "v_mov_b32 v49, v54\n"                               // v[48:49] = x1:x0
      "v_addc_co_u32_e64 v57, vcc, 0, v57, %2\n"           // H += cR 2^32                   v56 = hl, v57 = hh
      "v_mad_u64_u32 v[52:53], %2, v56, -1, v[48:49]\n"    // u = hl eps + x1:x0, carry c
      "s_nop 1\n"
      "v_subb_co_u32_e64 %0, vcc, v52, v57, %2\n"          // r0 = u.lo - hh - c, borrow b
      "v_addc_co_u32_e64 v53, %2, 0, v53, %2\n"            // u.hi + c
      "s_nop 0\n"
      "v_subbrev_co_u32_e32 %1, vcc, 0, v53, vcc\n"        // r1 = u.hi + c - b, borrow: the rare case
      "s_cbranch_vccz .Ldp_glm_%=\n"
      "v_cndmask_b32_e64 v50, 0, -1, vcc\n"
      "v_sub_co_u32_e32 %0, vcc, %0, v50\n"
      "s_nop 1\n"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n"
      ".Ldp_glm_%=:\n"
      : "=&v"(r0), "=&v"(r1), "=&s"(sc)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
  return ((u64)r1 << 32) | r0;
}
// a * b + d (d any u64), any representative: the addend rides in the first multiplier's accumulate, its carry in the column above
__device__ __forceinline__ u64 fma(u64 a, u64 b, u64 d) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u32 r0, r1; u64 sc;
  asm("v_mad_u64_u32 v[48:49], %2, %3, %5, %7\n"           // P + cP 2^64 = a0 b0 + d
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"             // {P.hi, 0}
      "s_nop 0\n"
      "v_addc_co_u32_e64 v51, vcc, 0, 0, %2\n"             // {P.hi, cP}
      "v_mad_u64_u32 v[52:53], vcc, %4, %5, v[50:51]\n"    // Q = a1 b0 + P.hi + cP 2^32   (< 2^64: it is the exact column sum)
      "v_mad_u64_u32 v[54:55], %2, %3, %6, v[52:53]\n"
      "v_lshrrev_b64 v[50:51], 32, v[54:55]\n"
      "v_mad_u64_u32 v[56:57], vcc, %4, %6, v[50:51]\n"
      "v_mov_b32 v49, v54\n"
      "v_addc_co_u32_e64 v57, vcc, 0, v57, %2\n"
      "v_mad_u64_u32 v[52:53], %2, v56, -1, v[48:49]\n"
      "s_nop 1\n"
      "v_subb_co_u32_e64 %0, vcc, v52, v57, %2\n"
      "v_addc_co_u32_e64 v53, %2, 0, v53, %2\n"
      "s_nop 0\n"
      "v_subbrev_co_u32_e32 %1, vcc, 0, v53, vcc\n"
      "s_cbranch_vccz .Ldp_glf_%=\n"
      "v_cndmask_b32_e64 v50, 0, -1, vcc\n"
      "v_sub_co_u32_e32 %0, vcc, %0, v50\n"
      "s_nop 1\n"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n"
      ".Ldp_glf_%=:\n"
      : "=&v"(r0), "=&v"(r1), "=&s"(sc)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(d)
      : "vcc", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
  return ((u64)r1 << 32) | r0;
}

// a * k for a 32-bit k (7 for the extension's X^2 = 7), any representative: the product has three limbs, 2^64 hl = eps hl, one carry
__device__ __forceinline__ u64 mul_small(u64 a, u32 k) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
  u64 r, sc;
  asm("v_mad_u64_u32 v[48:49], vcc, %2, %4, 0\n"
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"             // {P.hi, 0}
      "v_mad_u64_u32 v[52:53], vcc, %3, %4, v[50:51]\n"    // v52 = x1, v53 = hl (< k)
      "v_mov_b32 v49, v52\n"
      "v_mad_u64_u32 v[54:55], %1, v53, -1, v[48:49]\n"    // u = hl eps + x1:x0, carry c (then the wrapped u is < k 2^32)
      "s_nop 1\n"
      "v_cndmask_b32_e64 v50, 0, -1, %1\n"                 // {c ? eps : 0, 0}
      "v_lshl_add_u64 %0, v[50:51], 0, v[54:55]\n"
      : "=&v"(r), "=&s"(sc)
      : "v"(a0), "v"(a1), "v"(k)
      : "vcc", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  return r;
}

}  // namespace gx
}  // namespace dp
#endif
