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

#if DP_GX_ON
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

// ------------------------------------------------------------------------------------------------ forms for the ONE-WAVE sponge (kernels.inc, p2l_*)
// A lone wave pays for every instruction slot (4-5 cycles each, s_nop wait states included: profiles/r04_issue_rate_one_wave.txt) and ~38 cycles for
// the uniform branch on the rare borrow above. Here the rare condition of a block is written to an SGPR pair (`pend`) by the instruction that produces
// it and OR-ed into `flag` in the NEXT block, in a slot that would otherwise be a wait state; whoever runs a chain of these checks `flag | pend` once at
// the end and repeats the chain with the exact forms (FIX = true: the correction always executed, no flags) if it is set — probability ~2^-60 per
// permutation on real data, exact when it happens.
#define DP_GX_CLOB "vcc", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"
#define DP_GX_TAIL_FLAGS                                     \
      "v_mad_u64_u32 v[52:53], %[sc], v56, -1, v[48:49]\n"   \
      "s_or_b64 %[flag], %[flag], %[pend]\n"                 \
      "s_nop 0\n"                                            \
      "v_subb_co_u32_e64 %[r0], vcc, v52, v57, %[sc]\n"      \
      "v_addc_co_u32_e64 v53, %[sc], 0, v53, %[sc]\n"        \
      "s_nop 0\n"                                            \
      "v_subbrev_co_u32_e64 %[r1], %[pend], 0, v53, vcc\n"
#define DP_GX_TAIL_FIX                                       \
      "v_mad_u64_u32 v[52:53], %[sc], v56, -1, v[48:49]\n"   \
      "s_nop 1\n"                                            \
      "v_subb_co_u32_e64 %[r0], vcc, v52, v57, %[sc]\n"      \
      "v_addc_co_u32_e64 v53, %[sc], 0, v53, %[sc]\n"        \
      "s_nop 0\n"                                            \
      "v_subbrev_co_u32_e32 %[r1], vcc, 0, v53, vcc\n"       \
      "s_nop 1\n"                                            \
      "v_cndmask_b32_e64 v50, 0, -1, vcc\n"                  \
      "v_sub_co_u32_e32 %[r0], vcc, %[r0], v50\n"            \
      "s_nop 1\n"                                            \
      "v_subbrev_co_u32_e32 %[r1], vcc, 0, %[r1], vcc\n"
#define DP_GX_PRODUCT                                        \
      "v_mad_u64_u32 v[48:49], vcc, %[a0], %[b0], 0\n"       \
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"               \
      "v_mad_u64_u32 v[52:53], vcc, %[a1], %[b0], v[50:51]\n" \
      "v_mad_u64_u32 v[54:55], %[sc], %[a0], %[b1], v[52:53]\n" \
      "v_lshrrev_b64 v[50:51], 32, v[54:55]\n"               \
      "v_mad_u64_u32 v[56:57], vcc, %[a1], %[b1], v[50:51]\n" \
      "v_mov_b32 v49, v54\n"                                 \
      "v_addc_co_u32_e64 v57, vcc, 0, v57, %[sc]\n"
template <bool FIX> __device__ __forceinline__ u64 mul_f(u64 a, u64 b, u64& flag, u64& pend) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u32 r0, r1; u64 sc;
  if (FIX) asm(DP_GX_PRODUCT DP_GX_TAIL_FIX : [r0] "=&v"(r0), [r1] "=&v"(r1), [sc] "=&s"(sc) : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1) : DP_GX_CLOB);
  else asm(DP_GX_PRODUCT DP_GX_TAIL_FLAGS : [r0] "=&v"(r0), [r1] "=&v"(r1), [sc] "=&s"(sc), [flag] "+s"(flag), [pend] "+s"(pend) : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1) : DP_GX_CLOB, "scc");
  return ((u64)r1 << 32) | r0;
}
// x * d + A + B 2^22 + C 2^44 for d < p and limb sums A, B < 2^28, C < 2^24 (the internal layer's d_i x_i + row sum, the sum still in limbs):
// the limbs ride in the multipliers' accumulates (t = A + B 2^22 under the first product, C 2^12 in the 2^32 column); the two carries of that
// column have weight 2^96 and go into hh. 17 VALU instructions instead of a 6-instruction join and the 13-instruction fma.
#define DP_GX_FMA3_PRODUCT                                   \
      "v_mad_u64_u32 v[58:59], vcc, %[B], %[k22], 0\n"       \
      "v_mad_u64_u32 v[58:59], vcc, %[A], 1, v[58:59]\n"     \
      "v_mad_u64_u32 v[48:49], %[sc], %[a0], %[b0], v[58:59]\n" \
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"               \
      "s_nop 0\n"                                            \
      "v_addc_co_u32_e64 v51, vcc, 0, 0, %[sc]\n"            \
      "v_mad_u64_u32 v[50:51], vcc, %[C], %[k12], v[50:51]\n" \
      "v_mad_u64_u32 v[52:53], %[sq], %[a1], %[b0], v[50:51]\n" \
      "v_mad_u64_u32 v[54:55], %[sc], %[a0], %[b1], v[52:53]\n" \
      "v_lshrrev_b64 v[50:51], 32, v[54:55]\n"               \
      "v_mad_u64_u32 v[56:57], vcc, %[a1], %[b1], v[50:51]\n" \
      "v_mov_b32 v49, v54\n"                                 \
      "v_addc_co_u32_e64 v57, vcc, 0, v57, %[sq]\n"          \
      "v_addc_co_u32_e64 v57, vcc, 0, v57, %[sc]\n"
template <bool FIX> __device__ __forceinline__ u64 fma3_f(u64 x, u64 d, u32 A, u32 B, u32 C, u64& flag, u64& pend) {
  const u32 a0 = (u32)x, a1 = (u32)(x >> 32), b0 = (u32)d, b1 = (u32)(d >> 32), k22 = 1u << 22, k12 = 1u << 12;
  u32 r0, r1; u64 sc, sq;
  if (FIX) asm(DP_GX_FMA3_PRODUCT DP_GX_TAIL_FIX : [r0] "=&v"(r0), [r1] "=&v"(r1), [sc] "=&s"(sc), [sq] "=&s"(sq)
               : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [A] "v"(A), [B] "v"(B), [C] "v"(C), [k22] "s"(k22), [k12] "s"(k12) : DP_GX_CLOB, "v58", "v59");
  else asm(DP_GX_FMA3_PRODUCT DP_GX_TAIL_FLAGS : [r0] "=&v"(r0), [r1] "=&v"(r1), [sc] "=&s"(sc), [sq] "=&s"(sq), [flag] "+s"(flag), [pend] "+s"(pend)
           : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [A] "v"(A), [B] "v"(B), [C] "v"(C), [k22] "s"(k22), [k12] "s"(k12) : DP_GX_CLOB, "v58", "v59", "scc");
  return ((u64)r1 << 32) | r0;
}
// A + B 2^22 + C 2^44 (limb sums A, B, C < 2^28) -> any u64 representative: t = A + B 2^22, Hh = C 2^12 + (t >> 32) (< 2^41), value = t.lo + Hh.lo 2^32 + Hh.hi 2^64;
// the carry of the last multiply-add (needs the low words close to 2^64: rare) is the block's rare condition
template <bool FIX> __device__ __forceinline__ u64 join3_f(u32 A, u32 B, u32 C, u64& flag, u64& pend) {
  const u32 k22 = 1u << 22, k12 = 1u << 12;
  u32 r0, r1; u64 r, sc;
  if (FIX) {
    asm("v_mad_u64_u32 v[48:49], vcc, %[B], %[k22], 0\n"
        "v_mad_u64_u32 v[48:49], vcc, %[A], 1, v[48:49]\n"
        "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"
        "v_mad_u64_u32 v[50:51], vcc, %[C], %[k12], v[50:51]\n"
        "v_mov_b32 v49, v50\n"
        "v_mad_u64_u32 v[52:53], %[sc], v51, -1, v[48:49]\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 v50, 0, -1, %[sc]\n"
        "v_add_co_u32_e32 %[r0], vcc, v52, v50\n"
        "s_nop 1\n"
        "v_addc_co_u32_e32 %[r1], vcc, 0, v53, vcc\n"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [sc] "=&s"(sc) : [A] "v"(A), [B] "v"(B), [C] "v"(C), [k22] "s"(k22), [k12] "s"(k12) : "vcc", "v48", "v49", "v50", "v51", "v52", "v53");
    return ((u64)r1 << 32) | r0;
  }
  asm("v_mad_u64_u32 v[48:49], vcc, %[B], %[k22], 0\n"
      "v_mad_u64_u32 v[48:49], vcc, %[A], 1, v[48:49]\n"
      "s_or_b64 %[flag], %[flag], %[pend]\n"
      "v_lshrrev_b64 v[50:51], 32, v[48:49]\n"
      "v_mad_u64_u32 v[50:51], vcc, %[C], %[k12], v[50:51]\n"
      "v_mov_b32 v49, v50\n"
      "v_mad_u64_u32 %[r], %[pend], v51, -1, v[48:49]\n"
      : [r] "=&v"(r), [flag] "+s"(flag), [pend] "+s"(pend) : [A] "v"(A), [B] "v"(B), [C] "v"(C), [k22] "s"(k22), [k12] "s"(k12) : "vcc", "scc", "v48", "v49", "v50", "v51");
  return r;
}

// The linear layers of the lane-parallel Poseidon2 on three limbs of 22 / 22 / 20 bits, written out so that every cross-lane operand is the DPP source of
// an add and no instruction reads through DPP a register written less than two instructions earlier (the gfx950 hazard that otherwise costs s_nop slots):
// the three limbs are interleaved. State word i of a permutation lives in lane i of its group of 8.
#define DP_DPP_XOR1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DP_DPP_ROT1 " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define DP_DPP_ROT2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
#define DP_DPP_REV " quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n"
#define DP_DPP_HMIR " row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define DP_GX_SPLIT                                          \
      "v_and_b32 %[A], 0x3fffff, %[x0]\n"                    \
      "v_alignbit_b32 %[B], %[x1], %[x0], 22\n"              \
      "v_lshrrev_b32 %[C], 12, %[x1]\n"                      \
      "v_and_b32 %[B], 0x3fffff, %[B]\n"
// x -> limbs -> external layer circ(2 M4, M4) (+ this lane's round constant, as limbs): out = 2 t + t(lane ^ 4) + rc, t_j = (quad sum) + v_j + 2 v_{j+1}
__device__ __forceinline__ void lin_full(u64 x, u32 rA, u32 rB, u32 rC, u32& oA, u32& oB, u32& oC) {
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  u32 A, B, C, pA, pB, pC, qA, qB, qC, nA, nB, nC;
  asm(DP_GX_SPLIT
      "v_add_u32_dpp %[pA], %[A], %[A]" DP_DPP_XOR1
      "v_add_u32_dpp %[pC], %[C], %[C]" DP_DPP_XOR1
      "v_mov_b32_dpp %[nA], %[A]" DP_DPP_ROT1
      "v_add_u32_dpp %[pB], %[B], %[B]" DP_DPP_XOR1
      "v_add_u32_dpp %[qA], %[pA], %[pA]" DP_DPP_ROT2
      "v_add_u32_dpp %[qC], %[pC], %[pC]" DP_DPP_ROT2
      "v_mov_b32_dpp %[nC], %[C]" DP_DPP_ROT1
      "v_add_u32_dpp %[qB], %[pB], %[pB]" DP_DPP_ROT2
      "v_mov_b32_dpp %[nB], %[B]" DP_DPP_ROT1
      "v_add_u32 %[qA], %[qA], %[A]\n"
      "v_add_u32 %[qC], %[qC], %[C]\n"
      "v_add_u32 %[qB], %[qB], %[B]\n"
      "v_lshl_add_u32 %[qA], %[nA], 1, %[qA]\n"
      "v_lshl_add_u32 %[qC], %[nC], 1, %[qC]\n"
      "v_lshl_add_u32 %[qB], %[nB], 1, %[qB]\n"
      "v_mov_b32_dpp %[pA], %[qA]" DP_DPP_HMIR
      "v_mov_b32_dpp %[pC], %[qC]" DP_DPP_HMIR
      "v_mov_b32_dpp %[pB], %[qB]" DP_DPP_HMIR
      "v_lshl_add_u32 %[qA], %[qA], 1, %[rA]\n"
      "v_lshl_add_u32 %[qC], %[qC], 1, %[rC]\n"
      "v_lshl_add_u32 %[qB], %[qB], 1, %[rB]\n"
      "v_add_u32_dpp %[oA], %[pA], %[qA]" DP_DPP_REV
      "v_add_u32_dpp %[oC], %[pC], %[qC]" DP_DPP_REV
      "v_add_u32_dpp %[oB], %[pB], %[qB]" DP_DPP_REV
      : [A] "=&v"(A), [B] "=&v"(B), [C] "=&v"(C), [pA] "=&v"(pA), [pB] "=&v"(pB), [pC] "=&v"(pC), [qA] "=&v"(qA), [qB] "=&v"(qB), [qC] "=&v"(qC),
        [nA] "=&v"(nA), [nB] "=&v"(nB), [nC] "=&v"(nC), [oA] "=&v"(oA), [oB] "=&v"(oB), [oC] "=&v"(oC)
      : [x0] "v"(x0), [x1] "v"(x1), [rA] "v"(rA), [rB] "v"(rB), [rC] "v"(rC));
}
// x -> limbs -> the sum over the 8 lanes of the group, limb by limb (the internal layer's row sum)
__device__ __forceinline__ void lin_sum8(u64 x, u32& oA, u32& oB, u32& oC) {
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  u32 A, B, C, pA, pB, pC, qA, qB, qC;
  asm(DP_GX_SPLIT
      "v_add_u32_dpp %[pA], %[A], %[A]" DP_DPP_XOR1
      "v_add_u32_dpp %[pC], %[C], %[C]" DP_DPP_XOR1
      "v_add_u32_dpp %[pB], %[B], %[B]" DP_DPP_XOR1
      "v_add_u32_dpp %[qA], %[pA], %[pA]" DP_DPP_ROT2
      "v_add_u32_dpp %[qC], %[pC], %[pC]" DP_DPP_ROT2
      "v_add_u32_dpp %[qB], %[pB], %[pB]" DP_DPP_ROT2
      "v_add_u32_dpp %[oA], %[qA], %[qA]" DP_DPP_HMIR
      "v_add_u32_dpp %[oC], %[qC], %[qC]" DP_DPP_HMIR
      "v_add_u32_dpp %[oB], %[qB], %[qB]" DP_DPP_HMIR
      : [A] "=&v"(A), [B] "=&v"(B), [C] "=&v"(C), [pA] "=&v"(pA), [pB] "=&v"(pB), [pC] "=&v"(pC), [qA] "=&v"(qA), [qB] "=&v"(qB), [qC] "=&v"(qC),
        [oA] "=&v"(oA), [oB] "=&v"(oB), [oC] "=&v"(oC)
      : [x0] "v"(x0), [x1] "v"(x1));
}

}  // namespace gx
}  // namespace dp
#endif
