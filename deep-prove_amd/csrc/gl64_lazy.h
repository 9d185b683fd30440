// Goldilocks / Ext2 arithmetic for the VALU-bound streaming kernels (k_sc_fused): on gfx950 v_mad_u64_u32 is full rate, so a
// 64 x 64 -> 128 product costs ~11 instructions and a 128 -> 64 reduction ~13; the canonical ex_mul of gl64.h (Karatsuba:
// 3 products, 3 reductions, 5 canonical adds of ~9 instructions, a gl_mul7) is ~140. Here an extension product is schoolbook —
// 4 wide products accumulated as exact integers in 64 + 64 + 32 bits — with TWO reductions, and its result is any u64
// representative (callers canonicalise what they store). Same field elements, bit for bit, as gl64.h.
#pragma once
#include "gl64.h"
#include "poseidon2_fast.h"  // p2f::red128, p2f::canon

namespace dp {
namespace lz {

struct A3 { u64 w0, w1; u32 w2; };  // exact integer w0 + w1 * 2^64 + w2 * 2^128 (w2 small)

DP_HD A3 a3_zero() { A3 a; a.w0 = 0; a.w1 = 0; a.w2 = 0; return a; }
DP_HD A3 a3_of(u64 x) { A3 a; a.w0 = x; a.w1 = 0; a.w2 = 0; return a; }
DP_HD void a3_add(A3& a, u64 lo, u64 hi) {
  bool c0 = __builtin_add_overflow(a.w0, lo, &a.w0);
  u64 t;
  bool c1 = __builtin_add_overflow(a.w1, hi, &t);
  bool c2 = __builtin_add_overflow(t, (u64)(c0 ? 1 : 0), &a.w1);
  a.w2 += (c1 ? 1u : 0u) + (c2 ? 1u : 0u);
}
DP_HD void a3_add_prod(A3& a, u64 x, u64 y) {
  unsigned __int128 p = (unsigned __int128)x * y;
  a3_add(a, (u64)p, (u64)(p >> 64));
}
// a += 7 * x * y  (X^2 = 7): 7 p = 8 p - p on three words
DP_HD void a3_add_prod7(A3& a, u64 x, u64 y) {
  unsigned __int128 p = (unsigned __int128)x * y;
  const u64 lo = (u64)p, hi = (u64)(p >> 64);
  u64 s0 = lo << 3, s1 = (hi << 3) | (lo >> 61); u32 s2 = (u32)(hi >> 61);   // 8 p
  u64 d0, d1;
  bool b0 = __builtin_sub_overflow(s0, lo, &d0);
  bool b1 = __builtin_sub_overflow(s1, hi, &d1);
  bool b2 = __builtin_sub_overflow(d1, (u64)(b0 ? 1 : 0), &d1);
  s2 -= (b1 ? 1u : 0u) + (b2 ? 1u : 0u);                                      // 7 p = (d0, d1, s2), s2 <= 6
  a3_add(a, d0, d1);
  a.w2 += s2;
}
// -> any u64 representative. 2^128 = (2^32 - 1)^2 = -2^32 (mod p): subtract w2 * 2^32 from the reduced low part
DP_HD u64 a3_reduce(const A3& a) {
  u64 r = p2f::red128(a.w0, a.w1);
  const u64 s = (u64)a.w2 << 32;  // < 2^40
  u64 d;
  bool br = __builtin_sub_overflow(r, s, &d);
  return d - (br ? GL_EPS : 0);   // wrapped d >= 2^64 - 2^40: no second borrow
}
// extension product, any-representative result
DP_HD Ext ex_mul(Ext a, Ext b) {
#ifdef DP_GFX950_ASM
  // c0 = a0 b0 + 7 a1 b1, c1 = a0 b1 + a1 b0 with the 12 / 13-instruction multiply and multiply-add of gl64_gfx950.h: 57 VALU instructions
  const u64 t7 = gx::mul_small(gx::mul(a.c1, b.c1), 7);
  return ex(gx::fma(a.c0, b.c0, t7), gx::fma(a.c0, b.c1, gx::mul(a.c1, b.c0)));
#endif
  A3 c0 = a3_zero(), c1 = a3_zero();
  a3_add_prod(c0, a.c0, b.c0); a3_add_prod7(c0, a.c1, b.c1);
  a3_add_prod(c1, a.c0, b.c1); a3_add_prod(c1, a.c1, b.c0);
  return ex(a3_reduce(c0), a3_reduce(c1));
}
// e + r * d  (the fold e0 + r (e1 - e0) with d = e1 - e0 computed by the caller), any-representative result
DP_HD Ext ex_fma(Ext r, Ext d, Ext e) {
#ifdef DP_GFX950_ASM
  const u64 r7 = gx::mul_small(r.c1, 7);  // (loop invariant where r is the round's challenge: the compiler hoists it)
  return ex(gx::fma(r.c0, d.c0, gx::fma(r7, d.c1, e.c0)), gx::fma(r.c0, d.c1, gx::fma(r.c1, d.c0, e.c1)));
#endif
  A3 c0 = a3_of(e.c0), c1 = a3_of(e.c1);
  a3_add_prod(c0, r.c0, d.c0); a3_add_prod7(c0, r.c1, d.c1);
  a3_add_prod(c1, r.c0, d.c1); a3_add_prod(c1, r.c1, d.c0);
  return ex(a3_reduce(c0), a3_reduce(c1));
}
// a + r * d with base-field a, d (first fold of a base table), any-representative result
DP_HD Ext ex_fma_base(Ext r, u64 d, u64 a) {
#ifdef DP_GFX950_ASM
  return ex(gx::fma(r.c0, d, a), gx::mul(r.c1, d));
#endif
  unsigned __int128 p0 = (unsigned __int128)r.c0 * d + a, p1 = (unsigned __int128)r.c1 * d;  // r.c0 d + a < 2^128: no overflow
  return ex(p2f::red128((u64)p0, (u64)(p0 >> 64)), p2f::red128((u64)p1, (u64)(p1 >> 64)));
}
DP_HD Ext ex_canon(Ext a) { return ex(p2f::canon(a.c0), p2f::canon(a.c1)); }
// running sums of any-representative extension values as exact integers (64 + 32 bits per limb), reduced once
struct ExAcc { p2f::W c0, c1; };
DP_HD ExAcc acc_zero() { ExAcc a; a.c0 = p2f::w_of(0); a.c1 = p2f::w_of(0); return a; }
DP_HD void acc_add(ExAcc& a, Ext x) { a.c0 = p2f::w_add64(a.c0, x.c0); a.c1 = p2f::w_add64(a.c1, x.c1); }
DP_HD Ext acc_value(const ExAcc& a) { return ex(p2f::canon(p2f::w_reduce(a.c0)), p2f::canon(p2f::w_reduce(a.c1))); }

}  // namespace lz
}  // namespace dp
