// Poseidon2-w8 over Goldilocks, throughput formulation for the gfx950 Merkle kernels (same permutation as poseidon2.h /
// ff_ext/src/lib.rs:167-236, bit for bit — field arithmetic is exact, only the representatives in flight differ):
//  * state words are ANY u64 representative (not canonical) between operations; canonical only at the very end;
//  * linear layers (external circ(2 M4, M4), internal 1 1^T + diag) are accumulated as exact integers in 64 + 32 bits
//    (row sums are 21 resp. 9 words, far below 2^32 carries) and reduced ONCE per output word together with the next
//    round constant — a modular addition costs ~9 VALU instructions on gfx950, a wide add 3;
//  * the internal diagonal product keeps its 128 bits, the row sum is added before the single 128 -> 64 reduction;
//  * hl * (2^32 - 1) is formed from 32-bit halves: the compiler otherwise emits a fifth v_mad_u64_u32 (quarter rate).
#pragma once
#include "gl64.h"
#include "gl64_gfx950.h"

namespace dp {
namespace p2f {

struct W { u64 lo; u32 hi; };  // exact integer lo + hi * 2^64, hi small

DP_HD W w_of(u64 x) { W w; w.lo = x; w.hi = 0; return w; }
DP_HD W w_add(W a, W b) {
  W r; bool c = __builtin_add_overflow(a.lo, b.lo, &r.lo);
  r.hi = a.hi + b.hi + (c ? 1u : 0u);
  return r;
}
DP_HD W w_add64(W a, u64 b) {
  W r; bool c = __builtin_add_overflow(a.lo, b, &r.lo);
  r.hi = a.hi + (c ? 1u : 0u);
  return r;
}
// lo + hi * 2^64 (hi < 2^31) -> any u64 representative: hi * 2^64 = hi * (2^32 - 1) < 2^63, one carry fix-up at most
DP_HD u64 w_reduce(W a) {
  u64 t = ((u64)a.hi << 32) - a.hi;
  u64 r;
  bool c = __builtin_add_overflow(a.lo, t, &r);
  return r + (c ? GL_EPS : 0);  // wrapped r < t < 2^63: cannot carry again
}
// lo + hi * 2^64 -> any u64 representative (hi arbitrary)
DP_HD u64 red128(u64 lo, u64 hi) {
  const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
  u64 t0;
  bool br = __builtin_sub_overflow(lo, (u64)hh, &t0);
  t0 -= br ? GL_EPS : 0;                                        // wrapped t0 >= 2^64 - 2^32 + 1: no second borrow
  const u64 t1 = ((u64)(hl - (hl != 0 ? 1u : 0u)) << 32) | (u64)(u32)(0u - hl);  // hl * (2^32 - 1)
  u64 r;
  bool c = __builtin_add_overflow(t0, t1, &r);
  return r + (c ? GL_EPS : 0);                                  // wrapped r < t1 <= 2^64 - 2^33 + 1: no second carry
}
DP_HD u64 mul(u64 a, u64 b) {
#ifdef DP_GFX950_ASM
  return gx::mul(a, b);  // 12 VALU instructions instead of the compiler's 23 (gl64_gfx950.h)
#else
  unsigned __int128 x = (unsigned __int128)a * b;
  return red128((u64)x, (u64)(x >> 64));
#endif
}
// a * b + sum (sum an exact integer of 64 + 32 bits), any representative: the internal layer's d_i x_i + sum
DP_HD u64 mul_add_w(u64 a, u64 b, u64 sum_reduced, W sum) {
#ifdef DP_GFX950_ASM
  (void)sum;
  return gx::fma(a, b, sum_reduced);
#else
  (void)sum_reduced;
  unsigned __int128 x = (unsigned __int128)a * b;  // < 2^64 * p: the high word stays below p after + sum
  u64 lo, hi = (u64)(x >> 64);
  bool c = __builtin_add_overflow((u64)x, sum.lo, &lo);
  hi += (u64)sum.hi + (c ? 1u : 0u);
  return red128(lo, hi);
#endif
}
DP_HD u64 sbox(u64 x) {
  u64 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
  return mul(x3, x4);
}
DP_HD u64 canon(u64 x) { return x >= GL_P ? x - GL_P : x; }

// [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on exact integers
DP_HD void mat4(W& a, W& b, W& c, W& d) {
  W t01 = w_add(a, b), t23 = w_add(c, d);
  W t0123 = w_add(t01, t23);
  W t01123 = w_add(t0123, b), t01233 = w_add(t0123, d);
  W n3 = w_add(t01233, w_add(a, a));
  W n1 = w_add(t01123, w_add(c, c));
  W n0 = w_add(t01123, t01);
  W n2 = w_add(t01233, t23);
  a = n0; b = n1; c = n2; d = n3;
}
// external linear layer on u64 words -> exact integers (each < 21 * 2^64)
DP_HD void mds_wide(const u64* s, W* w) {
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = w_of(s[i]);
  mat4(w[0], w[1], w[2], w[3]);
  mat4(w[4], w[5], w[6], w[7]);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    W sum = w_add(w[k], w[k + 4]);
    w[k] = w_add(w[k], sum);
    w[k + 4] = w_add(w[k + 4], sum);
  }
}
// rc: the 94-word table of poseidon2.h (canonical constants)
DP_HD void permute(u64* s, const u64* rc) {
  W w[8];
  mds_wide(s, w);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = sbox(w_reduce(w_add64(w[i], rc[r * 8 + i])));
    mds_wide(s, w);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = w_reduce(w[i]);
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    s[0] = sbox(w_reduce(w_add64(w_of(s[0]), rc[32 + r])));
    W sum = w_of(s[0]);
#pragma unroll
    for (int i = 1; i < 8; i++) sum = w_add64(sum, s[i]);
#ifdef DP_GFX950_ASM
    const u64 sr = w_reduce(sum);  // one reduction of the row sum per round, then eight 13-instruction multiply-adds
#else
    const u64 sr = 0;
#endif
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = mul_add_w(s[i], rc[86 + i], sr, sum);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = w_of(s[i]);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = sbox(w_reduce(w_add64(w[i], rc[54 + r * 8 + i])));
    mds_wide(s, w);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = w_reduce(w[i]);
}
// compress(x, y) of poseidon2.h; x, y canonical, out canonical
DP_HD void compress(const u64* x, const u64* y, u64* out, const u64* rc) {
  u64 s[8] = {x[0], x[1], x[2], x[3], 0, 0, 0, 0};
  permute(s, rc);
  s[0] = y[0]; s[1] = y[1]; s[2] = y[2]; s[3] = y[3];
  permute(s, rc);
  out[0] = canon(s[3]); out[1] = canon(s[2]); out[2] = canon(s[1]); out[3] = canon(s[0]);
}

}  // namespace p2f
}  // namespace dp
