// Goldilocks (p = 2^64 - 2^32 + 1) and its quadratic extension F_p[X]/(X^2 - 7): one source of truth for the
// host orchestrator and the gfx950 kernels. Replaces the p3-goldilocks / BinomialExtensionField<Goldilocks,2>
// arithmetic the reference reaches through ff_ext (ff_ext/src/lib.rs:13). Values are always stored canonical (< p).
#pragma once
#include <cstdint>
#include <cstddef>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DP_HD __host__ __device__ __forceinline__
#else
#define DP_HD inline
#endif

// The hand-written instruction sequences of gl64_gfx950.h name gfx950 registers, encodings and hazards: they are compiled in the device pass FOR gfx950 only
// (the library is built with --offload-arch=gfx950); any other target, the host pass and -DDP_NO_GFX950_ASM get the portable forms of the same functions.
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__) && !defined(DP_NO_GFX950_ASM)
#define DP_GX_ON 1
#else
#define DP_GX_ON 0
#endif

namespace dp {
typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
constexpr u64 GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p = 2^32 - 1

// All primitives are branch-free selects on carry / borrow conditions: on gfx950 a data-dependent `if` around a 64-bit
// correction compiles to an EXEC-mask region (s_and_saveexec + branch + hazard nops) per operation, which more than doubles
// the instruction count of the compute-bound kernels (Poseidon2 Merkle layers, fused fold+sum); the select forms below are
// straight-line VALU.
DP_HD u64 gl_add(u64 a, u64 b) {
  // a,b < p.  s = a + b, t = s - p = s + EPS (mod 2^64).  a + b >= p  <=>  the first add carried, or s + EPS carries.
  u64 s, t;
  bool c1 = __builtin_add_overflow(a, b, &s);
  bool c2 = __builtin_add_overflow(s, GL_EPS, &t);
  return (c1 | c2) ? t : s;
}
DP_HD u64 gl_sub(u64 a, u64 b) {
  u64 d;
  bool br = __builtin_sub_overflow(a, b, &d);
  return d - (br ? GL_EPS : 0);  // + p (mod 2^64)
}
DP_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
DP_HD u64 gl_dbl(u64 a) { return gl_add(a, a); }

DP_HD void mul64wide(u64 a, u64 b, u64& lo, u64& hi) {
  unsigned __int128 x = (unsigned __int128)a * b;  // device: 4 v_mad_u64_u32
  lo = (u64)x;
  hi = (u64)(x >> 64);
}
// x = hi*2^64 + lo  ->  x mod p   (2^64 = 2^32 - 1, 2^96 = -1 mod p)
DP_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hh = hi >> 32, hl = hi & GL_EPS;
  u64 t0, r, t;
  bool br = __builtin_sub_overflow(lo, hh, &t0);
  t0 -= br ? GL_EPS : 0;            // lo - hh (mod p), any u64 representative
  u64 t1 = (hl << 32) - hl;         // hl * EPS, no overflow
  bool c = __builtin_add_overflow(t0, t1, &r);
  r += c ? GL_EPS : 0;              // cannot carry again: the wrapped sum is < t1 <= 2^64 - 2^32
  bool c2 = __builtin_add_overflow(r, GL_EPS, &t);
  return c2 ? t : r;                // r >= p  <=>  r + EPS carries
}
#if DP_GX_ON
namespace gx {  // gl64_gfx950.h: hand-written gfx950 sequences, any representative on output
__device__ __forceinline__ u64 mul(u64 a, u64 b);          // 12 VALU instructions
__device__ __forceinline__ u64 fma(u64 a, u64 b, u64 d);   // 13
__device__ __forceinline__ u64 mul_small(u64 a, u32 k);    // 7
}
#endif
DP_HD u64 gl_mul(u64 a, u64 b) {
#if DP_GX_ON
  const u64 r = gx::mul(a, b);
  u64 t;
  const bool c = __builtin_add_overflow(r, GL_EPS, &t);
  return c ? t : r;  // canonical
#else
  u64 lo, hi;
  mul64wide(a, b, lo, hi);
  return gl_reduce128(lo, hi);
#endif
}
DP_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
DP_HD u64 gl_mul7(u64 a) {  // 7a = 8a - a with the top three bits folded: (a>>61)*2^64 = (a>>61)*EPS
  u64 hi = a >> 61, lo = a << 3;
  u64 r = gl_reduce128(lo, hi);
  return gl_sub(r, a);
}
DP_HD u64 gl_pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_sqr(a);
    e >>= 1;
  }
  return r;
}
DP_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
// Fieldizer (zkml/src/quantization/mod.rs:210-220): negative -> p - |v|
DP_HD u64 gl_from_i64(int64_t v) { return v < 0 ? GL_P - (u64)(-(v + 1)) - 1 : (u64)v; }
DP_HD u64 gl_from_u64(u64 v) { return v >= GL_P ? v - GL_P : v; }

constexpr u64 GL_GENERATOR = 7;
constexpr u64 GL_G32 = 1753635133440165772ULL;  // 7^((p-1)/2^32): generator of the 2^32 subgroup

struct alignas(16) Ext {
  u64 c0, c1;
};
DP_HD Ext ex(u64 a, u64 b) { Ext r; r.c0 = a; r.c1 = b; return r; }
DP_HD Ext ex_zero() { return ex(0, 0); }
DP_HD Ext ex_one() { return ex(1, 0); }
DP_HD Ext ex_base(u64 b) { return ex(b, 0); }
DP_HD bool ex_eq(Ext a, Ext b) { return a.c0 == b.c0 && a.c1 == b.c1; }
DP_HD bool ex_is_zero(Ext a) { return (a.c0 | a.c1) == 0; }
DP_HD Ext ex_add(Ext a, Ext b) { return ex(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
DP_HD Ext ex_sub(Ext a, Ext b) { return ex(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
DP_HD Ext ex_neg(Ext a) { return ex(gl_neg(a.c0), gl_neg(a.c1)); }
DP_HD Ext ex_dbl(Ext a) { return ex_add(a, a); }
DP_HD u64 gl_canon_any(u64 r) { u64 t; const bool c = __builtin_add_overflow(r, GL_EPS, &t); return c ? t : r; }  // any representative -> canonical
DP_HD Ext ex_mul(Ext a, Ext b) {
#if DP_GX_ON
  // device: schoolbook on the 12 / 13-instruction multiply and multiply-add (4 products, the X^2 = 7 factor as a 7-instruction small multiple): ~68 VALU
  // instructions against ~95 for Karatsuba on canonical operations (its five modular additions cost as much as a product here). Same element, canonical.
  const u64 t7 = gx::mul_small(gx::mul(a.c1, b.c1), 7);
  return ex(gl_canon_any(gx::fma(a.c0, b.c0, t7)), gl_canon_any(gx::fma(a.c0, b.c1, gx::mul(a.c1, b.c0))));
#endif
  // Karatsuba: 3 base multiplications.  c0 = a0b0 + 7 a1b1 ; c1 = (a0+a1)(b0+b1) - a0b0 - a1b1
  u64 m0 = gl_mul(a.c0, b.c0), m1 = gl_mul(a.c1, b.c1);
  u64 s = gl_mul(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1));
  return ex(gl_add(m0, gl_mul7(m1)), gl_sub(gl_sub(s, m0), m1));
}
DP_HD Ext ex_mul_base(Ext a, u64 b) { return ex(gl_mul(a.c0, b), gl_mul(a.c1, b)); }
DP_HD Ext ex_sqr(Ext a) { return ex_mul(a, a); }
DP_HD Ext ex_inv(Ext a) {
  u64 n = gl_sub(gl_sqr(a.c0), gl_mul7(gl_sqr(a.c1)));
  u64 ni = gl_inv(n);
  return ex(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
DP_HD Ext ex_from_u64(u64 v) { return ex(gl_from_u64(v), 0); }
DP_HD Ext ex_from_i64(int64_t v) { return ex(gl_from_i64(v), 0); }
// a + r*(b-a) with base-field a,b (first sumcheck fold of a base table)
DP_HD Ext ex_lerp_base(u64 a, u64 b, Ext r) {
#if DP_GX_ON
  { const u64 dd = gl_sub(b, a); return ex(gl_canon_any(gx::fma(r.c0, dd, a)), gl_canon_any(gx::mul(r.c1, dd))); }
#endif
  u64 d = gl_sub(b, a);
  return ex(gl_add(gl_mul(r.c0, d), a), gl_mul(r.c1, d));
}
DP_HD Ext ex_lerp(Ext a, Ext b, Ext r) {
#if DP_GX_ON
  const Ext d = ex_sub(b, a);  // a + r d with the addends riding in the multiply-adds
  const u64 r7 = gx::mul_small(r.c1, 7);
  return ex(gl_canon_any(gx::fma(r.c0, d.c0, gx::fma(r7, d.c1, a.c0))), gl_canon_any(gx::fma(r.c0, d.c1, gx::fma(r.c1, d.c0, a.c1))));
#endif
  return ex_add(a, ex_mul(ex_sub(b, a), r));
}

DP_HD unsigned dp_ceil_log2(size_t x) {
  unsigned r = 0;
  while ((size_t(1) << r) < x) r++;
  return r;
}
DP_HD size_t dp_reverse_bits(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
}  // namespace dp
#include "gl64_gfx950.h"
