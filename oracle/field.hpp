// ORACLE (test infrastructure only — never linked into the product library).
// CPU restatement of the Goldilocks field and its degree-2 extension as used by the reference through
// Plonky3 (p3-goldilocks / BinomialExtensionField<Goldilocks,2>, pinned rev f37dc2a5, NOT on disk):
//   reference call sites: ff_ext/src/lib.rs:13 (GoldilocksExt2), :246-254 (canonical check),
//   :262-272 (bytes_to_field_elements), zkml/src/quantization/mod.rs:210-220 (Fieldizer).
// PARITY UNPINNED at this boundary: the reference holds no known-answer test for the arithmetic; constants are
// re-derived and self-checked in oracle_selftest().
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>
#include <cstring>
#include <stdexcept>
#include <string>

namespace orc {
using u64 = uint64_t;
using u128 = unsigned __int128;

constexpr u64 P = 0xFFFFFFFF00000001ULL;  // 2^64 - 2^32 + 1
constexpr u64 EPS = 0xFFFFFFFFULL;        // 2^64 mod P

static inline u64 fadd(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a || s >= P) s -= P;  // wrapping subtract handles both the carry and the >=P case
  return s;
}
static inline u64 fsub(u64 a, u64 b) { return a >= b ? a - b : a - b + P; }
static inline u64 fneg(u64 a) { return a ? P - a : 0; }
static inline u64 reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hh = hi >> 32, hl = hi & EPS;
  u64 t0 = lo - hh;
  if (lo < hh) t0 -= EPS;  // + P (wrapping)
  u64 t1 = hl * EPS;
  u64 r = t0 + t1;
  if (r < t1) r += EPS;
  if (r >= P) r -= P;
  return r;
}
static inline u64 fmul(u64 a, u64 b) { return reduce128((u128)a * b); }
static inline u64 fpow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = fmul(r, a);
    a = fmul(a, a);
    e >>= 1;
  }
  return r;
}
static inline u64 finv(u64 a) {
  if (a == 0) throw std::runtime_error("finv(0)");
  return fpow(a, P - 2);
}
// Fieldizer: negative i64 -> p - |v|   (zkml/src/quantization/mod.rs:210-220)
static inline u64 from_i64(int64_t v) { return v < 0 ? P - (u64)(-(v + 1)) - 1 : (u64)v; }
static inline u64 from_u64(u64 v) { return v >= P ? v - P : v; }

constexpr u64 GENERATOR = 7;                  // multiplicative generator
constexpr u64 G32 = 1753635133440165772ULL;   // 7^((p-1)/2^32)
static inline u64 two_adic_generator(unsigned bits) {
  assert(bits <= 32);
  u64 g = G32;
  for (unsigned i = bits; i < 32; i++) g = fmul(g, g);
  return g;
}

// Degree-2 extension: c0 + c1*X, X^2 = 7.
struct E {
  u64 c0, c1;
  bool operator==(const E& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const E& o) const { return !(*this == o); }
};
constexpr u64 W = 7;
static inline E e_zero() { return {0, 0}; }
static inline E e_one() { return {1, 0}; }
static inline E e_from(u64 b) { return {b, 0}; }
static inline E e_from_u64(u64 v) { return {from_u64(v), 0}; }
static inline E e_from_i64(int64_t v) { return {from_i64(v), 0}; }
static inline E eadd(E a, E b) { return {fadd(a.c0, b.c0), fadd(a.c1, b.c1)}; }
static inline E esub(E a, E b) { return {fsub(a.c0, b.c0), fsub(a.c1, b.c1)}; }
static inline E eneg(E a) { return {fneg(a.c0), fneg(a.c1)}; }
static inline E emul(E a, E b) {
  u64 a0b0 = fmul(a.c0, b.c0), a1b1 = fmul(a.c1, b.c1);
  u64 c0 = fadd(a0b0, fmul(W, a1b1));
  u64 c1 = fadd(fmul(a.c0, b.c1), fmul(a.c1, b.c0));
  return {c0, c1};
}
static inline E emul_base(E a, u64 b) { return {fmul(a.c0, b), fmul(a.c1, b)}; }
static inline E edbl(E a) { return eadd(a, a); }
static inline E einv(E a) {
  // (a0 - a1 X) / (a0^2 - 7 a1^2)
  u64 n = fsub(fmul(a.c0, a.c0), fmul(W, fmul(a.c1, a.c1)));
  u64 ni = finv(n);
  return {fmul(a.c0, ni), fmul(fneg(a.c1), ni)};
}
static inline E epow(E a, u64 e) {
  E r = e_one();
  while (e) {
    if (e & 1) r = emul(r, a);
    a = emul(a, a);
    e >>= 1;
  }
  return r;
}
static inline bool e_is_zero(E a) { return a.c0 == 0 && a.c1 == 0; }

// SmallField::bytes_to_field_elements (ff_ext/src/lib.rs:262-272): 8-byte LE chunks, zero padded.
static inline std::vector<u64> bytes_to_field_elements(const uint8_t* b, size_t n) {
  std::vector<u64> out;
  for (size_t i = 0; i < n; i += 8) {
    uint8_t a[8] = {0};
    size_t m = n - i < 8 ? n - i : 8;
    memcpy(a, b + i, m);
    u64 v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | a[k];
    out.push_back(from_u64(v));  // from_canonical_u64 (labels are ASCII so always < p)
  }
  return out;
}

static inline unsigned ceil_log2(size_t x) {  // sumcheck/src/util.rs:205-210
  assert(x > 0);
  unsigned r = 0;
  while ((size_t(1) << r) < x) r++;
  return r;
}
static inline unsigned log2_strict(size_t x) {
  unsigned r = ceil_log2(x);
  if ((size_t(1) << r) != x) throw std::runtime_error("log2_strict: not a power of two");
  return r;
}
static inline size_t reverse_bits(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
template <class T>
static inline void reverse_index_bits_in_place(std::vector<T>& v) {
  unsigned lg = log2_strict(v.size());
  for (size_t i = 0; i < v.size(); i++) {
    size_t j = reverse_bits(i, lg);
    if (i < j) std::swap(v[i], v[j]);
  }
}
}  // namespace orc
