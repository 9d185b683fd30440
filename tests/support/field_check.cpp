// Test support: the product's field header (deep-prove_amd/csrc/gl64.h — the very functions the gfx950 kernels inline),
// compiled for the host and compared with plain 128-bit `%` arithmetic on edge values and a random sweep.
#include "../../deep-prove_amd/csrc/gl64.h"
#include <cstdio>
#include <vector>
using namespace dp;
typedef unsigned __int128 u128;
static u64 s_ = 0x1234567;
static u64 rnd() { s_ += 0x9E3779B97F4A7C15ULL; u64 z = s_; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
int main() {
  std::vector<u64> edge = {0, 1, 2, 6, 7, 8, GL_EPS - 1, GL_EPS, GL_EPS + 1, GL_EPS + 2, 1ULL << 32, (1ULL << 32) + 1, 1ULL << 61, 1ULL << 63, (1ULL << 63) + 1,
                           GL_P - 1, GL_P - 2, GL_P - GL_EPS, GL_P - GL_EPS - 1, GL_P - (1ULL << 32), GL_P / 2, GL_P / 2 + 1, 0xFFFFFFFE00000000ULL, 0xFFFFFFFEFFFFFFFFULL};
  size_t bad = 0, n = 0;
  auto chk2 = [&](u64 a, u64 b) {
    n++;
    if (gl_add(a, b) != (u64)(((u128)a + b) % GL_P)) bad++;
    if (gl_sub(a, b) != (u64)(((u128)a + GL_P - b) % GL_P)) bad++;
    if (gl_mul(a, b) != (u64)(((u128)a * b) % GL_P)) bad++;
    if (gl_mul7(a) != (u64)(((u128)a * 7) % GL_P)) bad++;
    if (gl_dbl(a) != (u64)(((u128)a * 2) % GL_P)) bad++;
    if (gl_neg(a) != (u64)((GL_P - a) % GL_P)) bad++;
  };
  for (u64 a : edge) for (u64 b : edge) chk2(a, b);
  for (int i = 0; i < 2000000; i++) {
    u64 a = gl_from_u64(rnd()), b = gl_from_u64(rnd());
    if (i % 7 == 0) a = edge[rnd() % edge.size()];
    if (i % 11 == 0) b = GL_P - 1 - (rnd() & 0xFFFF);
    if (i % 13 == 0) b = rnd() & GL_EPS;
    chk2(a, b);
  }
  // the 128-bit reduction on arbitrary (hi, lo), not only products of canonical values
  for (int i = 0; i < 2000000; i++) {
    u64 lo = rnd(), hi = rnd();
    if (i % 5 == 0) hi |= 0xFFFFFFFF00000000ULL;
    if (i % 3 == 0) lo = (i & 1) ? ~0ULL - (rnd() & 0xFF) : (rnd() & 0xFF);
    n++;
    if (gl_reduce128(lo, hi) != (u64)((((u128)hi << 64) | lo) % GL_P)) bad++;
  }
  for (u64 lo : edge) for (u64 hi : edge) { n++; if (gl_reduce128(lo, hi) != (u64)((((u128)hi << 64) | lo) % GL_P)) bad++; if (gl_reduce128(~lo, ~hi) != (u64)((((u128)(~hi) << 64) | (~lo)) % GL_P)) bad++; }
  // extension: (a0 + a1 X)(b0 + b1 X) mod X^2 - 7, inverse
  for (int i = 0; i < 200000; i++) {
    Ext a = ex(gl_from_u64(rnd()), gl_from_u64(rnd())), b = ex(gl_from_u64(rnd()), gl_from_u64(rnd()));
    if (i % 9 == 0) a = ex(GL_P - 1, GL_P - 1);
    if (i % 10 == 0) b = ex(GL_P - 1, GL_P - 2);
    Ext c = ex_mul(a, b);
    u64 c0 = (u64)((((u128)a.c0 * b.c0) % GL_P + 7 * (((u128)a.c1 * b.c1) % GL_P)) % GL_P);
    u64 c1 = (u64)((((u128)a.c0 * b.c1) % GL_P + ((u128)a.c1 * b.c0) % GL_P) % GL_P);
    n++;
    if (c.c0 != c0 || c.c1 != c1) bad++;
    if (i < 2000 && !ex_is_zero(a)) { Ext one = ex_mul(a, ex_inv(a)); if (!ex_eq(one, ex_one())) bad++; }
    Ext l = ex_lerp_base(a.c0, b.c0, b);  // a0 + r (b0 - a0)
    Ext d = ex_mul_base(b, gl_sub(b.c0, a.c0));
    if (!ex_eq(l, ex_add(d, ex_base(a.c0)))) bad++;
  }
  printf("field checks=%zu bad=%zu\n", n, bad);
  return bad != 0;
}
