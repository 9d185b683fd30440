// Poseidon2-w8 over Goldilocks for the HOST transcript, AVX-512: the eight state words are the eight 64-bit lanes of one register.
// Same permutation as hostnc::permute (poseidon2.h; ff_ext/src/lib.rs:167-236 wiring, p3 constants) — word for word after the final
// canonicalisation: intermediate values are arbitrary representatives < 2^64, as in the scalar code.
//   vmulred: 64x64 -> 128 from four 32x32 products (carry-free schoolbook), then 2^64 = 2^32 - 1, 2^96 = -1 (mod p);
//   external layer: M4 inside each group of four lanes by lane rotations, then the two halves mixed — on the 32-bit halves of the words,
//     so the sums (coefficients up to 21) need no modular adds until the recombination;
//   internal layer: S-box of word 0 in scalar code, the sum of the words by horizontal adds on the halves, diag multiply on all lanes.
// Built into libdeepprove_hip.so; hostnc::permute dispatches here when the CPU has AVX-512F/DQ (dp_p2_install). The test harnesses of
// tests/ that include poseidon2.h alone keep the scalar code (tests/test_host_poseidon2.py compares the two).
#include "poseidon2.h"
#include <immintrin.h>

namespace dp {
namespace {
#define P2V __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
P2V __m512i veps() { return _mm512_set1_epi64((long long)GL_EPS); }
P2V __m512i vadd(__m512i a, __m512i b) {
  const __m512i eps = veps();
  __m512i s = _mm512_add_epi64(a, b);
  __mmask8 c = _mm512_cmplt_epu64_mask(s, a);
  __m512i s2 = _mm512_mask_add_epi64(s, c, s, eps);
  __mmask8 c2 = (__mmask8)(c & _mm512_cmplt_epu64_mask(s2, eps));
  return _mm512_mask_add_epi64(s2, c2, s2, eps);
}
P2V __m512i vmulred(__m512i a, __m512i b) {
  const __m512i eps = veps();
  __m512i ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  __m512i p00 = _mm512_mul_epu32(a, b), p01 = _mm512_mul_epu32(a, bh), p10 = _mm512_mul_epu32(ah, b), p11 = _mm512_mul_epu32(ah, bh);
  __m512i t = _mm512_add_epi64(p01, _mm512_srli_epi64(p00, 32));
  __m512i u = _mm512_add_epi64(p10, _mm512_and_si512(t, eps));
  __m512i lo = _mm512_or_si512(_mm512_slli_epi64(u, 32), _mm512_and_si512(p00, eps));
  __m512i hi = _mm512_add_epi64(_mm512_add_epi64(p11, _mm512_srli_epi64(t, 32)), _mm512_srli_epi64(u, 32));
  __m512i hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, eps);
  __mmask8 brw = _mm512_cmplt_epu64_mask(lo, hh);
  __m512i lo2 = _mm512_sub_epi64(lo, hh);
  lo2 = _mm512_mask_sub_epi64(lo2, brw, lo2, eps);
  __m512i t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);
  __m512i r = _mm512_add_epi64(lo2, t1);
  __mmask8 cr = _mm512_cmplt_epu64_mask(r, t1);
  return _mm512_mask_add_epi64(r, cr, r, eps);
}
P2V __m512i vsbox(__m512i x) { __m512i x2 = vmulred(x, x), x3 = vmulred(x2, x), x4 = vmulred(x2, x2); return vmulred(x3, x4); }
// half = the low or the high 32 bits of every word, as 64-bit lanes: M4 (rows 2 3 1 1 / 1 2 3 1 / 1 1 2 3 / 3 1 1 2) in each group of four, then out = 2 n + swap(n)
P2V __m512i mds_half(__m512i h) {
  __m512i r1 = _mm512_permutex_epi64(h, 0x39), r2 = _mm512_permutex_epi64(h, 0x4E), r3 = _mm512_permutex_epi64(h, 0x93);
  __m512i n = _mm512_add_epi64(_mm512_add_epi64(_mm512_add_epi64(h, h), _mm512_add_epi64(r1, _mm512_add_epi64(r1, r1))), _mm512_add_epi64(r2, r3));
  return _mm512_add_epi64(_mm512_add_epi64(n, n), _mm512_shuffle_i64x2(n, n, 0x4E));
}
P2V __m512i vmds(__m512i v) {
  const __m512i eps = veps();
  __m512i ol = mds_half(_mm512_and_si512(v, eps)), oh = mds_half(_mm512_srli_epi64(v, 32));  // each < 21 * 2^32
  __m512i ohh = _mm512_srli_epi64(oh, 32);                                                      // < 32: its weight 2^64 = 2^32 - 1
  __m512i x = _mm512_slli_epi64(_mm512_and_si512(oh, eps), 32);
  __m512i y = _mm512_add_epi64(ol, _mm512_sub_epi64(_mm512_slli_epi64(ohh, 32), ohh));
  return vadd(x, y);
}
}  // namespace

__attribute__((target("avx512f,avx512dq"))) void p2_permute_avx512(u64* s) {
  const u64* rc = POSEIDON2_RC_HOST;
  __m512i v = _mm512_loadu_si512((const void*)s);
  const __m512i diag = _mm512_loadu_si512((const void*)(rc + 86));
  v = vmds(v);
  for (int r = 0; r < 4; r++) v = vmds(vsbox(vadd(v, _mm512_loadu_si512((const void*)(rc + r * 8)))));
  const __m512i eps = veps();
  for (int r = 0; r < 22; r++) {
    u64 s0 = (u64)_mm_cvtsi128_si64(_mm512_castsi512_si128(v));
    s0 = hostnc::sbox(hostnc::add(s0, rc[32 + r]));
    v = _mm512_mask_set1_epi64(v, 0x01, (long long)s0);
    const u64 sl = (u64)_mm512_reduce_add_epi64(_mm512_and_si512(v, eps)), sh = (u64)_mm512_reduce_add_epi64(_mm512_srli_epi64(v, 32));
    const u64 sum = hostnc::red((unsigned __int128)sl + ((unsigned __int128)sh << 32));
    v = vadd(vmulred(v, diag), _mm512_set1_epi64((long long)sum));
  }
  for (int r = 0; r < 4; r++) v = vmds(vsbox(vadd(v, _mm512_loadu_si512((const void*)(rc + 54 + r * 8)))));
  const __m512i p = _mm512_set1_epi64((long long)GL_P);
  __mmask8 ge = _mm512_cmpge_epu64_mask(v, p);
  v = _mm512_mask_sub_epi64(v, ge, v, p);
  _mm512_storeu_si512((void*)s, v);
}
// Eight INDEPENDENT permutations at once: state word i of the eight instances in the eight lanes of v[i] (no cross-lane traffic at all) —
// for the verifier's Merkle paths, where thousands of compressions wait side by side (pcs.h merkle_jobs_ok). Throughput-bound instead
// of latency-bound: ~3x the one-state code per permutation.
namespace {
P2V void vmat4(__m512i& a, __m512i& b, __m512i& c, __m512i& d) {
  __m512i t01 = vadd(a, b), t23 = vadd(c, d), t0123 = vadd(t01, t23);
  __m512i t01123 = vadd(t0123, b), t01233 = vadd(t0123, d);
  __m512i n3 = vadd(t01233, vadd(a, a)), n1 = vadd(t01123, vadd(c, c)), n0 = vadd(t01123, t01), n2 = vadd(t01233, t23);
  a = n0; b = n1; c = n2; d = n3;
}
P2V void vmds8(__m512i* s) {
  vmat4(s[0], s[1], s[2], s[3]); vmat4(s[4], s[5], s[6], s[7]);
  for (int k = 0; k < 4; k++) { __m512i sum = vadd(s[k], s[k + 4]); s[k] = vadd(s[k], sum); s[k + 4] = vadd(s[k + 4], sum); }
}
__attribute__((target("avx512f,avx512dq"))) void permute8(__m512i* s) {
  const u64* rc = POSEIDON2_RC_HOST;
  vmds8(s);
  for (int r = 0; r < 4; r++) { for (int i = 0; i < 8; i++) s[i] = vsbox(vadd(s[i], _mm512_set1_epi64((long long)rc[r * 8 + i]))); vmds8(s); }
  for (int r = 0; r < 22; r++) {
    s[0] = vsbox(vadd(s[0], _mm512_set1_epi64((long long)rc[32 + r])));
    __m512i sum = vadd(vadd(vadd(s[0], s[1]), vadd(s[2], s[3])), vadd(vadd(s[4], s[5]), vadd(s[6], s[7])));
    for (int i = 0; i < 8; i++) s[i] = vadd(vmulred(s[i], _mm512_set1_epi64((long long)rc[86 + i])), sum);
  }
  for (int r = 0; r < 4; r++) { for (int i = 0; i < 8; i++) s[i] = vsbox(vadd(s[i], _mm512_set1_epi64((long long)rc[54 + r * 8 + i]))); vmds8(s); }
  const __m512i p = _mm512_set1_epi64((long long)GL_P);
  for (int i = 0; i < 8; i++) { __mmask8 ge = _mm512_cmpge_epu64_mask(s[i], p); s[i] = _mm512_mask_sub_epi64(s[i], ge, s[i], p); }
}
}  // namespace
// out[j] = compress(left[j], right[j]) for eight pairs of digests (host_compress of poseidon2.h: the two-permutation sponge of
// poseidon/src/digest.rs two_to_one, digest words reversed)
__attribute__((target("avx512f,avx512dq"))) void p2_compress8_avx512(const u64 (*left)[4], const u64 (*right)[4], u64 (*out)[4]) {
  __m512i s[8];
  alignas(64) u64 tmp[8];
  for (int q = 0; q < 4; q++) { for (int j = 0; j < 8; j++) tmp[j] = left[j][q]; s[q] = _mm512_load_si512((const void*)tmp); s[q + 4] = _mm512_setzero_si512(); }
  permute8(s);
  for (int q = 0; q < 4; q++) { for (int j = 0; j < 8; j++) tmp[j] = right[j][q]; s[q] = _mm512_load_si512((const void*)tmp); }
  permute8(s);
  for (int q = 0; q < 4; q++) { _mm512_store_si512((void*)tmp, s[3 - q]); for (int j = 0; j < 8; j++) out[j][q] = tmp[j]; }
}
// dst[i] = src[i] and returns sum_i (i + 1) * src[i] (mod 2^64) — the copy out of the download staging area with the chunk checksum of k_download (hip_dev.hip d2h): the scalar
// loop costs ~1 ns per word, 0.7 ms of the proving thread for the 5.8 MB query section of every Dense-4M proof (the members of a cohort take turns on that thread)
__attribute__((target("avx512f,avx512dq"))) u64 dl_copy_sum_avx512(const u64* src, u64* dst, size_t n) {
  __m512i acc = _mm512_setzero_si512();
  __m512i idx = _mm512_set_epi64(8, 7, 6, 5, 4, 3, 2, 1);
  const __m512i eight = _mm512_set1_epi64(8);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m512i v = _mm512_loadu_si512((const void*)(src + i));
    _mm512_storeu_si512((void*)(dst + i), v);
    acc = _mm512_add_epi64(acc, _mm512_mullo_epi64(v, idx));
    idx = _mm512_add_epi64(idx, eight);
  }
  u64 cs = (u64)_mm512_reduce_add_epi64(acc);
  for (; i < n; i++) { const u64 v = src[i]; dst[i] = v; cs += (u64)(i + 1) * v; }
  return cs;
}
bool p2_cpu_has_avx512() { __builtin_cpu_init(); return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

}  // namespace dp
