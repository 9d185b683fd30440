// Poseidon2 width-8 permutation over Goldilocks as the reference wires it (ff_ext/src/lib.rs:167-236): HorizenLabs round
// constants + p3 MDSMat4 external layer + MATRIX_DIAG_8_GOLDILOCKS internal layer, x^7 S-box, R_F = 8, R_P = 22;
// the duplex sponge (DuplexChallenger<F,P,8,4>: poseidon/src/challenger.rs:14-46), `compress` / `hash_or_noop`
// (poseidon/src/poseidon_hash.rs:17-70) and BasicTranscript (transcript/src/basic.rs:8-54).
// The round-constant table below is a literal (tests re-derive it with the Poseidon Grain LFSR in oracle/ and compare).
#pragma once
#include "gl64.h"
#include <cstring>
#include <vector>
#include <string>

namespace dp {

#define DP_POSEIDON2_RC_WORDS 94
static const u64 POSEIDON2_RC_HOST[DP_POSEIDON2_RC_WORDS] = {
// [0..31] initial external (4x8), [32..53] internal, [54..85] terminal external (4x8), [86..93] internal diagonal minus one
0xdd5743e7f2a5a5d9ULL, 0xcb3a864e58ada44bULL, 0xffa2449ed32f8cdcULL, 0x42025f65d6bd13eeULL,
0x7889175e25506323ULL, 0x34b98bb03d24b737ULL, 0xbdcc535ecc4faa2aULL, 0x5b20ad869fc0d033ULL,
0xf1dda5b9259dfcb4ULL, 0x27515210be112d59ULL, 0x4227d1718c766c3fULL, 0x26d333161a5bd794ULL,
0x49b938957bf4b026ULL, 0x4a56b5938b213669ULL, 0x1120426b48c8353dULL, 0x6b323c3f10a56cadULL,
0xce57d6245ddca6b2ULL, 0xb1fc8d402bba1eb1ULL, 0xb5c5096ca959bd04ULL, 0x6db55cd306d31f7fULL,
0xc49d293a81cb9641ULL, 0x1ce55a4fe979719fULL, 0xa92e60a9d178a4d1ULL, 0x002cc64973bcfd8cULL,
0xcea721cce82fb11bULL, 0xe5b55eb8098ece81ULL, 0x4e30525c6f1ddd66ULL, 0x43c6702827070987ULL,
0xaca68430a7b5762aULL, 0x3674238634df9c93ULL, 0x88cee1c825e33433ULL, 0xde99ae8d74b57176ULL,
0x488897d85ff51f56ULL, 0x1140737ccb162218ULL, 0xa7eeb9215866ed35ULL, 0x9bd2976fee49fcc9ULL,
0xc0c8f0de580a3fccULL, 0x4fb2dae6ee8fc793ULL, 0x343a89f35f37395bULL, 0x223b525a77ca72c8ULL,
0x56ccb62574aaa918ULL, 0xc4d507d8027af9edULL, 0xa080673cf0b7e95cULL, 0xf0184884eb70dcf8ULL,
0x044f10b0cb3d5c69ULL, 0xe9e3f7993938f186ULL, 0x1b761c80e772f459ULL, 0x606cec607a1b5facULL,
0x14a0c2e1d45f03cdULL, 0x4eace8855398574fULL, 0xf905ca7103eff3e6ULL, 0xf8c8f8d20862c059ULL,
0xb524fe8bdd678e5aULL, 0xfbb7865901a1ec41ULL, 0x014ef1197d341346ULL, 0x9725e20825d07394ULL,
0xfdb25aef2c5bae3bULL, 0xbe5402dc598c971eULL, 0x93a5711f04cdca3dULL, 0xc45a9a5b2f8fb97bULL,
0xfe8946a924933545ULL, 0x2af997a27369091cULL, 0xaa62c88e0b294011ULL, 0x058eb9d810ce9f74ULL,
0xb3cb23eced349ae4ULL, 0xa3648177a77b4a84ULL, 0x43153d905992d95dULL, 0xf4e2a97cda44aa4bULL,
0x5baa2702b908682fULL, 0x082923bdf4f750d1ULL, 0x98ae09a325893803ULL, 0xf8a6475077968838ULL,
0xceb0735bf00b2c5fULL, 0x0a1a5d953888e072ULL, 0x2fcb190489f94475ULL, 0xb5be06270dec69fcULL,
0x739cb934b09acf8bULL, 0x537750b75ec7f25bULL, 0xe9dd318bae1f3961ULL, 0xf7462137299efe1aULL,
0xb1f6b8eee9adb940ULL, 0xbdebcc8a809dfe6bULL, 0x40fc1f791b178113ULL, 0x3ac1c3362d014864ULL,
0x9a016184bdb8aebaULL, 0x95f2394459fbc25eULL, 0xa98811a1fed4e3a5ULL, 0x1cc48b54f377e2a0ULL,
0xe40cd4f6c5609a26ULL, 0x11de79ebca97a4a3ULL, 0x9177c73d8b7e929cULL, 0x2a6fe8085797e791ULL,
0x3de6e93329f8d5adULL, 0x3f7af9125da962feULL, 
};

DP_HD u64 p2_sbox(u64 x) {
  u64 x2 = gl_sqr(x), x3 = gl_mul(x2, x), x4 = gl_sqr(x2);
  return gl_mul(x3, x4);
}
DP_HD void p2_mat4(u64& a, u64& b, u64& c, u64& d) {  // [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
  u64 t01 = gl_add(a, b), t23 = gl_add(c, d);
  u64 t0123 = gl_add(t01, t23);
  u64 t01123 = gl_add(t0123, b), t01233 = gl_add(t0123, d);
  u64 n3 = gl_add(t01233, gl_dbl(a));
  u64 n1 = gl_add(t01123, gl_dbl(c));
  u64 n0 = gl_add(t01123, t01);
  u64 n2 = gl_add(t01233, t23);
  a = n0; b = n1; c = n2; d = n3;
}
DP_HD void p2_mds_light(u64* s) {
  p2_mat4(s[0], s[1], s[2], s[3]);
  p2_mat4(s[4], s[5], s[6], s[7]);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    u64 sum = gl_add(s[k], s[k + 4]);
    s[k] = gl_add(s[k], sum);
    s[k + 4] = gl_add(s[k + 4], sum);
  }
}
// rc: the 94-word table above (host: POSEIDON2_RC_HOST, device: a __constant__ copy)
DP_HD void poseidon2_permute(u64* s, const u64* rc) {
  p2_mds_light(s);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = p2_sbox(gl_add(s[i], rc[r * 8 + i]));
    p2_mds_light(s);
  }
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    s[0] = p2_sbox(gl_add(s[0], rc[32 + r]));
    u64 sum = s[0];
#pragma unroll
    for (int i = 1; i < 8; i++) sum = gl_add(sum, s[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = gl_add(gl_mul(s[i], rc[86 + i]), sum);
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = p2_sbox(gl_add(s[i], rc[54 + r * 8 + i]));
    p2_mds_light(s);
  }
}
// compress(x,y) (poseidon_hash.rs:65-70): absorb x (overwrite lanes 0..3, permute), absorb y (overwrite, permute),
// squeeze pops from the back: digest = [s3,s2,s1,s0].
DP_HD void poseidon2_compress(const u64* x, const u64* y, u64* out, const u64* rc) {
  u64 s[8] = {x[0], x[1], x[2], x[3], 0, 0, 0, 0};
  poseidon2_permute(s, rc);
  s[0] = y[0]; s[1] = y[1]; s[2] = y[2]; s[3] = y[3];
  poseidon2_permute(s, rc);
  out[0] = s[3]; out[1] = s[2]; out[2] = s[1]; out[3] = s[0];
}

// ---------------------------------------------------------------- host-only sponge / transcript
struct Digest {
  u64 v[4];
  bool operator==(const Digest& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
  bool operator!=(const Digest& o) const { return !(*this == o); }
};
// Host fast path: the same permutation with lazily reduced intermediates (any u64 is accepted as a residue, results
// are canonicalised once at the end). Field arithmetic is exact, so the output equals poseidon2_permute's bit for bit
// (tests compare both against the oracle); it only removes the per-operation canonicalisation branches from the
// Fiat-Shamir critical path that sits between two device round trips.
namespace hostnc {
inline u64 add(u64 a, u64 b) {
  u64 s; bool c = __builtin_add_overflow(a, b, &s);
  u64 adj = c ? GL_EPS : 0; bool c2 = __builtin_add_overflow(s, adj, &s);
  return s + (c2 ? GL_EPS : 0);
}
inline u64 red(unsigned __int128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hh = hi >> 32, hl = hi & GL_EPS;
  u64 t0; bool b = __builtin_sub_overflow(lo, hh, &t0);
  t0 -= b ? GL_EPS : 0;
  u64 t1 = hl * GL_EPS;
  u64 r; bool c = __builtin_add_overflow(t0, t1, &r);
  return r + (c ? GL_EPS : 0);
}
inline u64 mul(u64 a, u64 b) { return red((unsigned __int128)a * b); }
inline u64 sbox(u64 x) { u64 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2); return mul(x3, x4); }
inline void mat4(u64& a, u64& b, u64& c, u64& d) {
  u64 t01 = add(a, b), t23 = add(c, d), t0123 = add(t01, t23);
  u64 t01123 = add(t0123, b), t01233 = add(t0123, d);
  u64 n3 = add(t01233, add(a, a)), n1 = add(t01123, add(c, c)), n0 = add(t01123, t01), n2 = add(t01233, t23);
  a = n0; b = n1; c = n2; d = n3;
}
inline void mds(u64* s) {
  mat4(s[0], s[1], s[2], s[3]); mat4(s[4], s[5], s[6], s[7]);
  for (int k = 0; k < 4; k++) { u64 sum = add(s[k], s[k + 4]); s[k] = add(s[k], sum); s[k + 4] = add(s[k + 4], sum); }
}
inline void permute_scalar(u64* s) {
  const u64* rc = POSEIDON2_RC_HOST;
  mds(s);
  for (int r = 0; r < 4; r++) { for (int i = 0; i < 8; i++) s[i] = sbox(add(s[i], rc[r * 8 + i])); mds(s); }
  for (int r = 0; r < 22; r++) {
    s[0] = sbox(add(s[0], rc[32 + r]));
    unsigned __int128 acc = 0;  // eight u64 terms: < 2^67
    for (int i = 0; i < 8; i++) acc += s[i];
    u64 sum = red(acc);
    for (int i = 0; i < 8; i++) s[i] = red((unsigned __int128)s[i] * rc[86 + i] + sum);  // < 2^128: no overflow
  }
  for (int r = 0; r < 4; r++) { for (int i = 0; i < 8; i++) s[i] = sbox(add(s[i], rc[54 + r * 8 + i])); mds(s); }
  for (int i = 0; i < 8; i++) s[i] = s[i] >= GL_P ? s[i] - GL_P : s[i];
}
}  // namespace hostnc
// the vectorised permutation of p2_avx512.cpp (the eight state words in the eight lanes of an AVX-512 register: 3x the scalar code),
// installed by the library when the CPU has AVX-512F/DQ (capi.cpp; DP_NO_AVX512=1 keeps the scalar code). Word-for-word equal results.
void p2_permute_avx512(u64* s);
bool p2_cpu_has_avx512();
u64 dl_copy_sum_avx512(const u64* src, u64* dst, size_t n);  // (p2_avx512.cpp) copy + sum (i + 1) * word_i: the download path's chunk check
inline u64 (*&dl_copy_sum_fast())(const u64*, u64*, size_t) { static u64 (*f)(const u64*, u64*, size_t) = nullptr; return f; }  // installed with p2_fast() when the CPU has AVX-512
void p2_compress8_avx512(const u64 (*left)[4], const u64 (*right)[4], u64 (*out)[4]);  // eight compressions side by side (lane = pair)
inline void (*&p2_fast())(u64*) { static void (*f)(u64*) = nullptr; return f; }
inline void (*&p2_fast_compress8())(const u64 (*)[4], const u64 (*)[4], u64 (*)[4]) { static void (*f)(const u64 (*)[4], const u64 (*)[4], u64 (*)[4]) = nullptr; return f; }
namespace hostnc {
inline void permute(u64* s) { if (void (*f)(u64*) = p2_fast()) f(s); else permute_scalar(s); }
}  // namespace hostnc

struct Challenger {
  u64 state[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 in_buf[4];
  int in_len = 0;
  u64 out_buf[4];
  int out_len = 0;
  void duplexing() {
    for (int i = 0; i < in_len; i++) state[i] = in_buf[i];
    in_len = 0;
    hostnc::permute(state);
    for (int i = 0; i < 4; i++) out_buf[i] = state[i];
    out_len = 4;
  }
  void observe(u64 v) {
    out_len = 0;
    in_buf[in_len++] = v;
    if (in_len == 4) duplexing();
  }
  u64 sample() {
    if (in_len != 0 || out_len == 0) duplexing();
    return out_buf[--out_len];
  }
};
inline Digest host_compress(const Digest& x, const Digest& y) {
  u64 s[8] = {x.v[0], x.v[1], x.v[2], x.v[3], 0, 0, 0, 0};
  hostnc::permute(s);
  s[0] = y.v[0]; s[1] = y.v[1]; s[2] = y.v[2]; s[3] = y.v[3];
  hostnc::permute(s);
  Digest d; d.v[0] = s[3]; d.v[1] = s[2]; d.v[2] = s[1]; d.v[3] = s[0];
  return d;
}
// Transcript<E> with the BasicTranscript behaviour; label bytes -> 8-byte LE words (ff_ext/src/lib.rs:262-272)
class Transcript {
 public:
  Transcript() {}
  explicit Transcript(const char* label) { append_message(label); }
  void append_field_element(u64 v) { ch_.observe(v); }
  void append_message(const char* s) { append_message((const uint8_t*)s, strlen(s)); }
  void append_message(const uint8_t* b, size_t n) {
    for (size_t i = 0; i < n; i += 8) {
      u64 v = 0;
      size_t m = n - i < 8 ? n - i : 8;
      for (size_t k = 0; k < m; k++) v |= (u64)b[i + k] << (8 * k);
      ch_.observe(gl_from_u64(v));
    }
  }
  void append_usize(u64 v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    append_message(b, 8);
  }
  void append_ext(Ext e) { ch_.observe(e.c0); ch_.observe(e.c1); }
  void append_exts(const std::vector<Ext>& v) { for (const Ext& e : v) append_ext(e); }
  void append_digest(const Digest& d) { for (int i = 0; i < 4; i++) ch_.observe(d.v[i]); }
  Ext read_challenge() { u64 a = ch_.sample(); u64 b = ch_.sample(); return ex(a, b); }
  Ext get_and_append_challenge(const char* label) { append_message(label); return read_challenge(); }
  Challenger& challenger() { return ch_; }  // for devices that run rounds of the transcript themselves (Dev::sc_tail)
  std::vector<Ext> read_challenges(size_t n) {
    std::vector<Ext> v;
    for (size_t i = 0; i < n; i++) v.push_back(read_challenge());
    return v;
  }
 private:
  Challenger ch_;
};
inline Transcript default_transcript() { return Transcript("m2vec"); }  // zkml/src/lib.rs:96-98

}  // namespace dp
