// Host-side algebra of the zkCNN FFT-convolution protocol used by the reference's Convolution layer: the DFT the
// quantised inference runs (zkml/src/tensor.rs:220-327), the FFT / iFFT matrix reduced at a random point with its
// intermediate tables (Prover::phi_pow_init / phi_g_init, zkml/src/iop/prover.rs:214-284) and the verifier's closed
// forms (pow_two_omegas / phi_eval, zkml/src/layers/convolution.rs:1442-1482). All of it is O(n) or O(n log n) work on
// vectors of at most a few thousand elements and stays on the host, next to the transcript; the O(channels x n)
// contractions of the protocol run on the device (Dev::fix_high, Dev::sc_round).
#pragma once
#include "sumcheck.h"

namespace dp {

// get_root_of_unity (tensor.rs:220-231): the 2^32-th root squared (32 - n) times
inline u64 gl_root_of_unity(unsigned n) {
  u64 r = GL_G32;
  for (unsigned i = n; i < 32; i++) r = gl_sqr(r);
  return r;
}
// The DFT of tensor.rs:261-323 over the base field (every value the inference transforms is a base-field element):
// out[k] = sum_t v[t] w^(t k), w = root_of_unity(log n) (inverse: w^-1 and a final 1/n). Decimation in frequency
// followed by one bit-reversal pass; exact arithmetic, so the result is the reference's whatever the butterfly order.
inline void gl_fft(u64* v, size_t n, bool inverse) {
  unsigned lg = dp_ceil_log2(n);
  DP_REQUIRE((size_t(1) << lg) == n, DP_ERR_SHAPE, "fft: length must be a power of two");
  if (n == 1) return;
  u64 w = gl_root_of_unity(lg);
  if (inverse) w = gl_inv(w);
  std::vector<u64> tw(n / 2);
  tw[0] = 1;
  for (size_t i = 1; i < n / 2; i++) tw[i] = gl_mul(tw[i - 1], w);
  for (size_t half = n / 2, step = 1; half >= 1; half >>= 1, step <<= 1) {
    for (size_t base = 0; base < n; base += 2 * half)
      for (size_t j = 0; j < half; j++) {
        u64 a = v[base + j], b = v[base + j + half];
        v[base + j] = gl_add(a, b);
        v[base + j + half] = gl_mul(gl_sub(a, b), tw[j * step]);
      }
  }
  for (size_t i = 0; i < n; i++) { size_t j = dp_reverse_bits(i, lg); if (j > i) std::swap(v[i], v[j]); }
  if (inverse) { u64 ni = gl_inv(gl_from_u64(n)); for (size_t i = 0; i < n; i++) v[i] = gl_mul(v[i], ni); }
}

// powers 1, w, w^2, .. of the 2^n-th root of unity (of its inverse when `is_fft` — the reference names the flag after
// the direction of the *inverse* transform it is used for; prover.rs:214-227)
inline std::vector<u64> phi_pow_init(unsigned n, bool is_fft) {
  u64 phi = gl_root_of_unity(n);
  if (is_fft) phi = gl_inv(phi);
  std::vector<u64> pm(size_t(1) << n);
  pm[0] = 1;
  for (size_t i = 1; i < pm.size(); i++) pm[i] = gl_mul(pm[i - 1], phi);
  return pm;
}
// phi_g_init (prover.rs:231-284): phi_g[i] = F(rx, i) for the (i)FFT matrix F, and the intermediate tables the
// delegation sumchecks need (mid[i-1] has 2^i entries)
inline void phi_g_init(std::vector<Ext>& phi_g, std::vector<std::vector<Ext>>& mid, const std::vector<Ext>& rx, Ext scale, unsigned n, bool is_fft) {
  std::vector<u64> phi_mul = phi_pow_init(n, is_fft);
  auto step = [&](unsigned i) {
    unsigned m = n - i;
    for (size_t b = 0; b < (size_t(1) << (i - 1)); b++) {
      size_t l = b, r = b ^ (size_t(1) << (i - 1));
      Ext tmp1 = ex_sub(ex_one(), rx[m]), tmp2 = ex_mul_base(rx[m], phi_mul[b << m]);
      phi_g[r] = ex_mul(phi_g[l], ex_sub(tmp1, tmp2));
      phi_g[l] = ex_mul(phi_g[l], ex_add(tmp1, tmp2));
    }
  };
  if (is_fft) {
    phi_g[0] = scale; phi_g[1] = scale;
    for (unsigned i = 1; i < n + 1; i++) {
      step(i);
      if (i < n) mid[i - 1].assign(phi_g.begin(), phi_g.begin() + (size_t(1) << i));
    }
  } else {
    phi_g[0] = scale;
    for (unsigned i = 1; i < n; i++) {
      step(i);
      mid[i - 1].assign(phi_g.begin(), phi_g.begin() + (size_t(1) << i));
    }
    Ext tmp1 = ex_sub(ex_one(), rx[0]);
    for (size_t b = 0; b < (size_t(1) << (n - 1)); b++) phi_g[b] = ex_mul(phi_g[b], ex_add(tmp1, ex_mul_base(rx[0], phi_mul[b])));
  }
}
// the table `phi` of one delegation round (prover.rs:178-195): l counts down from fm-1, fm = |f_middle|
inline std::vector<Ext> delegation_phi(size_t len, size_t l, size_t fm, const std::vector<Ext>& r1, Ext r2_last, const std::vector<u64>& omegas, bool is_fft) {
  std::vector<Ext> phi(len);
  Ext r1e = r1[(fm - 1) - l];
  Ext one_m = ex_sub(ex_one(), r1e);
  if (!is_fft && l == fm - 1) {
    Ext f = ex_sub(ex_one(), r2_last);
    for (size_t i = 0; i < len; i++) phi[i] = ex_mul(f, ex_add(one_m, ex_mul_base(r1e, omegas[i << ((fm - 1) - l)])));
  } else {
    Ext f = ex_mul(ex_sub(ex_one(), ex_dbl(r2_last)), r1e);
    for (size_t i = 0; i < len; i++) phi[i] = ex_add(one_m, ex_mul_base(f, omegas[i << ((fm - 1) - l)]));
  }
  return phi;
}

// ---- verifier side closed forms
// pow_two_omegas (convolution.rs:1442-1454): w, w^2, w^4, .. (n - 1 entries) for the 2^n-th root (inverted if is_fft)
inline std::vector<u64> pow_two_omegas(unsigned n, bool is_fft) {
  std::vector<u64> pows(n - 1);
  u64 rou = gl_root_of_unity(n);
  if (is_fft) rou = gl_inv(rou);
  pows[0] = rou;
  for (unsigned i = 1; i + 1 < n; i++) pows[i] = gl_sqr(pows[i - 1]);
  return pows;
}
// phi_eval (convolution.rs:1456-1476)
inline Ext phi_eval(const std::vector<Ext>& r, Ext rand1, Ext rand2, const std::vector<u64>& exponents, bool first_iter) {
  DP_REQUIRE(r.size() <= exponents.size(), DP_ERR_VERIFY, "phi_eval: point longer than the exponent table");
  Ext eval = ex_one();
  for (size_t i = 0; i < r.size(); i++)
    eval = ex_mul(eval, ex_add(ex_sub(ex_one(), r[i]), ex_mul_base(r[i], exponents[exponents.size() - r.size() + i])));
  if (first_iter) return ex_mul(ex_sub(ex_one(), rand2), ex_add(ex_sub(ex_one(), rand1), ex_mul(rand1, eval)));
  return ex_add(ex_sub(ex_one(), rand1), ex_mul(ex_mul(ex_sub(ex_one(), ex_dbl(rand2)), rand1), eval));
}
// IntoElement::to_element (quantization/mod.rs:225-242) on a base-field value
inline int64_t gl_to_element(u64 e) { return e <= (GL_P >> 1) ? (int64_t)e : -(int64_t)(GL_P - e); }

}  // namespace dp
