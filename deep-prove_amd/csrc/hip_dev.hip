// HipDev: the MI355X (gfx950 / CDNA4) implementation of dp::Dev — hand-written HIP kernels for every O(n) loop of
// the deep-prove sumcheck / logup-GKR / Basefold hot path. All arithmetic is 64-bit modular
// integer work over Goldilocks: no MFMA anywhere; the kernels are HBM-streaming (K1-K7, K9-K13) or VALU-integer bound
// (K8 Poseidon2). Wave = 64 lanes, blocks of 256 threads, grids sized to >= a few waves per SIMD on 256 CUs.
//
// Layout in HBM: a table of n field elements is a dense array in natural (little-endian index) order; base elements
// are canonical u64, extension elements are 16-byte {c0,c1} pairs so one `global_load_dwordx4` per lane fetches one
// element. A sumcheck pair (2b, 2b+1) is therefore 32 contiguous bytes per lane.
#include "dev.h"
#include "poseidon2_fast.h"
#include "gl64_lazy.h"
#include "sumcheck.h"
#include "fiber.h"
#include "logup_tail.h"
#include "classic_tail.h"
#include "dense_tail.h"
#include "eqsum_tail.h"
#include "commit_tail.h"
#include "sponge_host.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <deque>
#include <type_traits>
#include <chrono>
#include <map>
#include <vector>
#include <string>

namespace dp {

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw DpError(DP_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

__constant__ u64 c_rc[DP_POSEIDON2_RC_WORDS];
__constant__ unsigned long long c_poll_timeout_ticks = 2000000000ull;  // 20 s of the 100 MHz constant clock (DP_POLL_TIMEOUT_S)
__device__ __forceinline__ unsigned long long dp_realtime() { return __builtin_amdgcn_s_memrealtime(); }
__constant__ int c_poll_sleep = 1;
// DIAGNOSTIC BUILDS ONLY (DP_HIPCC_EXTRA=-DDP_DIAG_SKIP_HASH, then DP_DEBUG_SKIP_HASH=1 at run time): wide Merkle layers copy instead of
// hashing — a timing experiment whose proofs do not verify. The release library does not contain the switch.
#ifdef DP_DIAG_SKIP_HASH
__constant__ int c_dbg_skip_hash = 0;
#define DP_SKIP_HASH_ON() (c_dbg_skip_hash)
#else
#define DP_SKIP_HASH_ON() (false)
#endif

constexpr int TPB = 256;
constexpr int MAX_TABS = 32;
constexpr int MAX_TERMS = 48;
constexpr int MAX_PT = 32;

struct PointArg { Ext p[MAX_PT]; };

// ------------------------------------------------------------------------------------------------ launch forms
// Every kernel of this file is written once, as a device function (KBODY), and gets TWO entry points:
//   kg<Body>  one proof: the arguments arrive by value in the kernarg segment;
//   kc<Body>  a cohort of proofs in lock step: ONE launch serves every proof of the cohort, blockIdx.z selects the proof and
//             the workgroup fetches ITS arguments from a table of argument packs (one ArgPack per proof, written by the host
//             into a mapped ring; tools/argsrc.hip: uniform reads of such a table cost what kernarg reads cost).
// blockIdx.x / blockIdx.y keep their meaning inside the body in both forms. See `Cohort` below for the host side.
#define KBODY __device__ __forceinline__ void
template <class... A> struct ArgPack;
template <> struct ArgPack<> {
  template <class F, class... B> __device__ __forceinline__ void call(F f, const B&... b) const { f(b...); }
};
template <class H, class... T> struct ArgPack<H, T...> {
  H h; ArgPack<T...> t;
  ArgPack() = default;
  ArgPack(const H& h_, const T&... t_) : h(h_), t(t_...) {}
  template <class F, class... B> __device__ __forceinline__ void call(F f, const B&... b) const { t.call(f, b..., h); }
};
// parameter list of a body with references stripped: what travels (a body takes its large argument structs by const
// reference so that both entry points read them in place — kernarg segment / pack table — instead of copying them to scratch)
template <class T> struct KArgs;
template <class... A> struct KArgs<void (*)(A...)> {};
// FLAGS select how a one-workgroup (latency-critical) body shares the chip:
//   KF_CLAIM  latency mode, ONE proof on the GPU: the workgroup claims the whole register file of its CU (16 waves x 128 VGPRs;
//             with the LDS reservation of the launch no wave of another kernel can share the CU);
//   KF_PRIO   throughput mode, hundreds of proofs in flight: NO reservation at all — 256 threads, no LDS beyond what the body
//             needs — and raised wave priority instead (s_setprio 3: the SIMD's arbiter issues these waves first, the
//             Poseidon2 waves of the Merkle layers fill the remaining issue slots). A workgroup that needs an EMPTY CU stalls
//             the workgroup dispatcher until one drains, and the chip idles meanwhile: tools/hol.hip — two streams of
//             whole-CU workgroups (16 CUs, 6 % of the chip) halve the throughput of 14 streams of wide kernels, the same
//             serial work in 256-thread workgroups costs 9 %.
enum { KF_NONE = 0, KF_CLAIM = 1, KF_PRIO = 2 };
template <int FLAGS> __device__ __forceinline__ void kf_prologue() {
  if (FLAGS & KF_CLAIM) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (FLAGS & KF_PRIO) __builtin_amdgcn_s_setprio(3);
}
template <auto Body, int MAXT, int FLAGS, class... A> __global__ void __launch_bounds__(MAXT) kg(A... a) { kf_prologue<FLAGS>(); Body(a...); }
#ifdef DP_WG_TIMES
// DIAGNOSTIC BUILD ONLY (DP_HIPCC_EXTRA=-DDP_WG_TIMES, tools/wg_times.py): every workgroup of k_logup_tail records when it entered and
// left, on which CU, and which merged launch it belonged to (the address of the launch's argument packs) — how much of a merged
// launch's duration is the spread between its fastest and its slowest member?
__shared__ unsigned long long s_dbg_launch;
__device__ unsigned long long g_wgt[4 * 65536];
__device__ unsigned g_wgt_n;
__device__ __forceinline__ void dbg_wg_record(unsigned long long t_in) {
  unsigned i = atomicAdd(&g_wgt_n, 1u);
  if (i >= 65536) return;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, 4 bits
  g_wgt[4 * i] = t_in; g_wgt[4 * i + 1] = dp_realtime(); g_wgt[4 * i + 2] = s_dbg_launch;
  g_wgt[4 * i + 3] = (unsigned long long)hw | ((unsigned long long)xcc << 32) | ((unsigned long long)blockIdx.z << 40);
}
template <auto Body, int MAXT, int FLAGS, class... A> __global__ void __launch_bounds__(MAXT) kc(const ArgPack<A...>* __restrict__ packs) { kf_prologue<FLAGS>(); if (threadIdx.x == 0) s_dbg_launch = (unsigned long long)packs; packs[blockIdx.z].call(Body); }
#else
template <auto Body, int MAXT, int FLAGS, class... A> __global__ void __launch_bounds__(MAXT) kc(const ArgPack<A...>* __restrict__ packs) { kf_prologue<FLAGS>(); packs[blockIdx.z].call(Body); }
#endif

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ u64 shfl_down_u64(u64 v, int d) {
  int lo = __shfl_down((int)(u32)v, d, 64);
  int hi = __shfl_down((int)(u32)(v >> 32), d, 64);
  return ((u64)(u32)hi << 32) | (u64)(u32)lo;
}
__device__ __forceinline__ Ext wave_reduce_ext(Ext v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    Ext o = ex(shfl_down_u64(v.c0, d), shfl_down_u64(v.c1, d));
    v = ex_add(v, o);
  }
  return v;
}
// sum over the block; result valid in thread 0. `sm` must hold TPB/64 Ext values.
__device__ __forceinline__ Ext block_reduce_ext(Ext v, Ext* sm) {
  v = wave_reduce_ext(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  Ext r = ex_zero();
  if (threadIdx.x == 0) {
    r = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = ex_add(r, sm[i]);
  }
  return r;
}
__device__ __forceinline__ Ext ld_elem(const void* p, bool ext, size_t i) {
  if (ext) return ((const Ext*)p)[i];
  return ex_base(((const u64*)p)[i]);
}

// ------------------------------------------------------------------------------------------------ elementwise
KBODY k_copy_words(u64* dst, const u64* src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
KBODY k_zero_words(u64* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 0;
}
KBODY k_fieldize(const int64_t* in, u64* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = gl_from_i64(in[i]);
}
KBODY k_pow_table(u64* out, u64 base, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = gl_pow(base, i);
}
// K4: eq(x, pt) computed per index as a product over its bits (k ext multiplications per element, no log-k passes)
KBODY k_eq_table(Ext* out, const PointArg& pt, unsigned k, Ext scale, int acc, size_t n) {  // n > 2^k: the table repeats
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Ext v = scale;
    for (unsigned t = 0; t < k; t++) {
      Ext r = pt.p[t];
      v = ex_mul(v, ((i >> t) & 1) ? r : ex_sub(ex_one(), r));
    }
    out[i] = acc ? ex_add(out[i], v) : v;
  }
}
struct EvalArgs { const void* f[8]; int ext[8]; int nf; };
// out partial[block][f] = sum over the block's x of f(x) * eq(x, pt)
KBODY k_mle_eval_partial(const EvalArgs& a, const PointArg& pt, unsigned k, Ext* partial) {
  __shared__ Ext sm[TPB / 64];
  size_t n = size_t(1) << k;
  Ext acc[8];
#pragma unroll
  for (int f = 0; f < 8; f++) acc[f] = ex_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Ext e = ex_one();
    for (unsigned t = 0; t < k; t++) {
      Ext r = pt.p[t];
      e = ex_mul(e, ((i >> t) & 1) ? r : ex_sub(ex_one(), r));
    }
#pragma unroll
    for (int f = 0; f < 8; f++)
      if (f < a.nf) acc[f] = ex_add(acc[f], a.ext[f] ? ex_mul(e, ((const Ext*)a.f[f])[i]) : ex_mul_base(e, ((const u64*)a.f[f])[i]));
  }
#pragma unroll
  for (int f = 0; f < 8; f++) {
    if (f < a.nf) {
      Ext r = block_reduce_ext(acc[f], sm);
      if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 8 + f] = r;
    }
  }
}
// generic second stage: out[j] = sum_b partial[b*stride + j], one block per j
KBODY k_reduce_partials(const Ext* partial, size_t nblocks, size_t stride, Ext* out) {
  __shared__ Ext sm[TPB / 64];
  size_t j = blockIdx.x;
  Ext acc = ex_zero();
  for (size_t b = threadIdx.x; b < nblocks; b += blockDim.x) acc = ex_add(acc, partial[b * stride + j]);
  Ext r = block_reduce_ext(acc, sm);
  if (threadIdx.x == 0) out[j] = r;
}
// K2 one pass: partial[split][c] = sum over the split's rows of eq[r] * W[r*C + c]
KBODY k_fix_high_partial(const u64* W, const Ext* eq, size_t R, size_t C, size_t rows_per_split, Ext* partial) {
  size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  size_t r0 = blockIdx.y * rows_per_split, r1 = min(R, r0 + rows_per_split);
  Ext acc = ex_zero();
  for (size_t r = r0; r < r1; r++) acc = ex_add(acc, ex_mul_base(eq[r], W[r * C + c]));
  partial[(size_t)blockIdx.y * C + c] = acc;
}
// Dev::fix_low: out[r] = sum_c eq[c] * W[r][c] — one wave per row of a row-major base table, lanes stride along the row (coalesced);
// HBM-bound: the table is read once (8 R C bytes), eq (16 C bytes) stays in cache
KBODY k_fix_low(const u64* W, const Ext* eq, Ext* out, size_t R, size_t C) {
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < R; r += nwaves) {
    const u64* row = W + r * C;
    Ext acc = ex_zero();
    for (size_t c = lane; c < C; c += 64) acc = ex_add(acc, ex_mul_base(eq[c], row[c]));
    acc = wave_reduce_ext(acc);
    if (lane == 0) out[r] = acc;
  }
}
KBODY k_colsum(const Ext* partial, size_t nsplit, size_t C, Ext* out) {
  size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  Ext acc = ex_zero();
  for (size_t s = 0; s < nsplit; s++) acc = ex_add(acc, partial[s * C + c]);
  out[c] = acc;
}

// ------------------------------------------------------------------------------------------------ sumcheck (K1, K3)
struct FoldArgs { const void* in[MAX_TABS]; Ext* out[MAX_TABS]; int ext[MAX_TABS]; size_t half[MAX_TABS]; };
// K1: out[i] = in[2i] + r (in[2i+1] - in[2i]); blockIdx.y selects the table
KBODY k_fold(const FoldArgs& a, Ext r) {
  int t = blockIdx.y;
  size_t h = a.half[t];
  const void* in = a.in[t];
  Ext* out = a.out[t];
  if (a.ext[t]) {
    const Ext* p = (const Ext*)in;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) out[i] = ex_lerp(p[2 * i], p[2 * i + 1], r);
  } else {
    const u64* p = (const u64*)in;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) out[i] = ex_lerp_base(p[2 * i], p[2 * i + 1], r);
  }
}
constexpr int SC_SLOTS = SC_MAXK + 1;  // evaluations at t = 0..k of a degree-k term, k <= SC_MAXK
// The round sums of ONE product term over the pairs start, start + stride, .. < npairs: acc[t] += prod_j (lo_j + t (hi_j - lo_j)).
// `L(j, b, lo, hi)` loads pair b of the j-th factor. Degrees 1..3 (every sumcheck of the Dense / logup / Basefold path)
// keep their hand-scheduled bodies; degrees 4 and 5 (the maxpool zero-check) walk t with forward differences.
// HI = false: degrees 1..3 only (every sumcheck of the Dense / logup / Basefold path); HI = true adds degrees 4 and 5 (the
// maxpool zero-check), walking t = 0..5 with forward differences. Kernels are instantiated for both so that the register
// needs of the high-degree body never weigh on the common case.
template <bool HI, class PairLoader>
__device__ __forceinline__ void sc_accumulate(int k, PairLoader L, size_t start, size_t stride, size_t npairs, Ext (&acc)[SC_SLOTS]) {
#pragma unroll
  for (int t = 0; t < SC_SLOTS; t++) acc[t] = ex_zero();
  if (k == 1) {
    for (size_t b = start; b < npairs; b += stride) { Ext a0, b0; L(0, b, a0, b0); acc[0] = ex_add(acc[0], a0); acc[1] = ex_add(acc[1], b0); }
  } else if (k == 2) {
    for (size_t b = start; b < npairs; b += stride) {
      Ext a0, b0, a1, b1; L(0, b, a0, b0); L(1, b, a1, b1);
      Ext c0 = ex_sub(ex_dbl(b0), a0), c1 = ex_sub(ex_dbl(b1), a1);  // value at t = 2
      acc[0] = ex_add(acc[0], ex_mul(a0, a1)); acc[1] = ex_add(acc[1], ex_mul(b0, b1)); acc[2] = ex_add(acc[2], ex_mul(c0, c1));
    }
  } else if (!HI || k == 3) {
    for (size_t b = start; b < npairs; b += stride) {
      Ext a0, b0, a1, b1, a2, b2; L(0, b, a0, b0); L(1, b, a1, b1); L(2, b, a2, b2);
      Ext d0 = ex_sub(b0, a0), d1 = ex_sub(b1, a1), d2 = ex_sub(b2, a2);
      Ext c0 = ex_add(b0, d0), c1 = ex_add(b1, d1), c2 = ex_add(b2, d2);  // t = 2
      Ext f0 = ex_add(c0, d0), f1 = ex_add(c1, d1), f2 = ex_add(c2, d2);  // t = 3
      acc[0] = ex_add(acc[0], ex_mul(ex_mul(a0, a1), a2)); acc[1] = ex_add(acc[1], ex_mul(ex_mul(b0, b1), b2));
      acc[2] = ex_add(acc[2], ex_mul(ex_mul(c0, c1), c2)); acc[3] = ex_add(acc[3], ex_mul(ex_mul(f0, f1), f2));
    }
  } else if (HI) {
    for (size_t b = start; b < npairs; b += stride) {
      Ext cur[SC_MAXK], d[SC_MAXK];
#pragma unroll
      for (int j = 0; j < SC_MAXK; j++) {
        if (j < k) { Ext lo, hi; L(j, b, lo, hi); cur[j] = lo; d[j] = ex_sub(hi, lo); }
        else { cur[j] = ex_one(); d[j] = ex_zero(); }
      }
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) {
        Ext p = ex_mul(ex_mul(cur[0], cur[1]), ex_mul(cur[2], cur[3]));
        if (k == 5) p = ex_mul(p, cur[4]);
        acc[t] = ex_add(acc[t], p);
#pragma unroll
        for (int j = 0; j < SC_MAXK; j++) cur[j] = ex_add(cur[j], d[j]);
      }
    }
  }
}
// number of factor slots a kernel instantiation has to wire up
template <bool HI> struct ScW { static constexpr int K = HI ? SC_MAXK : 3; };
// pair loader over tables in global memory (natural order: pair b = elements 2b, 2b+1), base or extension per table
struct GlobalPairs {
  const void* p[SC_MAXK]; bool e[SC_MAXK];
  __device__ __forceinline__ void operator()(int j, size_t b, Ext& lo, Ext& hi) const { lo = ld_elem(p[j], e[j], 2 * b); hi = ld_elem(p[j], e[j], 2 * b + 1); }
};
// pair loader over LDS-resident tables kept in bit-reversed order: pair q = positions q, q + h
struct LdsPairs {
  const Ext* p[SC_MAXK]; size_t h;
  __device__ __forceinline__ void operator()(int j, size_t q, Ext& lo, Ext& hi) const { lo = p[j][q]; hi = p[j][q + h]; }
};
struct TermArgs { const void* tab[MAX_TABS]; int ext[MAX_TABS]; int k[MAX_TERMS]; int t[MAX_TERMS][SC_MAXK]; size_t npairs; };
// K3: partial[(term*gridDim.x + block)*SC_SLOTS + t] = sum over the block's pairs of prod_j (a_j + t d_j), t = 0..k
template <bool HI>
KBODY k_sc_terms(const TermArgs& a, Ext* partial) {
  __shared__ Ext sm[TPB / 64];
  int term = blockIdx.y;
  int k = a.k[term];
  Ext acc[SC_SLOTS];
  bool all_base = k <= 3;
  for (int j = 0; j < k; j++) all_base = all_base && !a.ext[a.t[term][j]];
  if (all_base) {
    // every factor is a base-field table (first round of a sumcheck over committed columns): stay in the base field,
    // as the reference macro does (sumcheck_macro/src/lib.rs:283-296), one 16-byte load per table and pair
    const void* p0 = a.tab[a.t[term][0]]; const void* p1 = a.tab[a.t[term][k > 1 ? 1 : 0]]; const void* p2 = a.tab[a.t[term][k > 2 ? 2 : 0]];
    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < a.npairs; b += (size_t)gridDim.x * blockDim.x) {
      ulonglong2 x0 = ((const ulonglong2*)p0)[b];
      if (k == 1) { s0 = gl_add(s0, x0.x); s1 = gl_add(s1, x0.y); }
      else if (k == 2) {
        ulonglong2 x1 = ((const ulonglong2*)p1)[b];
        u64 c0 = gl_sub(gl_dbl(x0.y), x0.x), c1 = gl_sub(gl_dbl(x1.y), x1.x);
        s0 = gl_add(s0, gl_mul(x0.x, x1.x)); s1 = gl_add(s1, gl_mul(x0.y, x1.y)); s2 = gl_add(s2, gl_mul(c0, c1));
      } else {
        ulonglong2 x1 = ((const ulonglong2*)p1)[b], x2 = ((const ulonglong2*)p2)[b];
        u64 d0 = gl_sub(x0.y, x0.x), d1 = gl_sub(x1.y, x1.x), d2 = gl_sub(x2.y, x2.x);
        u64 c0 = gl_add(x0.y, d0), c1 = gl_add(x1.y, d1), c2 = gl_add(x2.y, d2);
        u64 g0 = gl_add(c0, d0), g1 = gl_add(c1, d1), g2 = gl_add(c2, d2);
        s0 = gl_add(s0, gl_mul(gl_mul(x0.x, x1.x), x2.x)); s1 = gl_add(s1, gl_mul(gl_mul(x0.y, x1.y), x2.y));
        s2 = gl_add(s2, gl_mul(gl_mul(c0, c1), c2)); s3 = gl_add(s3, gl_mul(gl_mul(g0, g1), g2));
      }
    }
    acc[0] = ex_base(s0); acc[1] = ex_base(s1); acc[2] = ex_base(s2); acc[3] = ex_base(s3); acc[4] = ex_zero(); acc[5] = ex_zero();
  } else {
    GlobalPairs L;
#pragma unroll
    for (int j = 0; j < ScW<HI>::K; j++) { int ti = a.t[term][j < k ? j : 0]; L.p[j] = a.tab[ti]; L.e[j] = a.ext[ti]; }
    sc_accumulate<HI>(k, L, blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, a.npairs, acc);
  }
  size_t base = ((size_t)term * gridDim.x + blockIdx.x) * SC_SLOTS;
#pragma unroll
  for (int t = 0; t < SC_SLOTS; t++) {
    if (t <= k) { Ext r = block_reduce_ext(acc[t], sm); if (threadIdx.x == 0) partial[base + t] = r; }
    else if (threadIdx.x == 0) partial[base + t] = ex_zero();
  }
}
// out[term*4 + t] = sum_b partial[(term*nblocks + b)*4 + t]; one block per (term, t)
KBODY k_reduce_terms(const Ext* partial, size_t nblocks, Ext* out) {
  __shared__ Ext sm[TPB / 64];
  size_t term = blockIdx.x >> 2, t = blockIdx.x & 3;
  Ext acc = ex_zero();
  for (size_t b = threadIdx.x; b < nblocks; b += blockDim.x) acc = ex_add(acc, partial[(term * nblocks + b) * 4 + t]);
  Ext r = block_reduce_ext(acc, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}
// K3' (SURVEY.md 2.3): fused fold + round sums for ONE product of K equally typed large tables. Each lane takes 4
// consecutive elements of every table (32 contiguous bytes for base tables, 64 for extension tables), folds them with r
// into 2 values, stores those (32 contiguous bytes) and immediately accumulates the next round's sums on that pair —
// every table byte is read once and every folded byte written once per round (the unfused path reads the folded
// table a second time). partial[block*4 + t] = sum over the block's pairs of prod_j (f0_j + t (f1_j - f0_j)).
// SKIP1: the caller knows the round's claimed sum s(0) + s(1), so the t = 1 products are not computed (slot 1 stays 0
// and the host sets s(1) = claim - s(0)): 2 of the 8 extension products per pair less.
// `counter` non-null: the LAST workgroup to finish (a device-wide ticket) adds up the partials of all workgroups and publishes
// the four sums to the host itself — no k_reduce_publish launch between two rounds of a large sumcheck. The partials cross
// XCDs (per-XCD L2s are not coherent with each other): every workgroup releases at agent scope before it takes its ticket, the
// last one acquires at agent scope before it reads (MI355X_MICROARCH.md, correctness boundaries).
__device__ void sc_publish_vals_fwd(Ext* result, const Ext* src, size_t stride, int n, unsigned long long* flag, unsigned long long seq, int lane);
template <int K, bool BASE, bool SKIP1>
KBODY k_sc_fused(const void* in0, const void* in1, const void* in2, Ext* out0, Ext* out1, Ext* out2,
                                                  size_t nquads, Ext r, Ext* partial, unsigned* counter, Ext* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext sm[TPB / 64];
  const void* in[3] = {in0, in1, in2};
  Ext* out[3] = {out0, out1, out2};
  // The kernel is VALU-bound (93 % of the issue slots in its base-table round, profiles/r02_pmc_sq_sumcheck24.json), so the
  // field arithmetic is the lazy kind of gl64_lazy.h: folds as one fused multiply-add with a single reduction per limb,
  // extension products schoolbook with two reductions, running sums as exact integers reduced once per thread. What is STORED is
  // canonical (other kernels read the folded tables), what is multiplied is any representative.
  lz::ExAcc acc0 = lz::acc_zero(), acc1 = lz::acc_zero(), acc2 = lz::acc_zero(), acc3 = lz::acc_zero();
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nquads; q += (size_t)gridDim.x * blockDim.x) {
    Ext f0[K], f1[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (BASE) {
        const ulonglong2* p = (const ulonglong2*)((const u64*)in[j] + 4 * q);
        ulonglong2 a = p[0], b = p[1];
        f0[j] = lz::ex_canon(lz::ex_fma_base(r, gl_sub(a.y, a.x), a.x));
        f1[j] = lz::ex_canon(lz::ex_fma_base(r, gl_sub(b.y, b.x), b.x));
      } else {
        const Ext* p = (const Ext*)in[j] + 4 * q;
        Ext e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
        f0[j] = lz::ex_canon(lz::ex_fma(r, ex_sub(e1, e0), e0));
        f1[j] = lz::ex_canon(lz::ex_fma(r, ex_sub(e3, e2), e2));
      }
      out[j][2 * q] = f0[j];
      out[j][2 * q + 1] = f1[j];
    }
    if (K == 1) { lz::acc_add(acc0, f0[0]); if (!SKIP1) lz::acc_add(acc1, f1[0]); }
    else if (K == 2) {
      Ext c0 = ex_sub(ex_dbl(f1[0]), f0[0]), c1 = ex_sub(ex_dbl(f1[1]), f0[1]);
      lz::acc_add(acc0, lz::ex_mul(f0[0], f0[1])); if (!SKIP1) lz::acc_add(acc1, lz::ex_mul(f1[0], f1[1])); lz::acc_add(acc2, lz::ex_mul(c0, c1));
    } else {
      Ext d0 = ex_sub(f1[0], f0[0]), d1 = ex_sub(f1[1], f0[1]), d2 = ex_sub(f1[2], f0[2]);
      Ext c0 = ex_add(f1[0], d0), c1 = ex_add(f1[1], d1), c2 = ex_add(f1[2], d2);
      Ext g0 = ex_add(c0, d0), g1 = ex_add(c1, d1), g2 = ex_add(c2, d2);
      lz::acc_add(acc0, lz::ex_mul(lz::ex_mul(f0[0], f0[1]), f0[2])); if (!SKIP1) lz::acc_add(acc1, lz::ex_mul(lz::ex_mul(f1[0], f1[1]), f1[2]));
      lz::acc_add(acc2, lz::ex_mul(lz::ex_mul(c0, c1), c2)); lz::acc_add(acc3, lz::ex_mul(lz::ex_mul(g0, g1), g2));
    }
  }
  size_t base = (size_t)blockIdx.x * 4;
  Ext v;
  v = block_reduce_ext(lz::acc_value(acc0), sm); if (threadIdx.x == 0) partial[base + 0] = v;
  v = block_reduce_ext(lz::acc_value(acc1), sm); if (threadIdx.x == 0) partial[base + 1] = v;
  v = block_reduce_ext(lz::acc_value(acc2), sm); if (threadIdx.x == 0) partial[base + 2] = v;
  v = block_reduce_ext(lz::acc_value(acc3), sm); if (threadIdx.x == 0) partial[base + 3] = v;
  if (counter) {
    __shared__ int s_last;
    __shared__ Ext res[4];
    __threadfence();  // release: this workgroup's partials are visible device-wide
    if (threadIdx.x == 0) s_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {
      __threadfence();  // acquire: read what the other workgroups (other XCDs) wrote
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      if (wave < 4) {
        Ext acc = ex_zero();
        for (unsigned b = lane; b < gridDim.x; b += 64) acc = ex_add(acc, partial[(size_t)b * 4 + wave]);
        acc = wave_reduce_ext(acc);
        if (lane == 0) res[wave] = acc;
      }
      __syncthreads();
      if (wave == 0) sc_publish_vals_fwd(result, res, 1, 4, flag, seq, lane);
      if (threadIdx.x == 0) *counter = 0;  // the next launch on this stream starts from zero
    }
  }
}
// last fold of a sumcheck: every table has 2 elements; results go to one contiguous array
KBODY k_finish(const FoldArgs& a, Ext r, int ntabs, Ext* out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntabs) return;
  Ext v;
  if (a.ext[t]) { const Ext* p = (const Ext*)a.in[t]; v = ex_lerp(p[0], p[1], r); }
  else { const u64* p = (const u64*)a.in[t]; v = ex_lerp_base(p[0], p[1], r); }
  out[t] = v;
}

// ------------------------------------------------------------------------------------------------ logup (K13)
struct ColsArg { const u64* col[16]; int n; };
KBODY k_logup_den(Ext* out, const ColsArg& cols, Ext c, Ext chi, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Ext acc = c, pw = ex_one();
    for (int j = 0; j < cols.n; j++) {
      acc = ex_add(acc, ex_mul_base(pw, cols.col[j][i]));
      pw = ex_mul(pw, chi);
    }
    out[i] = acc;
  }
}
// (n1/d1 + n2/d2) pairing index i with i + half;  num_mode: 0 = all numerators are -1, 1 = base numerators, 2 = ext
KBODY k_logup_layer(const void* num, int num_mode, const Ext* den, Ext* num_out, Ext* den_out, size_t half) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Ext d1 = den[i], d2 = den[i + half];
    Ext nn;
    if (num_mode == 0) nn = ex_neg(ex_add(d1, d2));
    else if (num_mode == 1) { const u64* p = (const u64*)num; nn = ex_add(ex_mul_base(d2, p[i]), ex_mul_base(d1, p[i + half])); }
    else { const Ext* p = (const Ext*)num; nn = ex_add(ex_mul(p[i], d2), ex_mul(d1, p[i + half])); }
    num_out[i] = nn;
    den_out[i] = ex_mul(d1, d2);
  }
}

struct LogupTreeDesc { const u64* col[8]; int ncols; int num_mode; const void* num0; Ext* den_all; Ext* num_all; };
// K13 fused: denominators and every layer of the fractional-sum tree of one instance per workgroup (small tables).
// den_all: layer j at offset 2n - (2n >> j) (lengths n, n/2, .., 2); num_all: layer j >= 1 at offset n - (2n >> j).
// out[4*inst..] = [num_last[0], num_last[1], den_last[0], den_last[1]].
KBODY k_logup_tree(const LogupTreeDesc* d, size_t n, Ext c, Ext chi, Ext* out) {
  LogupTreeDesc t = d[blockIdx.x];
  int tid = threadIdx.x, nt = blockDim.x;
  Ext* den = t.den_all;
  for (size_t i = tid; i < n; i += nt) {
    Ext acc = c, pw = ex_one();
    for (int j = 0; j < t.ncols; j++) { acc = ex_add(acc, ex_mul_base(pw, t.col[j][i])); pw = ex_mul(pw, chi); }
    den[i] = acc;
  }
  __syncthreads();
  const void* num = t.num0;
  int mode = t.num_mode;
  size_t len = n, doff = 0, noff = 0;
  Ext* num_out = t.num_all;
  while (len > 2) {
    size_t half = len / 2;
    Ext* dcur = den + doff;
    Ext* dnext = den + doff + len;
    for (size_t i = tid; i < half; i += nt) {
      Ext d1 = dcur[i], d2 = dcur[i + half], nn;
      if (mode == 0) nn = ex_neg(ex_add(d1, d2));
      else if (mode == 1) { const u64* p = (const u64*)num; nn = ex_add(ex_mul_base(d2, p[i]), ex_mul_base(d1, p[i + half])); }
      else { const Ext* p = (const Ext*)num; nn = ex_add(ex_mul(p[i], d2), ex_mul(d1, p[i + half])); }
      num_out[noff + i] = nn;
      dnext[i] = ex_mul(d1, d2);
    }
    __syncthreads();
    num = (const void*)(num_out + noff); mode = 2;
    doff += len; noff += half; len = half;
  }
  if (tid < 4) {
    // len == 2 here: the last layer
    const Ext* nl = (const Ext*)num; const Ext* dl = den + doff;
    Ext v;
    if (tid < 2) { if (mode == 0) v = ex_neg(ex_one()); else if (mode == 1) v = ex_base(((const u64*)num)[tid]); else v = nl[tid]; }
    else v = dl[tid - 2];
    out[4 * blockIdx.x + tid] = v;
  }
}

// ------------------------------------------------------------------------------------------------ RS code / NTT (K5-K7)
template <bool EXT>
KBODY k_mobius_stage(void* data, size_t n, unsigned lg_half) {
  size_t half = size_t(1) << lg_half;
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < n / 2; b += (size_t)gridDim.x * blockDim.x) {
    size_t lo = ((b >> lg_half) << (lg_half + 1)) | (b & (half - 1));
    if (EXT) { Ext* p = (Ext*)data; p[lo + half] = ex_sub(p[lo + half], p[lo]); }
    else { u64* p = (u64*)data; p[lo + half] = gl_sub(p[lo + half], p[lo]); }
  }
}
// cw[2i] = cw[2i+1] = coeff[i] * shift^{bitrev_nv(i)}: bit-reversed, zero-padded, coset-scaled DIT input with the
// first (trivial, zero-tail) butterfly stage already applied (rs.rs:129-173 "r = 1")
template <bool EXT>
KBODY k_rs_prepare(const void* coeff, void* cw, const u64* pow7, unsigned nv, unsigned L) {
  size_t n = size_t(1) << nv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t j = __brevll((unsigned long long)i) >> (64 - nv);
    u64 s = pow7[j << (L - nv)];
    if (EXT) { Ext v = ex_mul_base(((const Ext*)coeff)[i], s); ((Ext*)cw)[2 * i] = v; ((Ext*)cw)[2 * i + 1] = v; }
    else { u64 v = gl_mul(((const u64*)coeff)[i], s); ((u64*)cw)[2 * i] = v; ((u64*)cw)[2 * i + 1] = v; }
  }
}
// one radix-2 DIT stage; twiddle w_{2^(lg_half+1)}^j = tw[j << (L - lg_half)], tw[i] = w_{2^(L+1)}^i
template <bool EXT>
KBODY k_ntt_stage(void* data, size_t N, unsigned lg_half, const u64* tw, unsigned L) {
  size_t half = size_t(1) << lg_half;
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N / 2; b += (size_t)gridDim.x * blockDim.x) {
    size_t j = b & (half - 1);
    size_t lo = ((b >> lg_half) << (lg_half + 1)) | j;
    u64 w = tw[j << (L - lg_half)];
    if (EXT) { Ext* p = (Ext*)data; Ext t = ex_mul_base(p[lo + half], w), u = p[lo]; p[lo] = ex_add(u, t); p[lo + half] = ex_sub(u, t); }
    else { u64* p = (u64*)data; u64 t = gl_mul(p[lo + half], w), u = p[lo]; p[lo] = gl_add(u, t); p[lo + half] = gl_sub(u, t); }
  }
}
template <bool EXT>
KBODY k_bitrev(void* dst, const void* src, unsigned lg) {
  size_t n = size_t(1) << lg;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    size_t j = lg ? (__brevll((unsigned long long)i) >> (64 - lg)) : 0;
    if (EXT) ((Ext*)dst)[j] = ((const Ext*)src)[i]; else ((u64*)dst)[j] = ((const u64*)src)[i];
  }
}

// ------------------------------------------------------------------------------------------------ Merkle (K8)
// layer 0: digest = the two leaves verbatim (hash_or_noop on <= 4 base elements)
template <bool EXT>
KBODY k_merkle_leaves(const void* leaves, u64* nodes, size_t npairs) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) {
    u64 d0, d1, d2, d3;
    if (EXT) { Ext a = ((const Ext*)leaves)[2 * i], b = ((const Ext*)leaves)[2 * i + 1]; d0 = a.c0; d1 = a.c1; d2 = b.c0; d3 = b.c1; }
    else { d0 = ((const u64*)leaves)[2 * i]; d1 = ((const u64*)leaves)[2 * i + 1]; d2 = 0; d3 = 0; }
    u64* o = nodes + 4 * i;
    o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
  }
}
// the node hash of the one-node-per-lane Merkle kernels
__device__ __forceinline__ void merkle_compress(const u64* x, const u64* y, u64* o) {
  if (DP_SKIP_HASH_ON()) { for (int k = 0; k < 4; k++) o[k] = x[k] ^ y[k]; return; }
  p2f::compress(x, y, o, c_rc);  // poseidon2_fast.h: the same permutation with wide accumulation and any-representative words (1.33x, tools/p2bench.hip)
}
// one Poseidon2 compress (2 permutations) per lane; state held in 8 VGPR pairs, round constants in constant memory
KBODY k_merkle_layer(const u64* in, u64* out, size_t cnt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2* p = (const ulonglong2*)(in + 8 * i);
    ulonglong2 x01 = p[0], x23 = p[1], y01 = p[2], y23 = p[3];
    u64 x[4] = {x01.x, x01.y, x23.x, x23.y}, y[4] = {y01.x, y01.y, y23.x, y23.y}, o[4];
    merkle_compress(x, y, o);
    ulonglong2* q = (ulonglong2*)(out + 4 * i);
    q[0] = make_ulonglong2(o[0], o[1]);
    q[1] = make_ulonglong2(o[2], o[3]);
  }
}

// The row hashes of MerkleTree::from_batch_leaves (merkle_tree.rs:261-329; util/hash.rs:32-41): several polynomials of one size share ONE
// tree whose leaf j is the row [cw_0[j], .., cw_{k-1}[j]]. Lane j writes hash_or_noop(row j) (poseidon_hash.rs:22-28: up to four base words
// are the digest themselves, zero padded; more go through the sponge: overwrite four lanes, permute, the digest is popped from the back)
// as the two extension "leaves" 2j, 2j+1 of `out`: the ordinary tree over `out` (k_merkle_leaves packs pairs, layers compress) then IS the
// batch tree from its first hashed layer up — hash_two_digests(hash(row 2i), hash(row 2i+1)) — so nothing else is specific to batches.
// Reads are coalesced per polynomial (consecutive lanes, consecutive elements). Not on the zkml path (dp_pcs_batch_commit only).
constexpr int BATCH_ROW_MAX = 32;
struct BatchRowPtrs { const u64* cw[BATCH_ROW_MAX]; };
template <bool EXT>
KBODY k_batch_row_hash(BatchRowPtrs a, int k, u64* out, size_t n) {
  constexpr int W = EXT ? 2 : 1;
  const int m = k * W;
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    u64 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c0 = 0; c0 < m; c0 += 4) {
#pragma unroll
      for (int i = 0; i < 4; i++) { const int t = c0 + i; if (t < m) s[i] = a.cw[t / W][W * j + t % W]; }
      if (m > 4) poseidon2_permute(s, c_rc);
    }
    u64* q = out + 4 * j;
    if (m > 4) { q[0] = s[3]; q[1] = s[2]; q[2] = s[1]; q[3] = s[0]; }
    else { q[0] = s[0]; q[1] = s[1]; q[2] = s[2]; q[3] = s[3]; }
  }
}

// Several consecutive Merkle layers in ONE launch (DP_MERKLE_FUSE=levels, experiment, default off): workgroup b hashes the
// 2 * blockDim.x digests [b * 2 blockDim.x, ..) of the input layer down `levels` layers — every layer it produces is consumed
// by itself only, so a block barrier between layers is all the synchronisation there is — and writes each layer to its place
// in the tree. With hundreds of proofs in flight a launch costs more than the 24 us of a lane-serial compress.
// in: the input layer (cnt digests), out: where the layer above it starts (its cnt / 2 digests; the next ones follow).
KBODY k_merkle_layers(const u64* in, u64* out, size_t cnt, int levels) {
  const int tid = threadIdx.x;
  size_t nout = blockDim.x;                              // digests this block produces in the current layer
  size_t bo = (size_t)blockIdx.x * blockDim.x;           // index of its first one
  size_t layer_cnt = cnt;
  for (int l = 0; l < levels; l++) {
    if ((size_t)tid < nout) {
      const size_t i = bo + tid;
      u64 o[4];
      poseidon2_compress(in + 8 * i, in + 8 * i + 4, o, c_rc);
      out[4 * i] = o[0]; out[4 * i + 1] = o[1]; out[4 * i + 2] = o[2]; out[4 * i + 3] = o[3];
    }
    __syncthreads();
    in = out; out += 4 * (layer_cnt / 2); layer_cnt /= 2; nout /= 2; bo /= 2;
  }
}

// ------------------------------------------------------------------------------------------------ Basefold opening (K9-K12, K14)
struct PolyDesc { const void* f; const Ext* eq; Ext* fout; Ext* eqout; size_t n; int fext; int pad; };
// fold every (f, eq) pair of length > 1 by r; blockIdx.y = polynomial
KBODY k_classic_fold(const PolyDesc* d, Ext r) {
  PolyDesc p = d[blockIdx.y];
  if (p.n <= 1) return;
  size_t h = p.n / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) {
    p.eqout[i] = ex_lerp(p.eq[2 * i], p.eq[2 * i + 1], r);
    if (p.fext) p.fout[i] = ex_lerp(((const Ext*)p.f)[2 * i], ((const Ext*)p.f)[2 * i + 1], r);
    else p.fout[i] = ex_lerp_base(((const u64*)p.f)[2 * i], ((const u64*)p.f)[2 * i + 1], r);
  }
}
// partial[(poly*gridDim.x + block)*2 + {0,1}]: c0 = sum f0*e0, c2 = sum (f1-f0)(e1-e0)   (coeff.rs:236-345)
KBODY k_classic_sums(const PolyDesc* d, Ext* partial) {
  __shared__ Ext sm[TPB / 64];
  PolyDesc p = d[blockIdx.y];
  Ext c0 = ex_zero(), c2 = ex_zero();
  if (p.n == 1) {
    if (blockIdx.x == 0 && threadIdx.x == 0) c0 = ex_mul(ld_elem(p.f, p.fext, 0), p.eq[0]);
  } else {
    size_t h = p.n / 2;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < h; j += (size_t)gridDim.x * blockDim.x) {
      Ext l0 = p.eq[2 * j], l1 = p.eq[2 * j + 1];
      if (p.fext) {
        Ext r0 = ((const Ext*)p.f)[2 * j], r1 = ((const Ext*)p.f)[2 * j + 1];
        c0 = ex_add(c0, ex_mul(l0, r0));
        c2 = ex_add(c2, ex_mul(ex_sub(l1, l0), ex_sub(r1, r0)));
      } else {
        u64 r0 = ((const u64*)p.f)[2 * j], r1 = ((const u64*)p.f)[2 * j + 1];
        c0 = ex_add(c0, ex_mul_base(l0, r0));
        c2 = ex_add(c2, ex_mul_base(ex_sub(l1, l0), gl_sub(r1, r0)));
      }
    }
  }
  size_t base = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
  Ext r;
  r = block_reduce_ext(c0, sm); if (threadIdx.x == 0) partial[base] = r;
  r = block_reduce_ext(c2, sm); if (threadIdx.x == 0) partial[base + 1] = r;
}
KBODY k_reduce_pairs(const Ext* partial, size_t nblocks, Ext* out) {
  __shared__ Ext sm[TPB / 64];
  size_t poly = blockIdx.x >> 1, t = blockIdx.x & 1;
  Ext acc = ex_zero();
  for (size_t b = threadIdx.x; b < nblocks; b += blockDim.x) acc = ex_add(acc, partial[(poly * nblocks + b) * 2 + t]);
  Ext r = block_reduce_ext(acc, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}
// K11: acc[j*rep + q] += x[j] * coeff
KBODY k_axpy_rep(Ext* acc, const void* x, int xext, Ext coeff, size_t n_acc, unsigned lg_rep) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_acc; i += (size_t)gridDim.x * blockDim.x) {
    size_t j = i >> lg_rep;
    Ext m = xext ? ex_mul(((const Ext*)x)[j], coeff) : ex_mul_base(coeff, ((const u64*)x)[j]);
    acc[i] = ex_add(acc[i], m);
  }
}
// K10 message on evaluation-form pairs: [sum a*ea, sum ((b-a)*ea + a*(eb-ea)), sum (b-a)(eb-ea)]
KBODY k_bf_msg(const Ext* f, const Ext* eq, size_t npairs, Ext* partial) {
  __shared__ Ext sm[TPB / 64];
  Ext c0 = ex_zero(), c1 = ex_zero(), c2 = ex_zero();
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < npairs; j += (size_t)gridDim.x * blockDim.x) {
    Ext a = f[2 * j], b = ex_sub(f[2 * j + 1], a), ea = eq[2 * j], eb = ex_sub(eq[2 * j + 1], ea);
    c0 = ex_add(c0, ex_mul(a, ea));
    c1 = ex_add(c1, ex_add(ex_mul(b, ea), ex_mul(a, eb)));
    c2 = ex_add(c2, ex_mul(b, eb));
  }
  size_t base = (size_t)blockIdx.x * 4;
  Ext r;
  r = block_reduce_ext(c0, sm); if (threadIdx.x == 0) partial[base] = r;
  r = block_reduce_ext(c1, sm); if (threadIdx.x == 0) partial[base + 1] = r;
  r = block_reduce_ext(c2, sm); if (threadIdx.x == 0) { partial[base + 2] = r; partial[base + 3] = ex_zero(); }
}
// K9: out[i] = y0 + (ch - x0)(y1 - y0) w,  x0 = gamma * w_{2^(level+1)}^{bitrev(i)},  w = -1/(2 x0)   (rs.rs:377-410)
KBODY k_fri_fold(const Ext* in, Ext* out, size_t nout, unsigned level, const u64* tw, unsigned L, u64 gamma, u64 neg_inv2gamma, Ext ch) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nout; i += (size_t)gridDim.x * blockDim.x) {
    size_t b = level ? (__brevll((unsigned long long)i) >> (64 - level)) : 0;
    u64 root = tw[b << (L - level)];
    u64 x0 = gl_mul(root, gamma);
    u64 rinv = b == 0 ? 1 : gl_neg(tw[((size_t(1) << level) - b) << (L - level)]);
    u64 w = gl_mul(neg_inv2gamma, rinv);
    Ext y0 = in[2 * i], y1 = in[2 * i + 1];
    Ext t = ex_mul(ex(gl_sub(ch.c0, x0), ch.c1), ex_sub(y1, y0));
    out[i] = ex_add(y0, ex_mul_base(t, w));
  }
}
struct GatherDesc { const void* leaves; const u64* nodes; size_t nleaves; size_t p0; size_t out_off; int ext; int height; };
// K14: one wave per (query, tree): leaf pair then the sibling digests bottom-up
KBODY k_query_gather(const GatherDesc* d, size_t nd, u64* out) {
  size_t q = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= nd) return;
  int lane = threadIdx.x & 63;
  GatherDesc g = d[q];
  u64* o = out + g.out_off;
  int nleafw = g.ext ? 4 : 2;
  if (lane < nleafw) o[lane] = ((const u64*)g.leaves)[g.p0 * (g.ext ? 2 : 1) + lane];
  int npath = g.height - 1;
  for (int w = lane; w < npath * 4; w += 64) {
    int l = w >> 2;
    size_t off = g.nleaves - (g.nleaves >> l);
    size_t idx = (g.p0 >> (l + 1)) ^ 1;
    o[nleafw + w] = g.nodes[4 * (off + idx) + (w & 3)];
  }
}


// ---- relaxed system-scope publication helpers (see sc_publish below)
__device__ __forceinline__ unsigned long long pub_mix(unsigned long long seq) { return seq * 0x9E3779B97F4A7C15ull + 0x7F4A7C159E3779B9ull; }
__device__ __forceinline__ void pub_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// one wave: lane l owns words l, l+64, ...; returns (on lane 0) the payload checksum
__device__ __forceinline__ unsigned long long pub_wave_sum(unsigned long long local) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local += shfl_down_u64(local, d);
  return local;
}

// ------------------------------------------------------------------------------------------------ lane-parallel Poseidon2
// One permutation spread over 8 adjacent lanes (lane i holds state[i]): used where there are too few hashes to fill
// the machine with one-hash-per-lane (the top layers of every Merkle tree), cutting the serial latency of a compress
// from 2 x ~520 dependent multiplications to 2 x ~100.
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  int lo = __shfl((int)(u32)v, src, 64);
  int hi = __shfl((int)(u32)(v >> 32), src, 64);
  return ((u64)(u32)hi << 32) | (u64)(u32)lo;
}
// DPP cross-lane moves (no LDS crossbar round trip): quad permutes and the 8-lane half-row mirror
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
  int lo = __builtin_amdgcn_update_dpp(0, (int)(u32)v, CTRL, 0xF, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), CTRL, 0xF, 0xF, false);
  return ((u64)(u32)hi << 32) | (u64)(u32)lo;
}
constexpr int DPP_QUAD_ROT1 = 0x39;     // quad_perm [1,2,3,0]: lane j reads j+1 (mod 4)
constexpr int DPP_QUAD_ROT2 = 0x4E;     // quad_perm [2,3,0,1]: lane j reads j+2 == j^2
constexpr int DPP_QUAD_ROT3 = 0x93;     // quad_perm [3,0,1,2]: lane j reads j+3
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_REV = 0x1B;      // quad_perm [3,2,1,0]
constexpr int DPP_HALF_MIRROR = 0x141;  // row_half_mirror: lane i reads 7-i within its group of 8
constexpr int DPP_QUAD_BCAST0 = 0x00;   // quad_perm [0,0,0,0]: every lane of a quad reads the quad's lane 0
// The 8-lane permutation in the formulation of poseidon2_fast.h: state words are any u64 representative, the linear layers
// are accumulated as exact integers (64 + 32 bits) THROUGH the DPP moves and reduced once together with the next round
// constant. One wave on this dependent chain runs at ~8 cycles per instruction, so instructions are what counts: a modular
// add is ~9 of them, a wide add 3. (The sponge of every fused protocol kernel and the narrow Merkle layers run on this.)
template <int CTRL> __device__ __forceinline__ p2f::W dpp_w(p2f::W a) { p2f::W r; r.lo = dpp_u64<CTRL>(a.lo); r.hi = (u32)dpp_u64<CTRL>((u64)a.hi); return r; }
// external layer circ(2 M4, M4) on the word of this lane: exact, < 21 * 2^64
__device__ __forceinline__ p2f::W p2l_mds_wide(u64 s) {
  using namespace p2f;
  W ws = w_of(s), b = w_of(dpp_u64<DPP_QUAD_ROT1>(s)), c = w_of(dpp_u64<DPP_QUAD_ROT2>(s)), d = w_of(dpp_u64<DPP_QUAD_ROT3>(s));
  W t = w_add(w_add(w_add(ws, ws), w_add(b, w_add(b, b))), w_add(c, d));  // row j of circ(2,3,1,1): 2s + 3b + c + d
  W o = dpp_w<DPP_QUAD_REV>(dpp_w<DPP_HALF_MIRROR>(t));                     // lane i reads i ^ 4
  return w_add(w_add(t, t), o);
}
__device__ __forceinline__ u64 p2l_mds_light(u64 s, int lane) { (void)lane; return p2f::canon(p2f::w_reduce(p2l_mds_wide(s))); }
__device__ __forceinline__ u64 p2l_permute(u64 s, int lane) {
  using namespace p2f;
  const int i = lane & 7;
  W w = p2l_mds_wide(s);
  for (int r = 0; r < 4; r++) { s = sbox(w_reduce(w_add64(w, c_rc[r * 8 + i]))); w = p2l_mds_wide(s); }
  s = w_reduce(w);
  const u64 diag = c_rc[86 + i];
  for (int r = 0; r < 22; r++) {
    // the words that keep their value this round, summed over the group (independent of the S-box chain below)
    W rest = i == 0 ? w_of(0) : w_of(s);
    rest = w_add(rest, dpp_w<DPP_QUAD_XOR1>(rest));
    rest = w_add(rest, dpp_w<DPP_QUAD_ROT2>(rest));
    rest = w_add(rest, dpp_w<DPP_HALF_MIRROR>(rest));  // every lane of a quad holds the quad sum: any lane of the other quad will do
    // x^7 of word 0 (every lane computes, lane 0's counts), broadcast to the group of 8
    const u64 x7 = sbox(w_reduce(w_add64(w_of(s), c_rc[32 + r])));
    const u64 q = dpp_u64<DPP_QUAD_BCAST0>(x7), m = dpp_u64<DPP_HALF_MIRROR>(q);
    const u64 x7b = i < 4 ? q : m;
    const W sum = w_add64(rest, x7b);
    // y_i = d_i x_i + sum, one 128 -> 64 reduction (the high word of the product stays below p after + sum)
    unsigned __int128 pr = (unsigned __int128)(i == 0 ? x7b : s) * diag;
    u64 lo, hi = (u64)(pr >> 64);
    bool cy = __builtin_add_overflow((u64)pr, sum.lo, &lo);
    hi += (u64)sum.hi + (cy ? 1u : 0u);
    s = red128(lo, hi);
  }
  w = w_of(s);
  for (int r = 0; r < 4; r++) { s = sbox(w_reduce(w_add64(w, c_rc[54 + r * 8 + i]))); w = p2l_mds_wide(s); }
  return canon(w_reduce(w));  // canonical: the host resumes transcripts from these words
}
// in: 8 words (two digests), out: 4 words; executed by the 8 lanes of one group together
__device__ __forceinline__ void p2l_compress(const u64* in, u64* out, int lane) {
  int i = lane & 7;
  u64 s = i < 4 ? in[i] : 0;
  s = p2l_permute(s, lane);
  if (i < 4) s = in[4 + i];
  s = p2l_permute(s, lane);
  if (i < 4) out[3 - i] = s;
}
// one Merkle layer with 8 lanes per node: for layers too narrow to hide the latency of a one-lane compress
KBODY k_merkle_layer_lp(const u64* in, u64* out, size_t cnt) {
  size_t g = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3;
  size_t stride = ((size_t)gridDim.x * blockDim.x) >> 3;
  if (DP_SKIP_HASH_ON()) { for (; g < cnt; g += stride) if ((threadIdx.x & 7) < 4) out[4 * g + (threadIdx.x & 7)] = in[8 * g + (threadIdx.x & 7)]; return; }
  for (; g < cnt; g += stride) p2l_compress(in + 8 * g, out + 4 * g, threadIdx.x & 63);
}
// verifier: one Merkle path per lane, from the leaf-pair digest up to the root (authenticate_merkle_path_root,
// mpcs/src/util/merkle_tree.rs:331-420). meta[3j..] = {index of the leaf pair, first digest of the path in `pool`, depth};
// bad[0] counts the paths that do not authenticate, bad[1] = the smallest index of one
KBODY k_merkle_paths(const u64* leaf, const u64* root, const u64* meta, const u64* pool, size_t n, unsigned long long* bad) {
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    u64 h[4] = {leaf[4 * j], leaf[4 * j + 1], leaf[4 * j + 2], leaf[4 * j + 3]};
    u64 x = meta[3 * j];
    const u64* p = pool + 4 * meta[3 * j + 1];
    const unsigned depth = (unsigned)meta[3 * j + 2];
    for (unsigned l = 0; l < depth; l++) {
      u64 sib[4] = {p[4 * l], p[4 * l + 1], p[4 * l + 2], p[4 * l + 3]}, o[4];
      if (x & 1) p2f::compress(sib, h, o, c_rc); else p2f::compress(h, sib, o, c_rc);
      h[0] = o[0]; h[1] = o[1]; h[2] = o[2]; h[3] = o[3];
      x >>= 1;
    }
    if (h[0] != root[4 * j] || h[1] != root[4 * j + 1] || h[2] != root[4 * j + 2] || h[3] != root[4 * j + 3]) { atomicAdd(&bad[0], 1ull); atomicMin(&bad[1], (unsigned long long)j); }
  }
}
struct TailDesc { u64* nodes; size_t off; size_t cnt; };
// All Merkle layers above an already computed layer of `cnt` (<= 2048) digests, one workgroup per tree, no relaunch
// between layers. Wide layers hash one node per lane, narrow ones use the 8-lane permutation. roots[4*tree..] = root.
KBODY k_merkle_tail(const TailDesc* d, u64* roots, u64* host_result, unsigned long long* flag, unsigned long long seq) {
  TailDesc t = d[blockIdx.x];
  u64* nd = t.nodes;
  size_t off = t.off, cnt = t.cnt;
  int tid = threadIdx.x;
  while (cnt > 1) {
    size_t next = cnt / 2;
    const u64* in = nd + 4 * off;
    u64* out = nd + 4 * (off + cnt);
    if (next > blockDim.x) {  // very wide layer: one node per lane; otherwise 8 lanes per node (the DPP permutation is ~8x lower latency)
      for (size_t i = tid; i < next; i += blockDim.x) {
        u64 o[4];
        poseidon2_compress(in + 8 * i, in + 8 * i + 4, o, c_rc);
        out[4 * i] = o[0]; out[4 * i + 1] = o[1]; out[4 * i + 2] = o[2]; out[4 * i + 3] = o[3];
      }
    } else {
      for (size_t g = tid >> 3; g < next; g += (blockDim.x >> 3)) p2l_compress(in + 8 * g, out + 4 * g, tid & 63);
    }
    __syncthreads();
    off += cnt; cnt = next;
  }
  if (tid < 4) roots[4 * blockIdx.x + tid] = nd[4 * off + tid];
  if (host_result && tid < 64) {  // single tree: publish the root (4 words) directly
    unsigned long long cs = 0;
    if (tid < 4) { u64 v = nd[4 * off + tid]; pub_store(host_result + tid, v); cs = (unsigned long long)(tid + 1) * v; }
    cs = pub_wave_sum(cs);
    if (tid == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
  }
}
struct SmallCommitDesc { const void* evals; void* cw; void* bh; u64* nodes; void* tmp; };
// layer 0 of many equally sized trees: blockIdx.y = tree
template <bool EXT>
KBODY k_merkle_leaves_many(const SmallCommitDesc* d, size_t npairs) {
  const void* leaves = d[blockIdx.y].cw;
  u64* nodes = d[blockIdx.y].nodes;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npairs; i += (size_t)gridDim.x * blockDim.x) {
    u64 d0, d1, d2, d3;
    if (EXT) { Ext a = ((const Ext*)leaves)[2 * i], b = ((const Ext*)leaves)[2 * i + 1]; d0 = a.c0; d1 = a.c1; d2 = b.c0; d3 = b.c1; }
    else { d0 = ((const u64*)leaves)[2 * i]; d1 = ((const u64*)leaves)[2 * i + 1]; d2 = 0; d3 = 0; }
    u64* o = nodes + 4 * i;
    o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
  }
}
template <bool EXT> struct ElemOps;
template <> struct ElemOps<false> {
  typedef u64 T;
  static __device__ __forceinline__ T sub(T a, T b) { return gl_sub(a, b); }
  static __device__ __forceinline__ T add(T a, T b) { return gl_add(a, b); }
  static __device__ __forceinline__ T mulb(T a, u64 b) { return gl_mul(a, b); }
};
template <> struct ElemOps<true> {
  typedef Ext T;
  static __device__ __forceinline__ T sub(T a, T b) { return ex_sub(a, b); }
  static __device__ __forceinline__ T add(T a, T b) { return ex_add(a, b); }
  static __device__ __forceinline__ T mulb(T a, u64 b) { return ex_mul_base(a, b); }
};
// Stages [s_lo, s_hi) of the Moebius transform (NTT = false: p[lo + half] -= p[lo]) or of the radix-2 DIT NTT (NTT = true:
// butterfly with w = tw[j << (L - s)], j = the index bits below s) in ONE launch, LDS-tiled: a tile is the 2^(s_hi - s_lo + lgc)
// elements whose index is (hi << s_hi) | (m << s_lo) | (lowblk << lgc) | c for all m < 2^(s_hi - s_lo), c < 2^lgc — the bits
// the stages act on (m) plus 2^lgc consecutive elements, so that global accesses stay 128-byte segments when the stride
// 2^s_lo is large; lgc <= s_lo, and with lgc == s_lo the tile is one contiguous block. Tiles partition the array, every stage of
// the range only pairs elements of one tile: a 2^20-coefficient polynomial takes 2 Moebius + 2 NTT passes instead of 20 + 20
// per-stage launches over HBM (K5 / K7). In LDS the element sits at (m << lgc) | c.
template <bool EXT, bool NTT>
KBODY k_butterfly_pass(void* data, unsigned s_lo, unsigned s_hi, unsigned lgc, const u64* tw, unsigned L) {
  extern __shared__ __align__(16) unsigned char lds_bp[];
  typedef typename ElemOps<EXT>::T T;
  T* A = (T*)lds_bp;
  T* p = (T*)data;
  const unsigned span = s_hi - s_lo, lgt = span + lgc;
  const size_t tile = blockIdx.x;
  const size_t nlowblk = size_t(1) << (s_lo - lgc);           // tiles per value of the high bits
  const size_t hi = tile >> (s_lo - lgc), lowblk = tile & (nlowblk - 1);
  const size_t base = (hi << s_hi) | (lowblk << lgc);
  const int tid = threadIdx.x, nt = blockDim.x;
  const size_t telems = size_t(1) << lgt;
  for (size_t q = tid; q < telems; q += nt) { size_t m = q >> lgc, c = q & ((size_t(1) << lgc) - 1); A[q] = p[base | (m << s_lo) | c]; }
  __syncthreads();
  for (unsigned s = s_lo; s < s_hi; s++) {
    const unsigned b = s - s_lo;                               // the bit of m this stage pairs
    for (size_t q = tid; q < telems / 2; q += nt) {
      // q enumerates the pairs: insert a 0 at bit (b + lgc) of the LDS position
      const size_t lowmask = (size_t(1) << (b + lgc)) - 1;
      const size_t lo = ((q & ~lowmask) << 1) | (q & lowmask), hi2 = lo | (size_t(1) << (b + lgc));
      if (NTT) {
        const size_t m = lo >> lgc, c = lo & ((size_t(1) << lgc) - 1);
        const size_t j = ((m & ((size_t(1) << b) - 1)) << s_lo) | (lowblk << lgc) | c;   // index bits below s
        const u64 w = tw[j << (L - s)];
        T t = ElemOps<EXT>::mulb(A[hi2], w), u = A[lo];
        A[lo] = ElemOps<EXT>::add(u, t); A[hi2] = ElemOps<EXT>::sub(u, t);
      } else A[hi2] = ElemOps<EXT>::sub(A[hi2], A[lo]);
    }
    __syncthreads();
  }
  for (size_t q = tid; q < telems; q += nt) { size_t m = q >> lgc, c = q & ((size_t(1) << lgc) - 1); p[base | (m << s_lo) | c] = A[q]; }
}
// K5+K6+K7 for a small polynomial entirely in LDS: evaluations -> coefficients (Moebius), coset scale, zero-pad,
// radix-2 DIT NTT on 2n points, bit-reversed store; also the bit-reversed copy of the evaluations. One workgroup per
// polynomial (blockIdx.x), dynamic LDS = 3n elements.
template <bool EXT>
KBODY k_commit_small(const SmallCommitDesc* d, unsigned nv, unsigned L, const u64* tw, const u64* pow7) {
  typedef typename ElemOps<EXT>::T T;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  T* A = (T*)lds_raw;  // n coefficients
  size_t n = size_t(1) << nv, N = 2 * n;
  T* Bf = A + n;       // 2n NTT buffer
  SmallCommitDesc pd = d[blockIdx.x];
  const T* ev = (const T*)pd.evals;
  T* bh = (T*)pd.bh;
  T* cw = (T*)pd.cw;
  int tid = threadIdx.x, nt = blockDim.x;
  for (size_t i = tid; i < n; i += nt) {
    T v = ev[i];
    A[i] = v;
    bh[__brev((unsigned)i) >> (32 - nv)] = v;
  }
  __syncthreads();
  for (unsigned s = 0; s < nv; s++) {
    size_t half = size_t(1) << s;
    for (size_t b = tid; b < n / 2; b += nt) {
      size_t lo = ((b >> s) << (s + 1)) | (b & (half - 1));
      A[lo + half] = ElemOps<EXT>::sub(A[lo + half], A[lo]);
    }
    __syncthreads();
  }
  for (size_t i = tid; i < n; i += nt) {
    size_t j = __brev((unsigned)i) >> (32 - nv);
    T v = ElemOps<EXT>::mulb(A[i], pow7[j << (L - nv)]);
    Bf[2 * i] = v; Bf[2 * i + 1] = v;
  }
  __syncthreads();
  for (unsigned s = 1; s <= nv; s++) {
    size_t half = size_t(1) << s;
    for (size_t b = tid; b < n; b += nt) {
      size_t j = b & (half - 1);
      size_t lo = ((b >> s) << (s + 1)) | j;
      T t = ElemOps<EXT>::mulb(Bf[lo + half], tw[j << (L - s)]), u = Bf[lo];
      Bf[lo] = ElemOps<EXT>::add(u, t);
      Bf[lo + half] = ElemOps<EXT>::sub(u, t);
    }
    __syncthreads();
  }
  for (size_t o = tid; o < N; o += nt) cw[o] = Bf[__brev((unsigned)o) >> (32 - (nv + 1))];
}

// ---- medium polynomials (2^12..2^14 base elements: the witness columns of a convolution layer), many at once: blockIdx.y
// (or .x) = polynomial. Four launches replace the 2 nv + 4 per-stage launches of the generic path, for the whole group.
constexpr unsigned MED_NTT_LG = 13;  // an NTT block of 2^13 base elements (64 KB) lives in LDS
// K5 + K6 + coset scaling: evaluations -> LDS, all Moebius stages there, then tmp[2i] = tmp[2i+1] = coeff[i] * shift^bitrev(i)
// (the zero-padded, bit-reversed DIT input after its trivial first stage) and bh[bitrev(i)] = evals[i]
KBODY k_med_prepare(const SmallCommitDesc* d, unsigned nv, unsigned L, const u64* pow7) {
  extern __shared__ __align__(16) unsigned char lds_med[];
  u64* A = (u64*)lds_med;
  SmallCommitDesc pd = d[blockIdx.x];
  const u64* ev = (const u64*)pd.evals; u64* bh = (u64*)pd.bh; u64* tmp = (u64*)pd.tmp;
  size_t n = size_t(1) << nv;
  int tid = threadIdx.x, nt = blockDim.x;
  for (size_t i = tid; i < n; i += nt) { u64 v = ev[i]; A[i] = v; bh[__brev((unsigned)i) >> (32 - nv)] = v; }
  __syncthreads();
  for (unsigned s = 0; s < nv; s++) {
    size_t half = size_t(1) << s;
    for (size_t b = tid; b < n / 2; b += nt) { size_t lo = ((b >> s) << (s + 1)) | (b & (half - 1)); A[lo + half] = gl_sub(A[lo + half], A[lo]); }
    __syncthreads();
  }
  for (size_t i = tid; i < n; i += nt) {
    size_t j = __brev((unsigned)i) >> (32 - nv);
    u64 v = gl_mul(A[i], pow7[j << (L - nv)]);
    ((ulonglong2*)tmp)[i] = make_ulonglong2(v, v);
  }
}
// DIT stages 1..smax (butterfly span <= 2^MED_NTT_LG) of the 2n-point NTT, one LDS-resident block per workgroup
KBODY k_med_ntt_local(const SmallCommitDesc* d, unsigned lgblk, unsigned smax, const u64* tw, unsigned L) {
  extern __shared__ __align__(16) unsigned char lds_med[];
  u64* B = (u64*)lds_med;
  size_t blk = size_t(1) << lgblk;
  u64* p = (u64*)d[blockIdx.y].tmp + blockIdx.x * blk;
  int tid = threadIdx.x, nt = blockDim.x;
  for (size_t i = tid; i < blk; i += nt) B[i] = p[i];
  __syncthreads();
  for (unsigned s = 1; s <= smax; s++) {
    size_t half = size_t(1) << s;
    for (size_t b = tid; b < blk / 2; b += nt) {
      size_t j = b & (half - 1), lo = ((b >> s) << (s + 1)) | j;
      u64 t = gl_mul(B[lo + half], tw[j << (L - s)]), u = B[lo];
      B[lo] = gl_add(u, t); B[lo + half] = gl_sub(u, t);
    }
    __syncthreads();
  }
  for (size_t i = tid; i < blk; i += nt) p[i] = B[i];
}
KBODY k_ntt_stage_many(const SmallCommitDesc* d, size_t N, unsigned lg_half, const u64* tw, unsigned L) {
  u64* p = (u64*)d[blockIdx.y].tmp;
  size_t half = size_t(1) << lg_half;
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < N / 2; b += (size_t)gridDim.x * blockDim.x) {
    size_t j = b & (half - 1), lo = ((b >> lg_half) << (lg_half + 1)) | j;
    u64 t = gl_mul(p[lo + half], tw[j << (L - lg_half)]), u = p[lo];
    p[lo] = gl_add(u, t); p[lo + half] = gl_sub(u, t);
  }
}
KBODY k_bitrev_many(const SmallCommitDesc* d, unsigned lg) {
  const u64* src = (const u64*)d[blockIdx.y].tmp; u64* dst = (u64*)d[blockIdx.y].cw;
  size_t n = size_t(1) << lg;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[__brevll((unsigned long long)i) >> (64 - lg)] = src[i];
}
// one Merkle layer of many equally shaped trees (one Poseidon2 compress per lane): blockIdx.y = tree
KBODY k_merkle_layer_many(const TailDesc* td, size_t off, size_t cnt) {
  u64* nd = td[blockIdx.y].nodes;
  const u64* in = nd + 4 * off; u64* out = nd + 4 * (off + cnt);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cnt / 2; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2* p = (const ulonglong2*)(in + 8 * i);
    ulonglong2 x01 = p[0], x23 = p[1], y01 = p[2], y23 = p[3];
    u64 x[4] = {x01.x, x01.y, x23.x, x23.y}, y[4] = {y01.x, y01.y, y23.x, y23.y}, o[4];
    merkle_compress(x, y, o);
    ulonglong2* q = (ulonglong2*)(out + 4 * i);
    q[0] = make_ulonglong2(o[0], o[1]); q[1] = make_ulonglong2(o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------ single-launch sumcheck round
__device__ void sc_publish_fwd(Ext* result, const Ext* part, const int* tk, const int* toff, int nterms, int wpt, unsigned long long* flag, unsigned long long seq, int lane);
__device__ void sc_wait_challenge_fwd(const unsigned long long* mailbox, unsigned long long seq, unsigned long long* chal);
struct ScSmallArgs {
  const void* in[MAX_TABS]; Ext* out[MAX_TABS]; int in_ext[MAX_TABS];
  int k[MAX_TERMS]; int t[MAX_TERMS][SC_MAXK]; int off[MAX_TERMS];  // off[i] = sum_{j<i} (k_j + 1): slot of term i in the published message
  int ntabs, nterms, has_r; size_t n_after; Ext r;
};
__device__ __forceinline__ Ext block_reduce_ext_n(Ext v, Ext* sm) {
  v = wave_reduce_ext(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  Ext r = ex_zero();
  if (threadIdx.x == 0) { r = sm[0]; for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = ex_add(r, sm[i]); }
  return r;
}
// K1 + K3 + final reduction in ONE workgroup for small tables: fold with r (if any), then every term's round sums.
// Terms are spread over the waves of the block (a term with many pairs is split over several waves), so the only
// block-wide barriers are the one after the fold and the one before the final combine; wave 0 then writes
// result[term*4 + t] straight into host-mapped memory and releases `flag = seq`.
template <bool HI>
KBODY k_sc_small(const ScSmallArgs& a, Ext* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext part[64 * SC_SLOTS];  // [slot][t], slot = term * wpt + sub  (<= 64 slots)
  int tid = threadIdx.x, nt = blockDim.x;
  size_t n = a.n_after;
  if (a.has_r) {
    for (int t = 0; t < a.ntabs; t++) {
      Ext* o = a.out[t];
      if (a.in_ext[t]) { const Ext* p = (const Ext*)a.in[t]; for (size_t i = tid; i < n; i += nt) o[i] = ex_lerp(p[2 * i], p[2 * i + 1], a.r); }
      else { const u64* p = (const u64*)a.in[t]; for (size_t i = tid; i < n; i += nt) o[i] = ex_lerp_base(p[2 * i], p[2 * i + 1], a.r); }
    }
    __syncthreads();
  }
  size_t npairs = n / 2;
  int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  int wpt = a.nterms >= W ? 1 : W / a.nterms;  // waves per term
  for (int term = wave / wpt; term < a.nterms; term += (wpt == 1 ? W : a.nterms + W)) {
    int sub = wave % wpt;
    int k = a.k[term];
    GlobalPairs L;
#pragma unroll
    for (int j = 0; j < ScW<HI>::K; j++) { int ti = a.t[term][j < k ? j : 0]; L.p[j] = a.has_r ? (const void*)a.out[ti] : a.in[ti]; L.e[j] = a.has_r ? true : a.in_ext[ti]; }
    Ext acc[SC_SLOTS];
    sc_accumulate<HI>(k, L, (size_t)sub * 64 + lane, (size_t)wpt * 64, npairs, acc);
#pragma unroll
    for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
    if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
    if (wpt > 1) break;  // with several waves per term every wave owns exactly one (term, sub)
  }
  __syncthreads();
  if (wave == 0) sc_publish_fwd(result, part, a.k, a.off, a.nterms, wpt, flag, seq, lane);
}

// ------------------------------------------------------------------------------------------------ device-side Fiat-Shamir
// With many proofs in flight the per-round trip to the host (publish the round sums, the host runs the sponge, the kernel
// polls the mailbox) is what a proof spends its time on: the host threads serve several proofs each and every hop crosses
// PCIe. A persistent sumcheck can instead keep the transcript to itself: it gets the sponge state of the host transcript
// (DuplexChallenger<F,P,8,4>, poseidon/src/challenger.rs:14-46 — poseidon2.h `Challenger`), combines the term sums into the
// round message exactly as sumcheck_prove does (coefficients, extrapolation to max_degree + 1 points: prover.rs:498-585),
// absorbs it, squeezes the challenge (transcript/src/basic.rs:8-54) and goes on; at the end ONE publication carries all
// round messages, all challenges, the final evaluations and the sponge state back to the host transcript.
// Field arithmetic is exact and canonical, so the messages and challenges are the host's bit for bit.
// The sponge runs on wave 0 with the lane-parallel permutation (p2l_permute: state[i] in lane i of every group of 8).
// the reply poll of wc_request: bounded in time like sc_wait_challenge (the emulator of tests/ defines both macros itself and serves
// the request from inside the poll)
#ifndef WC_POLL_PAUSE
#define WC_POLL_BEGIN const unsigned long long wc_t0 = dp_realtime();
#define WC_POLL_PAUSE(spin) wc_poll_pause(wc_t0, spin)
__device__ __forceinline__ bool wc_poll_pause(unsigned long long t0, unsigned spin) {
  if ((spin & 63) == 63 && dp_realtime() - t0 > c_poll_timeout_ticks) return true;
  for (int q = 0; q < c_poll_sleep; q++) __builtin_amdgcn_s_sleep(16);
  return false;
}
#endif
struct ScFsArgs {
  u64 state[8]; u64 in_buf[4]; int in_len, out_len;  // the host Challenger at the start of the first round
  int md, rounds;                                    // max_degree of the virtual polynomial, rounds the kernel runs
  u64 label[2]; int nlabel, pad;                     // "Internal round" as transcript words
  Ext coeff[MAX_TERMS];                              // coefficient of every product term
};
__constant__ u64 c_extrap[(SC_MAXK + 1) * (SC_MAXK + 1) * (SC_MAXK + 1)];  // [k][at][i]: extrapolation_coeffs(k, at)[i] of sumcheck.h
// Host mode (req != nullptr, DP_HOST_SPONGE=1; sponge_host.h): the sponge stays in the host transcript. Observed words are staged in the
// mapped request area by lane 0; a sample posts the request (tag = sequence + checksum: the host re-reads until it is complete) and
// every lane polls the reply area for the sponge's output buffer (uniform decision: the tag and length lane 0 read, the four outputs
// lanes 0..3 read, validated by the reply's own tag). 2.9 us per round trip against ~12 us per permutation on an 8-lane wave.
struct WaveChallenger {
  u64 st, ib; int in_len, out_len;
  u64* req = nullptr; const u64* rep = nullptr; unsigned n = 0, consumed = 0, failed = 0; unsigned long long rseq = 0, cs = 0; u64 cache = 0;
};
__device__ __forceinline__ void wc_host_init(WaveChallenger& c, u64* req, const u64* rep, unsigned long long seq0) {
  c.req = req; c.rep = rep; c.rseq = seq0; c.n = c.consumed = c.failed = 0; c.cs = 0; c.cache = 0;
  if (req) c.out_len = 0;  // nothing cached: the first sample asks the host
}
__device__ __forceinline__ void wc_request(WaveChallenger& c, int lane, int want) {
  c.rseq++;
  if (lane == 0) {
    pub_store(c.req + 1, (u64)c.n); pub_store(c.req + 2, (u64)c.consumed); pub_store(c.req + 3, (u64)want);
    pub_store(c.req, wc_req_mix(c.rseq) + c.cs + 3ull * c.n + 5ull * c.consumed + 7ull * (u64)want);
  }
  const unsigned long long base = wc_rep_mix(c.rseq);
  u64 ol = 0, o = 0;
  bool ok = false;
  WC_POLL_BEGIN
  u64 seen = __hip_atomic_load(c.rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the previous reply's tag)
  seen = shfl_u64(seen, 0);
  for (unsigned spin = 0;; spin++) {
    // one PCIe read per poll (every lane asks for the same word): the payload is only fetched once the tag word has changed —
    // 264 workgroups polling three words each saturate the link's read rate and slow every poll down
    u64 tag = __hip_atomic_load(c.rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    tag = shfl_u64(tag, 0);
    if (tag != seen || spin == 0) {
      ol = __hip_atomic_load(c.rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      o = __hip_atomic_load(c.rep + 2 + (lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ol = shfl_u64(ol, 0);
      const u64 w = o * (u64)((lane & 3) + 1);
      const u64 sum = ol + shfl_u64(w, 0) + shfl_u64(w, 1) + shfl_u64(w, 2) + shfl_u64(w, 3);
      if (tag == base + sum && ol <= 4) { ok = true; break; }
    }
    if (WC_POLL_PAUSE(spin)) break;
  }
  c.n = 0; c.cs = 0; c.consumed = 0;
  if (ok) { c.out_len = (int)ol; c.cache = o; } else { c.failed = 1; c.out_len = 4; c.cache = 0; }
}
// end of a kernel: the words observed since the last sample, and the samples popped since the last reply, reach the host before the
// kernel's own message does (the host transcript is complete when the proof's thread sees that message)
__device__ __forceinline__ void wc_finish(WaveChallenger& c, int lane) { if (c.req) wc_request(c, lane, 0); }
__device__ __forceinline__ void wc_duplex(WaveChallenger& c, int lane) {
  if ((lane & 7) < c.in_len) c.st = c.ib;
  c.in_len = 0;
  c.st = p2l_permute(c.st, lane);
  c.out_len = 4;
}
__device__ __forceinline__ void wc_observe(WaveChallenger& c, u64 v, int lane) {  // v uniform over the wave
  if (c.req) {
    c.out_len = 0;
    if (lane == 0) { pub_store(c.req + 4 + c.n, v); c.cs += (unsigned long long)(c.n + 1) * v; }
    if (++c.n == WC_REQ_CAP) wc_request(c, lane, 0);
    return;
  }
  c.out_len = 0;
  if ((lane & 7) == c.in_len) c.ib = v;
  if (++c.in_len == 4) wc_duplex(c, lane);
}
__device__ __forceinline__ u64 wc_sample(WaveChallenger& c, int lane) {
  if (c.req) {
    if (c.n != 0 || c.out_len == 0) wc_request(c, lane, 1);
    --c.out_len; c.consumed++;
    return shfl_u64(c.cache, c.out_len);
  }
  if (c.in_len != 0 || c.out_len == 0) wc_duplex(c, lane);
  --c.out_len;
  return shfl_u64(c.st, c.out_len);
}
__device__ __forceinline__ Ext sc_term_sum(const Ext* part, int term, int t, int wpt) {
  if (wpt == 1) return part[(size_t)term * SC_SLOTS + t];
  Ext v = ex_zero();
  for (int sb = 0; sb < wpt; sb++) v = ex_add(v, part[(size_t)(term * wpt + sb) * SC_SLOTS + t]);
  return v;
}
// host result area of a device-driven sumcheck, in words: [rounds x (md+1) message values][rounds challenges][ntabs finals][14 sponge words]
__device__ __forceinline__ size_t fs_msg_word(const ScFsArgs& f, int round, int j) { return ((size_t)round * (f.md + 1) + j) * 2; }
__device__ __forceinline__ size_t fs_chal_word(const ScFsArgs& f, int round) { return ((size_t)f.rounds * (f.md + 1) + round) * 2; }
__device__ __forceinline__ size_t fs_final_word(const ScFsArgs& f) { return ((size_t)f.rounds * (f.md + 2)) * 2; }
// One round on wave 0 (all 64 lanes): part[] holds the raw term sums. Returns the challenge; `cs` accumulates the checksum
// of the words this lane has sent to the host.
__device__ __forceinline__ Ext sc_fs_round(WaveChallenger& wc, const ScFsArgs& f, const Ext* part, const int* tk, int nterms, int wpt, u64* rw, int round, unsigned long long& cs, int lane) {
  const int j = lane & 7, tg = lane >> 3;
  Ext acc = ex_zero();
  if (j <= f.md) {
    for (int t = tg; t < nterms; t += 8) {
      int k = tk[t];
      Ext v;
      if (j <= k) v = sc_term_sum(part, t, j, wpt);
      else {
        const u64* c = c_extrap + ((size_t)k * (SC_MAXK + 1) + j) * (SC_MAXK + 1);
        v = ex_zero();
        for (int i = 0; i <= k; i++) v = ex_add(v, ex_mul_base(sc_term_sum(part, t, i, wpt), c[i]));
      }
      acc = ex_add(acc, ex_mul(v, f.coeff[t]));
    }
  }
#pragma unroll
  for (int d = 8; d <= 32; d <<= 1) {
    Ext o = ex(shfl_u64(acc.c0, lane ^ d), shfl_u64(acc.c1, lane ^ d));
    acc = ex_add(acc, o);
  }
  for (int jj = 0; jj <= f.md; jj++) {
    u64 c0 = shfl_u64(acc.c0, jj), c1 = shfl_u64(acc.c1, jj);
    wc_observe(wc, c0, lane); wc_observe(wc, c1, lane);
    if (lane == 0) { size_t w = fs_msg_word(f, round, jj); pub_store(rw + w, c0); pub_store(rw + w + 1, c1); cs += (unsigned long long)(w + 1) * c0 + (unsigned long long)(w + 2) * c1; }
  }
  for (int q = 0; q < f.nlabel; q++) wc_observe(wc, f.label[q], lane);
  u64 r0 = wc_sample(wc, lane), r1 = wc_sample(wc, lane);
  if (lane == 0) { size_t w = fs_chal_word(f, round); pub_store(rw + w, r0); pub_store(rw + w + 1, r1); cs += (unsigned long long)(w + 1) * r0 + (unsigned long long)(w + 2) * r1; }
  return ex(r0, r1);
}
// the last message of a device-driven sumcheck: final evaluation of every table (src[e * stride]), the sponge, the tag
__device__ __forceinline__ void sc_fs_finish(const WaveChallenger& wc, const ScFsArgs& f, const Ext* const* srcs, const Ext* src, size_t stride, int ntabs, u64* rw, unsigned long long* flag, unsigned long long seq, unsigned long long cs, int lane) {
  size_t w0 = fs_final_word(f);
  for (int e = lane; e < ntabs; e += 64) {
    Ext v = srcs ? srcs[e][0] : src[(size_t)e * stride];
    size_t w = w0 + 2 * e;
    pub_store(rw + w, v.c0); pub_store(rw + w + 1, v.c1);
    cs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
  }
  size_t ws = w0 + 2 * (size_t)ntabs;
  if (lane < 8) { pub_store(rw + ws + lane, wc.st); cs += (unsigned long long)(ws + lane + 1) * wc.st; }
  if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rw + ws + 8 + lane, v); cs += (unsigned long long)(ws + 8 + lane + 1) * v; }
  if (lane == 0) {
    u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
    pub_store(rw + ws + 12, a); pub_store(rw + ws + 13, b);
    cs += (unsigned long long)(ws + 13) * a + (unsigned long long)(ws + 14) * b;
  }
  cs = pub_wave_sum(cs);
  if (lane == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
}
__device__ __forceinline__ void wc_load(WaveChallenger& wc, const ScFsArgs& f, int lane) {
  wc.st = f.state[lane & 7]; wc.ib = f.in_buf[lane & 3]; wc.in_len = f.in_len; wc.out_len = f.out_len;
}

// ------------------------------------------------------------------------------------------------ persistent sumcheck
// A whole (tail of a) sumcheck in ONE launch of ONE workgroup: per round the kernel publishes the raw term sums to
// host-mapped memory, the host runs the Fiat-Shamir sponge and posts the challenge into a host-mapped mailbox which
// the kernel polls; then the kernel folds every table and goes on. No kernel launch, no stream synchronisation and no
// memcpy on the per-round critical path — only two PCIe hops. Ends by publishing the final evaluation of every table.
struct ScPersistArgs {
  const void* in[MAX_TABS]; int in_ext[MAX_TABS];
  Ext* bufA[MAX_TABS]; Ext* bufB[MAX_TABS];
  int k[MAX_TERMS]; int t[MAX_TERMS][SC_MAXK]; int off[MAX_TERMS];
  int ntabs, nterms, has_r0; size_t n0; Ext r0;
  unsigned long long* dbg;  // optional: per-phase cycle counters (DP_SC_DEBUG=1)
  int eq_tab, eq_k;         // eq_tab >= 0: table eq_tab (an extension buffer in global memory) is eq(., eq_pt) and is built here first
  Ext eq_pt[MAX_PT];
  // multi-workgroup phase (k_sc_persist only): workgroup g of nwg owns the contiguous slice [g n0/nwg, (g+1) n0/nwg) of
  // every table, publishes the sums of its slice into result slot g (slot_ext extension values apart, flag word g) and
  // leaves after `rounds_a` folds; the host adds the shares. nwg = 1, rounds_a = 0: the whole sumcheck in one workgroup.
  int nwg, rounds_a, slot_ext;
};
// out[i] = prod_t (i_t ? pt[t] : 1 - pt[t]) for i < 2^k, by the whole workgroup (ends with a barrier)
__device__ __forceinline__ void wg_build_eq(Ext* out, const Ext* pt, int k) {
  size_t n = size_t(1) << k;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    Ext v = ex_one();
    for (int t = 0; t < k; t++) { Ext r = pt[t]; v = ex_mul(v, ((i >> t) & 1) ? r : ex_sub(ex_one(), r)); }
    out[i] = v;
  }
  __syncthreads();
}
__device__ __forceinline__ void sc_fold_all(const ScPersistArgs& a, const void* const* cur, const int* cur_ext, Ext* const* dst, size_t n_after, Ext r, size_t dst_off = 0) {
  int tid = threadIdx.x, nt = blockDim.x;
  for (int t = 0; t < a.ntabs; t++) {
    Ext* o = dst[t] + dst_off;
    if (cur_ext[t]) { const Ext* p = (const Ext*)cur[t]; for (size_t i = tid; i < n_after; i += nt) o[i] = ex_lerp(p[2 * i], p[2 * i + 1], r); }
    else { const u64* p = (const u64*)cur[t]; for (size_t i = tid; i < n_after; i += nt) o[i] = ex_lerp_base(p[2 * i], p[2 * i + 1], r); }
  }
}
template <bool HI>
KBODY k_sc_persist(const ScPersistArgs& a, Ext* result, unsigned long long* flag, const unsigned long long* mailbox, unsigned long long seq0, const ScFsArgs* fs) {
  __shared__ Ext part[64 * SC_SLOTS];
  __shared__ unsigned long long chal[3];
  __shared__ const void* cur[MAX_TABS];
  __shared__ int cur_ext[MAX_TABS];
  __shared__ Ext* dstA[MAX_TABS];
  __shared__ Ext* dstB[MAX_TABS];
  __shared__ ScFsArgs fsl;
  int tid = threadIdx.x, nt = blockDim.x;
  int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  const bool autofs = fs != nullptr;  // device-side Fiat-Shamir (single workgroup only): no host round trips
  if (autofs) for (int i = tid; i < (int)(sizeof(ScFsArgs) / 8); i += nt) ((u64*)&fsl)[i] = ((const u64*)fs)[i];
  if (a.eq_tab >= 0) wg_build_eq((Ext*)a.in[a.eq_tab], a.eq_pt, a.eq_k);
  const size_t g = blockIdx.x;
  size_t n = a.n0 / (size_t)a.nwg;  // local slice length
  if (tid < a.ntabs) { cur[tid] = (const char*)a.in[tid] + g * n * (a.in_ext[tid] ? 16 : 8); cur_ext[tid] = a.in_ext[tid]; dstA[tid] = a.bufA[tid]; dstB[tid] = a.bufB[tid]; }
  result += g * (size_t)a.slot_ext; flag += g;
  int folds = 0;
  size_t lvl_off = 0;
  __syncthreads();
  unsigned long long seq = seq0;
  bool useA = true;
  WaveChallenger wc; wc.st = wc.ib = 0; wc.in_len = wc.out_len = 0;
  if (autofs) wc_load(wc, fsl, lane);
  unsigned long long fcs = 0; int round = 0;
  if (a.has_r0) {
    // (several workgroups: the fold with the pending challenge fills this workgroup's slice of the first level region of bufA;
    // the regions of the later levels follow it — see the comment at the round fold below)
    const size_t off0 = a.nwg > 1 ? g * (n / 2) : 0;
    sc_fold_all(a, cur, cur_ext, dstA, n / 2, a.r0, off0);
    __syncthreads();
    if (tid < a.ntabs) { cur[tid] = dstA[tid] + off0; cur_ext[tid] = 1; }
    __syncthreads();
    n /= 2; useA = false;
    if (a.nwg > 1) lvl_off = (size_t)a.nwg * n;
  }
  int wpt = a.nterms >= W ? 1 : W / a.nterms;
  for (;;) {
    size_t npairs = n / 2;
    for (int term = wave / wpt; term < a.nterms; term += (wpt == 1 ? W : a.nterms + W)) {
      int sub = wave % wpt;
      int k = a.k[term];
      GlobalPairs L;
#pragma unroll
      for (int j = 0; j < ScW<HI>::K; j++) { int ti = a.t[term][j < k ? j : 0]; L.p[j] = cur[ti]; L.e[j] = cur_ext[ti]; }
      Ext acc[SC_SLOTS];
      sc_accumulate<HI>(k, L, (size_t)sub * 64 + lane, (size_t)wpt * 64, npairs, acc);
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
      if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
      if (wpt > 1) break;
    }
    __syncthreads();
    ++seq;
    if (wave == 0) {
      if (autofs) { Ext rr = sc_fs_round(wc, fsl, part, a.k, a.nterms, wpt, (u64*)result, round, fcs, lane); if (lane == 0) { chal[0] = 1; chal[1] = rr.c0; chal[2] = rr.c1; } }
      else { sc_publish_fwd(result, part, a.k, a.off, a.nterms, wpt, flag, seq, lane); if (lane == 0) sc_wait_challenge_fwd(mailbox, seq, chal); }
    }
    round++;
    __syncthreads();
    if (chal[0] == 0) {  // host never answered: publish an abort marker and leave
      if (tid == 0) pub_store((u64*)flag, ~0ull);
      return;
    }
    Ext r = ex(chal[1], chal[2]);
    // One workgroup: ping-pong between bufA and bufB. Several workgroups: the folded slice goes to its place in the compact
    // folded table of this level, and every level has its own region of bufA (level 1 at 0, level 2 behind it, ..): the
    // workgroups sit on different XCDs whose L2s are not coherent with each other, so no address may be written by two
    // workgroups during the life of the kernel (a stale dirty line of an old level could be written back over a new one).
    Ext* const* dst = (a.nwg > 1 || useA) ? dstA : dstB;
    size_t off = a.nwg > 1 ? lvl_off + g * (n / 2) : 0;
    sc_fold_all(a, cur, cur_ext, dst, n / 2, r, off);
    __syncthreads();
    if (tid < a.ntabs) { cur[tid] = dst[tid] + off; cur_ext[tid] = 1; }
    __syncthreads();
    lvl_off += (size_t)a.nwg * (n / 2);
    n /= 2; useA = !useA;
    if (++folds == a.rounds_a) return;  // end of the multi-workgroup phase: the compact folded tables are complete
    if (n == 1) {
      ++seq;
      if (wave == 0 && autofs) sc_fs_finish(wc, fsl, (const Ext* const*)cur, nullptr, 0, a.ntabs, (u64*)result, flag, seq0 + 1, fcs, lane);
      else if (wave == 0) {
        unsigned long long cs = 0; u64* rw = (u64*)result;
        for (int e = lane; e < a.ntabs; e += 64) { Ext v = ((const Ext*)cur[e])[0]; pub_store(rw + 2 * e, v.c0); pub_store(rw + 2 * e + 1, v.c1); cs += (unsigned long long)(2 * e + 1) * v.c0 + (unsigned long long)(2 * e + 2) * v.c1; }
        cs = pub_wave_sum(cs);
        if (lane == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
      }
      return;
    }
  }
}

// ------------------------------------------------------------------------------------------------ whole logup-GKR layer loop
// Dev::logup_tail (dev.h): every layer of a logup-GKR batch proof (logup_layers of logup.h: absorb the claim, the batched
// layer sumcheck with its Fiat-Shamir rounds, the three layer challenges, the next claim) in ONE launch of one workgroup —
// one device wait per lookup argument instead of one per tree layer; in full mode (Dev::logup_full, DP_DEVICE_LOGUP=2) also
// the trees, the circuit outputs, the initial challenges and the column claims. Default in throughput mode (DP_DEVICE_LOGUP=0 / 1 select the
// layer-by-layer path / the layer loop only); written against the contract pinned by the CPU double (tests/support/cpu_dev.hpp),
// checked on the SIMT emulator and on MI355X (tests/test_gpu_fused.py).
// Tree layers stay where k_logup_tree / k_logup_layer left them (global memory, read once per layer); folded tables
// ping-pong through bufA / bufB like k_sc_persist. Result area, in words, one block per layer lv = 1..L followed by the
// sponge: [lv x 4 message values][lv challenges][batching][final evaluations without eq] ... [8 state, 4 input buffer,
// in_len, out_len]; the tag is mix(seq) + sum over blocks of sum_i (i + 1) * word_i with i relative to the block.
static_assert(LT_MAX_TABS == MAX_TABS, "logup_tail.h: table capacity");
KBODY k_logup_tail(const LogupTailDesc* dp, u64* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext part[64 * SC_SLOTS];
  __shared__ unsigned long long chal[3];
  __shared__ const void* cur[MAX_TABS];
  __shared__ int cur_ext[MAX_TABS];
  __shared__ int tk[MAX_TERMS];
  __shared__ int tt[MAX_TERMS][3];
  __shared__ int s_ntab, s_nterm;
  __shared__ ScFsArgs fsl;
  __shared__ LogupTailDesc dl;
  __shared__ Ext pt[MAX_PT];
  __shared__ Ext glue[4];  // batching, alpha, lambda, claim of the layer at hand
  __shared__ Ext outs[LT_MAXI * 4];  // full mode: [n0, n1, d0, d1] of every instance
  const int tid = threadIdx.x, nt = blockDim.x;
  const int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
#ifdef DP_WG_TIMES
  const unsigned long long dbg_t_in = dp_realtime();
#endif
  for (int i = tid; i < (int)(sizeof(LogupTailDesc) / 8); i += nt) ((u64*)&dl)[i] = ((const u64*)dp)[i];
  __syncthreads();
  if (tid == 0) {
    glue[0] = dl.batching; glue[1] = dl.alpha; glue[2] = dl.lambda; glue[3] = dl.claim; pt[0] = dl.batching;
    fsl.md = 3; fsl.label[0] = dl.lab_round[0]; fsl.label[1] = dl.lab_round[1]; fsl.nlabel = 2; fsl.pad = 0;
  }
  WaveChallenger wc;
  wc.st = dl.state[lane & 7]; wc.ib = dl.in_buf[lane & 3]; wc.in_len = dl.in_len; wc.out_len = dl.out_len;
  wc_host_init(wc, dl.sp_req, dl.sp_rep, dl.sp_seq);
  unsigned long long fcs = 0;
  size_t wbase = 0;
  __syncthreads();
  if (dl.full) {
    // ---- full mode (Dev::logup_full). The fractional-sum tree of every instance, as k_logup_tree builds it: denominators
    // c + sum_j chi^j col_j, then layer by layer (n1 d2 + d1 n2, d1 d2) over the pairs (i, i + half)
    const size_t n = dl.n;
    for (int s = 0; s < dl.ninst; s++) {
      Ext* den = dl.den_all[s];
      Ext* num_out = dl.num_all[s];
      for (size_t i = tid; i < n; i += nt) {
        Ext acc = dl.c, pw = ex_one();
        for (int j = 0; j < dl.cpi; j++) { acc = ex_add(acc, ex_mul_base(pw, dl.col[s][j][i])); pw = ex_mul(pw, dl.chi); }
        den[i] = acc;
      }
      __syncthreads();
      const void* num = dl.mult;
      int mode = dl.is_table ? 1 : 0;  // 0: all numerators are -1 (lookup), 1: base-field multiplicities (table), 2: extension
      size_t len = n, doff = 0, noff = 0;
      while (len > 2) {
        const size_t half = len / 2;
        const Ext* dcur = den + doff;
        Ext* dnext = den + doff + len;
        for (size_t i = tid; i < half; i += nt) {
          Ext d1 = dcur[i], d2 = dcur[i + half], nn;
          if (mode == 0) nn = ex_neg(ex_add(d1, d2));
          else if (mode == 1) { const u64* q = (const u64*)num; nn = ex_add(ex_mul_base(d2, q[i]), ex_mul_base(d1, q[i + half])); }
          else { const Ext* q = (const Ext*)num; nn = ex_add(ex_mul(q[i], d2), ex_mul(d1, q[i + half])); }
          num_out[noff + i] = nn;
          dnext[i] = ex_mul(d1, d2);
        }
        __syncthreads();
        num = (const void*)(num_out + noff); mode = 2;
        doff += len; noff += half; len = half;
      }
      if (tid < 4) {  // the 2-element top layer: [n0, n1, d0, d1]
        Ext v;
        if (tid < 2) { if (mode == 0) v = ex_neg(ex_one()); else if (mode == 1) v = ex_base(((const u64*)num)[tid]); else v = ((const Ext*)num)[tid]; }
        else v = (den + doff)[tid - 2];
        outs[4 * s + tid] = v;
      }
      __syncthreads();
    }
    if (tid == 0)
      for (int s = 0; s < dl.ninst; s++)
        for (int li = 0; li < dl.nlayers; li++) {
          dl.den[s][li] = dl.den_all[s] + (2 * n - ((2 * n) >> li));
          dl.num[s][li] = li == 0 ? (const void*)dl.mult : (const void*)(dl.num_all[s] + (n - ((2 * n) >> li)));
        }
    // transcript: the number of instances, their outputs, the three initial challenges; the first claim
    if (wave == 0) {
      wc_observe(wc, (u64)dl.ninst, lane);
      for (int q = 0; q < 4 * dl.ninst; q++) { Ext v = outs[q]; wc_observe(wc, v.c0, lane); wc_observe(wc, v.c1, lane); }
      u64 b0, b1;
      wc_observe(wc, dl.lab_ibatching[0], lane); wc_observe(wc, dl.lab_ibatching[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext bt = ex(b0, b1);
      wc_observe(wc, dl.lab_ialpha[0], lane); wc_observe(wc, dl.lab_ialpha[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext al = ex(b0, b1);
      wc_observe(wc, dl.lab_ilambda[0], lane); wc_observe(wc, dl.lab_ilambda[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext la = ex(b0, b1);
      Ext claim = ex_zero(), ac = ex_one();
      for (int s = 0; s < dl.ninst; s++) {
        const Ext* e = outs + 4 * s;
        Ext a = ex_add(ex_mul(bt, ex_sub(e[1], e[0])), e[0]);
        Ext b = ex_add(ex_mul(bt, ex_sub(e[3], e[2])), e[2]);
        claim = ex_add(claim, ex_mul(ac, ex_add(a, ex_mul(la, b))));
        ac = ex_mul(ac, al);
      }
      if (lane == 0) { glue[0] = bt; glue[1] = al; glue[2] = la; glue[3] = claim; pt[0] = bt; }
      for (int e = lane; e < 4 * dl.ninst; e += 64) {
        Ext v = outs[e];
        size_t w = 2 * (size_t)e;
        pub_store(result + w, v.c0); pub_store(result + w + 1, v.c1);
        fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
      }
    }
    wbase = (size_t)dl.ninst * 8;
    __syncthreads();
  }
  for (int lv = 1; lv <= dl.total_layers; lv++) {
    const size_t half = size_t(1) << lv;
    const int li = dl.nlayers - 1 - lv;  // layers().iter().rev().skip(1)
    // transcript: the running claim, then the header of the layer's sumcheck (num_vars, max_degree)
    if (wave == 0) {
      Ext c = glue[3];
      wc_observe(wc, c.c0, lane); wc_observe(wc, c.c1, lane);
      wc_observe(wc, (u64)lv, lane); wc_observe(wc, (u64)3, lane);
    }
    wg_build_eq(dl.eq, pt, lv);  // eq(point, .) over the layer's lv variables (ends with a barrier)
    if (tid == 0) {
      cur[0] = dl.eq; cur_ext[0] = 1;
      int ntab = 1, nterm = 0;
      Ext ca = ex_one();
      const Ext al = glue[1], la = glue[2];
      const bool init_lk = !dl.is_table && li == 0;   // initial lookup layer: all numerators are -1
      const bool nbase = dl.is_table && li == 0;      // table layer 0: base-field multiplicities
      for (int i = 0; i < dl.ninst; i++) {
        const Ext* dlo = dl.den[i][li];
        const Ext* dhi = dlo + half;
        if (!init_lk) {
          const void* nlo = dl.num[i][li];
          const void* nhi = (const char*)nlo + half * (nbase ? 8 : 16);
          const int t_nlo = ntab, t_dhi = ntab + 1, t_nhi = ntab + 2, t_dlo = ntab + 3;
          cur[t_nlo] = nlo; cur_ext[t_nlo] = nbase ? 0 : 1;
          cur[t_dhi] = dhi; cur_ext[t_dhi] = 1;
          cur[t_nhi] = nhi; cur_ext[t_nhi] = nbase ? 0 : 1;
          cur[t_dlo] = dlo; cur_ext[t_dlo] = 1;
          ntab += 4;
          tk[nterm] = 3; tt[nterm][0] = 0; tt[nterm][1] = t_nlo; tt[nterm][2] = t_dhi; fsl.coeff[nterm] = ca; nterm++;
          tk[nterm] = 3; tt[nterm][0] = 0; tt[nterm][1] = t_nhi; tt[nterm][2] = t_dlo; fsl.coeff[nterm] = ca; nterm++;
          tk[nterm] = 3; tt[nterm][0] = 0; tt[nterm][1] = t_dlo; tt[nterm][2] = t_dhi; fsl.coeff[nterm] = ex_mul(ca, la); nterm++;
        } else {
          const int t_dhi = ntab, t_dlo = ntab + 1;
          cur[t_dhi] = dhi; cur_ext[t_dhi] = 1;
          cur[t_dlo] = dlo; cur_ext[t_dlo] = 1;
          ntab += 2;
          tk[nterm] = 2; tt[nterm][0] = 0; tt[nterm][1] = t_dhi; tt[nterm][2] = 0; fsl.coeff[nterm] = ex_neg(ca); nterm++;
          tk[nterm] = 2; tt[nterm][0] = 0; tt[nterm][1] = t_dlo; tt[nterm][2] = 0; fsl.coeff[nterm] = ex_neg(ca); nterm++;
          tk[nterm] = 3; tt[nterm][0] = 0; tt[nterm][1] = t_dlo; tt[nterm][2] = t_dhi; fsl.coeff[nterm] = ex_mul(ca, la); nterm++;
        }
        ca = ex_mul(ca, al);
      }
      s_ntab = ntab; s_nterm = nterm; fsl.rounds = lv;
    }
    __syncthreads();
    const int ntab = s_ntab, nterm = s_nterm;
    const int wpt = nterm >= W ? 1 : W / nterm;
    u64* rw = result + wbase;
    size_t n = half;
    bool useA = true;
    for (int round = 0; round < lv; round++) {
      const size_t npairs = n / 2;
      for (int term = wave / wpt; term < nterm; term += (wpt == 1 ? W : nterm + W)) {
        const int sub = wave % wpt;
        const int k = tk[term];
        GlobalPairs L;
#pragma unroll
        for (int j = 0; j < 3; j++) { int ti = tt[term][j < k ? j : 0]; L.p[j] = cur[ti]; L.e[j] = cur_ext[ti] != 0; }
        Ext acc[SC_SLOTS];
        sc_accumulate<false>(k, L, (size_t)sub * 64 + lane, (size_t)wpt * 64, npairs, acc);
#pragma unroll
        for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
        if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
        if (wpt > 1) break;  // with several waves per term every wave owns exactly one (term, sub)
      }
      __syncthreads();
      if (wave == 0) {
        Ext rr = sc_fs_round(wc, fsl, part, tk, nterm, wpt, rw, round, fcs, lane);
        if (lane == 0) { chal[1] = rr.c0; chal[2] = rr.c1; pt[round] = rr; }  // the eq table of this layer is built: pt may take the new point
      }
      __syncthreads();
      const Ext r = ex(chal[1], chal[2]);
      Ext* const* dst = useA ? dl.bufA : dl.bufB;
      for (int t = 0; t < ntab; t++) {
        Ext* o = dst[t];
        if (cur_ext[t]) { const Ext* q = (const Ext*)cur[t]; for (size_t i = tid; i < npairs; i += nt) o[i] = ex_lerp(q[2 * i], q[2 * i + 1], r); }
        else { const u64* q = (const u64*)cur[t]; for (size_t i = tid; i < npairs; i += nt) o[i] = ex_lerp_base(q[2 * i], q[2 * i + 1], r); }
      }
      __syncthreads();
      if (tid < ntab) { cur[tid] = dst[tid]; cur_ext[tid] = 1; }
      __syncthreads();
      n = npairs; useA = !useA;
    }
    // every table is one value now: the layer's evaluations, the three layer challenges, the next claim
    if (wave == 0) {
      u64 b0, b1;
      wc_observe(wc, dl.lab_batching[0], lane); wc_observe(wc, dl.lab_batching[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext nb = ex(b0, b1);
      wc_observe(wc, dl.lab_alpha[0], lane); wc_observe(wc, dl.lab_alpha[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext na = ex(b0, b1);
      wc_observe(wc, dl.lab_lambda[0], lane); wc_observe(wc, dl.lab_lambda[1], lane); b0 = wc_sample(wc, lane); b1 = wc_sample(wc, lane);
      const Ext nl = ex(b0, b1);
      const size_t wb = (size_t)lv * 10;  // behind lv * 4 message values and lv challenges
      if (lane == 0) { pub_store(rw + wb, nb.c0); pub_store(rw + wb + 1, nb.c1); fcs += (unsigned long long)(wb + 1) * nb.c0 + (unsigned long long)(wb + 2) * nb.c1; }
      for (int e = lane; e < ntab - 1; e += 64) {
        Ext v = ((const Ext*)cur[e + 1])[0];
        size_t w = wb + 2 + 2 * (size_t)e;
        pub_store(rw + w, v.c0); pub_store(rw + w + 1, v.c1);
        fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
      }
      const bool lookup_final = lv == dl.total_layers && !dl.is_table;  // final_round_claim (prover.rs:201-237)
      Ext acc = ex_zero(), acomb = ex_one();
      int tb = 1;
      for (int i = 0; i < dl.ninst; i++) {
        if (!lookup_final) {
          Ext e0 = ((const Ext*)cur[tb])[0], e1 = ((const Ext*)cur[tb + 1])[0], e2 = ((const Ext*)cur[tb + 2])[0], e3 = ((const Ext*)cur[tb + 3])[0];
          Ext a = ex_add(ex_mul(nb, ex_sub(e2, e0)), e0);
          Ext b = ex_add(ex_mul(nb, ex_sub(e1, e3)), e3);
          acc = ex_add(acc, ex_mul(acomb, ex_add(a, ex_mul(nl, b))));
          tb += 4;
        } else {
          Ext e0 = ((const Ext*)cur[tb])[0], e1 = ((const Ext*)cur[tb + 1])[0];
          acc = ex_add(acc, ex_mul(acomb, ex_add(ex_mul(nb, ex_sub(e0, e1)), e1)));
          tb += 2;
        }
        acomb = ex_mul(acomb, na);
      }
      if (lane == 0) { glue[0] = nb; glue[1] = na; glue[2] = nl; glue[3] = acc; pt[lv] = nb; }
    }
    wbase += ((size_t)lv * 5 + 1 + (size_t)(ntab - 1)) * 2;
    __syncthreads();
  }
  if (dl.full) {
    // ---- full mode: the output claims — [multiplicities,] columns evaluated at the final point (sum_i eq(point, i) col[i])
    wg_build_eq(dl.eqn, pt, dl.nlayers);
    const int tcol = dl.is_table ? 1 : 0;
    const int ncol = tcol + dl.ninst * dl.cpi;
    u64* rw = result + wbase;
    for (int cidx = 0; cidx < ncol; cidx++) {
      const u64* col = cidx < tcol ? dl.mult : dl.col[(cidx - tcol) / dl.cpi][(cidx - tcol) % dl.cpi];
      Ext acc = ex_zero();
      for (size_t i = tid; i < dl.n; i += nt) acc = ex_add(acc, ex_mul_base(dl.eqn[i], col[i]));
      acc = wave_reduce_ext(acc);
      if (lane == 0) part[wave] = acc;
      __syncthreads();
      if (wave == 0) {
        Ext v = lane < W ? part[lane] : ex_zero();
        v = wave_reduce_ext(v);
        if (lane == 0) { size_t w = 2 * (size_t)cidx; pub_store(rw + w, v.c0); pub_store(rw + w + 1, v.c1); fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1; }
      }
      __syncthreads();
    }
    wbase += 2 * (size_t)ncol;
  }
  if (wave == 0) {  // the sponge goes back to the host transcript; the tag closes the message
    u64* rw = result + wbase;
    if (lane < 8) { pub_store(rw + lane, wc.st); fcs += (unsigned long long)(lane + 1) * wc.st; }
    if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rw + 8 + lane, v); fcs += (unsigned long long)(8 + lane + 1) * v; }
    if (lane == 0) {
      u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
      pub_store(rw + 12, a); pub_store(rw + 13, b);
      fcs += (unsigned long long)13 * a + (unsigned long long)14 * b;
    }
    fcs = pub_wave_sum(fcs);
    wc_finish(wc, lane);  // (host sponge: the transcript is complete before this message is)
    if (lane == 0) pub_store((u64*)flag, wc.failed ? ~0ull : pub_mix(seq) + fcs);
#ifdef DP_WG_TIMES
    if (tid == 0) dbg_wg_record(dbg_t_in);
#endif
  }
}

// ------------------------------------------------------------------------------------------------ batch-opening sumcheck tail
// Dev::classic_tail (dev.h, classic_tail.h): the remaining rounds of the batch-opening sumcheck of pcs_batch_open — per round
// fold every (f, eq) pair, the per-pair sums of Dev::classic_round, the 3-coefficient message of classic_round_message
// (pcs.h), absorb, squeeze "sumcheck round" — in ONE launch of one workgroup once every table is short. A pair belongs to
// one wave per phase; the phases of a round are separated by barriers. Default in throughput mode (DP_DEVICE_CLASSIC=0 turns
// it off); checked on the SIMT emulator of tests/ and on MI355X (tests/test_gpu_fused.py).
KBODY k_classic_tail(const ClassicTailDesc* dp, u64* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ ClassicTailDesc dl;
  __shared__ Ext raw[2 * CT_MAXP];
  __shared__ const void* curf[CT_MAXP];
  __shared__ const Ext* cure[CT_MAXP];
  __shared__ unsigned clen[CT_MAXP];
  __shared__ unsigned cext[CT_MAXP];
  __shared__ unsigned toA[CT_MAXP];
  __shared__ unsigned long long chal[3];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (int)(sizeof(ClassicTailDesc) / 8); i += nt) ((u64*)&dl)[i] = ((const u64*)dp)[i];
  __syncthreads();
  const int np = dl.np;
  for (int i = tid; i < np; i += nt) { curf[i] = dl.f[i]; cure[i] = dl.eq[i]; clen[i] = dl.len[i]; cext[i] = dl.f_ext[i]; toA[i] = 1; }
  WaveChallenger wc;
  wc.st = dl.state[lane & 7]; wc.ib = dl.in_buf[lane & 3]; wc.in_len = dl.in_len; wc.out_len = dl.out_len;
  wc_host_init(wc, dl.sp_req, dl.sp_rep, dl.sp_seq);
  unsigned long long fcs = 0;
  const int R = (int)(dl.num_vars - dl.round);
  Ext r = dl.r, sum = dl.sum;
  bool have_r = dl.has_r != 0;
  __syncthreads();
  for (int q = 0; q < R; q++) {
    if (have_r) {
      // fold f and eq of every pair that is longer than one value
      for (int i = wave; i < np; i += W) {
        const unsigned n = clen[i];
        if (n <= 1) continue;
        const unsigned half = n / 2;
        Ext* df = toA[i] ? dl.fA[i] : dl.fB[i];
        Ext* de = toA[i] ? dl.eA[i] : dl.eB[i];
        const Ext* e = cure[i];
        if (cext[i]) { const Ext* f = (const Ext*)curf[i]; for (unsigned j = lane; j < half; j += 64) df[j] = ex_lerp(f[2 * j], f[2 * j + 1], r); }
        else { const u64* f = (const u64*)curf[i]; for (unsigned j = lane; j < half; j += 64) df[j] = ex_lerp_base(f[2 * j], f[2 * j + 1], r); }
        for (unsigned j = lane; j < half; j += 64) de[j] = ex_lerp(e[2 * j], e[2 * j + 1], r);
      }
      __syncthreads();
      for (int i = tid; i < np; i += nt)
        if (clen[i] > 1) { curf[i] = toA[i] ? dl.fA[i] : dl.fB[i]; cure[i] = toA[i] ? dl.eA[i] : dl.eB[i]; cext[i] = 1; clen[i] /= 2; toA[i] ^= 1; }
      __syncthreads();
    }
    // per pair: c0 = sum_j f[2j] eq[2j], c2 = sum_j (f[2j+1] - f[2j]) (eq[2j+1] - eq[2j]); a single value gives (f eq, 0)
    for (int i = wave; i < np; i += W) {
      const unsigned n = clen[i];
      const Ext* e = cure[i];
      Ext c0 = ex_zero(), c2 = ex_zero();
      if (n == 1) { if (lane == 0) c0 = ex_mul(ld_elem(curf[i], cext[i] != 0, 0), e[0]); }
      else
        for (unsigned j = lane; j < n / 2; j += 64) {
          Ext l0 = e[2 * j], l1 = e[2 * j + 1], r0 = ld_elem(curf[i], cext[i] != 0, 2 * j), r1 = ld_elem(curf[i], cext[i] != 0, 2 * j + 1);
          c0 = ex_add(c0, ex_mul(l0, r0));
          c2 = ex_add(c2, ex_mul(ex_sub(l1, l0), ex_sub(r1, r0)));
        }
      c0 = wave_reduce_ext(c0); c2 = wave_reduce_ext(c2);
      if (lane == 0) { raw[2 * i] = c0; raw[2 * i + 1] = c2; }
    }
    __syncthreads();
    if (wave == 0) {
      // the message [h0, h1, h2] (classic_round_message of pcs.h), the transcript, the next claim
      const size_t size = size_t(1) << (dl.num_vars - (dl.round + (unsigned)q) - 1);
      Ext h0 = ex_zero(), h2 = ex_zero();
      for (int i = lane; i < np; i += 64) {
        const size_t poly_len = clen[i];
        Ext c0 = raw[2 * i], c2 = raw[2 * i + 1];
        size_t multiple;
        if (poly_len == 1) multiple = size;
        else if (size < poly_len || size == 1) multiple = 1;
        else multiple = size / (poly_len >> 1);
        if (multiple != 1) { Ext m = ex_from_u64((u64)multiple); c0 = ex_mul(c0, m); c2 = ex_mul(c2, m); }
        h0 = ex_add(h0, ex_mul(dl.eq_xt[i], c0));
        h2 = ex_add(h2, ex_mul(dl.eq_xt[i], c2));
      }
      h0 = wave_reduce_ext(h0); h2 = wave_reduce_ext(h2);
      h0 = ex(shfl_u64(h0.c0, 0), shfl_u64(h0.c1, 0)); h2 = ex(shfl_u64(h2.c0, 0), shfl_u64(h2.c1, 0));
      const Ext h1 = ex_sub(ex_sub(sum, ex_dbl(h0)), h2);
      wc_observe(wc, h0.c0, lane); wc_observe(wc, h0.c1, lane);
      wc_observe(wc, h1.c0, lane); wc_observe(wc, h1.c1, lane);
      wc_observe(wc, h2.c0, lane); wc_observe(wc, h2.c1, lane);
      wc_observe(wc, dl.lab[0], lane); wc_observe(wc, dl.lab[1], lane);
      const u64 r0 = wc_sample(wc, lane), r1 = wc_sample(wc, lane);
      const Ext ch = ex(r0, r1);
      if (lane == 0) {
        const Ext m[3] = {h0, h1, h2};
        for (int j = 0; j < 3; j++) {
          size_t w = ((size_t)q * 3 + j) * 2;
          pub_store(result + w, m[j].c0); pub_store(result + w + 1, m[j].c1);
          fcs += (unsigned long long)(w + 1) * m[j].c0 + (unsigned long long)(w + 2) * m[j].c1;
        }
        size_t w = ((size_t)R * 3 + q) * 2;
        pub_store(result + w, r0); pub_store(result + w + 1, r1);
        fcs += (unsigned long long)(w + 1) * r0 + (unsigned long long)(w + 2) * r1;
        chal[1] = r0; chal[2] = r1;
      }
      sum = ex_add(h0, ex_mul(ch, ex_add(h1, ex_mul(ch, h2))));
    }
    __syncthreads();
    r = ex(chal[1], chal[2]);
    have_r = true;
  }
  if (wave == 0) {  // the sponge goes back to the host transcript; the tag closes the message
    u64* rw = result + (size_t)R * 8;
    if (lane < 8) { pub_store(rw + lane, wc.st); fcs += (unsigned long long)(lane + 1) * wc.st; }
    if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rw + 8 + lane, v); fcs += (unsigned long long)(8 + lane + 1) * v; }
    if (lane == 0) {
      u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
      pub_store(rw + 12, a); pub_store(rw + 13, b);
      fcs += (unsigned long long)13 * a + (unsigned long long)14 * b;
    }
    fcs = pub_wave_sum(fcs);
    wc_finish(wc, lane);  // (host sponge: the transcript is complete before this message is)
    if (lane == 0) pub_store((u64*)flag, wc.failed ? ~0ull : pub_mix(seq) + fcs);
  }
}

// ------------------------------------------------------------------------------------------------ dense layer in one launch
// Dev::dense_tail (dev.h, dense_tail.h): the bias at the output point, W(point, .) — the one-pass fix_high over the base-field
// weights — and the degree-2 sumcheck of sum_c W(point, c) in(c) with its transcript, in ONE launch of one workgroup: a
// 1024 x 1024 layer is 8 MB of weights through one CU (~0.15 ms), cheaper than the five dispatches it replaces when the
// GPU serves hundreds of proofs. Default in throughput mode (DP_DEVICE_DENSE=0 turns it off); checked on the SIMT
// emulator of tests/ and on MI355X (tests/test_gpu_fused.py).
KBODY k_dense_tail(const DenseTailDesc* dp, u64* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ DenseTailDesc dl;
  __shared__ Ext part[64 * SC_SLOTS];
  __shared__ unsigned long long chal[3];
  __shared__ int tk[1];
  __shared__ ScFsArgs fsl;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (int)(sizeof(DenseTailDesc) / 8); i += nt) ((u64*)&dl)[i] = ((const u64*)dp)[i];
  __syncthreads();
  if (tid == 0) { fsl.md = 2; fsl.rounds = (int)dl.lgC; fsl.label[0] = dl.lab_round[0]; fsl.label[1] = dl.lab_round[1]; fsl.nlabel = 2; fsl.pad = 0; fsl.coeff[0] = ex_one(); tk[0] = 2; }
  WaveChallenger wc;
  wc.st = dl.state[lane & 7]; wc.ib = dl.in_buf[lane & 3]; wc.in_len = dl.in_len; wc.out_len = dl.out_len;
  wc_host_init(wc, dl.sp_req, dl.sp_rep, dl.sp_seq);
  unsigned long long fcs = 0;
  const size_t R = dl.R, C = dl.C;
  wg_build_eq(dl.eqr, dl.pt, (int)dl.lgR);  // eq(point, .) over the rows (ends with a barrier)
  // bias at the point
  {
    Ext acc = ex_zero();
    for (size_t i = tid; i < R; i += nt) acc = ex_add(acc, ex_mul_base(dl.eqr[i], dl.bias[i]));
    acc = wave_reduce_ext(acc);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (wave == 0) {
      Ext v = lane < W ? part[lane] : ex_zero();
      v = wave_reduce_ext(v);
      if (lane == 0) { pub_store(result, v.c0); pub_store(result + 1, v.c1); fcs += (unsigned long long)1 * v.c0 + (unsigned long long)2 * v.c1; }
    }
    __syncthreads();
  }
  // W(point, c) = sum_r eq(point, r) W[r][c]: consecutive threads read consecutive columns of a row
  for (size_t c = tid; c < C; c += nt) {
    Ext acc = ex_zero();
    for (size_t r = 0; r < R; r++) acc = ex_add(acc, ex_mul_base(dl.eqr[r], dl.W[r * C + c]));
    dl.mat[c] = acc;
  }
  __syncthreads();
  // the sumcheck of mat * in: header, then log2(C) rounds
  if (wave == 0) { wc_observe(wc, (u64)dl.lgC, lane); wc_observe(wc, (u64)2, lane); }
  const Ext* cur[2] = {dl.mat, dl.in};
  u64* rw = result + 2;
  size_t n = C;
  bool useA = true;
  for (int round = 0; round < (int)dl.lgC; round++) {
    const size_t npairs = n / 2;
    {
      GlobalPairs L;
      L.p[0] = cur[0]; L.e[0] = true; L.p[1] = cur[1]; L.e[1] = true; L.p[2] = cur[0]; L.e[2] = true;
      Ext acc[SC_SLOTS];
      sc_accumulate<false>(2, L, (size_t)tid, (size_t)nt, npairs, acc);  // every wave takes a share of the one term
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) if (t <= 2) acc[t] = wave_reduce_ext(acc[t]);
      if (lane == 0) { Ext* o = part + (size_t)wave * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
    }
    __syncthreads();
    if (wave == 0) {
      Ext rr = sc_fs_round(wc, fsl, part, tk, 1, W, rw, round, fcs, lane);  // wpt = W: the term's sums are spread over all waves
      if (lane == 0) { chal[1] = rr.c0; chal[2] = rr.c1; }
    }
    __syncthreads();
    const Ext r = ex(chal[1], chal[2]);
    Ext* const* dst = useA ? dl.bufA : dl.bufB;
    for (int t = 0; t < 2; t++) { const Ext* q = cur[t]; Ext* o = dst[t]; for (size_t i = tid; i < npairs; i += nt) o[i] = ex_lerp(q[2 * i], q[2 * i + 1], r); }
    __syncthreads();
    cur[0] = dst[0]; cur[1] = dst[1];
    n = npairs; useA = !useA;
  }
  if (wave == 0) {
    const size_t wf = (size_t)dl.lgC * 8;  // behind 3 evaluations and one challenge per round
    if (lane < 2) { Ext v = cur[lane][0]; size_t w = wf + 2 * (size_t)lane; pub_store(rw + w, v.c0); pub_store(rw + w + 1, v.c1); fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1; }
    u64* rs = result + (1 + (size_t)dl.lgC * 4 + 2) * 2;
    if (lane < 8) { pub_store(rs + lane, wc.st); fcs += (unsigned long long)(lane + 1) * wc.st; }
    if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rs + 8 + lane, v); fcs += (unsigned long long)(8 + lane + 1) * v; }
    if (lane == 0) {
      u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
      pub_store(rs + 12, a); pub_store(rs + 13, b);
      fcs += (unsigned long long)13 * a + (unsigned long long)14 * b;
    }
    fcs = pub_wave_sum(fcs);
    wc_finish(wc, lane);  // (host sponge: the transcript is complete before this message is)
    if (lane == 0) pub_store((u64*)flag, wc.failed ? ~0ull : pub_mix(seq) + fcs);
  }
}

// ------------------------------------------------------------------------------------------------ eq tables + sumcheck in one launch
// Dev::eqsum_tail (dev.h, eqsum_tail.h): the eq tables of an accumulation sumcheck (Requant: three plain tables; same_poly:
// one table accumulated from scaled eq's) and the whole sumcheck over them with its transcript, in ONE launch of one
// workgroup. Default in throughput mode (DP_DEVICE_EQSUM=0 turns it off); checked on the SIMT emulator and on MI355X.
KBODY k_eqsum_tail(const EqSumDesc* dp, u64* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ EqSumDesc dl;
  __shared__ Ext part[64 * SC_SLOTS];
  __shared__ unsigned long long chal[3];
  __shared__ const void* cur[ES_MAXT];
  __shared__ int cur_ext[ES_MAXT];
  __shared__ ScFsArgs fsl;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (int)(sizeof(EqSumDesc) / 8); i += nt) ((u64*)&dl)[i] = ((const u64*)dp)[i];
  __syncthreads();
  const int ntab = dl.ntabs, nterm = dl.nterms;
  if (tid == 0) { fsl.md = (int)dl.md; fsl.rounds = (int)dl.nv; fsl.label[0] = dl.lab_round[0]; fsl.label[1] = dl.lab_round[1]; fsl.nlabel = 2; fsl.pad = 0; }
  for (int i = tid; i < nterm; i += nt) fsl.coeff[i] = dl.coeff[i];
  for (int i = tid; i < ntab; i += nt) { cur[i] = dl.tab[i]; cur_ext[i] = dl.tab_ext[i]; }
  WaveChallenger wc;
  wc.st = dl.state[lane & 7]; wc.ib = dl.in_buf[lane & 3]; wc.in_len = dl.in_len; wc.out_len = dl.out_len;
  wc_host_init(wc, dl.sp_req, dl.sp_rep, dl.sp_seq);
  unsigned long long fcs = 0;
  const size_t n0 = size_t(1) << dl.nv;
  // the eq tables, in order: out[idx] (+)= scale * prod_t (idx_t ? pt[t] : 1 - pt[t])
  for (int j = 0; j < dl.njobs; j++) {
    Ext* out = dl.job_out[j];
    const Ext sc = dl.job_scale[j];
    for (size_t i = tid; i < n0; i += nt) {
      Ext v = sc;
      for (unsigned t = 0; t < dl.nv; t++) { Ext r = dl.job_pt[j][t]; v = ex_mul(v, ((i >> t) & 1) ? r : ex_sub(ex_one(), r)); }
      out[i] = dl.job_acc[j] ? ex_add(out[i], v) : v;
    }
    __syncthreads();
  }
  if (wave == 0) { wc_observe(wc, (u64)dl.nv, lane); wc_observe(wc, (u64)dl.md, lane); }  // header of the sumcheck
  const int wpt = nterm >= W ? 1 : W / nterm;
  size_t n = n0;
  bool useA = true;
  for (int round = 0; round < (int)dl.nv; round++) {
    const size_t npairs = n / 2;
    for (int term = wave / wpt; term < nterm; term += (wpt == 1 ? W : nterm + W)) {
      const int sub = wave % wpt;
      const int k = dl.tk[term];
      GlobalPairs L;
#pragma unroll
      for (int j = 0; j < 3; j++) { int ti = dl.tt[term][j < k ? j : 0]; L.p[j] = cur[ti]; L.e[j] = cur_ext[ti] != 0; }
      Ext acc[SC_SLOTS];
      sc_accumulate<false>(k, L, (size_t)sub * 64 + lane, (size_t)wpt * 64, npairs, acc);
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
      if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
      if (wpt > 1) break;  // with several waves per term every wave owns exactly one (term, sub)
    }
    __syncthreads();
    if (wave == 0) {
      Ext rr = sc_fs_round(wc, fsl, part, dl.tk, nterm, wpt, result, round, fcs, lane);
      if (lane == 0) { chal[1] = rr.c0; chal[2] = rr.c1; }
    }
    __syncthreads();
    const Ext r = ex(chal[1], chal[2]);
    Ext* const* dst = useA ? dl.bufA : dl.bufB;
    for (int t = 0; t < ntab; t++) {
      Ext* o = dst[t];
      if (cur_ext[t]) { const Ext* q = (const Ext*)cur[t]; for (size_t i = tid; i < npairs; i += nt) o[i] = ex_lerp(q[2 * i], q[2 * i + 1], r); }
      else { const u64* q = (const u64*)cur[t]; for (size_t i = tid; i < npairs; i += nt) o[i] = ex_lerp_base(q[2 * i], q[2 * i + 1], r); }
    }
    __syncthreads();
    if (tid < ntab) { cur[tid] = dst[tid]; cur_ext[tid] = 1; }
    __syncthreads();
    n = npairs; useA = !useA;
  }
  if (wave == 0) {
    const size_t wf = (size_t)dl.nv * (dl.md + 2) * 2;
    for (int e = lane; e < ntab; e += 64) {
      Ext v = ((const Ext*)cur[e])[0];
      size_t w = wf + 2 * (size_t)e;
      pub_store(result + w, v.c0); pub_store(result + w + 1, v.c1);
      fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
    }
    u64* rs = result + wf + 2 * (size_t)ntab;
    if (lane < 8) { pub_store(rs + lane, wc.st); fcs += (unsigned long long)(lane + 1) * wc.st; }
    if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rs + 8 + lane, v); fcs += (unsigned long long)(8 + lane + 1) * v; }
    if (lane == 0) {
      u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
      pub_store(rs + 12, a); pub_store(rs + 13, b);
      fcs += (unsigned long long)13 * a + (unsigned long long)14 * b;
    }
    fcs = pub_wave_sum(fcs);
    wc_finish(wc, lane);  // (host sponge: the transcript is complete before this message is)
    if (lane == 0) pub_store((u64*)flag, wc.failed ? ~0ull : pub_mix(seq) + fcs);
  }
}

// ------------------------------------------------------------------------------------------------ Basefold commit-phase tail
// Dev::commit_tail (dev.h, commit_tail.h): the last rounds of the Basefold commit phase (commit_rounds of pcs.h) — per round the
// pending sumcheck message absorbed, the folding challenge, the merge of the committed codewords of the oracle's size, the FRI
// fold (k_fri_fold's formula), the fold of the sumcheck pairs, the next message (k_bf_msg's sums), the Merkle tree of the folded
// oracle (k_merkle_tail's layer loop) and its root absorbed; in the last round the final message absorbed — in ONE launch of
// one workgroup once the oracle is short. Default in throughput mode (DP_DEVICE_COMMIT=0 turns it off); checked on the SIMT
// emulator of tests/ and on MI355X (tests/test_gpu_fused.py).
KBODY k_commit_tail(const CommitTailDesc* dp, u64* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ CommitTailDesc dl;
  __shared__ Ext part[64 * 3];
  __shared__ unsigned long long chal[3];
  __shared__ Ext s_last[3];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (int)(sizeof(CommitTailDesc) / 8); i += nt) ((u64*)&dl)[i] = ((const u64*)dp)[i];
  __syncthreads();
  if (tid < 3) s_last[tid] = dl.last[tid];
  WaveChallenger wc;
  wc.st = dl.state[lane & 7]; wc.ib = dl.in_buf[lane & 3]; wc.in_len = dl.in_len; wc.out_len = dl.out_len;
  wc_host_init(wc, dl.sp_req, dl.sp_rep, dl.sp_seq);
  unsigned long long fcs = 0;
  const Ext* prev = dl.folded;   // the folded oracle of the previous round
  const Ext* eq = dl.eq;
  const Ext* f = dl.f;
  size_t m = dl.m;
  bool useA = true;
  __syncthreads();
  for (int j = 0; j < dl.rounds; j++) {
    const size_t n = (size_t)dl.n >> j, half = n / 2;
    const bool final_round = j == dl.rounds - 1;
    // transcript: the pending message, then the folding challenge
    if (wave == 0) {
      for (int q = 0; q < 3; q++) { Ext v = s_last[q]; wc_observe(wc, v.c0, lane); wc_observe(wc, v.c1, lane); }
      wc_observe(wc, dl.lab[0], lane); wc_observe(wc, dl.lab[1], lane);
      const u64 r0 = wc_sample(wc, lane), r1 = wc_sample(wc, lane);
      if (lane == 0) { chal[1] = r0; chal[2] = r1; }
    }
    __syncthreads();
    const Ext c = ex(chal[1], chal[2]);
    // the committed codewords as long as the oracle join it (into a fresh buffer: `prev` is the leaf array of a tree)
    const Ext* src = prev;
    if (dl.nmerge[j] > 0) {
      Ext* run = dl.run[j];
      for (size_t i = tid; i < n; i += nt) {
        Ext acc = prev[i];
        for (int k = 0; k < dl.nmerge[j]; k++) {
          const Ext co = dl.mcoeff[j][k];
          acc = ex_add(acc, dl.mext[j][k] ? ex_mul(((const Ext*)dl.mcw[j][k])[i], co) : ex_mul_base(co, ((const u64*)dl.mcw[j][k])[i]));
        }
        run[i] = acc;
      }
      src = run;
      __syncthreads();
    }
    // FRI fold (K9): out[i] = y0 + (c - x0)(y1 - y0) w, x0 = gamma * w_{2^(level+1)}^{bitrev(i)}, w = -1/(2 x0); the last round's
    // folded oracle is never used
    if (!final_round) {
      const unsigned level = dl.level[j], L = dl.L;
      Ext* out = dl.leaves[j];
      for (size_t i = tid; i < half; i += nt) {
        size_t b = level ? (size_t)(__brevll((unsigned long long)i) >> (64 - level)) : 0;
        u64 root = dl.tw[b << (L - level)];
        u64 x0 = gl_mul(root, dl.gamma[j]);
        u64 rinv = b == 0 ? 1 : gl_neg(dl.tw[((size_t(1) << level) - b) << (L - level)]);
        u64 w = gl_mul(dl.ninv[j], rinv);
        Ext y0 = src[2 * i], y1 = src[2 * i + 1];
        Ext t = ex_mul(ex(gl_sub(c.c0, x0), c.c1), ex_sub(y1, y0));
        out[i] = ex_add(y0, ex_mul_base(t, w));
      }
    }
    // the sumcheck pairs fold with the same challenge
    Ext* eqd = useA ? dl.eqA : dl.eqB;
    Ext* fd = useA ? dl.fA : dl.fB;
    for (size_t i = tid; i < m / 2; i += nt) { eqd[i] = ex_lerp(eq[2 * i], eq[2 * i + 1], c); fd[i] = ex_lerp(f[2 * i], f[2 * i + 1], c); }
    __syncthreads();
    eq = eqd; f = fd; m /= 2; useA = !useA;
    if (!final_round) {
      // the next message (K10 on evaluation-form pairs): [sum a ea, sum (b ea + a eb), sum b eb]; a single value: three times it
      Ext c0 = ex_zero(), c1 = ex_zero(), c2 = ex_zero();
      for (size_t q = tid; q < m / 2; q += nt) {
        Ext a = f[2 * q], b = ex_sub(f[2 * q + 1], a), ea = eq[2 * q], eb = ex_sub(eq[2 * q + 1], ea);
        c0 = ex_add(c0, ex_mul(a, ea));
        c1 = ex_add(c1, ex_add(ex_mul(b, ea), ex_mul(a, eb)));
        c2 = ex_add(c2, ex_mul(b, eb));
      }
      c0 = wave_reduce_ext(c0); c1 = wave_reduce_ext(c1); c2 = wave_reduce_ext(c2);
      if (lane == 0) { part[wave * 3] = c0; part[wave * 3 + 1] = c1; part[wave * 3 + 2] = c2; }
      // layer 0 of the tree: leaf pairs packed, no hashing
      u64* nd = dl.nodes[j];
      const Ext* lv = dl.leaves[j];
      for (size_t i = tid; i < half / 2; i += nt) { Ext a = lv[2 * i], b = lv[2 * i + 1]; u64* o = nd + 4 * i; o[0] = a.c0; o[1] = a.c1; o[2] = b.c0; o[3] = b.c1; }
      __syncthreads();
      if (wave == 0) {
        Ext v0 = lane < W ? part[lane * 3] : ex_zero(), v1 = lane < W ? part[lane * 3 + 1] : ex_zero(), v2 = lane < W ? part[lane * 3 + 2] : ex_zero();
        v0 = wave_reduce_ext(v0); v1 = wave_reduce_ext(v1); v2 = wave_reduce_ext(v2);
        if (lane == 0) {
          if (m == 1) { s_last[0] = f[0]; s_last[1] = f[0]; s_last[2] = f[0]; }
          else { s_last[0] = v0; s_last[1] = v1; s_last[2] = v2; }
        }
      }
      // upper layers: Poseidon2 compress, one node per lane while the layer is wide, 8 lanes per node when it is narrow
      size_t off = 0, cnt = half / 2;
      while (cnt > 1) {
        const size_t next = cnt / 2;
        const u64* in = nd + 4 * off;
        u64* out = nd + 4 * (off + cnt);
        if (next > (size_t)nt) {
          for (size_t i = tid; i < next; i += nt) {
            u64 o[4];
            poseidon2_compress(in + 8 * i, in + 8 * i + 4, o, c_rc);
            out[4 * i] = o[0]; out[4 * i + 1] = o[1]; out[4 * i + 2] = o[2]; out[4 * i + 3] = o[3];
          }
        } else {
          // every lane takes part in the permutation (no divergence around the cross-lane moves); idle groups redo the last node
          const size_t groups = (size_t)nt >> 3;
          for (size_t g0 = 0; g0 < next; g0 += groups) {
            const size_t g = g0 + ((size_t)tid >> 3);
            const size_t gg = g < next ? g : next - 1;
            const int i8 = lane & 7;
            u64 sv = i8 < 4 ? in[8 * gg + i8] : 0;
            sv = p2l_permute(sv, lane);
            if (i8 < 4) sv = in[8 * gg + 4 + i8];
            sv = p2l_permute(sv, lane);
            if (g < next && i8 < 4) out[4 * g + (3 - i8)] = sv;
          }
        }
        __syncthreads();
        off += cnt; cnt = next;
      }
      // the message and the root: published, and the root absorbed (the message is absorbed at the top of the next round)
      if (wave == 0) {
        const u64* root = nd + 4 * (half - 2);
        for (int q = 0; q < 4; q++) wc_observe(wc, root[q], lane);
        if (lane == 0) {
          u64* rw = result + (size_t)j * 10;
          for (int q = 0; q < 3; q++) {
            Ext v = s_last[q];
            size_t w = (size_t)j * 10 + 2 * q;
            pub_store(rw + 2 * q, v.c0); pub_store(rw + 2 * q + 1, v.c1);
            fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
          }
          for (int q = 0; q < 4; q++) { size_t w = (size_t)j * 10 + 6 + q; pub_store(rw + 6 + q, root[q]); fcs += (unsigned long long)(w + 1) * root[q]; }
        }
      }
      __syncthreads();
      prev = dl.leaves[j];
    } else {
      // the final message: the folded sum_evals in bit-reversed index order, absorbed in its natural order
      if (wave == 0) {
        unsigned lg = 0; while ((size_t(1) << lg) < m) lg++;
        u64* rw = result + (size_t)(dl.rounds - 1) * 10;
        for (size_t r = 0; r < m; r++) {
          size_t src_i = lg ? (size_t)(__brevll((unsigned long long)r) >> (64 - lg)) : 0;
          Ext v = f[src_i];
          wc_observe(wc, v.c0, lane); wc_observe(wc, v.c1, lane);
          if (lane == 0) {
            size_t w = (size_t)(dl.rounds - 1) * 10 + 2 * r;
            pub_store(rw + 2 * r, v.c0); pub_store(rw + 2 * r + 1, v.c1);
            fcs += (unsigned long long)(w + 1) * v.c0 + (unsigned long long)(w + 2) * v.c1;
          }
        }
      }
    }
  }
  if (wave == 0) {  // the sponge goes back to the host transcript; the tag closes the message
    u64* rs = result + (size_t)(dl.rounds - 1) * 10 + 2 * m;
    if (lane < 8) { pub_store(rs + lane, wc.st); fcs += (unsigned long long)(lane + 1) * wc.st; }
    if (lane < 4) { u64 v = lane < wc.in_len ? wc.ib : 0; pub_store(rs + 8 + lane, v); fcs += (unsigned long long)(8 + lane + 1) * v; }
    if (lane == 0) {
      u64 a = (u64)wc.in_len, b = (u64)wc.out_len;
      pub_store(rs + 12, a); pub_store(rs + 13, b);
      fcs += (unsigned long long)13 * a + (unsigned long long)14 * b;
    }
    fcs = pub_wave_sum(fcs);
    wc_finish(wc, lane);  // (host sponge: the transcript is complete before this message is)
    if (lane == 0) pub_store((u64*)flag, wc.failed ? ~0ull : pub_mix(seq) + fcs);
  }
}

// Same protocol as k_sc_persist, but the tables live in LDS after the first fold (bit-reversed index order, so a fold
// pairs positions q and q + m/2 and is done in place with no hazards): after the first round no table byte touches
// global memory again. Dynamic LDS = ntabs * (n0/2) extension elements.
// ---- device -> host publication without fences -------------------------------------------------------------------
// A system-scope release makes the wave wait for an L2 write-back; with many proofs in flight on one GPU that stalls
// every round behind other proofs' dirty lines. Instead every payload word goes out as a relaxed system-scope store
// (the result area is fine-grained host memory, nothing is cached) and the flag word carries a TAG that binds the
// sequence number to the payload: tag = mix(seq) + sum_i (i+1) * word_i. The host accepts a message only when the tag
// it recomputes from the words it reads matches, so any reordering of the posted writes is harmless.
__device__ __forceinline__ void sc_publish(Ext* result, const Ext* part, const int* tk, const int* toff, int nterms, int wpt, unsigned long long* flag, unsigned long long seq, int lane) {
  unsigned long long cs = 0;
  u64* rw = (u64*)result;
  for (int e = lane; e < nterms * SC_SLOTS; e += 64) {
    int term = e / SC_SLOTS, t = e - term * SC_SLOTS;
    if (t > tk[term]) continue;  // a degree-k term publishes k + 1 values, packed back to back
    Ext v = ex_zero();
    if (wpt == 1) v = part[(size_t)term * SC_SLOTS + t];
    else for (int sb = 0; sb < wpt; sb++) v = ex_add(v, part[(size_t)(term * wpt + sb) * SC_SLOTS + t]);
    int o = toff[term] + t;
    pub_store(rw + 2 * o, v.c0); pub_store(rw + 2 * o + 1, v.c1);
    cs += (unsigned long long)(2 * o + 1) * v.c0 + (unsigned long long)(2 * o + 2) * v.c1;
  }
  cs = pub_wave_sum(cs);
  if (lane == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
}
// publish `n` extension values held in `src` (LDS or global) by one wave
__device__ __forceinline__ void sc_publish_vals(Ext* result, const Ext* src, size_t stride, int n, unsigned long long* flag, unsigned long long seq, int lane) {
  unsigned long long cs = 0;
  u64* rw = (u64*)result;
  for (int e = lane; e < n; e += 64) {
    Ext v = src[(size_t)e * stride];
    pub_store(rw + 2 * e, v.c0); pub_store(rw + 2 * e + 1, v.c1);
    cs += (unsigned long long)(2 * e + 1) * v.c0 + (unsigned long long)(2 * e + 2) * v.c1;
  }
  cs = pub_wave_sum(cs);
  if (lane == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
}
// lane 0 of wave 0: poll the host mailbox [seq, c0, c1, tag] for `seq`, leave {ok, c0, c1} in chal[]. The seq word is polled
// with relaxed loads; once it matches, an acquire fence orders the payload loads after it, and the tag word — mix(seq) + c0 +
// 2*c1, written by the host with the payload (post_challenge) — is checked against what was read: a stale or torn payload is
// re-polled instead of becoming a wrong challenge (same protocol as the device-to-host direction, wait_flag). The wait is
// bounded in TIME (c_poll_timeout_ticks of the 100 MHz s_memrealtime clock; DP_POLL_TIMEOUT_S, default 20 s: below the host's
// own 30 s), not in spins: a host that serves hundreds of proofs from a few threads is slow, not gone.
__device__ __forceinline__ unsigned long long chal_mix(unsigned long long seq) { return seq * 0x9E3779B97F4A7C15ull + 0x7F4A7C159E3779B9ull; }
__device__ __forceinline__ void sc_wait_challenge(const unsigned long long* mailbox, unsigned long long seq, unsigned long long* chal) {
  const unsigned long long t0 = dp_realtime();
  bool ok = false;
  unsigned long long c0 = 0, c1 = 0;
  for (unsigned spin = 0;; spin++) {
    unsigned long long got = __hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (got == seq) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      c0 = __hip_atomic_load(mailbox + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      c1 = __hip_atomic_load(mailbox + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      unsigned long long tag = __hip_atomic_load(mailbox + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tag == chal_mix(seq) + c0 + 2 * c1) { ok = true; break; }
    }
    if ((spin & 63) == 63 && dp_realtime() - t0 > c_poll_timeout_ticks) break;
    for (int q = 0; q < c_poll_sleep; q++) __builtin_amdgcn_s_sleep(4);  // every poll is a PCIe read: keep the rate of all kernels in flight bounded
  }
  chal[0] = ok ? 1 : 0; chal[1] = c0; chal[2] = c1;
}
__device__ void sc_publish_fwd(Ext* result, const Ext* part, const int* tk, const int* toff, int nterms, int wpt, unsigned long long* flag, unsigned long long seq, int lane) { sc_publish(result, part, tk, toff, nterms, wpt, flag, seq, lane); }
__device__ void sc_publish_vals_fwd(Ext* result, const Ext* src, size_t stride, int n, unsigned long long* flag, unsigned long long seq, int lane) { sc_publish_vals(result, src, stride, n, flag, seq, lane); }
__device__ void sc_wait_challenge_fwd(const unsigned long long* mailbox, unsigned long long seq, unsigned long long* chal) { sc_wait_challenge(mailbox, seq, chal); }
template <bool HI>
KBODY k_sc_persist_lds(const ScPersistArgs& a, Ext* result, unsigned long long* flag, const unsigned long long* mailbox, unsigned long long seq0, const ScFsArgs* fs) {
  extern __shared__ __align__(16) unsigned char lds_dyn[];
  Ext* L = (Ext*)lds_dyn;
  __shared__ Ext part[64 * SC_SLOTS];
  __shared__ unsigned long long chal[3];
  __shared__ ScFsArgs fsl;
  int tid = threadIdx.x, nt = blockDim.x;
  int W = nt >> 6, wave = tid >> 6, lane = tid & 63;
  int wpt = a.nterms >= W ? 1 : W / a.nterms;
  const bool autofs = fs != nullptr;  // device-side Fiat-Shamir: no host round trips
  if (autofs) { for (int i = tid; i < (int)(sizeof(ScFsArgs) / 8); i += nt) ((u64*)&fsl)[i] = ((const u64*)fs)[i]; __syncthreads(); }
  WaveChallenger wc; wc.st = wc.ib = 0; wc.in_len = wc.out_len = 0;
  if (autofs) wc_load(wc, fsl, lane);
  unsigned long long fcs = 0; int round = 0;
  if (a.eq_tab >= 0) wg_build_eq((Ext*)a.in[a.eq_tab], a.eq_pt, a.eq_k);
  unsigned long long seq = seq0;
  size_t first = a.n0 / 2;
  unsigned lgf = 0; while ((size_t(1) << lgf) < first) lgf++;
  Ext r = a.r0;
  unsigned long long c_fold = 0, c_sums = 0, c_pub = 0, c_wait = 0, c_rounds = 0, tk = clock64();
  if (!a.has_r0) {
    // round on the tables as they sit in global memory
    size_t npairs = a.n0 / 2;
    for (int term = wave / wpt; term < a.nterms; term += (wpt == 1 ? W : a.nterms + W)) {
      int sub = wave % wpt;
      int k = a.k[term];
      GlobalPairs L;
#pragma unroll
      for (int j = 0; j < ScW<HI>::K; j++) { int ti = a.t[term][j < k ? j : 0]; L.p[j] = a.in[ti]; L.e[j] = a.in_ext[ti]; }
      Ext acc[SC_SLOTS];
      sc_accumulate<HI>(k, L, (size_t)sub * 64 + lane, (size_t)wpt * 64, npairs, acc);
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
      if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
      if (wpt > 1) break;
    }
    __syncthreads();
    ++seq;
    if (wave == 0) {
      if (autofs) { Ext rr = sc_fs_round(wc, fsl, part, a.k, a.nterms, wpt, (u64*)result, round, fcs, lane); if (lane == 0) { chal[0] = 1; chal[1] = rr.c0; chal[2] = rr.c1; } }
      else { sc_publish(result, part, a.k, a.off, a.nterms, wpt, flag, seq, lane); if (lane == 0) sc_wait_challenge(mailbox, seq, chal); }
    }
    round++;
    __syncthreads();
    if (chal[0] == 0) { if (tid == 0) pub_store((u64*)flag, ~0ull); return; }
    r = ex(chal[1], chal[2]);
  }
  // first fold: global -> LDS (bit-reversed positions)
  for (size_t idx = tid; idx < (size_t)a.ntabs * first; idx += nt) {
    int t = (int)(idx >> lgf); size_t i = idx & (first - 1);
    Ext v = a.in_ext[t] ? ex_lerp(((const Ext*)a.in[t])[2 * i], ((const Ext*)a.in[t])[2 * i + 1], r)
                        : ex_lerp_base(((const u64*)a.in[t])[2 * i], ((const u64*)a.in[t])[2 * i + 1], r);
    size_t pos = lgf ? (__brev((unsigned)i) >> (32 - lgf)) : 0;
    L[((size_t)t << lgf) + pos] = v;
  }
  __syncthreads();
  size_t m = first;
  for (;;) {
    if (m == 1) {
      ++seq;
      if (tid == 0 && a.dbg) { atomicAdd(a.dbg + 0, c_fold); atomicAdd(a.dbg + 1, c_sums); atomicAdd(a.dbg + 2, c_pub); atomicAdd(a.dbg + 3, c_wait); atomicAdd(a.dbg + 4, c_rounds); }
      if (wave == 0 && autofs) sc_fs_finish(wc, fsl, nullptr, L, size_t(1) << lgf, a.ntabs, (u64*)result, flag, seq0 + 1, fcs, lane);
      else if (wave == 0) sc_publish_vals(result, L, size_t(1) << lgf, a.ntabs, flag, seq, lane);
      return;
    }
    size_t h = m / 2;
    if (tid == 0) { unsigned long long now = clock64(); c_fold += now - tk; tk = now; }
    for (int term = wave / wpt; term < a.nterms; term += (wpt == 1 ? W : a.nterms + W)) {
      int sub = wave % wpt;
      int k = a.k[term];
      LdsPairs LP; LP.h = h;
#pragma unroll
      for (int j = 0; j < ScW<HI>::K; j++) LP.p[j] = L + ((size_t)a.t[term][j < k ? j : 0] << lgf);
      Ext acc[SC_SLOTS];
      sc_accumulate<HI>(k, LP, (size_t)sub * 64 + lane, (size_t)wpt * 64, h, acc);
#pragma unroll
      for (int t = 0; t < SC_SLOTS; t++) if (t <= k) acc[t] = wave_reduce_ext(acc[t]);
      if (lane == 0) { Ext* o = part + (size_t)(term * wpt + sub) * SC_SLOTS; for (int t = 0; t < SC_SLOTS; t++) o[t] = acc[t]; }
      if (wpt > 1) break;
    }
    __syncthreads();
    ++seq;
    if (tid == 0) { unsigned long long now = clock64(); c_sums += now - tk; tk = now; }
    if (wave == 0 && autofs) {
      Ext rr = sc_fs_round(wc, fsl, part, a.k, a.nterms, wpt, (u64*)result, round, fcs, lane);
      if (lane == 0) { chal[0] = 1; chal[1] = rr.c0; chal[2] = rr.c1; }
      if (tid == 0) { unsigned long long now = clock64(); c_wait += now - tk; tk = now; c_rounds++; }
    } else if (wave == 0) {
      sc_publish(result, part, a.k, a.off, a.nterms, wpt, flag, seq, lane);
      if (tid == 0) { unsigned long long now = clock64(); c_pub += now - tk; tk = now; }
      if (lane == 0) sc_wait_challenge(mailbox, seq, chal);
      if (tid == 0) { unsigned long long now = clock64(); c_wait += now - tk; tk = now; c_rounds++; }
    }
    round++;
    __syncthreads();
    if (chal[0] == 0) { if (tid == 0) pub_store((u64*)flag, ~0ull); return; }
    r = ex(chal[1], chal[2]);
    unsigned lgh = 0; while ((size_t(1) << lgh) < h) lgh++;
    for (size_t idx = tid; idx < (size_t)a.ntabs * h; idx += nt) {
      size_t t = idx >> lgh, q = idx & (h - 1);
      Ext* p = L + (t << lgf);
      p[q] = ex_lerp(p[q], p[q + h], r);
    }
    __syncthreads();
    m = h;
  }
}

// out[e] = sum_{b < nblocks} partial[(e / inner) * nblocks * inner + b * inner + (e % inner)], e < nout, computed by ONE
// workgroup (wave w owns outputs w, w+W, ..) and published straight to host-mapped memory with the tag protocol: the
// second stage of every block-partial reduction needs neither a separate publish launch nor a stream synchronisation.
KBODY k_reduce_publish(const Ext* partial, size_t nblocks, size_t inner, int nout, Ext* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext res[1024];
  int tid = threadIdx.x, W = blockDim.x >> 6, wave = tid >> 6, lane = tid & 63;
  for (int e = wave; e < nout; e += W) {
    size_t base = ((size_t)e / inner) * nblocks * inner + ((size_t)e % inner);
    Ext acc = ex_zero();
    for (size_t b = lane; b < nblocks; b += 64) acc = ex_add(acc, partial[base + b * inner]);
    acc = wave_reduce_ext(acc);
    if (lane == 0) res[e] = acc;
  }
  __syncthreads();
  if (wave == 0) sc_publish_vals(result, res, 1, nout, flag, seq, lane);
}
// ---- work-proportional grids for batched kernels over items of very different sizes ------------------------------------
// blockIdx.y = item wastes most of a launch when one item is 2^20 long and thirty others are 2^10 (every empty workgroup
// still has to fetch its descriptor before it can leave). Instead the grid is 1-D and `first[i] .. first[i+1]` are the
// workgroups of item i (first[] sits in front of the descriptors, n + 1 entries): one coalesced load finds the owner.
__device__ __forceinline__ int seg_find(const unsigned* first, int n, int* slot) {
  if (threadIdx.x < 64) {
    for (int base = 0; base < n; base += 64) {
      int i = base + (int)threadIdx.x;
      bool mine = i < n && first[i] <= blockIdx.x && blockIdx.x < first[i + 1];
      if (mine) *slot = i;
    }
  }
  __syncthreads();
  return *slot;
}
// One round of the batch-opening sumcheck (sum_check/classic.rs:232-285, coeff.rs:198-345) over every (f, eq) pair in one
// launch: fold both tables by r (if has_r and the pair is longer than 1; the folded tables go to fout / eqout) and, in the
// same pass, the two sums of the FOLDED pair the next message needs: c0 = sum f0 e0, c2 = sum (f1 - f0)(e1 - e0).
// partial[2 * workgroup + {0,1}]; k_classic_reduce adds the workgroups of each pair and publishes.
struct ClassicDesc { const void* f; const Ext* eq; Ext* fout; Ext* eqout; size_t n; int fext; int pad; };
KBODY k_classic_fused(const unsigned* first, const ClassicDesc* d, int np, Ext r, int has_r, Ext* partial) {
  __shared__ Ext sm[TPB / 64];
  __shared__ int slot;
  const int pi = seg_find(first, np, &slot);
  const ClassicDesc p = d[pi];
  const size_t b = blockIdx.x - first[pi], nb = first[pi + 1] - first[pi];
  Ext c0 = ex_zero(), c2 = ex_zero();
  if (has_r && p.n > 1) {
    const size_t h = p.n / 2;  // length after the fold
    if (h == 1) {
      if (b == 0 && threadIdx.x == 0) {
        Ext f = p.fext ? ex_lerp(((const Ext*)p.f)[0], ((const Ext*)p.f)[1], r) : ex_lerp_base(((const u64*)p.f)[0], ((const u64*)p.f)[1], r);
        Ext e = ex_lerp(p.eq[0], p.eq[1], r);
        p.fout[0] = f; p.eqout[0] = e;
        c0 = ex_mul(f, e);
      }
    } else {
      for (size_t j = b * blockDim.x + threadIdx.x; j < h / 2; j += nb * blockDim.x) {
        Ext f0, f1;
        if (p.fext) { const Ext* q = (const Ext*)p.f + 4 * j; f0 = ex_lerp(q[0], q[1], r); f1 = ex_lerp(q[2], q[3], r); }
        else { const u64* q = (const u64*)p.f + 4 * j; f0 = ex_lerp_base(q[0], q[1], r); f1 = ex_lerp_base(q[2], q[3], r); }
        const Ext* qe = p.eq + 4 * j;
        Ext e0 = ex_lerp(qe[0], qe[1], r), e1 = ex_lerp(qe[2], qe[3], r);
        p.fout[2 * j] = f0; p.fout[2 * j + 1] = f1; p.eqout[2 * j] = e0; p.eqout[2 * j + 1] = e1;
        c0 = ex_add(c0, ex_mul(f0, e0));
        c2 = ex_add(c2, ex_mul(ex_sub(f1, f0), ex_sub(e1, e0)));
      }
    }
  } else if (p.n == 1) {
    if (b == 0 && threadIdx.x == 0) c0 = ex_mul(ld_elem(p.f, p.fext, 0), p.eq[0]);
  } else {
    for (size_t j = b * blockDim.x + threadIdx.x; j < p.n / 2; j += nb * blockDim.x) {
      Ext l0 = p.eq[2 * j], l1 = p.eq[2 * j + 1];
      if (p.fext) {
        Ext r0 = ((const Ext*)p.f)[2 * j], r1 = ((const Ext*)p.f)[2 * j + 1];
        c0 = ex_add(c0, ex_mul(l0, r0));
        c2 = ex_add(c2, ex_mul(ex_sub(l1, l0), ex_sub(r1, r0)));
      } else {
        u64 r0 = ((const u64*)p.f)[2 * j], r1 = ((const u64*)p.f)[2 * j + 1];
        c0 = ex_add(c0, ex_mul_base(l0, r0));
        c2 = ex_add(c2, ex_mul_base(ex_sub(l1, l0), gl_sub(r1, r0)));
      }
    }
  }
  Ext t;
  t = block_reduce_ext(c0, sm); if (threadIdx.x == 0) partial[2 * (size_t)blockIdx.x] = t;
  t = block_reduce_ext(c2, sm); if (threadIdx.x == 0) partial[2 * (size_t)blockIdx.x + 1] = t;
}
// out[2 i + t] = sum over the workgroups of pair i of partial[2 w + t], published to the host (np <= 512)
KBODY k_classic_reduce(const unsigned* first, int np, const Ext* partial, Ext* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext res[1024];
  int tid = threadIdx.x, W = blockDim.x >> 6, wave = tid >> 6, lane = tid & 63;
  for (int i = wave; i < np; i += W) {
    Ext a0 = ex_zero(), a2 = ex_zero();
    for (unsigned w = first[i] + lane; w < first[i + 1]; w += 64) { a0 = ex_add(a0, partial[2 * (size_t)w]); a2 = ex_add(a2, partial[2 * (size_t)w + 1]); }
    a0 = wave_reduce_ext(a0); a2 = wave_reduce_ext(a2);
    if (lane == 0) { res[2 * i] = a0; res[2 * i + 1] = a2; }
  }
  __syncthreads();
  if (wave == 0) sc_publish_vals(result, res, 1, 2 * np, flag, seq, lane);
}
struct EqDesc { Ext* out; unsigned k; unsigned pad; Ext pt[MAX_PT]; };
// many eq tables in one launch: blockIdx.y selects the table (batch_open builds one per opened polynomial)
// eq(i, pt) = lo[i mod 2^kl] * hi[i >> kl]: a workgroup owns EQ_CHUNK consecutive entries of one table, builds the 2^kl
// low-part products (kl <= 8) and its EQ_CHUNK / 2^kl high-part products in LDS, then spends ONE multiplication per entry
// instead of k. (Products in the field are exact: the entries are those of build_eq_x_r_vec bit for bit.)
constexpr unsigned EQ_CHUNK = 4096;
KBODY k_eq_table_many(const unsigned* first, const EqDesc* d, int nd) {
  __shared__ Ext lo[256];
  __shared__ Ext hi[EQ_CHUNK / 256 > 16 ? EQ_CHUNK / 256 : 16];
  __shared__ int slot;
  const int di = seg_find(first, nd, &slot);
  const EqDesc& e = d[di];
  const size_t n = size_t(1) << e.k;
  const unsigned kl = e.k < 8 ? e.k : 8;
  const size_t base = (size_t)(blockIdx.x - first[di]) * EQ_CHUNK;
  const size_t cnt = n - base < EQ_CHUNK ? n - base : EQ_CHUNK;   // entries of this workgroup (a multiple of 2^kl)
  if (threadIdx.x < (1u << kl)) {
    Ext v = ex_one();
    for (unsigned t = 0; t < kl; t++) { Ext r = e.pt[t]; v = ex_mul(v, ((threadIdx.x >> t) & 1) ? r : ex_sub(ex_one(), r)); }
    lo[threadIdx.x] = v;
  }
  const size_t nhi = cnt >> kl;
  if (threadIdx.x < nhi) {
    size_t h = (base >> kl) + threadIdx.x;
    Ext v = ex_one();
    for (unsigned t = kl; t < e.k; t++) { Ext r = e.pt[t]; v = ex_mul(v, ((h >> (t - kl)) & 1) ? r : ex_sub(ex_one(), r)); }
    hi[threadIdx.x] = v;
  }
  __syncthreads();
  for (size_t q = threadIdx.x; q < cnt; q += blockDim.x) e.out[base + q] = ex_mul(lo[q & ((size_t(1) << kl) - 1)], hi[q >> kl]);
}
struct AxpyDesc { const void* x; int xext; unsigned lg_rep; size_t n_x; Ext coeff; };
// acc[i] = init[i] (or 0) + sum_d x_d[i >> lg_rep_d] * coeff_d over all descriptors (K11: every codeword / evaluation table merged into
// the running oracle in ONE pass over acc instead of one launch and one read-modify-write of acc per polynomial)
KBODY k_axpy_many(Ext* acc, const Ext* init, size_t n_acc, const AxpyDesc* d, int nd) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_acc; i += (size_t)gridDim.x * blockDim.x) {
    Ext a = init ? init[i] : ex_zero();
    for (int q = 0; q < nd; q++) {
      size_t j = i >> d[q].lg_rep;
      Ext m = d[q].xext ? ex_mul(((const Ext*)d[q].x)[j], d[q].coeff) : ex_mul_base(d[q].coeff, ((const u64*)d[q].x)[j]);
      a = ex_add(a, m);
    }
    acc[i] = a;
  }
}
// last fold of a sumcheck, results published directly
KBODY k_finish_publish(const FoldArgs& a, Ext r, int ntabs, Ext* result, unsigned long long* flag, unsigned long long seq) {
  __shared__ Ext res[MAX_TABS];
  int t = threadIdx.x;
  if (t < ntabs) {
    Ext v;
    if (a.ext[t]) { const Ext* p = (const Ext*)a.in[t]; v = ex_lerp(p[0], p[1], r); }
    else { const u64* p = (const u64*)a.in[t]; v = ex_lerp_base(p[0], p[1], r); }
    res[t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) sc_publish_vals(result, res, 1, ntabs, flag, seq, threadIdx.x);
}

// copy a small device result into host-mapped memory and publish it
// (launched with ONE wave so that payload stores and the releasing flag store come from the same wave)
KBODY k_publish(const u64* src, u64* dst, size_t nwords, unsigned long long* flag, unsigned long long seq) {
  unsigned long long cs = 0;
  for (size_t i = threadIdx.x; i < nwords; i += blockDim.x) { u64 v = src[i]; pub_store(dst + i, v); cs += (unsigned long long)(i + 1) * v; }
  cs = pub_wave_sum(cs);
  if (threadIdx.x == 0) pub_store((u64*)flag, pub_mix(seq) + cs);
}

// ================================================================================================ HipDev
// Grid size for grid-stride kernels. DP_MAX_GRID bounds every launch so that, with several proofs in flight on one GPU,
// a large kernel of one proof cannot occupy every wave slot while another proof's latency-critical one-block kernels
// wait for a CU.
static int g_max_grid = [] { const char* e = getenv("DP_MAX_GRID"); return e ? atoi(e) : 1 << 30; }();
static inline int grid_for(size_t n, int cap = 2048) {
  size_t b = (n + TPB - 1) / TPB;
  if (b < 1) b = 1;
  return (int)std::min<size_t>(std::min<size_t>(b, cap), (size_t)g_max_grid);
}

struct ProfRec { const char* name; double bytes; hipEvent_t a, b; };
// every launch of this file goes through HipDev::launch_: kg<Body> on the context's own stream, or — when the context is a
// member of a cohort — an argument pack handed to the cohort, which launches kc<Body> once for all its members
#define DPL_B(kern, maxt, flags, grid, block, lds, ...) do { prof_begin(#kern); LaunchTimer lt_(this); launch_<kern, maxt, flags>(KArgs<decltype(&kern)>(), #kern, grid, block, lds, __VA_ARGS__); lt_.stop(); prof_end(); } while (0)
#define DPL(kern, grid, block, ...) DPL_B(kern, 1024, KF_NONE, grid, block, 0, __VA_ARGS__)
#define DPL_LDS(kern, grid, block, lds, ...) DPL_B(kern, 1024, KF_NONE, grid, block, lds, __VA_ARGS__)
// one-workgroup-per-proof kernels (persistent sumchecks, fused protocol tails, Merkle tails): `threads` and the CU reservation
// apply in latency mode; in throughput mode the workgroup shrinks to shared_threads_ and reserves nothing (KF_PRIO above).
// `lds` = dynamic LDS the body really needs.
#define DPL_ONE(kern, grid, threads, lds, ...) do { if (shared_now()) { DPL_B(kern, 1024, KF_PRIO, grid, dim3(std::min<unsigned>((unsigned)(threads), (unsigned)shared_threads_)), (size_t)(lds), __VA_ARGS__); } \
                                                     else { DPL_B(kern, 1024, KF_CLAIM, grid, dim3(threads), std::max<size_t>((size_t)(lds), excl_now()), __VA_ARGS__); } } while (0)
#define DPL_ONE_HI(kern, hi, grid, threads, lds, ...) do { if (hi) { DPL_ONE((kern<true>), grid, threads, lds, __VA_ARGS__); } else { DPL_ONE((kern<false>), grid, threads, lds, __VA_ARGS__); } } while (0)
#define DPL_HI(kern, hi, grid, block, ...) do { if (hi) { DPL((kern<true>), grid, block, __VA_ARGS__); } else { DPL((kern<false>), grid, block, __VA_ARGS__); } } while (0)
#define DPL_LDS_HI(kern, hi, grid, block, lds, ...) do { if (hi) { DPL_LDS((kern<true>), grid, block, lds, __VA_ARGS__); } else { DPL_LDS((kern<false>), grid, block, lds, __VA_ARGS__); } } while (0)
#define DP_SET_LDS(kern, maxt, bytes) set_lds_<kern, maxt, KF_NONE>(KArgs<decltype(&kern)>(), (int)(bytes))
#define DP_SET_LDS_ONE(kern, maxt, bytes) do { set_lds_<kern, maxt, KF_CLAIM>(KArgs<decltype(&kern)>(), (int)(bytes)); set_lds_<kern, maxt, KF_PRIO>(KArgs<decltype(&kern)>(), (int)(bytes)); } while (0)

static const bool g_host_stats = getenv("DP_TIMING") && atoi(getenv("DP_TIMING"));

// ------------------------------------------------------------------------------------------------ cohorts
// A cohort is a set of proofs of the SAME model proved in lock step on one stream by one host thread (each proof a fiber,
// fiber.h). Proofs of one model issue the same sequence of launches with the same shapes — only pointers and challenges
// differ — so launch number i of every member is merged into ONE kc<Body> launch with gridDim.z = members: the per-launch
// costs of the command processor (dispatch, barrier, end-of-kernel cache maintenance — what bounds a GPU that serves two
// dozen independent streams of tiny kernels) are paid once per cohort step instead of once per proof step.
// A member never blocks at a launch: it drops its argument pack and goes on to its next wait (where it yields to the next
// member); whoever completes a launch's set of packs fires it. Everything is driven from the cohort's one host thread.
struct Cohort {
  struct Pending {
    void (*fire)(const Pending&, hipStream_t);  // also the identity of the kernel (one instantiation per Body)
    const char* name;
    dim3 g, b; size_t lds, pack_bytes;
    char* packs; const char* packs_dev;
    int count, expected;
    size_t ring_begin, ring_end;
  };
  hipStream_t s = nullptr;
  int members = 0;  // proofs currently in the cohort
  char* ring = nullptr; const char* ring_dev = nullptr;
  size_t ring_cap = 0, ring_off = 0;
  std::deque<Pending> q; size_t q_base = 0;            // q[i] = launch number q_base + i of the common sequence, not yet fired
  std::deque<std::pair<size_t, size_t>> inflight;      // (launch number, ring_begin) of fired launches not known to have run
  size_t executed = 0;                                 // every launch number < executed has run to completion
  size_t nfired = 0, npacks = 0;
  // DP_TIMING: where a cohort's wall time goes — "device phase" = from a fire to the first member that sees a result afterwards,
  // "host phase" = from that wake-up to the next fire (the members digest the result one after the other on the cohort's
  // thread; nothing of this cohort is queued on the GPU meanwhile)
  std::chrono::steady_clock::time_point t_fire_{}, t_wake_{}; bool awake_ = false, have_fire_ = false;
  double dev_phase_us = 0, host_phase_us = 0; size_t nwakes = 0;
  void note_wake() {
    if (awake_ || !have_fire_) return;
    awake_ = true; t_wake_ = std::chrono::steady_clock::now(); nwakes++;
    dev_phase_us += std::chrono::duration<double, std::micro>(t_wake_ - t_fire_).count();
  }
  void note_fire() {
    auto t = std::chrono::steady_clock::now();
    if (awake_) { host_phase_us += std::chrono::duration<double, std::micro>(t - t_wake_).count(); awake_ = false; }
    t_fire_ = t; have_fire_ = true;
  }

  // DP_COHORT_XCD=1 (experiment, default off): the cohort's stream is confined to ONE XCD (32 CUs, its own L2) with a CU
  // mask, cohorts dealt round robin over the 8 XCDs — every kernel of a proof then runs under one coherent L2 and cohorts on
  // different XCDs cannot take each other's CUs. Mask bit i of a multi-XCD device addresses XCD i % 8, CU i / 8.
  explicit Cohort(size_t ring_bytes = size_t(32) << 20) : ring_cap(ring_bytes) {
    static std::atomic<unsigned> next_xcd{0};
    const char* xe = getenv("DP_COHORT_XCD");
    if (xe && atoi(xe)) {
      unsigned x = next_xcd.fetch_add(1) % 8;
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (unsigned bit = x; bit < 256; bit += 8) mask[bit / 32] |= 1u << (bit % 32);
      HIP_CHECK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    } else
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_CHECK(hipHostMalloc((void**)&ring, ring_cap, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostGetDevicePointer((void**)&ring_dev, ring, 0));
  }
  ~Cohort() { if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); } if (ring) hipHostFree(ring); }
  Cohort(const Cohort&) = delete;
  Cohort& operator=(const Cohort&) = delete;

  // ring space for `bytes` (virtual offsets grow monotonically; physical = virtual mod capacity, regions never straddle
  // the end), never overlapping a region a launch may still read: those of unfired launches and of fired launches not yet
  // known to have run
  size_t ring_take(size_t bytes) {
    bytes = (bytes + 63) & ~size_t(63);
    while (!inflight.empty() && inflight.front().first < executed) inflight.pop_front();
    size_t head = ring_off;
    if (head % ring_cap + bytes > ring_cap) head += ring_cap - head % ring_cap;
    size_t tail = !inflight.empty() ? inflight.front().second : !q.empty() ? q.front().ring_begin : head;
    if (bytes > ring_cap || head + bytes - tail > ring_cap) throw DpError(DP_ERR_OOM, "cohort argument ring exhausted (DP_COHORT_RING_BYTES)");
    ring_off = head + bytes;
    return head;
  }
  // member `li`-th launch of its sequence: add its pack to launch number `li`, fire every launch whose set is complete
  void submit(size_t li, void (*fire)(const Pending&, hipStream_t), const char* name, dim3 g, dim3 b, size_t lds, const void* pack, size_t pack_bytes) {
    if (li < q_base) throw DpError(DP_ERR_SHAPE, std::string("cohort out of step: a member reached launch ") + name + " after it was fired (the proofs of a cohort must issue identical launch sequences)");
    size_t k = li - q_base;
    if (k > q.size()) throw DpError(DP_ERR_SHAPE, "cohort out of step: launch sequence gap");
    if (k == q.size()) {
      Pending p; p.fire = fire; p.name = name; p.g = g; p.b = b; p.lds = lds; p.pack_bytes = pack_bytes; p.count = 0; p.expected = members;
      p.ring_begin = ring_take(pack_bytes * (size_t)members); p.ring_end = ring_off;
      p.packs = ring + p.ring_begin % ring_cap; p.packs_dev = ring_dev + p.ring_begin % ring_cap;
      q.push_back(p);
    }
    Pending& p = q[k];
    if (p.fire != fire || p.g.x != g.x || p.g.y != g.y || p.b.x != b.x || p.lds != lds || p.pack_bytes != pack_bytes)
      throw DpError(DP_ERR_SHAPE, std::string("cohort out of step: launch ") + name + " of one member meets " + p.name + " of another (the proofs of a cohort must issue identical launch sequences)");
    if (p.count >= p.expected) throw DpError(DP_ERR_SHAPE, "cohort out of step: too many packs for one launch");
    memcpy(p.packs + (size_t)p.count * pack_bytes, pack, pack_bytes);
    p.count++; npacks++;
    flush();
  }
  void flush() {
    while (!q.empty() && q.front().count >= q.front().expected) {
      Pending& p = q.front();
      if (p.count > 0) {
        std::atomic_thread_fence(std::memory_order_release);
        if (g_host_stats) note_fire();
        p.fire(p, s);
        inflight.push_back({q_base, p.ring_begin});
        nfired++;
      }
      q.pop_front(); q_base++;
    }
  }
  // a member has seen the result of its launch number `li` (or of a later round of it): everything before it has run
  void note_executed(size_t li) { if (li > executed) executed = li; }
  void join() { if (!q.empty()) throw DpError(DP_ERR_SHAPE, "a proof cannot join a cohort in the middle of a step"); members++; }
  // a member leaves (its proofs are done, or it failed) having issued `li` launches: later launches no longer wait for it
  void leave(size_t li) {
    members--;
    for (size_t k = li > q_base ? li - q_base : 0; k < q.size(); k++) q[k].expected--;
    flush();
  }
  void drain() { HIP_CHECK(hipStreamSynchronize(s)); inflight.clear(); executed = q_base; }
};


class HipDev : public Dev {
  // host-side cost accounting (DP_TIMING=1): time inside hipLaunchKernel and number of launches / device waits
  double launch_us_ = 0; size_t nlaunch_ = 0, nwait_ = 0, nyield_ = 0;
  // host time between two device waits (the proof's own host work: no yield happens there) and time from the first poll
  // of a wait to its success (device latency + the other fibers of this thread)
  struct Chunk { size_t wait; double us; const char* first; const char* last; };
  std::vector<Chunk> chunks_; const char* first_launch_ = nullptr; const char* last_launch_ = nullptr;
  double work_us_ = 0, waitlat_us_ = 0; std::chrono::steady_clock::time_point last_exit_{}; bool have_exit_ = false;
  std::chrono::steady_clock::time_point wait_enter_() {
    auto t = std::chrono::steady_clock::now();
    if (g_host_stats && have_exit_) {
      double c = std::chrono::duration<double, std::micro>(t - last_exit_).count();
      work_us_ += c;
      if (c > 150.0 && chunks_.size() < 400) chunks_.push_back({nwait_, c, first_launch_, last_launch_});
    }
    first_launch_ = nullptr;
    return t;
  }
  void wait_exit_(std::chrono::steady_clock::time_point t0) {
    if (!g_host_stats) return;
    last_exit_ = std::chrono::steady_clock::now(); have_exit_ = true;
    waitlat_us_ += std::chrono::duration<double, std::micro>(last_exit_ - t0).count();
    if (co_) co_->note_wake();
  }
  std::map<const char*, size_t> by_name_;  // (keys: the string literals of the launch macros)
  struct LaunchTimer {
    HipDev* d; std::chrono::steady_clock::time_point t0;
    explicit LaunchTimer(HipDev* d_) : d(d_) { if (g_host_stats) t0 = std::chrono::steady_clock::now(); }
    void stop() { if (g_host_stats) { d->launch_us_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); d->nlaunch_++; } }
  };
  int device_;
  bool prof_ = false;
  double nb_ = 0;  // algorithmic bytes of the next launch (SURVEY.md 8d ledger), consumed by prof_begin
  std::vector<ProfRec> recs_;
  std::vector<hipEvent_t> prof_pool_;  // events of earlier profiled runs, reused: creating two events per launch between the launches is what a cold profiled run paid for
  hipEvent_t prof_event_() { hipEvent_t e; if (!prof_pool_.empty()) { e = prof_pool_.back(); prof_pool_.pop_back(); return e; } hipEventCreate(&e); return e; }
  void prof_begin(const char* name) {
    if (!prof_) { nb_ = 0; return; }
    ProfRec r; r.name = name; r.bytes = nb_; nb_ = 0;
    r.a = prof_event_(); r.b = prof_event_();
    hipEventRecord(r.a, s_);
    recs_.push_back(r);
  }
  void prof_end() { if (prof_) hipEventRecord(recs_.back().b, s_); }

  hipStream_t s_ = nullptr;
  Cohort* co_ = nullptr;  // non-null while this context proves as a member of a cohort: launches go to the cohort's stream
  size_t co_li_ = 0;      // number of launches this member has issued into the cohort's common sequence
  template <auto Body, int MAXT, int FLAGS, class... A>
  static void fire_(const Cohort::Pending& p, hipStream_t s) {
    hipLaunchKernelGGL((kc<Body, MAXT, FLAGS, std::decay_t<A>...>), dim3(p.g.x, p.g.y, (unsigned)p.count), p.b, p.lds, s, (const ArgPack<std::decay_t<A>...>*)p.packs_dev);
  }
  template <auto Body, int MAXT, int FLAGS, class... A, class... P>
  void launch_(KArgs<void (*)(A...)>, const char* name, dim3 g, dim3 b, size_t lds, P... args) {
    static_assert(sizeof...(A) == sizeof...(P), "kernel argument count");
    if (g_host_stats) { if (!first_launch_) first_launch_ = name; last_launch_ = name; by_name_[name]++; }
    if (!co_) { hipLaunchKernelGGL((kg<Body, MAXT, FLAGS, std::decay_t<A>...>), g, b, lds, s_, static_cast<std::decay_t<A>>(args)...); return; }
    DP_REQUIRE(g.z == 1, DP_ERR_SHAPE, "cohort launches use blockIdx.z for the proof");
    using Pack = ArgPack<std::decay_t<A>...>;
    static_assert(std::is_trivially_copyable<Pack>::value && std::is_trivially_destructible<Pack>::value, "argument packs travel as bytes");
    Pack pk(static_cast<std::decay_t<A>>(args)...);
    co_->submit(co_li_++, &fire_<Body, MAXT, FLAGS, A...>, name, g, b, lds, &pk, sizeof(Pack));
  }
  template <auto Body, int MAXT, int FLAGS, class... A>
  static void set_lds_(KArgs<void (*)(A...)>, int bytes) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kg<Body, MAXT, FLAGS, std::decay_t<A>...>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)kc<Body, MAXT, FLAGS, std::decay_t<A>...>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  }
  char* arena_ = nullptr;
  size_t arena_cap_ = 0, arena_off_ = 0, arena_peak_ = 0;
  u64* hres_ = nullptr;   // pinned, device-mapped host memory for small results (zero-copy readback)
  u64* hres_dev_ = nullptr;  // device view of hres_
  unsigned long long* hflag_ = nullptr;      // host view of the publish sequence number
  unsigned long long* hflag_dev_ = nullptr;  // device view
  unsigned long long seq_ = 0;
  unsigned long long last_tag_ = 0;
  bool zerocopy_ = true;  // DP_NO_ZEROCOPY=1 falls back to hipMemcpyAsync + hipStreamSynchronize
  bool persist_ = true;   // DP_NO_PERSIST=1 disables the persistent sumcheck kernel
  size_t excl_ = 0;       // dynamic LDS requested by one-workgroup kernels to keep a CU to themselves (DP_NO_EXCLUSIVE_CU=1: none)
  // Experiment knobs for the cohort regime (defaults = the measured configuration): with 192 proofs in flight
  // the exclusive workgroups hold ~100 CUs on average and the batched Merkle tail of a cohort asks for 360 at once.
  //   DP_COHORT_EXCL=0        members of a cohort do not reserve the CU (single proofs still do)
  //   DP_TAIL_MANY_EXCL=0     batched tails (several trees per launch) do not reserve the CU
  //   DP_TAIL_MANY_THREADS=n  workgroup size of batched tails (256 / 512 / 1024): 256 threads x 128 VGPRs = a quarter of a CU
  bool cohort_excl_ = !(getenv("DP_COHORT_EXCL") && !atoi(getenv("DP_COHORT_EXCL")));
  bool tail_many_excl_ = !(getenv("DP_TAIL_MANY_EXCL") && !atoi(getenv("DP_TAIL_MANY_EXCL")));
  int tail_many_threads_ = [] { const char* e = getenv("DP_TAIL_MANY_THREADS"); int v = e ? atoi(e) : 1024; return (v == 256 || v == 512) ? v : 1024; }();
  size_t excl_now() const { return (co_ && !cohort_excl_) ? 0 : excl_; }
  // throughput mode (several proofs in flight): one-workgroup kernels reserve nothing and run as 256-thread workgroups with
  // raised wave priority (KF_PRIO). DP_SHARED_TAILS=0 restores the whole-CU workgroups of round 1, DP_SHARED_THREADS = 64..1024.
  bool shared_tails_ = !(getenv("DP_SHARED_TAILS") && !atoi(getenv("DP_SHARED_TAILS")));
  int shared_threads_ = [] { const char* e = getenv("DP_SHARED_THREADS"); int v = e ? atoi(e) : 256; return (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) ? v : 256; }();
  bool throughput_mode_ = false;
  bool shared_now() const { return throughput_mode_ && shared_tails_; }
  //   DP_COHORT_PERSIST_THREADS=n  workgroup size of the one-workgroup sumcheck kernels of cohort members (256 / 512 / 1024;
  //                                default: 1024 when the CU is reserved, else by the amount of work)
  int cohort_persist_threads_ = [] { const char* e = getenv("DP_COHORT_PERSIST_THREADS"); int v = e ? atoi(e) : 0; return (v == 256 || v == 512 || v == 1024) ? v : 0; }();
  int persist_threads(size_t work) const {
    if (co_ && cohort_persist_threads_) return cohort_persist_threads_;
    return excl_now() ? 1024 : work >= 2048 ? 1024 : work >= 512 ? 512 : 256;  // exclusive CU: always the full 16 waves
  }
  unsigned long long* scdbg_ = nullptr;  // DP_SC_DEBUG=1: device cycle counters of the persistent sumcheck kernel
  unsigned long long* hmail_ = nullptr;      // host view of the challenge mailbox [seq, c0, c1]
  unsigned long long* hmail_dev_ = nullptr;  // device view
  struct ScSession { bool active = false; int ntabs = 0; size_t n = 0; unsigned long long seq = 0; std::vector<Ext*> a, b; bool nextA = true;
                     bool multi = false; int G = 0, rounds_a = 0, folds = 0, shift = 0; size_t slot_words = 0, n0 = 0; } sess_;
  static constexpr int MULTI_MAX_WG = 32;
  static constexpr size_t MULTI_MIN_N = 4096, MULTI_MAX_N = size_t(1) << 18, MULTI_TARGET_N = 1024;
  unsigned long long* hmflag_ = nullptr;      // host view of the per-workgroup flags
  unsigned long long* hmflag_dev_ = nullptr;  // device view
  unsigned long long last_tag_multi_[MULTI_MAX_WG];
  bool multi_ = true;  // DP_NO_MULTI=1 disables the multi-workgroup phase of large sumchecks
  bool multi_mid_ = getenv("DP_MULTI_MID") && atoi(getenv("DP_MULTI_MID"));
  bool persist_global_mid_ = getenv("DP_PERSIST_GLOBAL_MID") && atoi(getenv("DP_PERSIST_GLOBAL_MID"));
  static bool persist_flag_env(const char* name) { const char* e = getenv(name); return !(e && atoi(e)); }
  // all G workgroups have published round `seq`: every slot's tag matches its payload (same protocol as wait_flag)
  void wait_flags_multi(unsigned long long seq, size_t nwords, int G, size_t slot_words) {
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    int done = 0;
    while (done < G) {
      volatile unsigned long long* f = hmflag_ + done;
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent sumcheck (no challenge received)");
      if (tag != last_tag_multi_[done]) {
        std::atomic_thread_fence(std::memory_order_acquire);
        volatile u64* w = hres_ + (size_t)done * slot_words;
        unsigned long long cs = 0;
        for (size_t i = 0; i < nwords; i++) cs += (unsigned long long)(i + 1) * w[i];
        if (base + cs == tag) { last_tag_multi_[done] = tag; done++; continue; }
      }
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else __builtin_ia32_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, "timeout waiting for the device");
    }
    desc_off_ = 0; stage_off_ = 0;
    if (co_ && co_li_) co_->note_executed(co_li_ - 1);
    wait_exit_(t0);
  }
  u64* dres_ = nullptr;   // device result buffer
  unsigned* fused_ticket_ = nullptr;  // "last workgroup" ticket of k_sc_fused (device, zero between launches)
  void* hstage_ = nullptr;  // pinned + device-mapped staging: [0, DESC_BYTES) descriptor ring read by kernels over PCIe, rest = bulk copies
  char* hstage_dev_ = nullptr;
  size_t desc_off_ = 0;
  // Asynchronous uploads (throughput mode; DP_ASYNC_UPLOAD=0 turns them off, =1 forces them for single proofs too): small
  // host-to-device copies take successive slots of the bulk staging area and are not waited for — like descriptors, the slots
  // are recycled when the host observes a later publication of this stream (everything launched before it has run). Otherwise
  // every upload costs a copy launch, a publish launch and a device wait (~29 per Dense-4M proof). The slot arithmetic uses
  // ASYNC_STAGE, not the context's own staging size: the members of a cohort (the model's context has a larger staging buffer
  // than the batch workers) must take the same decisions, or the cohort falls out of step.
  size_t stage_off_ = 0;
  int async_upload_env_ = [] { const char* e = getenv("DP_ASYNC_UPLOAD"); return e ? atoi(e) : -1; }();
  bool async_upload_now() const { return async_upload_env_ < 0 ? throughput_mode_ : async_upload_env_ != 0; }
  static constexpr size_t ASYNC_STAGE = size_t(12) << 20, ASYNC_MAX = size_t(2) << 20;
  static constexpr size_t RES_WORDS = 1 << 16;
  size_t STAGE_BYTES = size_t(64) << 20;  // bulk staging of this context (workers of a batch get less: stage_bytes of the constructor)
  static constexpr size_t DESC_BYTES = 4 << 20;
  unsigned L_ = 0;  // full_message_size_log of the current PCS parameters
  u64* tw_ = nullptr;    // tw[i]   = w_{2^(L+1)}^i, i < 2^L   (all FFT root tables of rs.rs:31-68 in one array)
  u64* pow7_ = nullptr;  // pow7[i] = 7^i,          i < 2^L   (coset shifts)
  // The two tables are read-only after pcs_init: the workers of a model borrow the owning context's pair (pcs_share) — 2 x 8 B x 2^L
  // per worker otherwise (268 MB at L = 24), and one copy stays hot in L2 for every proof in flight instead of one copy per proof.
  struct PcsTables { u64* tw = nullptr; u64* pow7 = nullptr; int device = 0; ~PcsTables() { hipSetDevice(device); if (tw) hipFree(tw); if (pow7) hipFree(pow7); } };
  std::shared_ptr<PcsTables> pcs_tabs_;
  std::string name_;

  void* arena_alloc(size_t bytes) {
    size_t off = (arena_off_ + 255) & ~size_t(255);
    if (off + bytes > arena_cap_) throw DpError(DP_ERR_OOM, "device arena exhausted (raise DP_ARENA_BYTES / DP_WORKER_ARENA_BYTES)");
    arena_off_ = off + bytes;
    if (arena_off_ > arena_peak_) arena_peak_ = arena_off_;
    return arena_ + off;
  }
  static unsigned long long pub_mix(unsigned long long seq) { return seq * 0x9E3779B97F4A7C15ull + 0x7F4A7C159E3779B9ull; }
  // spin (bounded) until the message with sequence number `seq` and `nwords` payload words has fully landed in host
  // memory: the tag word must equal mix(seq) + sum (i+1)*word_i recomputed from what we read
  void wait_flag(unsigned long long seq, size_t nwords) {
    volatile unsigned long long* f = hflag_;
    volatile u64* w = hres_;
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    for (;;) {
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent sumcheck (no challenge received)");
      if (tag != last_tag_) {  // something new was written: check it against the payload
        std::atomic_thread_fence(std::memory_order_acquire);
        unsigned long long cs = 0;
        for (size_t i = 0; i < nwords; i++) cs += (unsigned long long)(i + 1) * w[i];
        if (base + cs == tag) { last_tag_ = tag; desc_off_ = 0; stage_off_ = 0; if (co_ && co_li_) co_->note_executed(co_li_ - 1); wait_exit_(t0); return; }
      }
      // inside a fiber the wait hands the host thread to the next proof in flight (fiber.h); otherwise spin
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else __builtin_ia32_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, "timeout waiting for the device");
    }
  }
  // wait_flag for a message made of blocks whose checksum runs over block-relative word indices (k_logup_tail)
  void wait_flag_blocks(unsigned long long seq, const std::vector<size_t>& block_words) {
    volatile unsigned long long* f = hflag_;
    volatile u64* w = hres_;
    auto t0 = wait_enter_();
    unsigned spins = 0;
    const unsigned long long base = pub_mix(seq);
    nwait_++;
    for (;;) {
      unsigned long long tag = *f;
      if (tag == ~0ull) throw DpError(DP_ERR_HIP, "device aborted a persistent kernel");
      if (tag != last_tag_) {
        std::atomic_thread_fence(std::memory_order_acquire);
        const unsigned long long cs = logup_tail_checksum(w, block_words);
        if (base + cs == tag) { last_tag_ = tag; desc_off_ = 0; stage_off_ = 0; if (co_ && co_li_) co_->note_executed(co_li_ - 1); wait_exit_(t0); return; }
      }
      const bool fib = fiber_active();
      if (fib) { nyield_++; fiber_yield(); } else __builtin_ia32_pause();
      if ((++spins & (fib ? 0x3FFu : 0xFFFFu)) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
        throw DpError(DP_ERR_HIP, "timeout waiting for the device");
    }
  }
  // descriptors for batched kernels: written by the host into the mapped ring and read by the kernel directly (no
  // H2D copy launch). The ring is recycled whenever the host has observed a publication, i.e. the stream is drained.
  template <class T> T* desc_alloc(size_t count, const T** dev_view) {
    size_t bytes = (count * sizeof(T) + 63) & ~size_t(63);
    DP_REQUIRE(bytes <= DESC_BYTES, DP_ERR_SHAPE, "descriptor batch too large");
    if (desc_off_ + bytes > DESC_BYTES) { stream_wait(); }
    T* h = (T*)((char*)hstage_ + desc_off_);
    *dev_view = (const T*)(hstage_dev_ + desc_off_);
    desc_off_ += bytes;
    return h;
  }
  char* bulk_stage() { return (char*)hstage_ + DESC_BYTES; }
  // wait until everything queued on the stream so far has executed, without entering hipStreamSynchronize (which
  // serialises against other host threads driving other proofs on the same GPU): a one-wave kernel posts a tag
  void stream_wait() {
    if (!zerocopy_) { HIP_CHECK(hipStreamSynchronize(s_)); desc_off_ = 0; stage_off_ = 0; return; }
    unsigned long long seq = ++seq_;
    nb_ = 0; DPL(k_publish, dim3(1), dim3(64), (const u64*)dres_, hres_dev_, (size_t)0, hflag_dev_, seq);
    wait_flag(seq, 0);
  }
  // second stage of a block-partial reduction, published straight to hres_ (see k_reduce_publish); nout <= 1024
  void reduce_publish(const Ext* partial, size_t nblocks, size_t inner, int nout) {
    DP_REQUIRE(nout >= 1 && nout <= 1024 && (size_t)nout * 2 <= RES_WORDS, DP_ERR_SHAPE, "reduce_publish: too many outputs");
    if (!zerocopy_) {
      if (inner == 4 && nout % 4 == 0 && nout > 4) DPL(k_reduce_terms, dim3(nout), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else if (inner == 2) DPL(k_reduce_pairs, dim3(nout), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else if (inner == 4 && nout <= 4) DPL(k_reduce_terms, dim3(4), dim3(TPB), partial, nblocks, (Ext*)dres_);
      else DPL(k_reduce_partials, dim3(nout), dim3(TPB), partial, nblocks, inner, (Ext*)dres_);
      fetch((size_t)nout * 2);
      return;
    }
    unsigned long long seq = ++seq_;
    int threads = nout >= 8 ? 1024 : nout >= 4 ? 256 : 64 * nout;
    DPL(k_reduce_publish, dim3(1), dim3(threads), partial, nblocks, inner, nout, (Ext*)hres_dev_, hflag_dev_, seq);
    wait_flag(seq, (size_t)nout * 2);
  }
  // bring `nwords` of dres_ to hres_
  void fetch(size_t nwords) {
    DP_REQUIRE(nwords <= RES_WORDS, DP_ERR_ARG, "result too large");
    if (zerocopy_) {
      unsigned long long seq = ++seq_;
      nb_ = 0; DPL(k_publish, dim3(1), dim3(64), (const u64*)dres_, hres_dev_, nwords, hflag_dev_, seq);
      wait_flag(seq, nwords);
    } else {
      HIP_CHECK(hipMemcpyAsync(hres_, dres_, nwords * 8, hipMemcpyDeviceToHost, s_));
      HIP_CHECK(hipStreamSynchronize(s_));
    }
  }
  static PointArg make_point(const Ext* pt, unsigned k) {
    DP_REQUIRE(k <= MAX_PT, DP_ERR_SHAPE, "point too long");
    PointArg p;
    for (unsigned i = 0; i < k; i++) p.p[i] = pt[i];
    for (unsigned i = k; i < MAX_PT; i++) p.p[i] = ex_zero();
    return p;
  }

 public:
  explicit HipDev(int device, size_t arena_bytes = 0, size_t stage_bytes = 0) : device_(device) {
    if (stage_bytes) STAGE_BYTES = stage_bytes;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) throw DpError(DP_ERR_NODEVICE, "no HIP device available: the MI355X path is mandatory, there is no CPU fallback");
    DP_REQUIRE(device >= 0 && device < cnt, DP_ERR_ARG, "bad device id");
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    name_ = std::string("hip:") + prop.name + ":" + prop.gcnArchName;
    HIP_CHECK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
    const char* env = getenv("DP_ARENA_BYTES");
    arena_cap_ = arena_bytes ? arena_bytes : env ? strtoull(env, nullptr, 10) : (size_t(12) << 30);
    HIP_CHECK(hipMalloc((void**)&arena_, arena_cap_));
    HIP_CHECK(hipHostMalloc((void**)&hres_, RES_WORDS * 8 + 1024, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostGetDevicePointer((void**)&hres_dev_, hres_, 0));
    if (host_sponge_) {  // request + reply areas of the host-side sponge service (sponge_host.h)
      HIP_CHECK(hipHostMalloc((void**)&hsp_, (WC_REQ_WORDS + WC_REP_WORDS) * 8, hipHostMallocMapped | hipHostMallocCoherent));
      memset(hsp_, 0, (WC_REQ_WORDS + WC_REP_WORDS) * 8);
      HIP_CHECK(hipHostGetDevicePointer((void**)&hsp_dev_, hsp_, 0));
      sp_slot_ = sponge_slot_new();
      sp_slot_->req = hsp_; sp_slot_->rep = hsp_ + WC_REQ_WORDS;
    }
    hflag_ = (unsigned long long*)(hres_ + RES_WORDS);
    hflag_dev_ = (unsigned long long*)(hres_dev_ + RES_WORDS);
    *hflag_ = 0;
    hmail_ = hflag_ + 8; hmail_dev_ = hflag_dev_ + 8;
    hmail_[0] = hmail_[1] = hmail_[2] = hmail_[3] = 0;
    hmflag_ = hflag_ + 32; hmflag_dev_ = hflag_dev_ + 32;  // one flag word per workgroup of a multi-workgroup sumcheck phase
    for (int i = 0; i < MULTI_MAX_WG; i++) { hmflag_[i] = 0; last_tag_multi_[i] = 0; }
    multi_ = persist_flag_env("DP_NO_MULTI");
    zerocopy_ = !(getenv("DP_NO_ZEROCOPY") && atoi(getenv("DP_NO_ZEROCOPY")));
    persist_ = zerocopy_ && !(getenv("DP_NO_PERSIST") && atoi(getenv("DP_NO_PERSIST")));
    if (getenv("DP_SC_DEBUG") && atoi(getenv("DP_SC_DEBUG"))) { HIP_CHECK(hipMalloc((void**)&scdbg_, 64)); HIP_CHECK(hipMemset(scdbg_, 0, 64)); }
    HIP_CHECK(hipMalloc((void**)&dres_, RES_WORDS * 8));
    HIP_CHECK(hipMalloc((void**)&fused_ticket_, 64)); HIP_CHECK(hipMemset(fused_ticket_, 0, 64));
    HIP_CHECK(hipHostMalloc(&hstage_, STAGE_BYTES + DESC_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_CHECK(hipHostGetDevicePointer((void**)&hstage_dev_, hstage_, 0));
    hstage_dev_ += 0;
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_rc), POSEIDON2_RC_HOST, sizeof(POSEIDON2_RC_HOST)));
    { std::vector<u64> ex((SC_MAXK + 1) * (SC_MAXK + 1) * (SC_MAXK + 1), 0);  // extrapolation_coeffs(k, at)[i] of sumcheck.h
      for (unsigned k = 1; k < (unsigned)SC_MAXK; k++) for (unsigned at = k + 1; at <= (unsigned)SC_MAXK; at++) for (unsigned i = 0; i <= k; i++)
        ex[((size_t)k * (SC_MAXK + 1) + at) * (SC_MAXK + 1) + i] = extrapolation_coeffs(k, at)[i];
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_extrap), ex.data(), ex.size() * 8)); }
    { double ts = getenv("DP_POLL_TIMEOUT_S") ? std::max(0.001, atof(getenv("DP_POLL_TIMEOUT_S"))) : 20.0; unsigned long long tk = (unsigned long long)(ts * 1e8); HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_poll_timeout_ticks), &tk, sizeof(tk))); }
#ifdef DP_DIAG_SKIP_HASH
    { int sk = getenv("DP_DEBUG_SKIP_HASH") ? atoi(getenv("DP_DEBUG_SKIP_HASH")) : 0; if (sk) HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_dbg_skip_hash), &sk, sizeof(int))); }
#endif
    { int ps = getenv("DP_POLL_SLEEP") ? std::max(0, atoi(getenv("DP_POLL_SLEEP"))) : 1; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_poll_sleep), &ps, sizeof(int))); }
    DP_SET_LDS_ONE((k_sc_persist_lds<false>), 1024, (int)SC_LDS_MAX);
    DP_SET_LDS_ONE((k_sc_persist_lds<true>), 1024, (int)SC_LDS_MAX);
    excl_ = (getenv("DP_NO_EXCLUSIVE_CU") && atoi(getenv("DP_NO_EXCLUSIVE_CU"))) ? 0 : EXCL_LDS;
    DP_SET_LDS_ONE((k_sc_persist<false>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_persist<true>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_small<false>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE((k_sc_small<true>), 1024, (int)EXCL_LDS);
    DP_SET_LDS_ONE(k_merkle_tail, 1024, (int)EXCL_LDS);
    if (devlogup_ || devlogup_full_) DP_SET_LDS_ONE(k_logup_tail, 1024, (int)EXCL_LDS);
    if (devclassic_) DP_SET_LDS_ONE(k_classic_tail, 1024, (int)EXCL_LDS);
    if (devdense_) DP_SET_LDS_ONE(k_dense_tail, 1024, (int)EXCL_LDS);
    if (deveqsum_) DP_SET_LDS_ONE(k_eqsum_tail, 1024, (int)EXCL_LDS);
    if (devcommit_) DP_SET_LDS_ONE(k_commit_tail, 1024, (int)EXCL_LDS);
    DP_SET_LDS((k_butterfly_pass<false, false>), 1024, 64 * 1024); DP_SET_LDS((k_butterfly_pass<false, true>), 1024, 64 * 1024);
    DP_SET_LDS((k_butterfly_pass<true, false>), 1024, 64 * 1024); DP_SET_LDS((k_butterfly_pass<true, true>), 1024, 64 * 1024);
    DP_SET_LDS(k_med_prepare, 1024, 128 * 1024);
    DP_SET_LDS(k_med_ntt_local, 1024, 64 * 1024);
  }
  ~HipDev() override {
    hipSetDevice(device_);
    if (s_) hipStreamSynchronize(s_);
    pcs_tabs_.reset();
    if (arena_) hipFree(arena_);
    if (dres_) hipFree(dres_);
    if (fused_ticket_) hipFree(fused_ticket_);
    if (sp_slot_) { sponge_disarm_(); sponge_slot_free(sp_slot_); sp_slot_ = nullptr; }
    if (hsp_) hipHostFree(hsp_);
    if (hres_) hipHostFree(hres_);
    if (hstage_) hipHostFree(hstage_);
    if (s_) hipStreamDestroy(s_);
  }
  const char* name() const override { return name_.c_str(); }
  size_t arena_peak() const { return arena_peak_; }
  void arena_peak_reset() { arena_peak_ = arena_off_; }
  // Dev::merkle_paths_check on the GPU: the job arrays go up in one staging pass, one launch, two words come back
  bool merkle_paths_check(const u64* leaf, const u64* root, const u64* x, const u64* path_off, const u64* depth, size_t n, const u64* pool, size_t pool_digests, size_t* first_bad) override {
    if (!n) return true;
    DP_REQUIRE(!co_, DP_ERR_ARG, "merkle_paths_check: not from inside a cohort");
    std::vector<u64> meta(3 * n);
    for (size_t j = 0; j < n; j++) { DP_REQUIRE(path_off[j] + depth[j] <= pool_digests, DP_ERR_ARG, "merkle_paths_check: path outside the pool"); meta[3 * j] = x[j]; meta[3 * j + 1] = path_off[j]; meta[3 * j + 2] = depth[j]; }
    struct ArenaMark { HipDev* d; size_t mk; ~ArenaMark() { d->release(mk); } } guard_{this, mark()};  // released on every exit: an adversarial proof must not leak the arena
    DBuf dl = alloc(4 * n, false), dr = alloc(4 * n, false), dm = alloc(3 * n, false), dp = alloc(std::max<size_t>(4 * pool_digests, 4), false), db = alloc(2, false);
    upload(dl, leaf); upload(dr, root); upload(dm, meta.data());
    if (pool_digests) upload(dp, pool);
    const u64 init[2] = {0, ~0ull};
    upload(db, init);
    nb_ = 96.0 * (double)pool_digests; DPL(k_merkle_paths, dim3(grid_for(n, 4096)), dim3(TPB), (const u64*)dl.p, (const u64*)dr.p, (const u64*)dm.p, (const u64*)dp.p, n, (unsigned long long*)db.p);
    u64 res[2];
    download(db, res);
    if (res[0] && first_bad) *first_bad = (size_t)res[1];
    return res[0] == 0;
  }
  // Poseidon2 compress() per second of the one-node-per-lane Merkle kernel on `nodes` nodes (chip-filling when nodes >> 458 752
  // = 256 CUs x 28 waves x 64 lanes): the VALU-integer peak bench.py prices the whole job's hashing against, measured with
  // HIP events on this context's stream in the same run. The input is whatever the arena holds: the arithmetic is branch-free.
  double probe_compress_rate(size_t nodes, int reps) {
    DP_REQUIRE(!co_ && nodes >= 1024 && reps >= 1, DP_ERR_ARG, "probe: bad arguments");
    const size_t mk = mark();
    DBuf in = alloc(8 * nodes, false), out = alloc(4 * nodes, false);
    nb_ = 0; DPL(k_zero_words, dim3(grid_for(8 * nodes)), dim3(TPB), (u64*)in.p, 8 * nodes);
    hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    nb_ = 96.0 * nodes; DPL(k_merkle_layer, dim3(grid_for(nodes, 4096)), dim3(TPB), (const u64*)in.p, (u64*)out.p, nodes);
    HIP_CHECK(hipEventRecord(a, s_));
    for (int r = 0; r < reps; r++) { nb_ = 96.0 * nodes; DPL(k_merkle_layer, dim3(grid_for(nodes, 4096)), dim3(TPB), (const u64*)in.p, (u64*)out.p, nodes); }
    HIP_CHECK(hipEventRecord(b, s_)); HIP_CHECK(hipEventSynchronize(b));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    release(mk);
    return ms > 0 ? (double)nodes * reps / (ms * 1e-3) : 0.0;
  }
  void set_latency_mode(bool on) { multi_ = on && persist_flag_env("DP_NO_MULTI"); devfs_ = devfs_env_ < 0 ? !on : devfs_env_ != 0; throughput_mode_ = !on; }
  void dump_host_stats() {
    if (!g_host_stats) return;
    fprintf(stderr, "[dp timing] device context: %zu launches, %.1f us of host time per launch (%.1f ms total), %zu device waits, %zu fiber yields; host work between waits %.1f ms, inside waits %.1f ms\n", nlaunch_, nlaunch_ ? launch_us_ / nlaunch_ : 0.0, launch_us_ / 1000.0, nwait_, nyield_, work_us_ / 1000.0, waitlat_us_ / 1000.0);
    if (sp_slot_) fprintf(stderr, "[dp timing] host sponge: %llu requests served for this context so far\n", (unsigned long long)sp_slot_->nserved.load());
    if (getenv("DP_LAUNCH_NAMES")) {  // launches by kernel since the last dump (DP_TIMING=1 DP_LAUNCH_NAMES=1)
      std::vector<std::pair<size_t, const char*>> v; for (auto& kv : by_name_) v.push_back({kv.second, kv.first});
      std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first > b.first; });
      for (auto& e : v) fprintf(stderr, "[dp launches] %6zu  %s\n", e.first, e.second);
    }
    by_name_.clear();
    if (getenv("DP_HOST_CHUNKS")) for (auto& c : chunks_) fprintf(stderr, "[dp chunk] before wait %zu: %.0f us of host work, launches %s .. %s\n", c.wait, c.us, c.first ? c.first : "-", c.last ? c.last : "-");
    chunks_.clear();
    launch_us_ = 0; nlaunch_ = nwait_ = nyield_ = 0; work_us_ = waitlat_us_ = 0; have_exit_ = false;
  }
  void dump_sc_debug() {
    if (!scdbg_) return;
    unsigned long long h[5]; hipStreamSynchronize(s_); hipMemcpy(h, scdbg_, 40, hipMemcpyDeviceToHost); hipMemset(scdbg_, 0, 64);
    fprintf(stderr, "[dp sc-debug] rounds %llu: cycles/round fold %.0f sums %.0f publish %.0f wait-for-challenge %.0f\n", h[4], (double)h[0] / h[4], (double)h[1] / h[4], (double)h[2] / h[4], (double)h[3] / h[4]);
  }
  void bind_thread() override { HIP_CHECK(hipSetDevice(device_)); }
  hipStream_t stream() const { return s_; }
  // per-kernel HIP-event timing on the launch stream (bench.py roofline). report: name -> (launches, total ms, total bytes)
  void profile_enable(bool on) {
    hipStreamSynchronize(s_);
    for (auto& r : recs_) { prof_pool_.push_back(r.a); prof_pool_.push_back(r.b); }
    recs_.clear();
    if (!on) { for (auto e : prof_pool_) hipEventDestroy(e); prof_pool_.clear(); }
    prof_ = on;
  }
  std::string profile_report() {
    hipStreamSynchronize(s_);
    struct Agg { size_t n = 0; double ms = 0, bytes = 0; };
    std::vector<std::pair<std::string, Agg>> agg;
    for (auto& r : recs_) {
      float ms = 0; hipEventElapsedTime(&ms, r.a, r.b);
      std::string nm = r.name;  // "(k_x<3, true>)" -> "k_x<3, true>"
      if (nm.size() > 2 && nm.front() == '(' && nm.back() == ')') nm = nm.substr(1, nm.size() - 2);
      size_t k = 0; for (; k < agg.size(); k++) if (agg[k].first == nm) break;
      if (k == agg.size()) agg.push_back({nm, Agg()});
      agg[k].second.n++; agg[k].second.ms += ms; agg[k].second.bytes += r.bytes;
    }
    std::string out = "[";
    for (size_t k = 0; k < agg.size(); k++) {
      char buf[512];
      snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"launches\": %zu, \"total_ms\": %.6f, \"alg_bytes\": %.0f}", k ? ", " : "", agg[k].first.c_str(), agg[k].second.n, agg[k].second.ms, agg[k].second.bytes);
      out += buf;
    }
    return out + "]";
  }

  // ---- memory
  DBuf alloc(size_t n, bool ext) override { DBuf b; b.n = n; b.ext = ext; b.p = arena_alloc(std::max<size_t>(n, 1) * (ext ? 16 : 8)); return b; }
  size_t mark() override { return arena_off_; }
  void release(size_t m) override { arena_off_ = m; }
  DBuf alloc_persistent(size_t n, bool ext) override {
    DBuf b; b.n = n; b.ext = ext;
    HIP_CHECK(hipSetDevice(device_));
    HIP_CHECK(hipMalloc(&b.p, std::max<size_t>(n, 1) * (ext ? 16 : 8)));
    return b;
  }
  void free_persistent(DBuf& b) override { if (b.p) { hipStreamSynchronize(s_); hipFree(b.p); b.p = nullptr; } }
  // host <-> device copies go through the pinned staging buffer: hipMemcpyAsync on pageable memory pins the user pages
  // on the fly, which costs tens of milliseconds per MB on this stack
  // A cohort member moves data with kernels (k_copy_words through the mapped staging buffer, k_zero_words): a memcpy
  // command queued by one member would overtake the merged launches its cohort has not fired yet.
  void h2d(void* dst, const void* src, size_t bytes) {
    if (async_upload_now() && zerocopy_ && bytes > 0 && bytes % 8 == 0 && bytes <= ASYNC_MAX && STAGE_BYTES >= ASYNC_STAGE) {
      const size_t need = (bytes + 255) & ~size_t(255);
      if (stage_off_ + need > ASYNC_STAGE) { stream_wait(); stage_off_ = 0; }
      char* slot = bulk_stage() + stage_off_;
      memcpy(slot, src, bytes);
      if (g_host_stats) by_name_["  (k_copy_words as upload)"]++;
      if (co_) { nb_ = 0; DPL(k_copy_words, dim3(grid_for(bytes / 8, 256)), dim3(TPB), (u64*)dst, (const u64*)(hstage_dev_ + DESC_BYTES + stage_off_), bytes / 8); }
      else { nb_ = 0; prof_begin("memcpy_h2d"); HIP_CHECK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, s_)); prof_end(); }
      stage_off_ += need;
      return;
    }
    if (stage_off_) { stream_wait(); stage_off_ = 0; }  // pending slots of earlier asynchronous uploads must have been read before slot 0 is reused
    for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
      size_t m = std::min(STAGE_BYTES, bytes - off);
      memcpy(bulk_stage(), (const char*)src + off, m);
      if (co_) {
        DP_REQUIRE(m % 8 == 0, DP_ERR_ARG, "copies are whole words");
        nb_ = 0; DPL(k_copy_words, dim3(grid_for(m / 8, 256)), dim3(TPB), (u64*)((char*)dst + off), (const u64*)(hstage_dev_ + DESC_BYTES), m / 8);
      } else { nb_ = 0; prof_begin("memcpy_h2d"); HIP_CHECK(hipMemcpyAsync((char*)dst + off, bulk_stage(), m, hipMemcpyHostToDevice, s_)); prof_end(); }
      stream_wait();
    }
  }
  void d2h(void* dst, const void* src, size_t bytes) {
    // (pending upload slots are read by copy kernels that precede this download's copy in the stream: no drain needed)
    for (size_t off = 0; off < bytes; off += STAGE_BYTES) {
      size_t m = std::min(STAGE_BYTES, bytes - off);
      if (g_host_stats) by_name_["  (k_copy_words as download)"]++;
      if (co_) {
        DP_REQUIRE(m % 8 == 0, DP_ERR_ARG, "copies are whole words");
        nb_ = 0; DPL(k_copy_words, dim3(grid_for(m / 8, 256)), dim3(TPB), (u64*)(hstage_dev_ + DESC_BYTES), (const u64*)((const char*)src + off), m / 8);
      } else { nb_ = 0; prof_begin("memcpy_d2h"); HIP_CHECK(hipMemcpyAsync(bulk_stage(), (const char*)src + off, m, hipMemcpyDeviceToHost, s_)); prof_end(); }
      stream_wait();
      memcpy((char*)dst + off, bulk_stage(), m);
    }
  }
  void upload(const DBuf& d, const u64* src) override { h2d(d.p, src, d.bytes()); }
  void upload_i64(const DBuf& d, const int64_t* src) override {
    DP_REQUIRE(!d.ext, DP_ERR_ARG, "upload_i64 needs a base buffer");
    size_t mk = mark();
    int64_t* tmp = (int64_t*)arena_alloc(d.n * 8);
    h2d(tmp, src, d.n * 8);
    DPL(k_fieldize, dim3(grid_for(d.n)), dim3(TPB), tmp, (u64*)d.p, d.n);
    stream_wait();
    release(mk);
  }
  void download(const DBuf& src, u64* dst) override { d2h(dst, src.p, src.bytes()); }
  void copy(const DBuf& d, const DBuf& s) override {
    nb_ = 2.0 * s.bytes();
    if (co_) { if (s.bytes()) DPL(k_copy_words, dim3(grid_for(s.bytes() / 8)), dim3(TPB), (u64*)d.p, (const u64*)s.p, s.bytes() / 8); return; }
    prof_begin("memcpy_d2d"); HIP_CHECK(hipMemcpyAsync(d.p, s.p, s.bytes(), hipMemcpyDeviceToDevice, s_)); prof_end();
  }
  void zero(const DBuf& d) override {
    nb_ = (double)d.bytes();
    if (co_) { if (d.bytes()) DPL(k_zero_words, dim3(grid_for(d.bytes() / 8)), dim3(TPB), (u64*)d.p, d.bytes() / 8); return; }
    prof_begin("memset"); HIP_CHECK(hipMemsetAsync(d.p, 0, d.bytes(), s_)); prof_end();
  }
  // ---- cohort membership (dp_model_prove_batch): while attached every launch of this context is one pack of a merged launch
  void cohort_attach(Cohort* co) {
    DP_REQUIRE(!co_ && !prof_ && zerocopy_, DP_ERR_ARG, "cohort members need the zero-copy publish path and no per-kernel profiling");
    HIP_CHECK(hipStreamSynchronize(s_));
    co->join(); co_ = co; co_li_ = co->q_base;
  }
  void cohort_detach() { if (co_) { Cohort* c = co_; co_ = nullptr; c->leave(co_li_); } }
  bool in_cohort() const { return co_ != nullptr; }
  void sync() override { stream_wait(); }

  // ---- MLE
  void eq_table_many(const EqJob* jobs, size_t n) override {
    if (!n) return;
    if (n * (sizeof(EqDesc) + 4) + 256 > DESC_BYTES) { Dev::eq_table_many(jobs, n); return; }
    if (desc_off_ + n * (sizeof(EqDesc) + 4) + 192 > DESC_BYTES) stream_wait();
    const EqDesc* dd = nullptr; const unsigned* fd = nullptr;
    unsigned* first = desc_alloc<unsigned>(n + 1, &fd);
    EqDesc* hd = desc_alloc<EqDesc>(n, &dd);
    unsigned nblk = 0; double bytes = 0;
    for (size_t i = 0; i < n; i++) {
      const EqJob& j = jobs[i];
      DP_REQUIRE(j.out.ext && j.out.n == (size_t(1) << j.k) && j.k <= (unsigned)MAX_PT, DP_ERR_SHAPE, "eq_table_many: output shape");
      hd[i].out = (Ext*)j.out.p; hd[i].k = j.k; hd[i].pad = 0;
      for (unsigned t = 0; t < j.k; t++) hd[i].pt[t] = j.pt[t];
      first[i] = nblk; nblk += (unsigned)((j.out.n + EQ_CHUNK - 1) / EQ_CHUNK);  // one workgroup per EQ_CHUNK entries
      bytes += 16.0 * j.out.n;
    }
    first[n] = nblk;
    nb_ = bytes; DPL(k_eq_table_many, dim3(nblk), dim3(TPB), fd, dd, (int)n);
  }
  void eq_table(const DBuf& out, const Ext* pt, unsigned k, Ext scale, bool acc) override {
    DP_REQUIRE(out.ext && out.n == (size_t(1) << k), DP_ERR_SHAPE, "eq_table: output shape");
    nb_ = 16.0 * out.n * (acc ? 2 : 1); DPL(k_eq_table, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, make_point(pt, k), k, scale, acc ? 1 : 0, out.n);
  }
  // ---- lazy eq: remembered here, built inside the persistent sumcheck kernel that consumes it (or materialised by a
  // plain launch if the next sumcheck takes another path)
  struct PendingEq { void* p = nullptr; unsigned k = 0; Ext pt[MAX_PT]; } pend_eq_;
  void eq_table_lazy(const DBuf& out, const Ext* pt, unsigned k) override {
    DP_REQUIRE(out.ext && out.n == (size_t(1) << k) && k <= (unsigned)MAX_PT, DP_ERR_SHAPE, "eq_table: output shape");
    flush_pending_eq();
    if (!persist_) { eq_table(out, pt, k, ex_one(), false); return; }
    pend_eq_.p = out.p; pend_eq_.k = k;
    for (unsigned i = 0; i < k; i++) pend_eq_.pt[i] = pt[i];
  }
  void flush_pending_eq() {
    if (!pend_eq_.p) return;
    DBuf b; b.p = pend_eq_.p; b.n = size_t(1) << pend_eq_.k; b.ext = true;
    pend_eq_.p = nullptr;
    eq_table(b, pend_eq_.pt, pend_eq_.k, ex_one(), false);
  }
  void eq_table_tiled(const DBuf& out, const Ext* pt, unsigned k) override {
    DP_REQUIRE(out.ext && out.n % (size_t(1) << k) == 0, DP_ERR_SHAPE, "eq_table_tiled: output shape");
    nb_ = 16.0 * out.n; DPL(k_eq_table, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, make_point(pt, k), k, ex_one(), 0, out.n);
  }
  void mle_eval_batch(const DBuf* fs, int nf, const Ext* pt, unsigned k, Ext* out) override {
    flush_pending_eq();
    size_t n = size_t(1) << k;
    PointArg p = make_point(pt, k);
    for (int s = 0; s < nf; s += 8) {
      EvalArgs a; a.nf = std::min(8, nf - s);
      for (int f = 0; f < 8; f++) { a.f[f] = nullptr; a.ext[f] = 0; }
      for (int f = 0; f < a.nf; f++) { DP_REQUIRE(fs[s + f].n == n, DP_ERR_SHAPE, "mle_eval: table size != 2^|point|"); a.f[f] = fs[s + f].p; a.ext[f] = fs[s + f].ext; }
      size_t mk = mark();
      int g = grid_for(n, 1024);
      Ext* partial = (Ext*)arena_alloc((size_t)g * 8 * 16);
      nb_ = [&] { double b = 0; for (int f = 0; f < a.nf; f++) b += fs[s + f].bytes(); return b; }(); DPL(k_mle_eval_partial, dim3(g), dim3(TPB), a, p, k, partial);
      reduce_publish(partial, (size_t)g, 8, a.nf);
      for (int f = 0; f < a.nf; f++) out[s + f] = ex(hres_[2 * f], hres_[2 * f + 1]);
      release(mk);
    }
  }
  void fix_high(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) override {
    DP_REQUIRE(!W.ext && W.n == R * C && out.ext && out.n == C, DP_ERR_SHAPE, "fix_high: shapes");
    unsigned k = dp_ceil_log2(R);
    size_t mk = mark();
    DBuf eq = alloc(R, true);
    eq_table(eq, pt, k, ex_one(), false);
    size_t nsplit = std::min<size_t>(R, 64);
    size_t rps = (R + nsplit - 1) / nsplit;
    Ext* partial = (Ext*)arena_alloc(nsplit * C * 16);
    dim3 g((unsigned)((C + TPB - 1) / TPB), (unsigned)nsplit);
    nb_ = 8.0 * R * C + 16.0 * R; DPL(k_fix_high_partial, g, dim3(TPB), (const u64*)W.p, (const Ext*)eq.p, R, C, rps, partial);
    DPL(k_colsum, dim3((unsigned)((C + TPB - 1) / TPB)), dim3(TPB), partial, nsplit, C, (Ext*)out.p);
    release(mk);  // safe: stream ordered, later allocations are only written by later kernels
  }

  void fix_low(const DBuf& out, const DBuf& W, size_t R, size_t C, const Ext* pt) override {
    DP_REQUIRE(!W.ext && W.n == R * C && out.ext && out.n == R && C >= 2 && (C & (C - 1)) == 0, DP_ERR_SHAPE, "fix_low: shapes");
    size_t mk = mark();
    DBuf eq = alloc(C, true);
    eq_table(eq, pt, dp_ceil_log2(C), ex_one(), false);
    nb_ = 8.0 * R * C + 16.0 * C + 16.0 * R; DPL(k_fix_low, dim3(grid_for(R * 64)), dim3(TPB), (const u64*)W.p, (const Ext*)eq.p, (Ext*)out.p, R, C);
    release(mk);  // safe: stream ordered, later allocations are only written by later kernels
  }

  // ---- sumcheck
  void fold_tables(DBuf* tabs, int nt, Ext r) {
    for (int s = 0; s < nt; s += MAX_TABS) {
      int m = std::min(MAX_TABS, nt - s);
      FoldArgs a; size_t maxh = 1;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.ext[i] = 0; a.half[i] = 0; }
      for (int i = 0; i < m; i++) {
        DBuf& t = tabs[s + i];
        DBuf o = alloc(t.n / 2, true);
        a.in[i] = t.p; a.out[i] = (Ext*)o.p; a.ext[i] = t.ext; a.half[i] = t.n / 2;
        maxh = std::max(maxh, t.n / 2);
        t = o;
      }
      nb_ = [&] { double b = 0; for (int i = 0; i < m; i++) b += a.half[i] * (a.ext[i] ? 32.0 : 16.0) + a.half[i] * 16.0; return b; }(); DPL(k_fold, dim3(grid_for(maxh), m), dim3(TPB), a, r);
    }
  }
  static constexpr size_t SC_LDS_MAX = 128 * 1024;  // dynamic LDS the LDS-resident sumcheck kernel may use
  // Latency-critical one-workgroup kernels ask for more than half of a CU's 160 KB of LDS even when they need none: two
  // such workgroups can then never share a CU. The workgroup dispatcher otherwise packs the small persistent kernels of
  // all proofs in flight onto the same first CUs (4 kernels of 256 threads fit on one), where they time-share the SIMDs.
  static constexpr size_t EXCL_LDS = 84 * 1024;
  static constexpr size_t SC_PERSIST_MAX = 16384;  // sumchecks whose tables are at most this long run in the persistent kernel
  void post_challenge(Ext r) {
    hmail_[1] = r.c0; hmail_[2] = r.c1; hmail_[3] = pub_mix(sess_.seq) + r.c0 + 2 * r.c1;  // tag: the device re-polls a torn payload
    std::atomic_thread_fence(std::memory_order_release);
    *(volatile unsigned long long*)hmail_ = sess_.seq;
    std::atomic_thread_fence(std::memory_order_seq_cst);
  }
  static constexpr size_t SC_SMALL_MAX = 8192;  // tables up to this length (after the fold) take the one-launch path
  const Ext* claim_hint_ = nullptr;  // set for the duration of sc_round_claim: the fused streaming path may skip t = 1
  void sc_round_claim(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, int nterms, const Ext* claim, Ext* out) override {
    claim_hint_ = nterms == 1 ? claim : nullptr;
    try { sc_round(tabs, nt, r, terms, nterms, out); } catch (...) { claim_hint_ = nullptr; throw; }
    claim_hint_ = nullptr;
  }
  // Every remaining round of a sumcheck in ONE launch with the Fiat-Shamir transcript on the device (ScFsArgs above): used
  // when several proofs are in flight (DP_DEVICE_FS: 0 never, 1 always, default: throughput mode only — alone, the host
  // sponge on a 5 GHz core plus two PCIe hops is faster per round than the lane-parallel permutation of one wave).
  int devfs_env_ = [] { const char* e = getenv("DP_DEVICE_FS"); return e ? atoi(e) : -1; }();
  bool devfs_ = devfs_env_ > 0;
  size_t nfs_ = 0;
  bool sc_tail(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned md, Challenger& ch,
               std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& point, Ext* finals) override {
    // latency mode (one proof on the GPU) keeps the host sponge: a round trip to the host costs 23-35 us per round, the 8-lane
    // wave sponge ~50-70 us (3-4 permutations of ~16 us: one wave on a dependent chain) — measured on the 2^24 sumcheck, whose
    // 14-round hand-over phase takes 0.49 ms with the host sponge and 0.99 ms with DP_SC_HANDOVER=1 (profiles/r02_sumcheck24_handover.txt)
    static const bool handover_env = getenv("DP_SC_HANDOVER") && atoi(getenv("DP_SC_HANDOVER"));
    const bool handover = handover_env && devfs_env_ < 0 && r && tabs[0].n >= 8192;
    if (!(devfs_ || handover) || !persist_ || !zerocopy_ || sess_.active) return false;
    if (nt > MAX_TABS || nterms > MAX_TERMS || nt <= 0 || nterms <= 0 || md < 1 || md > (unsigned)SC_MAXK) return false;
    size_t n_in = tabs[0].n;
    for (int i = 0; i < nt; i++) if (tabs[i].n != n_in) return false;
    size_t n_after = r ? n_in / 2 : n_in;
    if (n_after > SC_PERSIST_MAX || n_after < 4 || (n_after & (n_after - 1))) return false;
    bool hi = false;
    for (int i = 0; i < nterms; i++) { if (terms[i].k < 1 || terms[i].k > SC_MAXK || (unsigned)terms[i].k > md) return false; hi = hi || terms[i].k > 3; }
    unsigned rounds = 0; for (size_t m = n_after; m > 1; m >>= 1) rounds++;
    const size_t nwords = (size_t)rounds * (md + 2) * 2 + (size_t)nt * 2 + 14;
    if (nwords > RES_WORDS) return false;
    ScPersistArgs a;
    a.eq_tab = -1; a.eq_k = 0;
    if (pend_eq_.p && r) flush_pending_eq();
    if (pend_eq_.p) {
      for (int i = 0; i < nt; i++) if (tabs[i].p == pend_eq_.p && tabs[i].n == (size_t(1) << pend_eq_.k)) a.eq_tab = i;
      if (a.eq_tab >= 0) { a.eq_k = (int)pend_eq_.k; for (unsigned i = 0; i < pend_eq_.k; i++) a.eq_pt[i] = pend_eq_.pt[i]; pend_eq_.p = nullptr; }
      else flush_pending_eq();
    }
    for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
    for (int i = 0; i < MAX_TERMS; i++) { a.k[i] = 1; a.off[i] = 0; for (int j = 0; j < SC_MAXK; j++) a.t[i][j] = 0; }
    { int o = 0; for (int i = 0; i < nterms; i++) { a.k[i] = terms[i].k; for (int j = 0; j < SC_MAXK; j++) a.t[i][j] = j < terms[i].k ? terms[i].t[j] : 0; a.off[i] = o; o += terms[i].k + 1; } }
    const size_t lds = (size_t)nt * (n_in / 2) * 16;
    const bool in_lds = lds <= SC_LDS_MAX;
    size_t first = r ? n_after : n_after / 2;
    double tab_bytes = 0;
    for (int i = 0; i < nt; i++) {
      a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
      if (!in_lds) { a.bufA[i] = (Ext*)alloc(first, true).p; a.bufB[i] = (Ext*)alloc(std::max<size_t>(first / 2, 1), true).p; }
      tab_bytes += (double)n_in * (tabs[i].ext ? 16.0 : 8.0);
    }
    a.nwg = 1; a.rounds_a = 0; a.slot_ext = 0;
    a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = scdbg_;
    const ScFsArgs* fsd = nullptr;
    ScFsArgs* f = desc_alloc<ScFsArgs>(1, &fsd);
    for (int i = 0; i < 8; i++) f->state[i] = ch.state[i];
    for (int i = 0; i < 4; i++) f->in_buf[i] = i < ch.in_len ? ch.in_buf[i] : 0;
    f->in_len = ch.in_len; f->out_len = ch.out_len; f->md = (int)md; f->rounds = (int)rounds;
    { static const char lab[] = "Internal round"; size_t n = sizeof(lab) - 1; f->nlabel = 0; f->pad = 0; f->label[0] = f->label[1] = 0;
      for (size_t i = 0; i < n; i += 8) { u64 v = 0; size_t m = n - i < 8 ? n - i : 8; for (size_t q = 0; q < m; q++) v |= (u64)(uint8_t)lab[i + q] << (8 * q); f->label[f->nlabel++] = gl_from_u64(v); } }
    for (int i = 0; i < MAX_TERMS; i++) f->coeff[i] = i < nterms ? coeffs[i] : ex_zero();
    unsigned long long seq = ++seq_;
    size_t work = (size_t)nterms * (n_after / 2) + (size_t)nt * n_after / 4;
    int threads = persist_threads(work);
    if (in_lds) { nb_ = tab_bytes; DPL_ONE_HI(k_sc_persist_lds, hi, dim3(1), threads, lds, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, seq - 1, fsd); }
    else { nb_ = tab_bytes + 24.0 * (double)n_in * nt; DPL_ONE_HI(k_sc_persist, hi, dim3(1), threads, 0, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, seq - 1, fsd); }
    wait_flag(seq, nwords);
    const u64* w = hres_;
    for (unsigned q = 0; q < rounds; q++) {
      std::vector<Ext> m(md + 1);
      for (unsigned j = 0; j <= md; j++) { size_t o = ((size_t)q * (md + 1) + j) * 2; m[j] = ex(w[o], w[o + 1]); }
      msgs.push_back(std::move(m));
    }
    for (unsigned q = 0; q < rounds; q++) { size_t o = ((size_t)rounds * (md + 1) + q) * 2; point.push_back(ex(w[o], w[o + 1])); }
    size_t wf = (size_t)rounds * (md + 2) * 2;
    for (int i = 0; i < nt; i++) finals[i] = ex(w[wf + 2 * i], w[wf + 2 * i + 1]);
    size_t ws = wf + 2 * (size_t)nt;
    for (int i = 0; i < 8; i++) ch.state[i] = w[ws + i];
    ch.in_len = (int)w[ws + 12]; ch.out_len = (int)w[ws + 13];
    for (int i = 0; i < 4; i++) { ch.in_buf[i] = w[ws + 8 + i]; ch.out_buf[i] = ch.state[i]; }
    nfs_++;
    return true;
  }
  // ---- Dev::logup_tail: DP_DEVICE_LOGUP=1, see k_logup_tail. Declines (returns false) whenever the shape is
  // outside what the kernel was written for; the caller then runs the layers one by one (logup_layers).
  // fused protocol kernels: on by default whenever the device-side transcript is (throughput mode); DP_DEVICE_<X>=0 turns one off.
  // Validated on MI355X in round 2 (tests/test_gpu_fused.py: every knob alone and all together, Dense-4M and CNN-264k batches
  // against the sequential proofs; profiles/r02_fused_knob_sweep.jsonl).
  static int knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
  // ---- DP_HOST_SPONGE=1 (sponge_host.h): the fused protocol kernels keep the transcript's sponge on the HOST — a kernel posts the
  // words it absorbs and asks for challenges through a mapped mailbox, DP_SPONGE_THREADS server threads answer (the members of a
  // cohort ask together and sit with different servers). Off by default: new at the end of round 2 (the wave sponge costs ~12 us per permutation, the mailbox 2.9 us per round trip).
  bool host_sponge_ = knob("DP_HOST_SPONGE", 0) != 0;
  SpongeSlot* sp_slot_ = nullptr; u64* hsp_ = nullptr; u64* hsp_dev_ = nullptr; bool sp_active_ = false;
  void sponge_disarm_() {
    if (!sp_active_) return;
    while (sp_slot_->busy.load(std::memory_order_acquire)) __builtin_ia32_pause();
    sp_slot_->active.store(0, std::memory_order_release);
    while (sp_slot_->busy.load(std::memory_order_acquire)) __builtin_ia32_pause();  // a server that had passed the first check
    sponge_nactive().fetch_sub(1);
    sp_slot_->ch = nullptr; sp_active_ = false;
  }
  // between the descriptor fill and the launch: point the kernel at this context's mailbox and the service at the transcript; after the
  // wait the transcript is final (done()); the parse functions overwrite it with the kernel's unused sponge words (restore())
  struct SpongeArm {
    HipDev* dev; Challenger* ch; bool armed = false; Challenger fin;
    template <class D> SpongeArm(HipDev* dv, D* d, Challenger& c) : dev(dv), ch(&c) {
      if (!dv->host_sponge_ || !dv->sp_slot_) return;
      d->sp_req = dv->hsp_dev_; d->sp_rep = dv->hsp_dev_ + WC_REQ_WORDS; d->sp_seq = dv->sp_slot_->served;
      sponge_servers_start();
      dv->sp_slot_->ch = &c; dv->sp_slot_->active.store(1, std::memory_order_release); sponge_nactive().fetch_add(1); dv->sp_active_ = true;
      armed = true;
    }
    void done() { if (armed) { dev->sponge_disarm_(); fin = *ch; } }
    void restore() { if (armed) { *ch = fin; armed = false; } }
    ~SpongeArm() { if (armed) dev->sponge_disarm_(); }
  };
  bool devlogup_ = knob("DP_DEVICE_LOGUP", 2) == 1;
  size_t nlogup_tail_ = 0;
  bool logup_tail(const LogupTailArgs& a, Challenger& ch, std::vector<std::vector<std::vector<Ext>>>& layer_msgs,
                  std::vector<std::vector<Ext>>& layer_points, std::vector<std::vector<Ext>>& round_evals, std::vector<Ext>& point) override {
    if (!devlogup_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!logup_tail_accepts(a)) return false;
    const std::vector<size_t> blocks = logup_tail_blocks(a);
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    if (nwords > RES_WORDS) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const LogupTailDesc* dd = nullptr;
    LogupTailDesc* d = desc_alloc<LogupTailDesc>(1, &dd);
    logup_tail_fill(d, a, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (const LogupCircuitDev& c : *a.circuits) for (const DBuf& l : c.den) nb_ += 2.0 * 16.0 * (double)l.n;
    DPL_ONE(k_logup_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    logup_tail_parse(hres_, a, blocks, ch, layer_msgs, layer_points, round_evals, point);
    sponge.restore();
    release(mk);
    nlogup_tail_++;
    return true;
  }
  // ---- Dev::commit_tail: DP_DEVICE_COMMIT (default on): k_commit_tail, the last rounds of the Basefold commit phase
  bool devcommit_ = knob("DP_DEVICE_COMMIT", 1) != 0;
  bool commit_tail(const CommitTailArgs& a, Challenger& ch, CommitTailOut& out) override {
    if (!devcommit_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_ || !tw_) return false;
    if (!commit_tail_accepts(a, commit_tail_max_n(throughput_mode_)) || dp_ceil_log2(a.folded.n) - 1 > L_) return false;
    const std::vector<size_t> blocks = commit_tail_blocks(a);
    if (blocks[0] + blocks[1] > RES_WORDS) return false;
    const CommitTailDesc* dd = nullptr;
    CommitTailDesc* d = desc_alloc<CommitTailDesc>(1, &dd);
    std::vector<DevTree> trees;
    CommitTailDesc fill;
    commit_tail_fill(&fill, a, ch, *this, (const u64*)tw_, L_, trees);  // (the trees stay allocated: the query phase reads them)
    memcpy((void*)d, &fill, sizeof(CommitTailDesc));
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 16.0 * (double)a.folded.n * 2.0 + 32.0 * (double)a.sum_evals.n;
    DPL_ONE(k_commit_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    commit_tail_parse(hres_, a, ch, trees, out);
    sponge.restore();
    return true;
  }
  // ---- Dev::eqsum_tail: DP_DEVICE_EQSUM (default on): k_eqsum_tail, eq tables + accumulation sumcheck in one launch
  bool deveqsum_ = knob("DP_DEVICE_EQSUM", 1) != 0;
  bool eqsum_tail(const EqAccJob* jobs, int njobs, const DBuf* tabs, int ntabs, const ScTerm* terms, const Ext* coeffs, int nterms, unsigned nv, unsigned md,
                  Challenger& ch, EqSumOut& out) override {
    if (!deveqsum_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!eqsum_tail_accepts(jobs, njobs, tabs, ntabs, terms, nterms, nv, md)) return false;
    const std::vector<size_t> blocks = eqsum_tail_blocks(ntabs, nv, md);
    if (blocks[0] + blocks[1] > RES_WORDS) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const EqSumDesc* dd = nullptr;
    EqSumDesc* d = desc_alloc<EqSumDesc>(1, &dd);
    eqsum_tail_fill(d, jobs, njobs, tabs, ntabs, terms, coeffs, nterms, nv, md, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (int i = 0; i < ntabs; i++) nb_ += (double)tabs[i].bytes();
    DPL_ONE(k_eqsum_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    eqsum_tail_parse(hres_, ntabs, nv, md, ch, out);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::dense_tail: DP_DEVICE_DENSE (default on): k_dense_tail, a Dense layer's device work in one launch
  bool devdense_ = knob("DP_DEVICE_DENSE", 1) != 0;
  bool dense_tail(const DBuf& bias, const DBuf& W, size_t R, size_t C, const DBuf& in, const Ext* pt, Challenger& ch, DenseTailOut& out) override {
    if (!devdense_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!dense_tail_accepts(bias, W, R, C, in)) return false;
    const std::vector<size_t> blocks = dense_tail_blocks(C);
    flush_pending_eq();
    const size_t mk = mark();
    const DenseTailDesc* dd = nullptr;
    DenseTailDesc* d = desc_alloc<DenseTailDesc>(1, &dd);
    dense_tail_fill(d, bias, W, R, C, in, pt, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 8.0 * (double)R * (double)C + 16.0 * (double)R + 16.0 * (double)C * 4.0;
    DPL_ONE(k_dense_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    dense_tail_parse(hres_, C, ch, out);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::classic_tail: DP_DEVICE_CLASSIC (default on): k_classic_tail, the last rounds of the batch-opening sumcheck
  bool devclassic_ = knob("DP_DEVICE_CLASSIC", 1) != 0;
  bool classic_tail(const ClassicTailArgs& a, Challenger& ch, std::vector<std::vector<Ext>>& msgs, std::vector<Ext>& challenges) override {
    if (!devclassic_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    if (!classic_tail_accepts(a)) return false;
    const std::vector<size_t> blocks = classic_tail_blocks(a);
    if (blocks[0] + blocks[1] > RES_WORDS) return false;
    const size_t mk = mark();
    const ClassicTailDesc* dd = nullptr;
    ClassicTailDesc* d = desc_alloc<ClassicTailDesc>(1, &dd);
    classic_tail_fill(d, a, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = 0; for (int i = 0; i < a.np; i++) nb_ += a.fs[i].bytes() + a.eqs[i].bytes();
    DPL_ONE(k_classic_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    classic_tail_parse(hres_, a, ch, msgs, challenges);
    sponge.restore();
    release(mk);
    return true;
  }
  // ---- Dev::logup_full: DP_DEVICE_LOGUP=2 (the default): k_logup_tail in full mode — one launch and one device wait per
  // logup-GKR batch proof
  bool devlogup_full_ = knob("DP_DEVICE_LOGUP", 2) == 2;
  bool logup_full(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi, Challenger& ch, LogupFullOut& out) override {
    if (!devlogup_full_ || !devfs_ || !persist_ || !zerocopy_ || sess_.active || prof_) return false;
    size_t n = 0;
    if (!logup_full_accepts(cols, cpi, ninst, mult, &n)) return false;
    const std::vector<size_t> blocks = logup_full_blocks(n, cpi, ninst, !mult.null());
    size_t nwords = 0; for (size_t b : blocks) nwords += b;
    if (nwords > RES_WORDS) return false;
    flush_pending_eq();
    const size_t mk = mark();
    const LogupTailDesc* dd = nullptr;
    LogupTailDesc* d = desc_alloc<LogupTailDesc>(1, &dd);
    logup_full_fill(d, cols, cpi, ninst, mult, c, chi, ch, *this);
    SpongeArm sponge(this, d, ch);
    const unsigned long long seq = ++seq_;
    nb_ = (double)ninst * (8.0 * cpi * n + 16.0 * 3 * n) * 2.0;
    DPL_ONE(k_logup_tail, dim3(1), 1024, 0, dd, (u64*)hres_dev_, hflag_dev_, seq);
    wait_flag_blocks(seq, blocks);
    sponge.done();
    logup_full_parse(hres_, n, cpi, ninst, !mult.null(), blocks, ch, out);
    sponge.restore();
    release(mk);
    nlogup_tail_++;
    return true;
  }
  void sc_round(DBuf* tabs, int nt, const Ext* r, const ScTerm* terms, int nterms, Ext* out) override {
    DP_REQUIRE(nt <= MAX_TABS && nterms <= MAX_TERMS && nt > 0 && nterms > 0, DP_ERR_SHAPE, "sumcheck: too many tables/terms for one launch");
    size_t n_in = tabs[0].n;
    for (int i = 0; i < nt; i++) DP_REQUIRE(tabs[i].n == n_in, DP_ERR_SHAPE, "sumcheck: tables must have equal length");
    size_t n_after = r ? n_in / 2 : n_in;
    DP_REQUIRE(n_after >= 2, DP_ERR_SHAPE, "sumcheck: tables must keep length >= 2");
    // (n_in and r are rewritten below when a multi-workgroup phase hands its folded tables to a single-workgroup session)
    size_t nraw = 0;  // a degree-k term contributes k + 1 values; single-workgroup kernels publish them packed
    for (int i = 0; i < nterms; i++) { DP_REQUIRE(terms[i].k >= 1 && terms[i].k <= SC_MAXK, DP_ERR_SHAPE, "sumcheck: term degree must be 1..5"); nraw += terms[i].k + 1; }
    bool hi = false; for (int i = 0; i < nterms; i++) hi = hi || terms[i].k > 3;
    auto read_terms = [&]() { for (size_t o = 0; o < nraw; o++) out[o] = ex(hres_[2 * o], hres_[2 * o + 1]); };
    auto fill_terms = [&](int (*tk), int (*tt)[SC_MAXK], int* toff) {
      for (int i = 0; i < MAX_TERMS; i++) { tk[i] = 1; for (int j = 0; j < SC_MAXK; j++) tt[i][j] = 0; if (toff) toff[i] = 0; }
      int o = 0;
      for (int i = 0; i < nterms; i++) { tk[i] = terms[i].k; for (int j = 0; j < SC_MAXK; j++) tt[i][j] = j < terms[i].k ? terms[i].t[j] : 0; if (toff) toff[i] = o; o += terms[i].k + 1; }
    };
    auto read_shares = [&](int G, size_t slot_words) {  // the round sums are the sums of the workgroups' shares
      for (size_t o = 0; o < nraw; o++) {
        Ext acc = ex_zero();
        for (int g = 0; g < G; g++) { const u64* w = hres_ + (size_t)g * slot_words; acc = ex_add(acc, ex(w[2 * o], w[2 * o + 1])); }
        out[o] = acc;
      }
    };
    if (sess_.active && sess_.multi) {  // multi-workgroup phase: every workgroup folds its slice with this challenge
      DP_REQUIRE(r && nt == sess_.ntabs && n_in == sess_.n, DP_ERR_ARG, "sumcheck session out of sync");
      post_challenge(*r);
      sess_.folds++;
      const size_t lvl_off = sess_.n0 - (sess_.n0 >> (sess_.folds - 1 + sess_.shift));  // level j starts at n0 (1 - 2^-(j-1)); one level later when the phase began with a fold
      for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i] + lvl_off; tabs[i].n = n_after; tabs[i].ext = true; }
      sess_.n = n_after;
      if (sess_.folds < sess_.rounds_a) {
        wait_flags_multi(++sess_.seq, 2 * nraw, sess_.G, sess_.slot_words);
        read_shares(sess_.G, sess_.slot_words);
        return;
      }
      // the kernel leaves after this fold: the compact folded tables feed a single-workgroup session (stream ordered)
      sess_.active = false; sess_.multi = false;
      r = nullptr; n_in = n_after;
    }
    if (sess_.active) {  // the persistent kernel is waiting for this challenge
      DP_REQUIRE(r && nt == sess_.ntabs && n_in == sess_.n, DP_ERR_ARG, "sumcheck session out of sync");
      post_challenge(*r);
      wait_flag(++sess_.seq, 2 * nraw);
      sess_.n = n_after;
      for (int i = 0; i < nt; i++) { tabs[i].p = sess_.nextA ? sess_.a[i] : sess_.b[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      sess_.nextA = !sess_.nextA;
      read_terms();
      return;
    }
    if ((!r || multi_mid_) && multi_ && persist_ && n_after >= MULTI_MIN_N && n_in <= MULTI_MAX_N && nraw * 2 * MULTI_MAX_WG <= RES_WORDS) {
      // ---- multi-workgroup phase: G workgroups own contiguous slices, fold until the tables are MULTI_TARGET_N long. It may
      // begin in the middle of a sumcheck (r given: the streaming rounds of a large sumcheck hand over as soon as the tables
      // fit): every workgroup then first folds its slice with the pending challenge. Measured on the 2^24 sumcheck that costs
      // 75 us per round (32 workgroups x PCIe mailbox) against 23 us for one workgroup in LDS, so it is off unless DP_MULTI_MID=1:
      // the hand-over from the streaming rounds goes to the device-side transcript instead (sc_tail).
      flush_pending_eq();
      int G = (int)std::min<size_t>(MULTI_MAX_WG, n_after / 512);
      int rounds_a = (int)(dp_ceil_log2(n_after) - dp_ceil_log2(MULTI_TARGET_N));
      ScPersistArgs a;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
      fill_terms(a.k, a.t, a.off);
      sess_.a.assign(nt, nullptr); sess_.b.assign(nt, nullptr);
      double bytes = 0;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        sess_.a[i] = (Ext*)alloc(n_in, true).p; sess_.b[i] = nullptr;  // every fold level has its own region of this buffer
        a.bufA[i] = sess_.a[i]; a.bufB[i] = nullptr;
        bytes += tabs[i].bytes() + 3.0 * 16.0 * (n_in / 2);  // read once + the halving folded tables written and re-read
      }
      a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = nullptr; a.eq_tab = -1; a.eq_k = 0;
      a.nwg = G; a.rounds_a = rounds_a; a.slot_ext = (int)nraw;
      sess_.active = true; sess_.multi = true; sess_.G = G; sess_.rounds_a = rounds_a; sess_.folds = 0; sess_.shift = r ? 1 : 0; sess_.slot_words = 2 * nraw;
      sess_.ntabs = nt; sess_.n = n_after; sess_.n0 = n_in; sess_.seq = seq_;
      seq_ += (unsigned)rounds_a;  // one publication per round of the phase
      nb_ = bytes; if (hi) { DPL_B((k_sc_persist<true>), 1024, KF_CLAIM, dim3(G), dim3(1024), 0, a, (Ext*)hres_dev_, hmflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); } else { DPL_B((k_sc_persist<false>), 1024, KF_CLAIM, dim3(G), dim3(1024), 0, a, (Ext*)hres_dev_, hmflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }  // (latency mode only: G whole-CU workgroups)
      wait_flags_multi(++sess_.seq, 2 * nraw, G, sess_.slot_words);
      if (r) for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      read_shares(G, sess_.slot_words);
      return;
    }
    // A sumcheck that arrives here in the middle (r pending: the streaming rounds of a large one are handing over) enters the
    // persistent kernel only once its tables fit in LDS: the global-memory variant costs 35 us per round on 2^15..2^13-entry
    // tables against ~20 us for another streaming / one-launch round (DP_PERSIST_GLOBAL_MID=1 restores the early hand-over).
    const bool lds_fits = (size_t)nt * (n_in / 2) * 16 <= SC_LDS_MAX;
    const bool persist_here = persist_ && n_after <= SC_PERSIST_MAX && n_after >= 4 && 2 * nraw <= RES_WORDS && (!r || lds_fits || persist_global_mid_ || throughput_mode_);
    const bool take_persistent = !sess_.active && persist_here;
    if (pend_eq_.p && !(take_persistent && !r)) flush_pending_eq();
    if (persist_here) {
      ScPersistArgs a;
      a.eq_tab = -1; a.eq_k = 0;
      if (pend_eq_.p) {
        for (int i = 0; i < nt; i++) if (tabs[i].p == pend_eq_.p && tabs[i].n == (size_t(1) << pend_eq_.k)) a.eq_tab = i;
        if (a.eq_tab >= 0) { a.eq_k = (int)pend_eq_.k; for (unsigned i = 0; i < pend_eq_.k; i++) a.eq_pt[i] = pend_eq_.pt[i]; pend_eq_.p = nullptr; }
        else flush_pending_eq();
      }
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.in_ext[i] = 0; a.bufA[i] = nullptr; a.bufB[i] = nullptr; }
      fill_terms(a.k, a.t, a.off);
      sess_.a.assign(nt, nullptr); sess_.b.assign(nt, nullptr);
      // ping-pong buffers: A takes the first fold output, B the second, A the third, ...
      size_t first = r ? n_after : n_after / 2;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        sess_.a[i] = (Ext*)alloc(first, true).p; sess_.b[i] = (Ext*)alloc(std::max<size_t>(first / 2, 1), true).p;
        a.bufA[i] = sess_.a[i]; a.bufB[i] = sess_.b[i];
      }
      a.nwg = 1; a.rounds_a = 0; a.slot_ext = 0;
      a.ntabs = nt; a.nterms = nterms; a.has_r0 = r ? 1 : 0; a.n0 = n_in; a.r0 = r ? *r : ex_zero(); a.dbg = scdbg_;
      sess_.active = true; sess_.ntabs = nt; sess_.n = n_after; sess_.seq = seq_; sess_.nextA = r ? false : true;
      // reserve the sequence numbers of all rounds + the final message
      unsigned rounds = 0; for (size_t m = n_after; m > 1; m >>= 1) rounds++;
      seq_ += rounds + 1;
      size_t work = (size_t)nterms * (n_after / 2) + (size_t)nt * n_after / 4;
      int threads = persist_threads(work);
      size_t lds = (size_t)nt * (n_in / 2) * 16;
      double tab_bytes = 0; for (int i = 0; i < nt; i++) tab_bytes += (double)n_in * (tabs[i].ext && !r ? 16.0 : tabs[i].ext ? 16.0 : 8.0);
      // algorithmic HBM bytes of the launch: every table is read once (the LDS variant never touches HBM again; the
      // global variant also writes and re-reads the halving ping-pong buffers: + 3 x 16 B x n/2 per table in total)
      if (lds <= SC_LDS_MAX) { nb_ = tab_bytes; DPL_ONE_HI(k_sc_persist_lds, hi, dim3(1), threads, lds, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }
      else { nb_ = tab_bytes + 24.0 * (double)n_in * nt; DPL_ONE_HI(k_sc_persist, hi, dim3(1), threads, 0, a, (Ext*)hres_dev_, hflag_dev_, (const unsigned long long*)hmail_dev_, sess_.seq, (const ScFsArgs*)nullptr); }
      wait_flag(++sess_.seq, 2 * nraw);
      if (r) for (int i = 0; i < nt; i++) { tabs[i].p = sess_.a[i]; tabs[i].n = n_after; tabs[i].ext = true; }
      read_terms();
      return;
    }
    // (a single product still 4096+ entries long after the fold, one proof on the GPU: the streaming kernel below does the
    // round on many workgroups in ~15 us; the one-workgroup kernel here needs ~55 us for it)
    const bool stream_instead = !throughput_mode_ && r && nterms == 1 && terms[0].k == nt && nt <= 3 && n_after >= 4096;
    if (zerocopy_ && n_after <= SC_SMALL_MAX && 2 * nraw <= RES_WORDS && !stream_instead) {
      ScSmallArgs a;
      for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.in_ext[i] = 0; }
      fill_terms(a.k, a.t, a.off);
      double bytes = 0;
      for (int i = 0; i < nt; i++) {
        a.in[i] = tabs[i].p; a.in_ext[i] = tabs[i].ext;
        if (r) { DBuf o = alloc(n_after, true); a.out[i] = (Ext*)o.p; bytes += tabs[i].bytes() + o.bytes(); tabs[i] = o; }
      }
      for (int i = 0; i < nterms; i++) bytes += 16.0 * n_after * terms[i].k;
      a.ntabs = nt; a.nterms = nterms; a.has_r = r ? 1 : 0; a.n_after = n_after; a.r = r ? *r : ex_zero();
      unsigned long long seq = ++seq_;
      size_t work = (size_t)nterms * (n_after / 2) + (r ? (size_t)nt * n_after / 4 : 0);
      int threads = persist_threads(work);
      nb_ = bytes; DPL_ONE_HI(k_sc_small, hi, dim3(1), threads, 0, a, (Ext*)hres_dev_, hflag_dev_, seq);
      wait_flag(seq, 2 * nraw);
      read_terms();
      return;
    }
    if (r && nterms == 1 && terms[0].k == nt && nt <= 3 && n_in >= 8) {
      bool uniform = true, distinct = true;
      for (int i = 0; i < nt; i++) { uniform &= tabs[i].ext == tabs[0].ext; for (int j = 0; j < i; j++) distinct &= terms[0].t[i] != terms[0].t[j]; }
      if (uniform && distinct) {
        const void* in[3] = {nullptr, nullptr, nullptr}; Ext* outp[3] = {nullptr, nullptr, nullptr};
        bool base = !tabs[0].ext;
        double bytes = 0;
        for (int j = 0; j < nt; j++) {
          int ti = terms[0].t[j];
          DBuf o = alloc(n_after, true);
          in[j] = tabs[ti].p; outp[j] = (Ext*)o.p;
          bytes += tabs[ti].bytes() + o.bytes();
          tabs[ti] = o;
        }
        size_t nquads = n_in / 4;
        size_t mk = mark();
        int g = grid_for(nquads, 4096);
        Ext* partial = (Ext*)arena_alloc((size_t)g * 4 * 16);
        nb_ = bytes;
        const bool skip1 = claim_hint_ != nullptr;
        // one proof on the GPU: the kernel's last workgroup reduces and publishes (no second launch); cohort members keep the
        // separate reduction (their workgroups of one launch belong to different proofs)
        // ... which is OFF (DP_FUSED_TICKET=1): measured on the 2^24 sumcheck, the agent-scope release every workgroup needs before
        // it takes its ticket is an L2 write-back per workgroup — the fused rounds went from 162 / 40 us to 1122 / 230 us
        // (4096 write-backs per launch); a separate 14 us reduction launch per round is the cheaper way across XCDs.
        static const bool ticket_env = getenv("DP_FUSED_TICKET") && atoi(getenv("DP_FUSED_TICKET"));
        const bool inkernel = ticket_env && zerocopy_ && !co_ && fused_ticket_ != nullptr;
        unsigned* tick = inkernel ? fused_ticket_ : nullptr;
        const unsigned long long fseq = inkernel ? ++seq_ : 0;
        #define LAUNCH_FUSED2(KK, BB) do { if (skip1) DPL_B((k_sc_fused<KK, BB, true>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nquads, *r, partial, tick, (Ext*)hres_dev_, hflag_dev_, fseq); \
                                           else DPL_B((k_sc_fused<KK, BB, false>), 256, KF_NONE, dim3(g), dim3(TPB), 0, in[0], in[1], in[2], outp[0], outp[1], outp[2], nquads, *r, partial, tick, (Ext*)hres_dev_, hflag_dev_, fseq); } while (0)
        #define LAUNCH_FUSED(KK) do { if (base) LAUNCH_FUSED2(KK, true); else LAUNCH_FUSED2(KK, false); } while (0)
        if (nt == 1) LAUNCH_FUSED(1); else if (nt == 2) LAUNCH_FUSED(2); else LAUNCH_FUSED(3);
        #undef LAUNCH_FUSED
        #undef LAUNCH_FUSED2
        if (inkernel) wait_flag(fseq, 8); else reduce_publish(partial, (size_t)g, 4, 4);
        for (int t = 0; t <= terms[0].k; t++) out[t] = ex(hres_[2 * t], hres_[2 * t + 1]);
        if (skip1) out[1] = ex_sub(*claim_hint_, out[0]);  // s(0) + s(1) = claim, exactly
        release(mk);
        return;
      }
    }
    if (r) fold_tables(tabs, nt, *r);
    size_t n = tabs[0].n;
    TermArgs a;
    for (int i = 0; i < MAX_TABS; i++) { a.tab[i] = nullptr; a.ext[i] = 0; }
    for (int i = 0; i < nt; i++) { a.tab[i] = tabs[i].p; a.ext[i] = tabs[i].ext; }
    fill_terms(a.k, a.t, nullptr);
    a.npairs = n / 2;
    size_t mk = mark();
    int g = grid_for(a.npairs, 2048);
    DP_REQUIRE(nterms * SC_SLOTS <= 1024, DP_ERR_SHAPE, "sumcheck: too many terms for one reduction");
    Ext* partial = (Ext*)arena_alloc((size_t)nterms * g * SC_SLOTS * 16);
    nb_ = [&] { double b = 0; for (int i = 0; i < nterms; i++) for (int j = 0; j < terms[i].k; j++) b += tabs[terms[i].t[j]].bytes(); return b; }(); DPL_HI(k_sc_terms, hi, dim3(g, nterms), dim3(TPB), a, partial);
    reduce_publish(partial, (size_t)g, SC_SLOTS, nterms * SC_SLOTS);
    size_t o = 0;
    for (int i = 0; i < nterms; i++)
      for (int t = 0; t <= terms[i].k; t++) out[o++] = ex(hres_[(i * SC_SLOTS + t) * 2], hres_[(i * SC_SLOTS + t) * 2 + 1]);
    release(mk);
  }
  void sc_finish(DBuf* tabs, int nt, Ext r, Ext* finals) override {
    DP_REQUIRE(nt <= MAX_TABS, DP_ERR_SHAPE, "sumcheck: too many tables");
    flush_pending_eq();
    if (sess_.active) {
      DP_REQUIRE(nt == sess_.ntabs && sess_.n == 2, DP_ERR_ARG, "sumcheck session out of sync at finish");
      post_challenge(r);
      wait_flag(++sess_.seq, (size_t)nt * 2);
      for (int i = 0; i < nt; i++) finals[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
      sess_.active = false;
      return;
    }
    FoldArgs a;
    for (int i = 0; i < MAX_TABS; i++) { a.in[i] = nullptr; a.out[i] = nullptr; a.ext[i] = 0; a.half[i] = 0; }
    for (int i = 0; i < nt; i++) { DP_REQUIRE(tabs[i].n == 2, DP_ERR_SHAPE, "sc_finish: tables must have 2 entries"); a.in[i] = tabs[i].p; a.ext[i] = tabs[i].ext; }
    if (zerocopy_) { unsigned long long seq = ++seq_; DPL(k_finish_publish, dim3(1), dim3(64), a, r, nt, (Ext*)hres_dev_, hflag_dev_, seq); wait_flag(seq, 2 * (size_t)nt); }
    else { DPL(k_finish, dim3(1), dim3(64), a, r, nt, (Ext*)dres_); fetch(2 * (size_t)nt); }
    for (int i = 0; i < nt; i++) finals[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
  }

  // ---- logup
  void logup_den(const DBuf& out, const DBuf* cols, int nc, Ext c, Ext chi) override {
    DP_REQUIRE(nc >= 1 && nc <= 16 && out.ext, DP_ERR_SHAPE, "logup_den: 1..16 columns");
    ColsArg a; a.n = nc;
    for (int i = 0; i < 16; i++) a.col[i] = nullptr;
    for (int i = 0; i < nc; i++) { DP_REQUIRE(!cols[i].ext && cols[i].n == out.n, DP_ERR_SHAPE, "logup_den: column shape"); a.col[i] = (const u64*)cols[i].p; }
    nb_ = 8.0 * nc * out.n + 16.0 * out.n; DPL(k_logup_den, dim3(grid_for(out.n)), dim3(TPB), (Ext*)out.p, a, c, chi, out.n);
  }
  void logup_layer(const DBuf& ni, const DBuf& di, const DBuf& no, const DBuf& dout) override {
    size_t h = di.n / 2;
    DP_REQUIRE(di.ext && no.n == h && dout.n == h && (ni.null() || ni.n == di.n), DP_ERR_SHAPE, "logup_layer: shapes");
    int mode = ni.null() ? 0 : (ni.ext ? 2 : 1);
    nb_ = (mode == 0 ? 32.0 : mode == 1 ? 48.0 : 64.0) * h + 32.0 * h; DPL(k_logup_layer, dim3(grid_for(h)), dim3(TPB), (const void*)ni.p, mode, (const Ext*)di.p, (Ext*)no.p, (Ext*)dout.p, h);
  }

  void logup_build(const DBuf* cols, int cpi, int ninst, const DBuf& mult, Ext c, Ext chi,
                   std::vector<LogupCircuitDev>& circuits, std::vector<Ext>& outputs) override {
    size_t n = cols[0].n;
    if (n > 16384 || n < 4 || cpi > 8 || (size_t)ninst * sizeof(LogupTreeDesc) > DESC_BYTES || (size_t)ninst * 8 > RES_WORDS) {
      Dev::logup_build(cols, cpi, ninst, mult, c, chi, circuits, outputs);
      return;
    }
    circuits.clear(); outputs.clear();
    const LogupTreeDesc* dd = nullptr;
    LogupTreeDesc* hd = desc_alloc<LogupTreeDesc>((size_t)ninst, &dd);
    for (int s = 0; s < ninst; s++) {
      LogupCircuitDev cd;
      DBuf den_all = alloc(2 * n, true), num_all = alloc(n, true);
      LogupTreeDesc& t = hd[s];
      for (int j = 0; j < 8; j++) t.col[j] = nullptr;
      for (int j = 0; j < cpi; j++) { const DBuf& col = cols[(size_t)s * cpi + j]; DP_REQUIRE(!col.ext && col.n == n, DP_ERR_SHAPE, "logup: column shape"); t.col[j] = (const u64*)col.p; }
      t.ncols = cpi; t.num0 = mult.p; t.num_mode = mult.null() ? 0 : (mult.ext ? 2 : 1);
      t.den_all = (Ext*)den_all.p; t.num_all = (Ext*)num_all.p;
      size_t doff = 0, noff = 0;
      cd.num.push_back(mult);
      for (size_t len = n; len >= 2; len >>= 1) {
        cd.den.push_back(den_all.slice(doff, len));
        if (len < n) { cd.num.push_back(num_all.slice(noff, len)); noff += len; }
        doff += len;
      }
      circuits.push_back(cd);
    }
    size_t mk = mark();
    int threads = n >= 2048 ? 1024 : n >= 512 ? 512 : 256;
    nb_ = (double)ninst * (8.0 * cpi * n + 16.0 * 3 * n); DPL(k_logup_tree, dim3(ninst), dim3(threads), dd, n, c, chi, (Ext*)dres_);
    fetch((size_t)ninst * 8);
    for (int i = 0; i < 4 * ninst; i++) outputs.push_back(ex(hres_[2 * i], hres_[2 * i + 1]));
    release(mk);
  }

  // ---- PCS
  void pcs_init(unsigned L) override {
    if (L == L_ && tw_) return;
    DP_REQUIRE(L >= 1 && L <= 28, DP_ERR_ARG, "pcs_init: unsupported parameter size");
    HIP_CHECK(hipStreamSynchronize(s_));
    tw_ = pow7_ = nullptr;
    pcs_tabs_ = std::make_shared<PcsTables>();
    pcs_tabs_->device = device_;
    L_ = L;
    size_t n = size_t(1) << L;
    HIP_CHECK(hipMalloc((void**)&pcs_tabs_->tw, n * 8));
    HIP_CHECK(hipMalloc((void**)&pcs_tabs_->pow7, n * 8));
    tw_ = pcs_tabs_->tw; pow7_ = pcs_tabs_->pow7;
    u64 w = GL_G32;
    for (unsigned i = L + 1; i < 32; i++) w = gl_sqr(w);
    DPL(k_pow_table, dim3(grid_for(n)), dim3(TPB), tw_, w, n);
    DPL(k_pow_table, dim3(grid_for(n)), dim3(TPB), pow7_, GL_GENERATOR, n);
    HIP_CHECK(hipStreamSynchronize(s_));
  }
  void pcs_share(HipDev& owner) {
    DP_REQUIRE(owner.device_ == device_ && owner.pcs_tabs_, DP_ERR_ARG, "pcs_share: the owner has no tables on this device");
    HIP_CHECK(hipStreamSynchronize(s_));
    pcs_tabs_ = owner.pcs_tabs_; tw_ = pcs_tabs_->tw; pow7_ = pcs_tabs_->pow7; L_ = owner.L_;
  }
  void bitrev_copy(const DBuf& d, const DBuf& s) override {
    unsigned lg = dp_ceil_log2(s.n);
    if (s.ext) { nb_ = 32.0 * s.n; DPL(k_bitrev<true>, dim3(grid_for(s.n)), dim3(TPB), d.p, (const void*)s.p, lg); }
    else { nb_ = 16.0 * s.n; DPL(k_bitrev<false>, dim3(grid_for(s.n)), dim3(TPB), d.p, (const void*)s.p, lg); }
  }
  // Run k_merkle_tail over `nd` descriptors of the mapped ring and bring the roots to hres_[4*i..]. A single tree
  // publishes its root itself; several trees land in dres_ and are published together.
  void tails_to_host(const TailDesc* dd, size_t nd) {
    if (nd == 1 && zerocopy_) {
      unsigned long long seq = ++seq_;
      nb_ = 0; DPL_ONE(k_merkle_tail, dim3(1), 1024, 0, dd, dres_, hres_dev_, hflag_dev_, seq);
      wait_flag(seq, 4);
      return;
    }
    nb_ = 0; if (!shared_now() && !tail_many_excl_) { DPL_B(k_merkle_tail, 1024, KF_NONE, dim3((unsigned)nd), dim3(tail_many_threads_), 0, dd, dres_, (u64*)nullptr, (unsigned long long*)nullptr, 0ull); } else DPL_ONE(k_merkle_tail, dim3((unsigned)nd), tail_many_threads_, 0, dd, dres_, (u64*)nullptr, (unsigned long long*)nullptr, 0ull);
    fetch(4 * nd);
  }
  // layers of at most this many digests are finished by k_merkle_tail (one workgroup, no relaunch between layers); wider
  // layers get their own multi-workgroup launch: a 512-parent layer is 4 sequential passes inside the tail workgroup but one
  // pass spread over 16 CUs as a launch (DP_TAIL_MAX overrides)
  size_t TAIL_MAX = getenv("DP_TAIL_MAX") ? strtoull(getenv("DP_TAIL_MAX"), nullptr, 10) : 256;
  // Layers with at most lp_max_ parent nodes use the 8-lanes-per-node kernel (lowest latency per layer, but ~2.2x the
  // VALU work of one node per lane and a grid 8x as large: with many proofs in flight those grids fill the chip and
  // every other stream queues behind them), wider layers hash one node per lane. DP_MERKLE_LP_MAX overrides.
  int merkle_fuse_ = getenv("DP_MERKLE_FUSE") ? atoi(getenv("DP_MERKLE_FUSE")) : 0;  // layers per launch of k_merkle_layers (0 / 1: off)
  size_t lp_max_ = getenv("DP_MERKLE_LP_MAX") ? strtoull(getenv("DP_MERKLE_LP_MAX"), nullptr, 10) : (size_t(1) << 12);
  // `nodes` must hold 4*(n-1) words; synchronises (root is copied to the host)
  DevTree build_tree_into(const DBuf& leaves, const DBuf& nodes) {
    DevTree t; t.leaves = leaves; t.nleaves = leaves.n; t.nodes = nodes;
    size_t n = leaves.n;
    u64* nd = (u64*)t.nodes.p;
    if (leaves.ext) { nb_ = 16.0 * n + 16.0 * n; DPL(k_merkle_leaves<true>, dim3(grid_for(n / 2)), dim3(TPB), (const void*)leaves.p, nd, n / 2); }
    else { nb_ = 8.0 * n + 16.0 * n; DPL(k_merkle_leaves<false>, dim3(grid_for(n / 2)), dim3(TPB), (const void*)leaves.p, nd, n / 2); }
    size_t off = 0, cnt = n / 2;
    while (cnt > TAIL_MAX) {
      size_t next = cnt / 2;
      if (merkle_fuse_ > 1 && cnt >= 512) {  // several layers per launch (k_merkle_layers): as many as stay above the tail, at most merkle_fuse_
        int levels = 0;
        while (levels < merkle_fuse_ && levels < 8 && (cnt >> (levels + 1)) >= TAIL_MAX && (cnt >> (levels + 1)) >= 1) levels++;
        if (levels >= 2) {
          nb_ = 0; for (int l = 0; l < levels; l++) nb_ += 96.0 * (double)(cnt >> (l + 1));
          DPL(k_merkle_layers, dim3((unsigned)(cnt / 512)), dim3(256), (const u64*)(nd + 4 * off), nd + 4 * (off + cnt), cnt, levels);
          for (int l = 0; l < levels; l++) { off += cnt; cnt /= 2; }
          continue;
        }
      }
      if (next <= lp_max_) { nb_ = 96.0 * next; DPL(k_merkle_layer_lp, dim3((unsigned)std::min<size_t>(std::min<size_t>((next * 8 + 255) / 256, 8192), (size_t)std::max(1, g_max_grid))), dim3(256), (const u64*)(nd + 4 * off), nd + 4 * (off + cnt), next); }
      else { nb_ = 96.0 * next; DPL(k_merkle_layer, dim3(grid_for(next, 4096)), dim3(TPB), (const u64*)(nd + 4 * off), nd + 4 * (off + cnt), next); }
      off += cnt; cnt /= 2;
    }
    const TailDesc* dd = nullptr;
    TailDesc* hd = desc_alloc<TailDesc>(1, &dd);
    hd[0].nodes = nd; hd[0].off = off; hd[0].cnt = cnt;
    tails_to_host(dd, 1);
    for (int k = 0; k < 4; k++) t.root.v[k] = hres_[k];
    return t;
  }
  DevTree build_tree(const DBuf& leaves, bool persistent) {
    size_t n = leaves.n;
    DP_REQUIRE(n >= 2 && (n & (n - 1)) == 0, DP_ERR_SHAPE, "merkle: leaf count must be a power of two >= 2");
    DBuf nodes = persistent ? alloc_persistent(4 * (n - 1), false) : alloc(4 * (n - 1), false);
    return build_tree_into(leaves, nodes);
  }
  // stages [s0, s1) of the Moebius transform / DIT NTT of `buf` as LDS-tiled passes (k_butterfly_pass): tiles of 64 KB (2^13 base
  // or 2^12 extension elements); the first pass works on contiguous blocks, later ones on strided tiles of 128-byte segments
  bool ntt_stagewise_ = getenv("DP_NTT_STAGEWISE") && atoi(getenv("DP_NTT_STAGEWISE"));
  void butterfly_passes(const DBuf& buf, unsigned s0, unsigned s1, bool ntt) {
    const unsigned lgtile = buf.ext ? 12 : 13, lgseg = buf.ext ? 3 : 4;
    const unsigned lgn = dp_ceil_log2(buf.n);
    const size_t lds = (size_t(1) << lgtile) * (buf.ext ? 16 : 8);
    while (s0 < s1) {
      const unsigned lgc = std::min(s0, lgseg), span = std::min(s1 - s0, std::min(lgtile, lgn) - lgc), s_hi = s0 + span;
      const size_t tiles = buf.n >> (span + lgc);
      const size_t bytes = (size_t(1) << (span + lgc)) * (buf.ext ? 16 : 8);
      nb_ = 2.0 * (double)buf.bytes();
      if (buf.ext) { if (ntt) { DPL_LDS((k_butterfly_pass<true, true>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } else { DPL_LDS((k_butterfly_pass<true, false>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } }
      else { if (ntt) { DPL_LDS((k_butterfly_pass<false, true>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } else { DPL_LDS((k_butterfly_pass<false, false>), dim3((unsigned)tiles), dim3(TPB), bytes, buf.p, s0, s_hi, lgc, (const u64*)tw_, L_); } }
      (void)lds;
      s0 = s_hi;
    }
  }
  DevCommit commit(const DBuf& evals, bool persistent) override {
    DevCommit c; c.nv = dp_ceil_log2(evals.n); c.is_base = !evals.ext; c.evals = evals;
    DP_REQUIRE((size_t(1) << c.nv) == evals.n && evals.n >= 2, DP_ERR_SHAPE, "commit: polynomial length must be a power of two >= 2");
    if (c.nv <= 7) { c.bh_evals = evals; c.tree = build_tree(evals, persistent); return c; }
    DP_REQUIRE(tw_ && c.nv <= L_, DP_ERR_SHAPE, "commit: polynomial larger than the PCS parameters (PolynomialTooLarge)");
    size_t n = evals.n;
    auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
    DBuf cw = A(2 * n, evals.ext);
    c.bh_evals = A(n, evals.ext);
    DBuf nodes = A(4 * (2 * n - 1), false);
    size_t mk = mark();
    DBuf co = alloc(n, evals.ext);
    DBuf tmp = alloc(2 * n, evals.ext);
    copy(co, evals);
    bool E = evals.ext;
    if (ntt_stagewise_) {
      for (unsigned s = 0; s < c.nv; s++) {  // K5 evaluations -> multilinear coefficients, one stage per launch (DP_NTT_STAGEWISE=1)
        if (E) { nb_ = 32.0 * n; DPL(k_mobius_stage<true>, dim3(grid_for(n / 2)), dim3(TPB), co.p, n, s); }
        else { nb_ = 16.0 * n; DPL(k_mobius_stage<false>, dim3(grid_for(n / 2)), dim3(TPB), co.p, n, s); }
      }
    } else butterfly_passes(co, 0, c.nv, false);  // K5 in LDS tiles: 2 passes for 2^20
    if (E) { nb_ = 48.0 * n; DPL(k_rs_prepare<true>, dim3(grid_for(n)), dim3(TPB), (const void*)co.p, tmp.p, (const u64*)pow7_, c.nv, L_); }
    else { nb_ = 24.0 * n; DPL(k_rs_prepare<false>, dim3(grid_for(n)), dim3(TPB), (const void*)co.p, tmp.p, (const u64*)pow7_, c.nv, L_); }
    if (ntt_stagewise_) {
      for (unsigned s = 1; s <= c.nv; s++) {  // K7 remaining DIT stages on 2n points
        if (E) { nb_ = 64.0 * n; DPL(k_ntt_stage<true>, dim3(grid_for(n)), dim3(TPB), tmp.p, 2 * n, s, (const u64*)tw_, L_); }
        else { nb_ = 32.0 * n; DPL(k_ntt_stage<false>, dim3(grid_for(n)), dim3(TPB), tmp.p, 2 * n, s, (const u64*)tw_, L_); }
      }
    } else butterfly_passes(tmp, 1, c.nv + 1, true);  // K7 stages 1..nv of the 2n-point DIT
    bitrev_copy(cw, tmp);            // K6
    bitrev_copy(c.bh_evals, evals);  // K6
    c.tree = build_tree_into(cw, nodes);  // K8 (synchronises: root to host)
    release(mk);
    return c;
  }
  // Commit many polynomials at once (witness columns of one inference): equal-size groups share batched launches —
  // one LDS-resident Moebius+NTT workgroup per polynomial, one layer-0 launch, one fused Merkle-tail workgroup per tree.
  // Medium base-field polynomials (2^12..2^14: the witness columns of a convolution) of equal size: Moebius in LDS, the
  // NTT in LDS-resident blocks of 2^13 points plus at most two global stages, batched Merkle layers — a dozen launches for
  // the whole group instead of ~45 per polynomial.
  void commit_medium_group(const std::vector<DBuf>& evals, size_t first, std::vector<DevCommit>& out, std::vector<bool>& done, bool persistent) {
    const DBuf& e0 = evals[first];
    unsigned nv = dp_ceil_log2(e0.n);
    size_t n = e0.n, N = 2 * n;
    std::vector<size_t> grp;
    for (size_t j = first; j < evals.size(); j++) if (!done[j] && evals[j].n == n && !evals[j].ext) grp.push_back(j);
    size_t g = grp.size();
    DP_REQUIRE(g * (sizeof(SmallCommitDesc) + sizeof(TailDesc)) + 256 <= DESC_BYTES && 4 * g <= RES_WORDS && g <= 65535, DP_ERR_SHAPE, "commit_many: group too large");
    if (desc_off_ + g * (sizeof(SmallCommitDesc) + sizeof(TailDesc)) + 256 > DESC_BYTES) stream_wait();
    auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
    const SmallCommitDesc* dd = nullptr; const TailDesc* tdd = nullptr;
    SmallCommitDesc* hd = desc_alloc<SmallCommitDesc>(g, &dd);
    TailDesc* td = desc_alloc<TailDesc>(g, &tdd);
    std::vector<DBuf> cws(g), bhs(g), nodes(g);
    for (size_t q = 0; q < g; q++) { cws[q] = A(N, false); bhs[q] = A(n, false); nodes[q] = A(4 * (N - 1), false); }
    size_t mk = mark();
    // layers above TAIL_MAX digests are hashed by batched launches; the tail kernel finishes each tree
    size_t off = 0, cnt = N / 2;
    std::vector<std::pair<size_t, size_t>> layers;
    while (cnt > TAIL_MAX) { layers.push_back({off, cnt}); off += cnt; cnt /= 2; }
    for (size_t q = 0; q < g; q++) {
      DevCommit& c = out[grp[q]];
      const DBuf& ev = evals[grp[q]];
      c.nv = nv; c.is_base = true; c.evals = ev; c.bh_evals = bhs[q];
      c.tree.leaves = cws[q]; c.tree.nleaves = N; c.tree.nodes = nodes[q];
      DBuf tmp = alloc(N, false);
      hd[q].evals = ev.p; hd[q].cw = cws[q].p; hd[q].bh = bhs[q].p; hd[q].nodes = (u64*)nodes[q].p; hd[q].tmp = tmp.p;
      td[q].nodes = (u64*)nodes[q].p; td[q].off = off; td[q].cnt = cnt;
    }
    nb_ = g * 32.0 * n; DPL_LDS(k_med_prepare, dim3((unsigned)g), dim3(1024), n * 8, dd, nv, L_, (const u64*)pow7_);
    unsigned lgblk = std::min<unsigned>(nv + 1, MED_NTT_LG), smax = std::min<unsigned>(nv, lgblk - 1);
    nb_ = g * 32.0 * n; DPL_LDS(k_med_ntt_local, dim3((unsigned)(N >> lgblk), (unsigned)g), dim3(1024), (size_t(1) << lgblk) * 8, dd, lgblk, smax, (const u64*)tw_, L_);
    for (unsigned st = smax + 1; st <= nv; st++) { nb_ = g * 32.0 * n; DPL(k_ntt_stage_many, dim3(grid_for(n, 64), (unsigned)g), dim3(TPB), dd, N, st, (const u64*)tw_, L_); }
    nb_ = g * 32.0 * n; DPL(k_bitrev_many, dim3(grid_for(N, 64), (unsigned)g), dim3(TPB), dd, nv + 1);
    nb_ = g * 24.0 * N; DPL(k_merkle_leaves_many<false>, dim3(grid_for(N / 2, 64), (unsigned)g), dim3(TPB), dd, N / 2);
    for (auto& l : layers) { nb_ = g * 96.0 * (l.second / 2); DPL(k_merkle_layer_many, dim3(grid_for(l.second / 2, 64), (unsigned)g), dim3(TPB), tdd, l.first, l.second); }
    tails_to_host(tdd, g);
    for (size_t q = 0; q < g; q++) { for (int k = 0; k < 4; k++) out[grp[q]].tree.root.v[k] = hres_[4 * q + k]; done[grp[q]] = true; }
    release(mk);
  }
  std::vector<DevCommit> commit_many(const std::vector<DBuf>& evals, bool persistent) override {
    std::vector<DevCommit> out(evals.size());
    std::vector<bool> done(evals.size(), false);
    for (size_t i = 0; i < evals.size(); i++) {
      if (done[i]) continue;
      const DBuf& e0 = evals[i];
      unsigned nv = dp_ceil_log2(e0.n);
      bool small = (size_t(1) << nv) == e0.n && nv >= 1 && (nv <= 7 || (tw_ && nv <= L_ && nv <= (e0.ext ? 10u : 11u)));
      bool medium = !small && !e0.ext && (size_t(1) << nv) == e0.n && tw_ && nv <= L_ && nv >= 12 && nv <= 14;
      if (medium) { commit_medium_group(evals, i, out, done, persistent); continue; }
      if (!small) { out[i] = commit(e0, persistent); done[i] = true; continue; }
      std::vector<size_t> grp;
      for (size_t j = i; j < evals.size(); j++) if (!done[j] && evals[j].n == e0.n && evals[j].ext == e0.ext) grp.push_back(j);
      size_t g = grp.size(), n = e0.n;
      bool trivial = nv <= 7;
      size_t nleaves = trivial ? n : 2 * n;
      DP_REQUIRE(g * sizeof(SmallCommitDesc) + g * sizeof(TailDesc) + 128 <= DESC_BYTES && 4 * g <= RES_WORDS, DP_ERR_SHAPE, "commit_many: group too large");
      auto A = [&](size_t m, bool e) { return persistent ? alloc_persistent(m, e) : alloc(m, e); };
      if (desc_off_ + g * sizeof(SmallCommitDesc) + g * sizeof(TailDesc) + 128 > DESC_BYTES) stream_wait();
      const SmallCommitDesc* dd = nullptr; const TailDesc* tdd = nullptr;
      SmallCommitDesc* hd = desc_alloc<SmallCommitDesc>(g, &dd);
      TailDesc* td = desc_alloc<TailDesc>(g, &tdd);
      for (size_t q = 0; q < g; q++) {
        DevCommit& c = out[grp[q]];
        const DBuf& ev = evals[grp[q]];
        c.nv = nv; c.is_base = !ev.ext; c.evals = ev;
        DBuf cw = trivial ? ev : A(2 * n, ev.ext);
        c.bh_evals = trivial ? ev : A(n, ev.ext);
        DBuf nodes = A(4 * (nleaves - 1), false);
        c.tree.leaves = cw; c.tree.nleaves = nleaves; c.tree.nodes = nodes;
        hd[q].evals = ev.p; hd[q].cw = cw.p; hd[q].bh = c.bh_evals.p; hd[q].nodes = (u64*)nodes.p; hd[q].tmp = nullptr;
        td[q].nodes = (u64*)nodes.p; td[q].off = 0; td[q].cnt = nleaves / 2;
      }
      size_t mk = mark();
      if (!trivial) {
        size_t lds = 3 * n * (e0.ext ? 16 : 8);
        if (e0.ext) { nb_ = g * 64.0 * n; DPL_B((k_commit_small<true>), 256, KF_NONE, dim3((unsigned)g), dim3(256), lds, dd, nv, L_, (const u64*)tw_, (const u64*)pow7_); }
        else { nb_ = g * 32.0 * n; DPL_B((k_commit_small<false>), 256, KF_NONE, dim3((unsigned)g), dim3(256), lds, dd, nv, L_, (const u64*)tw_, (const u64*)pow7_); }
      }
      if (e0.ext) { nb_ = g * 32.0 * nleaves; DPL(k_merkle_leaves_many<true>, dim3(grid_for(nleaves / 2, 64), (unsigned)g), dim3(TPB), dd, nleaves / 2); }
      else { nb_ = g * 24.0 * nleaves; DPL(k_merkle_leaves_many<false>, dim3(grid_for(nleaves / 2, 64), (unsigned)g), dim3(TPB), dd, nleaves / 2); }
      tails_to_host(tdd, g);
      for (size_t q = 0; q < g; q++) { for (int k = 0; k < 4; k++) out[grp[q]].tree.root.v[k] = hres_[4 * q + k]; done[grp[q]] = true; }
      release(mk);
    }
    return out;
  }
  void free_commit(DevCommit& c) override {
    if (c.bh_evals.p && c.bh_evals.p != c.evals.p) free_persistent(c.bh_evals);
    if (c.tree.leaves.p && c.tree.leaves.p != c.evals.p) free_persistent(c.tree.leaves);
    free_persistent(c.tree.nodes);
    free_persistent(c.evals);
  }
  DevTree merkle_ext(const DBuf& leaves) override { return build_tree(leaves, false); }
  DevTree batch_tree(const DBuf* cws, int k, bool persistent) override {
    DP_REQUIRE(k >= 2 && k <= BATCH_ROW_MAX, DP_ERR_SHAPE, "batch_tree: 2..32 polynomials per tree");
    const size_t n = cws[0].n; const bool E = cws[0].ext;
    BatchRowPtrs a{};
    for (int q = 0; q < k; q++) { DP_REQUIRE(cws[q].n == n && cws[q].ext == E, DP_ERR_SHAPE, "batch_tree: equal sizes and one field expected"); a.cw[q] = (const u64*)cws[q].p; }
    DBuf rows = persistent ? alloc_persistent(2 * n, true) : alloc(2 * n, true);
    nb_ = (E ? 16.0 : 8.0) * n * k + 32.0 * n;
    if (E) { DPL(k_batch_row_hash<true>, dim3(grid_for(n)), dim3(TPB), a, k, (u64*)rows.p, n); }
    else { DPL(k_batch_row_hash<false>, dim3(grid_for(n)), dim3(TPB), a, k, (u64*)rows.p, n); }
    return build_tree(rows, persistent);
  }

  void classic_round(DBuf* fs, DBuf* eqs, int np, const Ext* r, Ext* out) override {
    DP_REQUIRE((size_t)np * (sizeof(PolyDesc) * 2 + sizeof(ClassicDesc) + 4) + 320 <= DESC_BYTES && (size_t)np * 4 <= RES_WORDS && np * 2 <= 1024, DP_ERR_SHAPE, "classic_round: too many polynomials");
    if (desc_off_ + (size_t)np * (sizeof(PolyDesc) * 2 + sizeof(ClassicDesc) + 4) + 320 > DESC_BYTES) stream_wait();
    const PolyDesc* dd = nullptr;
    PolyDesc* hd = desc_alloc<PolyDesc>((size_t)np, &dd);
    size_t maxn = 1;
    for (int i = 0; i < np; i++) {
      DP_REQUIRE(fs[i].n == eqs[i].n && eqs[i].ext, DP_ERR_SHAPE, "classic_round: f/eq shapes");
      hd[i].f = fs[i].p; hd[i].eq = (const Ext*)eqs[i].p; hd[i].n = fs[i].n; hd[i].fext = fs[i].ext; hd[i].pad = 0; hd[i].fout = nullptr; hd[i].eqout = nullptr;
      if (r && fs[i].n > 1) {
        DBuf fo = alloc(fs[i].n / 2, true), eo = alloc(fs[i].n / 2, true);
        hd[i].fout = (Ext*)fo.p; hd[i].eqout = (Ext*)eo.p;
        fs[i] = fo; eqs[i] = eo;
      }
      maxn = std::max(maxn, hd[i].n);
    }
    size_t mk = mark();
    if (zerocopy_) {
      // one fused launch (fold + sums of the folded tables) on a work-proportional grid, one reduction that publishes
      const unsigned* fd = nullptr; const ClassicDesc* cdd = nullptr;
      unsigned* first = desc_alloc<unsigned>((size_t)np + 1, &fd);
      ClassicDesc* cd = desc_alloc<ClassicDesc>((size_t)np, &cdd);
      unsigned nblk = 0; double bytes = 0;
      for (int i = 0; i < np; i++) {
        cd[i].f = hd[i].f; cd[i].eq = hd[i].eq; cd[i].fout = hd[i].fout; cd[i].eqout = hd[i].eqout; cd[i].n = hd[i].n; cd[i].fext = hd[i].fext; cd[i].pad = 0;
        const bool folds = r && hd[i].n > 1;
        size_t items = folds ? hd[i].n / 4 : hd[i].n / 2;  // loop iterations of the pair: 4 (2) entries of each table per iteration
        size_t nb = std::min<size_t>(std::max<size_t>((items + TPB * 4 - 1) / (TPB * 4), 1), (size_t)std::min(1024, g_max_grid));
        first[i] = nblk; nblk += (unsigned)nb;
        bytes += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + hd[i].n * 16.0 + (folds ? hd[i].n * 16.0 : 0.0);
      }
      first[np] = nblk;
      Ext* partial = (Ext*)arena_alloc((size_t)nblk * 2 * 16);
      unsigned long long seq = ++seq_;
      nb_ = bytes; DPL(k_classic_fused, dim3(nblk), dim3(TPB), fd, cdd, np, r ? *r : ex_zero(), r ? 1 : 0, partial);
      nb_ = 0; DPL(k_classic_reduce, dim3(1), dim3(np >= 8 ? 1024 : 256), fd, np, (const Ext*)partial, (Ext*)hres_dev_, hflag_dev_, seq);
      wait_flag(seq, (size_t)np * 4);
      for (int i = 0; i < 2 * np; i++) out[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
      release(mk);
      return;
    }
    if (r) {
      nb_ = [&] { double b = 0; for (int i = 0; i < np; i++) if (hd[i].fout) b += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + hd[i].n * 16.0 + hd[i].n * 16.0; return b; }(); DPL(k_classic_fold, dim3(grid_for(maxn / 2, 1024), np), dim3(TPB), dd, *r);
      // descriptors for the sums: the folded tables (a second ring slot — the fold may still be reading the first)
      hd = desc_alloc<PolyDesc>((size_t)np, &dd);
      maxn = 1;
      for (int i = 0; i < np; i++) { hd[i].f = fs[i].p; hd[i].eq = (const Ext*)eqs[i].p; hd[i].n = fs[i].n; hd[i].fext = fs[i].ext; hd[i].pad = 0; hd[i].fout = nullptr; hd[i].eqout = nullptr; maxn = std::max(maxn, hd[i].n); }
    }
    int g = grid_for(std::max<size_t>(maxn / 2, 1), 256);
    Ext* partial = (Ext*)arena_alloc((size_t)np * g * 2 * 16);
    nb_ = [&] { double b = 0; for (int i = 0; i < np; i++) b += hd[i].n * (hd[i].fext ? 16.0 : 8.0) + hd[i].n * 16.0; return b; }(); DPL(k_classic_sums, dim3(g, np), dim3(TPB), dd, partial);
    reduce_publish(partial, (size_t)g, 2, np * 2);
    for (int i = 0; i < 2 * np; i++) out[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
    release(mk);
  }
  void axpy_many(const DBuf& acc, const DBuf* init, const AxpyJob* jobs, size_t n) override {
    DP_REQUIRE(acc.ext && (!init || (init->ext && init->n == acc.n)), DP_ERR_SHAPE, "axpy_many: accumulator shape");
    if (n * sizeof(AxpyDesc) + 64 > DESC_BYTES) { Dev::axpy_many(acc, init, jobs, n); return; }
    const AxpyDesc* dd = nullptr;
    AxpyDesc* hd = desc_alloc<AxpyDesc>(std::max<size_t>(n, 1), &dd);
    double bytes = 16.0 * acc.n * (init ? 2 : 1);
    for (size_t i = 0; i < n; i++) {
      const AxpyJob& j = jobs[i];
      DP_REQUIRE(acc.n == j.x.n * j.rep && (j.rep & (j.rep - 1)) == 0, DP_ERR_SHAPE, "axpy_many: shapes");
      hd[i].x = j.x.p; hd[i].xext = j.x.ext; hd[i].lg_rep = dp_ceil_log2(j.rep); hd[i].n_x = j.x.n; hd[i].coeff = j.coeff;
      bytes += j.x.bytes();
    }
    nb_ = bytes; DPL(k_axpy_many, dim3(grid_for(acc.n)), dim3(TPB), (Ext*)acc.p, init ? (const Ext*)init->p : (const Ext*)nullptr, acc.n, dd, (int)n);
  }
  void axpy_rep(const DBuf& acc, const DBuf& x, Ext coeff, size_t rep) override {
    DP_REQUIRE(acc.ext && acc.n == x.n * rep && (rep & (rep - 1)) == 0, DP_ERR_SHAPE, "axpy_rep: shapes");
    nb_ = x.bytes() + 32.0 * acc.n; DPL(k_axpy_rep, dim3(grid_for(acc.n)), dim3(TPB), (Ext*)acc.p, (const void*)x.p, (int)x.ext, coeff, acc.n, dp_ceil_log2(rep));
  }
  void bf_round(DBuf& eq, DBuf& f, const Ext* ch, Ext* msg) override {
    DP_REQUIRE(eq.ext && f.ext && eq.n == f.n, DP_ERR_SHAPE, "bf_round: shapes");
    if (ch) { DBuf t[2] = {eq, f}; fold_tables(t, 2, *ch); eq = t[0]; f = t[1]; }
    if (!msg) return;
    if (f.n == 1) { u64 w[2]; download(f, w); msg[0] = msg[1] = msg[2] = ex(w[0], w[1]); return; }
    size_t mk = mark();
    int g = grid_for(f.n / 2, 512);
    Ext* partial = (Ext*)arena_alloc((size_t)g * 4 * 16);
    nb_ = 32.0 * f.n; DPL(k_bf_msg, dim3(g), dim3(TPB), (const Ext*)f.p, (const Ext*)eq.p, f.n / 2, partial);
    reduce_publish(partial, (size_t)g, 4, 3);
    for (int i = 0; i < 3; i++) msg[i] = ex(hres_[2 * i], hres_[2 * i + 1]);
    release(mk);
  }
  DBuf fri_fold(const DBuf& o, unsigned level, Ext ch) override {
    DP_REQUIRE(o.ext && o.n == (size_t(2) << level) && level <= L_, DP_ERR_SHAPE, "fri_fold: shapes");
    DBuf out = alloc(o.n / 2, true);
    u64 gam = GL_GENERATOR;
    for (unsigned i = 0; i < L_ + 1 - level - 1; i++) gam = gl_sqr(gam);
    u64 ninv = gl_neg(gl_inv(gl_dbl(gam)));  // -1/(2 gamma)
    nb_ = 16.0 * o.n + 8.0 * o.n; DPL(k_fri_fold, dim3(grid_for(out.n)), dim3(TPB), (const Ext*)o.p, (Ext*)out.p, out.n, level, (const u64*)tw_, L_, gam, ninv, ch);
    return out;
  }
  void query_gather(const QueryDesc* d, size_t nd, std::vector<std::vector<u64>>& out) override {
    out.resize(nd);
    if (!nd) return;
    const GatherDesc* dd = nullptr;
    GatherDesc* hd = desc_alloc<GatherDesc>(nd, &dd);
    size_t total = 0;
    for (size_t i = 0; i < nd; i++) {
      const DevTree& t = *d[i].tree;
      hd[i].leaves = t.leaves.p; hd[i].nodes = (const u64*)t.nodes.p; hd[i].nleaves = t.nleaves; hd[i].p0 = d[i].p0;
      hd[i].ext = t.leaves.ext; hd[i].height = (int)t.height(); hd[i].out_off = total;
      total += (t.leaves.ext ? 4 : 2) + 4 * (size_t)(t.height() - 1);
    }
    size_t mk = mark();
    u64* dout = (u64*)arena_alloc(total * 8);
    DPL(k_query_gather, dim3((unsigned)((nd + 3) / 4)), dim3(TPB), dd, nd, dout);
    std::vector<u64> flat(total);
    stream_wait();
    d2h(flat.data(), dout, total * 8);
    for (size_t i = 0; i < nd; i++) {
      size_t len = (hd[i].ext ? 4 : 2) + 4 * (size_t)(hd[i].height - 1);
      out[i].assign(flat.begin() + hd[i].out_off, flat.begin() + hd[i].out_off + len);
    }
    release(mk);
  }
};

Dev* make_hip_dev(int device) { return new HipDev(device); }
Dev* make_hip_worker(int device, size_t arena_bytes) { return new HipDev(device, arena_bytes, size_t(16) << 20); }
// cohorts (lock-step batches of proofs, see struct Cohort): created and driven by dp_model_prove_batch
Cohort* hip_cohort_new() { const char* e = getenv("DP_COHORT_RING_BYTES"); return e ? new Cohort(strtoull(e, nullptr, 10)) : new Cohort(); }
void hip_cohort_free(Cohort* c) { delete c; }
void hip_cohort_drain(Cohort* c) { c->drain(); }
void hip_cohort_stats(Cohort* c, size_t* fired, size_t* packs) {
  *fired = c->nfired; *packs = c->npacks; c->nfired = c->npacks = 0;
  if (g_host_stats && c->nwakes) fprintf(stderr, "[dp timing] cohort: %zu wake-ups; device phases (fire -> first member sees a result) %.1f ms, host phases (wake-up -> next fire) %.1f ms = %.1f us each\n",
                                        c->nwakes, c->dev_phase_us / 1000.0, c->host_phase_us / 1000.0, c->host_phase_us / c->nwakes);
  c->dev_phase_us = c->host_phase_us = 0; c->nwakes = 0; c->awake_ = false; c->have_fire_ = false;
}
void hip_dev_cohort_attach(Dev* d, Cohort* c) { static_cast<HipDev*>(d)->cohort_attach(c); }
void hip_dev_cohort_detach(Dev* d) { static_cast<HipDev*>(d)->cohort_detach(); }
void hip_dev_dump_sc_debug(Dev* d) { static_cast<HipDev*>(d)->dump_sc_debug(); }
size_t hip_dev_arena_peak(Dev* d) { return static_cast<HipDev*>(d)->arena_peak(); }
double hip_dev_probe_compress_rate(Dev* d, size_t nodes, int reps) { return static_cast<HipDev*>(d)->probe_compress_rate(nodes, reps); }
void hip_dev_arena_peak_reset(Dev* d) { static_cast<HipDev*>(d)->arena_peak_reset(); }
// free / total bytes of the device's HBM: dp_model_prove_batch sizes the number of proofs in flight against it
void hip_mem_info(int device, size_t* free_bytes, size_t* total_bytes) { HIP_CHECK(hipSetDevice(device)); HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes)); }
void hip_dev_dump_host_stats(Dev* d) { static_cast<HipDev*>(d)->dump_host_stats(); }
// latency mode (one proof on the GPU): large sumcheck rounds spread over several workgroups; throughput mode (many proofs
// in flight): one workgroup per sumcheck — spreading costs more CUs and host polls than it saves when the GPU is shared
// DIAGNOSTIC BUILD ONLY (DP_WG_TIMES): per merged launch of k_logup_tail, the spread between its members
void hip_dump_wg_times() {
#ifdef DP_WG_TIMES
  unsigned n = 0; if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgt_n), sizeof(n)) != hipSuccess) return;
  n = std::min(n, 65536u);
  std::vector<unsigned long long> w(4 * (size_t)n);
  if (n && hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_wgt), w.size() * 8) != hipSuccess) return;
  unsigned zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_wgt_n), &zero, sizeof(zero));
  // s_memrealtime ticks at 100 MHz. Packs addresses recur (a ring): a group = records with one address whose entries lie within 50 ms
  std::map<unsigned long long, std::vector<size_t>> by;
  for (size_t i = 0; i < n; i++) by[w[4 * i + 2]].push_back(i);
  double sum_kernel = 0, sum_med = 0, sum_spread_end = 0, sum_skew = 0, sum_minmax = 0; size_t groups = 0, members = 0, samecu = 0;
  for (auto& kv : by) {
    auto idx = kv.second;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return w[4 * a] < w[4 * b]; });
    size_t s = 0;
    while (s < idx.size()) {
      size_t e = s + 1;
      while (e < idx.size() && w[4 * idx[e]] - w[4 * idx[s]] < 5000000ull) e++;
      if (e - s >= 2) {
        unsigned long long t0 = ~0ull, t0max = 0, t1min = ~0ull, t1 = 0; std::vector<double> dur; std::map<unsigned long long, int> cu;
        for (size_t k = s; k < e; k++) {
          size_t i = idx[k];
          t0 = std::min(t0, w[4 * i]); t0max = std::max(t0max, w[4 * i]); t1min = std::min(t1min, w[4 * i + 1]); t1 = std::max(t1, w[4 * i + 1]);
          dur.push_back((double)(w[4 * i + 1] - w[4 * i]) / 100.0);
          unsigned long long id = w[4 * i + 3]; unsigned hw = (unsigned)id; unsigned xcc = (unsigned)(id >> 32) & 0xF;
          cu[((unsigned long long)xcc << 16) | ((hw >> 8) & 0xFF) | (((hw >> 13) & 0x7) << 12)]++;  // (xcc, se/sh, cu)
        }
        std::sort(dur.begin(), dur.end());
        sum_kernel += (double)(t1 - t0) / 100.0; sum_med += dur[dur.size() / 2]; sum_spread_end += (double)(t1 - t1min) / 100.0; sum_skew += (double)(t0max - t0) / 100.0;
        sum_minmax += dur.back() - dur.front();
        for (auto& c : cu) if (c.second > 1) samecu += c.second;
        groups++; members += e - s;
      }
      s = e;
    }
  }
  if (groups) fprintf(stderr, "[dp wg-times] k_logup_tail: %zu merged launches, %.1f members each: first entry -> last exit %.0f us; median member %.0f us; slowest - fastest member %.0f us; "
                              "start skew (last entry - first entry) %.0f us; last exit - first exit %.0f us; %.1f %% of the members shared a CU with another member of their launch\n",
                      groups, (double)members / groups, sum_kernel / groups, sum_med / groups, sum_minmax / groups, sum_skew / groups, sum_spread_end / groups, 100.0 * samecu / members);
#endif
}
void hip_dev_pcs_share(Dev* worker, Dev* owner) { static_cast<HipDev*>(worker)->pcs_share(*static_cast<HipDev*>(owner)); }
void hip_dev_set_latency_mode(Dev* d, bool on) { static_cast<HipDev*>(d)->set_latency_mode(on); }
void hip_dev_profile_enable(Dev* d, bool on) { static_cast<HipDev*>(d)->profile_enable(on); }
std::string hip_dev_profile_report(Dev* d) { return static_cast<HipDev*>(d)->profile_report(); }

}  // namespace dp
