// extern "C" surface of libdeepprove_hip.so (declared in include/deep_prove_hip.h).
#include "../../include/deep_prove_hip.h"
#include "zkml.h"
#include "blob.h"
#include "sharded.h"
#include "fiber.h"
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and enums only: the functions are resolved with dlopen (no link-time dependency on librccl)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <mutex>
#include <thread>

DP_FIBER_SWITCH_ASM
namespace dp { struct Cohort; Cohort* hip_cohort_new(); Cohort* hip_cohort_new_sharing(Cohort* with); void hip_cohort_free(Cohort* c); void hip_cohort_drain(Cohort* c); void hip_cohort_stats(Cohort* c, size_t* fired, size_t* packs); void hip_dev_cohort_attach(Dev* d, Cohort* c); void hip_dev_cohort_detach(Dev* d); void hip_dev_set_latency_mode(Dev* d, bool on); void hip_dev_pcs_share(Dev* worker, Dev* owner); void hip_dump_wg_times(); void hip_dev_dump_host_stats(Dev* d); size_t hip_dev_arena_peak(Dev* d); double hip_dev_probe_compress_rate(Dev* d, size_t nodes, int reps); void hip_dev_arena_peak_reset(Dev* d); void hip_mem_info(int device, size_t* free_bytes, size_t* total_bytes); void hip_dev_dump_sc_debug(Dev* d); Dev* make_hip_worker(int device, size_t arena_bytes); Dev* make_hip_dev(int device); void hip_dev_profile_enable(Dev* d, bool on); std::string hip_dev_profile_report(Dev* d); }
using namespace dp;

// `mu`: PCS::commit is called from rayon workers in the reference (zkml/src/commit/context.rs:79-103 into_par_iter over
// nodes, activation.rs:294-304 over columns): the entry points a host would call that way (uploads, commit, frees) take the
// context's lock and bind the device to the calling thread — the device work of one context is one stream, so concurrent
// callers are queued, not run in parallel.
struct dp_async;
struct dp_ctx { Dev* dev; int device_id; std::recursive_mutex mu; std::atomic<dp_async*> engine{nullptr}; };  // engine: dp_ctx_route_to_engine — the blocking seam calls of this context are submit + wait there
struct CtxLock { std::unique_lock<std::recursive_mutex> l; explicit CtxLock(dp_ctx* c) : l(c->mu) { c->dev->bind_thread(); } };
// ---- blocking seam calls routed through an engine (dp_ctx_route_to_engine): submit + wait. The ticket is freed whatever happens; a failed submit / wait /
// result call rethrows the library's error (g_err holds its message) so that the blocking entry point reports it like one of its own.
struct RoutedTicket {
  dp_ticket* t = nullptr;
  void wait();
  dp_ctx* t_ctx();
  ~RoutedTicket();
};
static void routed_check(int32_t rc);
static dp_async* routed_engine(dp_ctx* ctx);
struct dp_buf { DBuf b; };
struct dp_transcript { Transcript t; };
struct dp_commit { DevCommit c; };
struct dp_batch_commit { DevBatchCommit c; };
struct dp_model {
  dp_ctx* ctx; std::unique_ptr<Context> zk; std::vector<std::unique_ptr<Dev>> workers; std::vector<dp::Cohort*> cohorts;
  size_t last_in_flight = 0, in_flight_cap = 0; size_t prove_peak = 0;  // largest arena footprint a proof of this model has had so far (sizes the arenas of batch workers)
  ~dp_model() { for (size_t i = cohorts.size(); i-- > 0;) hip_cohort_free(cohorts[i]); }  // (last first: a cohort that shares a stream goes before the one that owns it)
};

// Every cohort stream needs a hardware queue of its own (24 are served without time slicing; the HIP runtime multiplexes streams
// onto GPU_MAX_HW_QUEUES = 4 by default and reads the variable when it initialises, i.e. at the first HIP call of the process).
// The library asks for 24 when it is loaded — a host that has not touched HIP yet needs to know nothing; one that has already
// initialised HIP keeps its setting (dp_model_prove_batch then simply shares queues: slower, not wrong, since no kernel of the
// throughput path waits for the host).
__attribute__((constructor)) static void dp_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "24", 0); }
// the host transcript's Poseidon2: the AVX-512 permutation of p2_avx512.cpp when the CPU has it (DP_NO_AVX512=1: the scalar code)
__attribute__((constructor)) static void dp_install_fast_poseidon2() {
  const char* e = getenv("DP_NO_AVX512");
  if (!(e && atoi(e)) && dp::p2_cpu_has_avx512()) { dp::p2_fast() = dp::p2_permute_avx512; dp::p2_fast_compress8() = dp::p2_compress8_avx512; dp::dl_copy_sum_fast() = dp::dl_copy_sum_avx512; }
}
static thread_local std::string g_err;
template <class F>
static int32_t guard(F f) {
  try { f(); return DP_OK; }
  catch (const DpError& e) { g_err = e.what(); return e.code; }
  catch (const std::bad_alloc&) { g_err = "host out of memory"; return DP_ERR_OOM; }
  catch (const std::exception& e) { g_err = e.what(); return DP_ERR_ARG; }
  catch (...) { g_err = "unknown error"; return DP_ERR_ARG; }
}
// Buffers handed to the caller (proof streams: 5.9 MB each for Dense-4M, 1 536 of them per bench step) come from a pool of
// recycled blocks: a fresh malloc of that size is an mmap whose pages fault in one by one under the worker that writes the
// proof, and its free is a munmap on the caller's thread — 9 GB of page faults and 1 536 serial munmaps per step otherwise.
// Blocks are rounded up to 64 KB classes and carry a 32-byte header (magic, capacity); dp_free returns them to the pool
// (bounded by DP_OUT_POOL_BYTES, default 24 GB of idle blocks; beyond that it is free()). dp_free takes ONLY pointers this
// library returned — as the header has always said.
namespace {
struct OutHeader { uint64_t magic, cap, pad0, pad1; };
constexpr uint64_t OUT_MAGIC = 0x44504F55544F4B31ULL, OUT_IDLE_MAGIC = 0x44504F5554465245ULL;  // handed out / idle in the pool
struct OutPool {
  std::mutex mu;
  std::multimap<size_t, void*> idle;  // capacity -> block (header address)
  std::set<void*> live;                // headers of the blocks currently handed out: dp_free validates against it BEFORE it touches a header
  size_t idle_bytes = 0, out_bytes = 0, out_peak = 0;
  // idle blocks kept: at most what was ever handed out at once (one batch of proofs comes back and is handed out again), at least
  // 1 GB, never more than DP_OUT_POOL_BYTES (default 24 GB)
  size_t limit = [] { const char* e = getenv("DP_OUT_POOL_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t(24) << 30); }();
  size_t keep() const { return std::min(limit, std::max<size_t>(size_t(1) << 30, out_peak)); }
  void* take(size_t bytes) {
    const size_t cap = (std::max<size_t>(bytes, 8) + 65535) & ~size_t(65535);
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = idle.lower_bound(cap);
      if (it != idle.end() && it->first <= cap + cap / 4) {
        OutHeader* h = (OutHeader*)it->second; idle_bytes -= it->first; idle.erase(it);
        h->magic = OUT_MAGIC; out_bytes += h->cap; out_peak = std::max(out_peak, out_bytes); live.insert(h);
        return (char*)h + sizeof(OutHeader);
      }
    }
    OutHeader* h = (OutHeader*)malloc(sizeof(OutHeader) + cap);
    if (!h) throw std::bad_alloc();
    h->magic = OUT_MAGIC; h->cap = cap; h->pad0 = h->pad1 = 0;
    { std::lock_guard<std::mutex> g(mu); out_bytes += cap; out_peak = std::max(out_peak, out_bytes); live.insert(h); }
    return (char*)h + sizeof(OutHeader);
  }
  // false: not a block this library handed out (or one already given back) — nothing is touched
  bool give(void* p) {
    OutHeader* h = (OutHeader*)((char*)p - sizeof(OutHeader));
    std::lock_guard<std::mutex> g(mu);
    auto lit = live.find((void*)h);
    if (lit == live.end()) return false;  // foreign pointer or double free: the header (possibly freed or unmapped memory) is never read
    live.erase(lit);
    if (h->magic != OUT_MAGIC) return false;
    out_bytes -= std::min<size_t>(out_bytes, h->cap);
    if (idle_bytes + h->cap > keep()) { h->magic = 0; free(h); return true; }
    h->magic = OUT_IDLE_MAGIC;
    idle.emplace((size_t)h->cap, (void*)h); idle_bytes += h->cap;
    return true;
  }
};
OutPool& out_pool() { static OutPool* p = new OutPool(); return *p; }  // (leaked on purpose: callers may free after static destruction)
}  // namespace
static void* out_alloc(size_t bytes) { return out_pool().take(bytes); }
static uint64_t* copy_out(const std::vector<u64>& w) {
  uint64_t* p = (uint64_t*)out_alloc(std::max<size_t>(w.size(), 1) * 8);
  if (!w.empty()) memcpy(p, w.data(), w.size() * 8);
  return p;
}
static std::vector<Ext> read_point(const uint64_t* w, size_t k) {
  std::vector<Ext> p(k);
  for (size_t i = 0; i < k; i++) { DP_REQUIRE(w[2 * i] < GL_P && w[2 * i + 1] < GL_P, DP_ERR_ARG, "non-canonical field element"); p[i] = ex(w[2 * i], w[2 * i + 1]); }
  return p;
}

extern "C" {

const char* dp_last_error(void) { return g_err.c_str(); }
void dp_free(void* p) {
  if (!p) return;
  if (!out_pool().give(p)) {  // a foreign pointer or a double free: refuse loudly, corrupt nothing
    g_err = "dp_free: not a live buffer of this library (foreign pointer or double free) — ignored";
    fprintf(stderr, "[deep-prove] %s\n", g_err.c_str());
  }
}  // every buffer the library hands out comes from out_alloc (copy_out, dp_profile_report)

int32_t dp_ctx_create(int32_t device_id, dp_ctx** out) {
  return guard([&] { DP_REQUIRE(out, DP_ERR_ARG, "null out"); Dev* d = make_hip_dev(device_id); dp_ctx* c = new dp_ctx(); c->dev = d; c->device_id = device_id; *out = c; });
}
int32_t dp_ctx_destroy(dp_ctx* ctx) { return guard([&] { if (ctx) { delete ctx->dev; delete ctx; } }); }

int32_t dp_ctx_set_throughput_mode(dp_ctx* ctx, int32_t on) {
  return guard([&] { DP_REQUIRE(ctx, DP_ERR_ARG, "null ctx"); CtxLock lk(ctx); ctx->dev->sync(); hip_dev_set_latency_mode(ctx->dev, on == 0); });
}
const char* dp_ctx_name(const dp_ctx* ctx) { return ctx ? ctx->dev->name() : ""; }

int32_t dp_profile_enable(dp_ctx* ctx, int32_t on) { return guard([&] { DP_REQUIRE(ctx, DP_ERR_ARG, "null ctx"); hip_dev_profile_enable(ctx->dev, on != 0); }); }
int32_t dp_profile_report(dp_ctx* ctx, char** json) {
  return guard([&] { DP_REQUIRE(ctx && json, DP_ERR_ARG, "bad arguments"); std::string r = hip_dev_profile_report(ctx->dev); char* p = (char*)out_alloc(r.size() + 1); memcpy(p, r.c_str(), r.size() + 1); *json = p; });
}
int32_t dp_probe_compress_rate(dp_ctx* ctx, size_t nodes, int32_t reps, double* per_second) {
  return guard([&] { DP_REQUIRE(ctx && per_second, DP_ERR_ARG, "bad arguments"); *per_second = hip_dev_probe_compress_rate(ctx->dev, nodes, reps); });
}
int32_t dp_buf_from_i64(dp_ctx* ctx, const int64_t* v, size_t n, dp_buf** out) {
  return guard([&] { DP_REQUIRE(ctx && v && out && n, DP_ERR_ARG, "bad arguments"); CtxLock lk(ctx); DBuf b = ctx->dev->alloc_persistent(n, false); ctx->dev->upload_i64(b, v); *out = new dp_buf{b}; });
}
int32_t dp_buf_upload(dp_ctx* ctx, const uint64_t* words, size_t n, int32_t is_ext, dp_buf** out) {
  return guard([&] {
    DP_REQUIRE(ctx && words && out && n, DP_ERR_ARG, "bad arguments");
    for (size_t i = 0; i < n * (is_ext ? 2 : 1); i++) DP_REQUIRE(words[i] < GL_P, DP_ERR_ARG, "non-canonical field element");
    CtxLock lk(ctx);
    DBuf b = ctx->dev->alloc_persistent(n, is_ext != 0); ctx->dev->upload(b, words); *out = new dp_buf{b};
  });
}
int32_t dp_buf_download(dp_ctx* ctx, const dp_buf* buf, uint64_t* o) { return guard([&] { DP_REQUIRE(ctx && buf && o, DP_ERR_ARG, "bad arguments"); CtxLock lk(ctx); ctx->dev->download(buf->b, o); }); }
size_t dp_buf_len(const dp_buf* buf) { return buf ? buf->b.n : 0; }
int32_t dp_buf_is_ext(const dp_buf* buf) { return buf && buf->b.ext; }
int32_t dp_buf_free(dp_ctx* ctx, dp_buf* buf) { return guard([&] { DP_REQUIRE(ctx || !buf, DP_ERR_ARG, "dp_buf_free: null context"); if (buf) { CtxLock lk(ctx); ctx->dev->free_persistent(buf->b); delete buf; } }); }

dp_transcript* dp_transcript_new(const char* label) { dp_transcript* t = new dp_transcript(); if (label) t->t.append_message(label); return t; }
void dp_transcript_free(dp_transcript* t) { delete t; }
int32_t dp_transcript_append_elements(dp_transcript* t, const uint64_t* e, size_t n) {
  return guard([&] { DP_REQUIRE(t && (e || !n), DP_ERR_ARG, "bad arguments"); for (size_t i = 0; i < n; i++) { DP_REQUIRE(e[i] < GL_P, DP_ERR_ARG, "non-canonical field element"); t->t.append_field_element(e[i]); } });
}
int32_t dp_transcript_append_message(dp_transcript* t, const uint8_t* b, size_t n) { return guard([&] { DP_REQUIRE(t && (b || !n), DP_ERR_ARG, "bad arguments"); t->t.append_message(b, n); }); }
int32_t dp_transcript_challenge(dp_transcript* t, const char* label, uint64_t out[2]) {
  return guard([&] { DP_REQUIRE(t && out, DP_ERR_ARG, "bad arguments"); Ext c = label ? t->t.get_and_append_challenge(label) : t->t.read_challenge(); out[0] = c.c0; out[1] = c.c1; });
}

int32_t dp_eq_table(dp_ctx* ctx, const uint64_t* point, uint32_t k, dp_buf** out) {
  return guard([&] {
    DP_REQUIRE(ctx && point && out && k <= 30, DP_ERR_ARG, "bad arguments");
    std::vector<Ext> p = read_point(point, k);
    DBuf b = ctx->dev->alloc_persistent(size_t(1) << k, true);
    ctx->dev->eq_table(b, p.data(), k, ex_one(), false);
    ctx->dev->sync();
    *out = new dp_buf{b};
  });
}
int32_t dp_mle_eval(dp_ctx* ctx, const dp_buf* f, const uint64_t* point, uint32_t k, uint64_t out[2]) {
  return guard([&] {
    DP_REQUIRE(ctx && f && point && out, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(k < 48 && f->b.n == (size_t(1) << k), DP_ERR_SHAPE, "MLE size does not match the point");
    if (dp_async* eng = routed_engine(ctx)) { RoutedTicket tk; routed_check(dp_mle_eval_submit(eng, f, point, k, &tk.t)); tk.wait(); routed_check(dp_ticket_values(tk.t, out, 2)); return; }
    std::vector<Ext> p = read_point(point, k); Ext r;
    ctx->dev->mle_eval_batch(&f->b, 1, p.data(), k, &r);
    out[0] = r.c0; out[1] = r.c1;
  });
}
int32_t dp_mle_fix_high(dp_ctx* ctx, const dp_buf* m, size_t rows, size_t cols, const uint64_t* point, dp_buf** out) {
  return guard([&] {
    DP_REQUIRE(ctx && m && point && out && is_pow2(rows) && is_pow2(cols) && !m->b.ext && m->b.n == rows * cols, DP_ERR_SHAPE, "fix_high: bad matrix shape");
    if (dp_async* eng = routed_engine(ctx)) { RoutedTicket tk; routed_check(dp_mle_fix_high_submit(eng, m, rows, cols, point, &tk.t)); tk.wait(); routed_check(dp_ticket_buf(tk.t, out)); return; }
    std::vector<Ext> p = read_point(point, dp_ceil_log2(rows));
    DBuf b = ctx->dev->alloc_persistent(cols, true);
    ctx->dev->fix_high(b, m->b, rows, cols, p.data());
    ctx->dev->sync();
    *out = new dp_buf{b};
  });
}

// (tables, ragged term lists) -> DevVP: `finals` follow the caller's table order, so every table enters in that order first
static void read_terms(DevVP& vp, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree, const int32_t* term_tables, int32_t nterms, const uint64_t* term_coeffs) {
  for (int i = 0; i < ntables; i++) {
    DP_REQUIRE(tables[i], DP_ERR_ARG, "null table");
    const size_t n = tables[i]->b.n;
    DP_REQUIRE(n >= 2 && (n & (n - 1)) == 0 && n <= (size_t(1) << vp.nv), DP_ERR_SHAPE, "table length must be 2^k, 1 <= k <= num_vars");
    vp.tabs.push_back(tables[i]->b);
  }
  size_t off = 0;
  for (int i = 0; i < nterms; i++) {
    const int k = term_degree[i];
    DP_REQUIRE(k >= 1 && k <= SC_MAXK, DP_ERR_SHAPE, "term degree must be 1..5");
    std::vector<DBuf> list;
    for (int j = 0; j < k; j++) { int ti = term_tables[off + j]; DP_REQUIRE(ti >= 0 && ti < ntables, DP_ERR_ARG, "term table index"); list.push_back(tables[ti]->b); }
    off += (size_t)k;
    vp.add_mle_list(list, term_coeffs ? read_point(term_coeffs + 2 * i, 1)[0] : ex_one());
  }
  DP_REQUIRE(vp.tabs.size() == (size_t)ntables, DP_ERR_ARG, "the same table was passed twice");
}
int32_t dp_sumcheck_prove(dp_ctx* ctx, uint32_t nv, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree,
                          const int32_t* term_tables, const uint64_t* term_coeffs, int32_t nterms, dp_transcript* t,
                          uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals) {
  return guard([&] {
    DP_REQUIRE(ctx && tables && term_degree && term_tables && term_coeffs && t && proof_words && proof_nwords && ntables > 0 && nterms > 0 && nv > 0, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(nv < 48, DP_ERR_SHAPE, "num_vars out of range");
    if (dp_async* eng = routed_engine(ctx)) {
      RoutedTicket tk; routed_check(dp_sumcheck_prove_submit(eng, nv, tables, ntables, term_degree, term_tables, term_coeffs, nterms, t, &tk.t)); tk.wait();
      routed_check(dp_ticket_words(tk.t, 0, proof_words, proof_nwords));
      if (finals) routed_check(dp_ticket_values(tk.t, finals, 2 * (size_t)ntables));
      return;
    }
    DevVP vp(nv);
    read_terms(vp, tables, ntables, term_degree, term_tables, nterms, term_coeffs);
    SumcheckOut so = sumcheck_prove(*ctx->dev, vp, t->t);
    Writer w; w.iop(so.proof);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
    if (finals) for (int i = 0; i < ntables; i++) { finals[2 * i] = so.finals[i].c0; finals[2 * i + 1] = so.finals[i].c1; }
  });
}

// ---- the sharded prover with its round loop in the library (csrc/sharded.h) and RCCL as the exchange
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// librccl is looked up at run time: a process that already holds an RCCL (PyTorch bundles one under the same soname) keeps
// using that copy, and a host that never shards a sumcheck never loads it
RcclApi* rccl() {  // (a pointer: the definition sits inside the extern "C" block of the ABI)
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; }
    if (!api.lib) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  });
  DP_REQUIRE(api.lib && api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy, DP_ERR_HIP, "librccl.so could not be loaded (sharded sumcheck over RCCL)");
  return &api;
}
#define RCCL_CHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw DpError(DP_ERR_HIP, std::string("RCCL: ") + (rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "error")); } while (0)
#define HIPRT_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw DpError(DP_ERR_HIP, std::string("HIP: ") + hipGetErrorString(e_)); } while (0)
}  // namespace
// one rank's communicator: device send / receive buffers the all-gather runs on, pinned host mirrors, its own stream
struct dp_dist : Exchange, Dev::ShareExchange {
  // Dev::ShareExchange: the shares never leave HBM — ncclAllGather on the proving stream itself (stream ordered behind the round's
  // reduction, in front of the kernel that adds the shares); DP_SHARDED_HOST_EXCHANGE=1 keeps round 2's host path (all_gather below)
  Dev::ShareExchange* device() override { static const bool host_only = getenv("DP_SHARDED_HOST_EXCHANGE") && atoi(getenv("DP_SHARDED_HOST_EXCHANGE")); return host_only ? nullptr : this; }
  void all_gather_device(const u64* dsend, size_t nwords, u64* drecv, void* stream_) override;
  dp_ctx* ctx = nullptr; ncclComm_t comm = nullptr; int rank_ = 0, world_ = 1; hipStream_t stream = nullptr;
  u64 *dsend = nullptr, *drecv = nullptr, *hsend = nullptr, *hrecv = nullptr; size_t cap = 0;  // words per rank
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  void reserve(size_t nwords) {
    if (nwords <= cap) return;
    release();
    cap = std::max<size_t>(nwords, 256);
    HIPRT_CHECK(hipMalloc((void**)&dsend, cap * 8)); HIPRT_CHECK(hipMalloc((void**)&drecv, cap * 8 * world_));
    HIPRT_CHECK(hipHostMalloc((void**)&hsend, cap * 8, 0)); HIPRT_CHECK(hipHostMalloc((void**)&hrecv, cap * 8 * world_, 0));
  }
  void release() { if (dsend) hipFree(dsend); if (drecv) hipFree(drecv); if (hsend) hipHostFree(hsend); if (hrecv) hipHostFree(hrecv); dsend = drecv = hsend = hrecv = nullptr; cap = 0; }
  // shares travel as u64 device words: ncclAllGather(ncclUint64) over xGMI; the mod-p sum happens on the host afterwards
  void all_gather(const u64* send, size_t nwords, u64* out) override {
    ctx->dev->bind_thread();
    reserve(nwords);
    memcpy(hsend, send, nwords * 8);
    HIPRT_CHECK(hipMemcpyAsync(dsend, hsend, nwords * 8, hipMemcpyHostToDevice, stream));
    RCCL_CHECK(rccl()->AllGather(dsend, drecv, nwords, ncclUint64, comm, stream));
    HIPRT_CHECK(hipMemcpyAsync(hrecv, drecv, nwords * 8 * world_, hipMemcpyDeviceToHost, stream));
    HIPRT_CHECK(hipStreamSynchronize(stream));
    memcpy(out, hrecv, nwords * 8 * world_);
  }
  ~dp_dist() override { release(); if (comm) rccl()->CommDestroy(comm); if (stream) hipStreamDestroy(stream); }
};
void dp_dist::all_gather_device(const u64* dsend, size_t nwords, u64* drecv, void* stream_) {
  RCCL_CHECK(rccl()->AllGather(dsend, drecv, nwords, ncclUint64, comm, (hipStream_t)stream_));
}
// the device exchange of `world` contexts of ONE process (dp_sumcheck_prove_sharded_local): device-to-device copies between the
// contexts' share buffers, rendezvous through the hub — the test double of ncclAllGather for a 1-GPU box. A rank's send buffer is
// reused every round, so nobody leaves before everyone's copies have run.
struct ThreadShareExchange : Dev::ShareExchange {
  ThreadExchangeHub& h; int r; std::vector<const u64*>& ptrs;
  ThreadShareExchange(ThreadExchangeHub& hub, int rank_, std::vector<const u64*>& p) : h(hub), r(rank_), ptrs(p) {}
  int world() const override { return h.world; }
  void barrier() {
    std::unique_lock<std::mutex> lk(h.mu);
    if (h.aborted) throw DpError(DP_ERR_HIP, "sharded sumcheck: another rank failed, the exchange is abandoned");
    const unsigned long long my = h.gen;
    if (++h.arrived == h.world) { h.arrived = 0; h.gen++; h.cv.notify_all(); }
    else { h.cv.wait(lk, [&] { return h.gen != my || h.aborted; }); if (h.gen == my) throw DpError(DP_ERR_HIP, "sharded sumcheck: another rank failed, the exchange is abandoned"); }
  }
  void all_gather_device(const u64* dsend, size_t nwords, u64* drecv, void* stream_) override {
    hipStream_t st = (hipStream_t)stream_;
    HIPRT_CHECK(hipStreamSynchronize(st));  // this rank's shares are written
    ptrs[r] = dsend;
    barrier();                               // everyone's are
    for (int g = 0; g < h.world; g++) HIPRT_CHECK(hipMemcpyAsync(drecv + (size_t)g * nwords, ptrs[g], nwords * 8, hipMemcpyDeviceToDevice, st));
    HIPRT_CHECK(hipStreamSynchronize(st));
    barrier();                               // everyone has copied: the send buffers may be overwritten
  }
};
extern "C" {
int32_t dp_dist_unique_id(uint8_t id[128]) {
  return guard([&] { DP_REQUIRE(id, DP_ERR_ARG, "null id"); static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId"); ncclUniqueId u; RCCL_CHECK(rccl()->GetUniqueId(&u)); memcpy(id, &u, 128); });
}
int32_t dp_dist_init(dp_ctx* ctx, const uint8_t id[128], int32_t rank, int32_t world, dp_dist** out) {
  return guard([&] {
    DP_REQUIRE(ctx && id && out && world >= 1 && rank >= 0 && rank < world && (world & (world - 1)) == 0, DP_ERR_ARG, "bad arguments (world must be a power of two)");
    std::unique_ptr<dp_dist> d(new dp_dist());
    d->ctx = ctx; d->rank_ = rank; d->world_ = world;
    ctx->dev->bind_thread();
    HIPRT_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    ncclUniqueId u; memcpy(&u, id, 128);
    RCCL_CHECK(rccl()->CommInitRank(&d->comm, world, u, rank));
    *out = d.release();
  });
}
int32_t dp_dist_free(dp_dist* d) { return guard([&] { delete d; }); }
static void sharded_out(const SumcheckOut& so, size_t ntables, uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals) {
  Writer w; w.iop(so.proof);
  *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  if (finals) for (size_t i = 0; i < ntables; i++) { finals[2 * i] = so.finals[i].c0; finals[2 * i + 1] = so.finals[i].c1; }
}
int32_t dp_sumcheck_prove_sharded(dp_ctx* ctx, dp_dist* dist, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree,
                                  const int32_t* term_tables, const uint64_t* term_coeffs, int32_t nterms, dp_transcript* t,
                                  uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals) {
  return guard([&] {
    DP_REQUIRE(ctx && tables && term_degree && term_tables && term_coeffs && t && proof_words && proof_nwords && ntables > 0 && nterms > 0 && num_vars > 0 && num_vars < 48, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(!dist || dist->ctx == ctx, DP_ERR_ARG, "the communicator belongs to another context");
    SoloExchange solo;
    Exchange& xch = dist ? static_cast<Exchange&>(*dist) : solo;
    unsigned k = 0; while ((1 << k) < xch.world()) k++;
    DP_REQUIRE(num_vars > k, DP_ERR_SHAPE, "more ranks than table entries");
    DevVP vp(num_vars - k);
    read_terms(vp, tables, ntables, term_degree, term_tables, nterms, term_coeffs);
    SumcheckOut so = sumcheck_prove_sharded(*ctx->dev, xch, num_vars, vp, t->t);
    sharded_out(so, (size_t)ntables, proof_words, proof_nwords, finals);
  });
}
/* `world` contexts driven from ONE process (one thread per rank, exchange in host memory): several contexts on one GPU, tests.
 * tables: world x ntables handles, rank-major; transcripts: one per rank, all in the same state. Rank 0's output is returned;
 * every rank's proof is compared with it (DP_ERR_HIP if they differ). */
int32_t dp_sumcheck_prove_sharded_local(dp_ctx* const* ctxs, int32_t world, uint32_t num_vars, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree,
                                        const int32_t* term_tables, const uint64_t* term_coeffs, int32_t nterms, dp_transcript* const* transcripts,
                                        uint64_t** proof_words, size_t* proof_nwords, uint64_t* finals) {
  return guard([&] {
    DP_REQUIRE(ctxs && tables && term_degree && term_tables && term_coeffs && transcripts && proof_words && proof_nwords && world >= 1 && (world & (world - 1)) == 0 && num_vars < 48, DP_ERR_ARG, "bad arguments");
    unsigned k = 0; while ((1 << k) < world) k++;
    DP_REQUIRE(num_vars > k, DP_ERR_SHAPE, "more ranks than table entries");
    // every rank's inputs are checked BEFORE a thread exists: a rank that cannot start must not leave the others in an exchange
    for (int g = 0; g < world; g++) {
      DP_REQUIRE(ctxs[g] && transcripts[g], DP_ERR_ARG, "null context / transcript");
      DevVP probe(num_vars - k);
      read_terms(probe, tables + (size_t)g * ntables, ntables, term_degree, term_tables, nterms, term_coeffs);
      for (const DBuf& b : probe.tabs) DP_REQUIRE(b.n == (size_t(1) << (num_vars - k)), DP_ERR_SHAPE, "sharded sumcheck: every local table has 2^(num_vars - log2 world) entries");
    }
    ThreadExchangeHub hub(world);
    std::vector<const u64*> share_ptrs(world, nullptr);
    static const bool host_only = getenv("DP_SHARDED_HOST_EXCHANGE") && atoi(getenv("DP_SHARDED_HOST_EXCHANGE"));
    std::vector<SumcheckOut> outs(world);
    std::vector<std::string> errs(world);
    std::vector<std::thread> th;
    for (int g = 0; g < world; g++) th.emplace_back([&, g] {
      try {
        DP_REQUIRE(ctxs[g] && transcripts[g], DP_ERR_ARG, "null context / transcript");
        ctxs[g]->dev->bind_thread(); ctxs[g]->dev->pin_thread();
        DevVP vp(num_vars - k);
        read_terms(vp, tables + (size_t)g * ntables, ntables, term_degree, term_tables, nterms, term_coeffs);
        ThreadExchange xch(hub, g);
        ThreadShareExchange sx(hub, g, share_ptrs);
        if (!host_only) xch.dev_x = &sx;  // the shares stay on the device, as over RCCL (k_shares_sum_publish: one host wait per round)
        outs[g] = sumcheck_prove_sharded(*ctxs[g]->dev, xch, num_vars, vp, transcripts[g]->t);
      } catch (const std::exception& e) { errs[g] = e.what(); if (errs[g].empty()) errs[g] = "error"; hub.abort(); }  // wakes the ranks waiting for this one
    });
    for (auto& x : th) x.join();
    for (int g = 0; g < world; g++) if (!errs[g].empty() && errs[g].find("another rank failed") == std::string::npos) throw DpError(DP_ERR_HIP, "rank " + std::to_string(g) + ": " + errs[g]);
    for (int g = 0; g < world; g++) if (!errs[g].empty()) throw DpError(DP_ERR_HIP, "rank " + std::to_string(g) + ": " + errs[g]);
    Writer w0; w0.iop(outs[0].proof);
    for (int g = 1; g < world; g++) { Writer w; w.iop(outs[g].proof); DP_REQUIRE(w.w == w0.w, DP_ERR_HIP, "ranks produced different proofs"); }
    sharded_out(outs[0], (size_t)ntables, proof_words, proof_nwords, finals);
  });
}
}  // extern "C"

// ---- round-level sumcheck (the unit a sharded prover exchanges between devices, sumcheck/src/prover.rs:37-321)
struct dp_sc_session { dp_ctx* ctx; std::vector<DBuf> tabs; std::vector<ScTerm> terms; size_t nraw; size_t mark; bool first; size_t len; };
int32_t dp_sc_session_new(dp_ctx* ctx, uint32_t nv, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree,
                          const int32_t* term_tables, int32_t nterms, dp_sc_session** out) {
  return guard([&] {
    DP_REQUIRE(ctx && tables && term_degree && term_tables && out && ntables > 0 && nterms > 0 && nv > 0, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(nv < 48, DP_ERR_SHAPE, "num_vars out of range");
    std::unique_ptr<dp_sc_session> s(new dp_sc_session());
    s->ctx = ctx; s->nraw = 0; s->first = true; s->len = size_t(1) << nv;
    DevVP vp(nv);
    read_terms(vp, tables, ntables, term_degree, term_tables, nterms, nullptr);
    s->terms = vp.terms;
    for (const ScTerm& st : s->terms) s->nraw += st.k + 1;
    s->mark = ctx->dev->mark();
    for (const DBuf& b : vp.tabs) s->tabs.push_back(tile_to(*ctx->dev, b, s->len));  // short tables: see tile_to (sumcheck.h)
    *out = s.release();
  });
}
/* fold every table with r_prev (NULL in the first round), then the raw per-term sums: (degree_i + 1) extension values per
 * term, terms back to back (2 words each). Fails once the tables are down to one element: call dp_sc_session_finish. */
int32_t dp_sc_session_round(dp_sc_session* s, const uint64_t* r_prev, uint64_t* raw_out, size_t* nraw_ext) {
  return guard([&] {
    DP_REQUIRE(s && raw_out && (s->first ? r_prev == nullptr : r_prev != nullptr), DP_ERR_ARG, "bad arguments");
    Ext r = ex_zero(); if (r_prev) r = read_point(r_prev, 1)[0];
    std::vector<Ext> raw(s->nraw);
    s->ctx->dev->sc_round(s->tabs.data(), (int)s->tabs.size(), r_prev ? &r : nullptr, s->terms.data(), (int)s->terms.size(), raw.data());
    s->first = false;
    for (size_t i = 0; i < raw.size(); i++) { raw_out[2 * i] = raw[i].c0; raw_out[2 * i + 1] = raw[i].c1; }
    if (nraw_ext) *nraw_ext = raw.size();
  });
}
/* the last fold: one value per table (get_mle_final_evaluations order); ends the session's device work */
int32_t dp_sc_session_finish(dp_sc_session* s, const uint64_t r_last[2], uint64_t* finals) {
  return guard([&] {
    DP_REQUIRE(s && r_last && finals && !s->first, DP_ERR_ARG, "bad arguments");
    std::vector<Ext> f(s->tabs.size());
    s->ctx->dev->sc_finish(s->tabs.data(), (int)s->tabs.size(), read_point(r_last, 1)[0], f.data());
    for (size_t i = 0; i < f.size(); i++) { finals[2 * i] = f[i].c0; finals[2 * i + 1] = f[i].c1; }
  });
}
int32_t dp_sc_session_free(dp_sc_session* s) { return guard([&] { if (s) { s->ctx->dev->release(s->mark); delete s; } }); }

int32_t dp_logup_prove(dp_ctx* ctx, const dp_buf* const* columns, int32_t ncols, int32_t cpi, const dp_buf* mult,
                       const uint64_t cc[2], const uint64_t csc[2], dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    DP_REQUIRE(ctx && columns && ncols > 0 && cc && csc && t && proof_words && proof_nwords && cpi > 0, DP_ERR_ARG, "bad arguments");
    if (dp_async* eng = routed_engine(ctx)) { RoutedTicket tk; routed_check(dp_logup_prove_submit(eng, columns, ncols, cpi, mult, cc, csc, t, &tk.t)); tk.wait(); routed_check(dp_ticket_words(tk.t, 0, proof_words, proof_nwords)); return; }
    LogUpInputDev in; in.is_table = mult != nullptr; in.columns_per_instance = cpi;
    for (int i = 0; i < ncols; i++) { DP_REQUIRE(columns[i], DP_ERR_ARG, "null column"); in.columns.push_back(columns[i]->b); }
    if (mult) { DP_REQUIRE(!mult->b.ext && mult->b.n == in.columns[0].n, DP_ERR_SHAPE, "multiplicities shape"); in.multiplicities = mult->b; }
    in.constant_challenge = read_point(cc, 1)[0]; in.column_separation_challenge = read_point(csc, 1)[0];
    LogUpProof p = logup_batch_prove(*ctx->dev, in, t->t);
    Writer w; w.logup(p);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}

int32_t dp_sumcheck_verify(uint32_t nv, uint32_t max_degree, const uint64_t claimed_sum[2], const uint64_t* proof_words, size_t proof_nwords,
                           dp_transcript* t, uint64_t* point, uint64_t expected_evaluation[2]) {
  return guard([&] {
    DP_REQUIRE(claimed_sum && proof_words && t && point && expected_evaluation && max_degree >= 1, DP_ERR_ARG, "bad arguments");
    Reader r(proof_words, proof_nwords); IOPProof p = r.iop();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    SubClaim sc = sumcheck_verify(read_point(claimed_sum, 1)[0], p, nv, max_degree, t->t);
    for (size_t i = 0; i < sc.point.size(); i++) { point[2 * i] = sc.point[i].c0; point[2 * i + 1] = sc.point[i].c1; }
    expected_evaluation[0] = sc.expected_evaluation.c0; expected_evaluation[1] = sc.expected_evaluation.c1;
  });
}
int32_t dp_logup_verify(const uint64_t* proof_words, size_t proof_nwords, int32_t num_instances, const uint64_t cc[2], const uint64_t csc[2],
                        dp_transcript* t, uint64_t* numerators, uint64_t* denominators, uint64_t** claims_words, size_t* claims_nwords) {
  return guard([&] {
    DP_REQUIRE(proof_words && num_instances > 0 && cc && csc && t, DP_ERR_ARG, "bad arguments");
    Reader r(proof_words, proof_nwords); LogUpProof p = r.logup();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    LogUpVerifierClaim vc = verify_logup_proof(p, (size_t)num_instances, read_point(cc, 1)[0], read_point(csc, 1)[0], t->t);
    for (size_t i = 0; i < vc.numerators.size(); i++) {
      if (numerators) { numerators[2 * i] = vc.numerators[i].c0; numerators[2 * i + 1] = vc.numerators[i].c1; }
      if (denominators) { denominators[2 * i] = vc.denominators[i].c0; denominators[2 * i + 1] = vc.denominators[i].c1; }
    }
    if (claims_words && claims_nwords) {
      Writer w; w.u(vc.claims.size()); for (auto& c : vc.claims) w.claim(c);
      *claims_words = copy_out(w.w); *claims_nwords = w.w.size();
    }
  });
}

int32_t dp_pcs_setup(dp_ctx* ctx, size_t max_poly_size) {
  return guard([&] { DP_REQUIRE(ctx && is_pow2(max_poly_size), DP_ERR_ARG, "max_poly_size must be a power of two"); ctx->dev->pcs_init(dp_ceil_log2(max_poly_size)); });
}
int32_t dp_pcs_commit(dp_ctx* ctx, const dp_buf* poly, dp_commit** out, uint64_t root[4]) {
  return guard([&] {
    DP_REQUIRE(ctx && poly && out, DP_ERR_ARG, "bad arguments");
    if (dp_async* eng = routed_engine(ctx)) { RoutedTicket tk; routed_check(dp_pcs_commit_submit(eng, poly, &tk.t)); tk.wait(); routed_check(dp_ticket_commit(tk.t, out, root)); return; }
    CtxLock lk(ctx);  // callable from several threads at once (queued on the context's stream)
    DevCommit c = ctx->dev->commit(poly->b, true);
    if (root) for (int k = 0; k < 4; k++) root[k] = c.tree.root.v[k];
    *out = new dp_commit{c};
  });
}
int32_t dp_pcs_commit_free(dp_ctx* ctx, dp_commit* c) {
  return guard([&] {
    if (!c) return;
    CtxLock lk(ctx);
    DevCommit& d = c->c;  // the evaluation table belongs to the caller's dp_buf
    if (d.bh_evals.p == d.evals.p) d.bh_evals.p = nullptr;
    if (d.tree.leaves.p == d.evals.p) d.tree.leaves.p = nullptr;
    d.evals.p = nullptr;
    ctx->dev->free_commit(d);
    delete c;
  });
}
/* PCS::get_pure_commitment (mpcs/src/basefold.rs:459-461) */
int32_t dp_pcs_commitment(const dp_commit* c, uint64_t root[4], uint32_t* num_vars, int32_t* is_base) {
  return guard([&] {
    DP_REQUIRE(c && root, DP_ERR_ARG, "bad arguments");
    for (int k = 0; k < 4; k++) root[k] = c->c.tree.root.v[k];
    if (num_vars) *num_vars = c->c.nv;
    if (is_base) *is_base = c->c.is_base ? 1 : 0;
  });
}
/* PCS::open (mpcs/src/basefold.rs:466-544): one committed polynomial at one point. Up to trivial_num_vars() = 7 variables (the only
 * case zkml itself reaches, zkml/src/commit/context.rs:395) the proof is the evaluation table and the transcript is not touched;
 * above, the commit phase (sumcheck interleaved with FRI folds) and 200 queries run on the device. */
int32_t dp_pcs_open(dp_ctx* ctx, const dp_commit* comm, const uint64_t* point, uint32_t num_vars, const uint64_t eval[2], dp_transcript* t,
                    uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    DP_REQUIRE(ctx && comm && point && proof_words && proof_nwords, DP_ERR_ARG, "bad arguments");
    (void)eval;  // "Opening does not need eval, except for sanity check" (basefold.rs:471)
    DP_REQUIRE(num_vars == comm->c.nv, DP_ERR_SHAPE, "point length != the polynomial's number of variables");
    DP_REQUIRE(comm->c.trivial() || t, DP_ERR_ARG, "a non-trivial opening needs the transcript");
    CtxLock lk(ctx);
    BasefoldProof p = comm->c.trivial() ? pcs_open_trivial(*ctx->dev, comm->c) : pcs_open(*ctx->dev, 64, comm->c, read_point(point, num_vars), t->t);  // (size check against the parameters: commit())
    Writer w; w.basefold(p);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
/* PCS::verify (mpcs/src/basefold.rs:863-962): trivial proof = Merkle root of the opened table + its evaluation (transcript untouched);
 * otherwise the commit-phase replay, 200 authenticated fold chains and the sumcheck chain. Host only. */
int32_t dp_pcs_verify(size_t max_poly_size, const uint64_t root[4], uint32_t num_vars, int32_t is_base, const uint64_t* point, const uint64_t eval[2],
                      const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t) {
  return guard([&] {
    DP_REQUIRE(root && point && eval && proof_words && num_vars <= 64 && is_pow2(max_poly_size), DP_ERR_ARG, "bad arguments");
    Commitment c; for (int k = 0; k < 4; k++) c.root.v[k] = root[k]; c.num_vars = num_vars; c.is_base = is_base != 0;
    Reader r(proof_words, proof_nwords); BasefoldProof p = r.basefold();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    if (p.is_trivial()) { DP_REQUIRE(num_vars <= PCS_BASECODE_LOG, DP_ERR_VERIFY, "trivial proof for a non-trivial commitment"); pcs_verify_trivial(c, read_point(point, num_vars), read_point(eval, 1)[0], p); return; }
    DP_REQUIRE(t, DP_ERR_ARG, "a non-trivial opening needs the transcript");
    VerifierParams vp; vp.full_log = dp_ceil_log2(max_poly_size);
    std::vector<MerkleJob> jobs;  // (the 200 x (rounds + 1) paths are recorded, then authenticated together: eight side by side on AVX-512 CPUs)
    merkle_sink() = &jobs;
    try { pcs_verify(vp, c, read_point(point, num_vars), read_point(eval, 1)[0], p, t->t); } catch (...) { merkle_sink() = nullptr; throw; }
    merkle_sink() = nullptr;
    DP_REQUIRE(merkle_jobs_ok(jobs, verify_threads()), DP_ERR_VERIFY, "merkle path does not authenticate against the root");
  });
}
static void read_claims(int32_t n, const uint64_t* points_flat, const uint64_t* evals, const std::vector<unsigned>& nvs, std::vector<std::vector<Ext>>& pts, std::vector<Ext>& evs) {
  size_t off = 0;
  for (int i = 0; i < n; i++) { DP_REQUIRE(nvs[i] <= 64, DP_ERR_ARG, "point too long"); pts.push_back(read_point(points_flat + off, nvs[i])); off += 2 * (size_t)nvs[i]; evs.push_back(read_point(evals + 2 * i, 1)[0]); }
}
int32_t dp_pcs_batch_open(dp_ctx* ctx, const dp_commit* const* comms, int32_t n, const uint64_t* points_flat, const uint64_t* evals,
                          dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    DP_REQUIRE(ctx && comms && n > 0 && points_flat && evals && t && proof_words && proof_nwords, DP_ERR_ARG, "bad arguments");
    if (dp_async* eng = routed_engine(ctx)) { RoutedTicket tk; routed_check(dp_pcs_batch_open_submit(eng, comms, n, points_flat, evals, t, &tk.t)); tk.wait(); routed_check(dp_ticket_words(tk.t, 0, proof_words, proof_nwords)); return; }
    CtxLock lk(ctx);
    std::vector<unsigned> nvs; for (int i = 0; i < n; i++) { DP_REQUIRE(comms[i], DP_ERR_ARG, "null commitment"); nvs.push_back(comms[i]->c.nv); }
    std::vector<std::vector<Ext>> pts; std::vector<Ext> evs;
    read_claims(n, points_flat, evals, nvs, pts, evs);
    std::vector<OpenClaim> oc;
    for (int i = 0; i < n; i++) oc.push_back({&comms[i]->c, pts[i], evs[i]});
    unsigned L = 0; for (auto v : nvs) L = std::max(L, v);
    BasefoldProof p = pcs_batch_open(*ctx->dev, 64, oc, t->t);  // the size check against the PCS parameters happens in commit()
    Writer w; w.basefold(p);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
int32_t dp_pcs_batch_verify(size_t max_poly_size, const uint64_t* roots, const uint32_t* num_vars, const int32_t* is_base, int32_t n,
                            const uint64_t* points_flat, const uint64_t* evals, const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t) {
  return guard([&] {
    DP_REQUIRE(roots && num_vars && is_base && n > 0 && points_flat && evals && proof_words && t && is_pow2(max_poly_size), DP_ERR_ARG, "bad arguments");
    std::vector<unsigned> nvs(num_vars, num_vars + n);
    std::vector<std::vector<Ext>> pts; std::vector<Ext> evs;
    read_claims(n, points_flat, evals, nvs, pts, evs);
    std::vector<VerifyClaim> vc;
    for (int i = 0; i < n; i++) { Commitment c; for (int k = 0; k < 4; k++) c.root.v[k] = roots[4 * i + k]; c.num_vars = nvs[i]; c.is_base = is_base[i] != 0; vc.push_back({c, pts[i], evs[i]}); }
    Reader r(proof_words, proof_nwords); BasefoldProof p = r.basefold();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    VerifierParams vp; vp.full_log = dp_ceil_log2(max_poly_size);
    std::vector<MerkleJob> jobs;
    merkle_sink() = &jobs;
    try { pcs_batch_verify(vp, vc, p, t->t); } catch (...) { merkle_sink() = nullptr; throw; }
    merkle_sink() = nullptr;
    DP_REQUIRE(merkle_jobs_ok(jobs, verify_threads()), DP_ERR_VERIFY, "merkle path does not authenticate against the root");
  });
}

/* PCS::batch_open / batch_verify over a general Evaluation list (mpcs/src/basefold.rs:546-770, 964-1098) */
static void read_eval_lists(const uint64_t* points_flat, const uint32_t* point_num_vars, int32_t n_points, const uint32_t* eval_poly, const uint32_t* eval_point, const uint64_t* eval_values,
                            int32_t n_evals, int32_t n_polys, std::vector<std::vector<Ext>>& pts, std::vector<size_t>& ep, std::vector<size_t>& eq, std::vector<Ext>& ev) {
  size_t off = 0;
  for (int i = 0; i < n_points; i++) { DP_REQUIRE(point_num_vars[i] <= 64, DP_ERR_ARG, "point too long"); pts.push_back(read_point(points_flat + off, point_num_vars[i])); off += 2 * (size_t)point_num_vars[i]; }
  for (int i = 0; i < n_evals; i++) {
    DP_REQUIRE(eval_poly[i] < (uint32_t)n_polys && eval_point[i] < (uint32_t)n_points, DP_ERR_ARG, "evaluation refers to a missing polynomial or point");
    ep.push_back(eval_poly[i]); eq.push_back(eval_point[i]); ev.push_back(read_point(eval_values + 2 * (size_t)i, 1)[0]);
  }
}
int32_t dp_pcs_batch_open_evals(dp_ctx* ctx, const dp_commit* const* comms, int32_t n_polys, const uint64_t* points_flat, const uint32_t* point_num_vars, int32_t n_points,
                                const uint32_t* eval_poly, const uint32_t* eval_point, const uint64_t* eval_values, int32_t n_evals, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    DP_REQUIRE(ctx && comms && n_polys > 0 && points_flat && point_num_vars && n_points > 0 && eval_poly && eval_point && eval_values && n_evals > 0 && t && proof_words && proof_nwords, DP_ERR_ARG, "bad arguments");
    std::vector<std::vector<Ext>> pts; std::vector<size_t> ep, eq; std::vector<Ext> ev;
    read_eval_lists(points_flat, point_num_vars, n_points, eval_poly, eval_point, eval_values, n_evals, n_polys, pts, ep, eq, ev);
    std::vector<const DevCommit*> cs;
    for (int i = 0; i < n_polys; i++) { DP_REQUIRE(comms[i], DP_ERR_ARG, "null commitment"); cs.push_back(&comms[i]->c); }
    std::vector<EvalClaim> evals;
    for (int i = 0; i < n_evals; i++) evals.push_back({ep[i], eq[i], ev[i]});
    CtxLock lk(ctx);
    BasefoldProof p = pcs_batch_open_evals(*ctx->dev, 64, cs, pts, evals, t->t);
    Writer w; w.basefold(p);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
// ================================================================================================ asynchronous seam calls
// The reference reaches its seams from rayon workers: PCS::commit per witness column (zkml/src/commit/context.rs:79-103, layers/activation.rs:294-304),
// one prove_parallel per layer (sumcheck/src/prover.rs:498-501), logup batch_prove per lookup, one batch_open per proof (mpcs/src/lib.rs:111-226). A
// host bound to these seams with one blocking call per thread keeps a dozen calls in flight and gets a sixth of dp_model_prove_batch's rate (bench.py
// `seam_level`, round 3): what makes the batch path fast — hundreds of calls in flight per host thread (fibers, fiber.h) and launches merged across
// calls of the same shape (cohorts, hip_dev.hip) — sits above the seams. dp_async gives it to seam-level hosts: submit returns a ticket at once; engine
// threads run the calls as fibers on worker contexts (own arena, own stream, the PCS tables of the owning context) in throughput mode (device-side
// Fiat-Shamir, fused protocol tails) and prove calls of IDENTICAL SHAPE that are queued together in lock step, launch for launch merged into one
// (the launch sequence of every seam depends on shapes only, never on data). Results are bit-identical to the blocking calls.
struct dp_ticket {
  dp_ctx* owner = nullptr;    // the context of the engine the call was submitted to (its device pool holds what the call allocated)
  std::atomic<int> state{0};  // 0 queued / running, 1 done, < 0 failed
  std::string err;
  std::function<void(Dev&, dp_ticket&)> body;
  uint64_t sig = 0;
  std::vector<uint64_t> words[2]; std::vector<uint64_t> finals;  // results: up to two word streams (the seam's malloc'ed outputs) + fixed-size values
  dp_commit* commit = nullptr; uint64_t root[4] = {0, 0, 0, 0}; dp_buf* buf = nullptr;
  std::chrono::steady_clock::time_point t_submit;
};
struct dp_async {
  dp_ctx* ctx = nullptr; size_t max_in_flight = 0, arena = 0; size_t group_max = 32, groups_per_thread = 1; double linger_us = 100.0;  // measured (profiles/r04_seam_async.txt): groups per thread 1 against 3: 111 against 52 proofs/s; groups of up to 64, a linger of 300 us, 22 engine threads: no gain
  std::mutex mu; std::condition_variable cv; std::deque<dp_ticket*> queue; bool stop = false;
  std::vector<std::unique_ptr<Dev>> workers; std::vector<Dev*> free_workers;
  std::vector<std::thread> threads;
  std::atomic<size_t> ngroups{0}, njobs{0}, nmerged{0}, qsize{0}; std::atomic<unsigned long long> body_us{0}, queue_us{0};  // (summed over calls: time inside the call, time queued before it)  // qsize: queue length, readable without the lock
  ~dp_async() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
    for (auto& t : threads) if (t.joinable()) t.join();
  }
};
namespace {
uint64_t sig_mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0x100000001B3ull; }
uint64_t sig_buf(uint64_t h, const DBuf& b) { return sig_mix(sig_mix(h, b.n), b.ext ? 2 : 1); }
// One engine thread: a fiber scheduler that never blocks in a call. It holds up to `groups_per_thread` GROUPS at a time (a group = the queued calls of one
// shape it found together, proved in lock step on one cohort stream), admits a new group whenever it has room and workers are free, and gives every live
// fiber a turn per pass. A thread with nothing to run sleeps on the queue's condition variable.
struct AsyncGroup { dp::Cohort* cohort = nullptr; bool merged = false; std::vector<Dev*> devs; size_t live = 0; };
void async_thread(dp_async* a) {
  a->ctx->dev->pin_thread();  // (an engine thread is the library's own: it keeps to the CPUs of the GPU's NUMA node)
  std::vector<dp::Cohort*> idle_cohorts;
  std::vector<std::unique_ptr<AsyncGroup>> groups;
  FiberSched sched;
  FiberSched*& cur = fiber_current_sched();
  cur = &sched;
  auto admit = [&](bool may_sleep) {
    std::vector<dp_ticket*> group; std::vector<Dev*> devs;
    {
      std::unique_lock<std::mutex> lk(a->mu);
      if (may_sleep) a->cv.wait(lk, [&] { return a->stop || (!a->queue.empty() && !a->free_workers.empty()); });
      if (a->queue.empty() || a->free_workers.empty()) return !(a->stop && a->queue.empty());
      const uint64_t sig = a->queue.front()->sig;
      auto take = [&] {
        for (auto it = a->queue.begin(); it != a->queue.end() && group.size() < a->group_max && !a->free_workers.empty();) {
          if ((*it)->sig == sig) { group.push_back(*it); devs.push_back(a->free_workers.back()); a->free_workers.pop_back(); it = a->queue.erase(it); a->qsize--; } else ++it;
        }
      };
      take();
      // linger (only a thread with nothing else to run, only for a lone call): a host that drives many proofs in step submits the same call of each of
      // them within microseconds — give the rest of the wave a moment to arrive
      if (may_sleep && group.size() == 1 && a->linger_us > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds((long long)a->linger_us);
        while (group.size() < a->group_max && !a->stop && std::chrono::steady_clock::now() < until) { a->cv.wait_until(lk, until); take(); }
      }
    }
    if (group.empty()) return true;
    a->ngroups++; a->njobs += group.size(); if (group.size() > 1) a->nmerged += group.size();
    std::unique_ptr<AsyncGroup> g(new AsyncGroup());
    g->merged = group.size() > 1; g->devs = devs; g->live = group.size();
    try {
      if (g->merged) { if (idle_cohorts.empty()) idle_cohorts.push_back(hip_cohort_new()); g->cohort = idle_cohorts.back(); idle_cohorts.pop_back(); }
      for (Dev* d : devs) { hip_dev_set_latency_mode(d, false); if (g->merged) hip_dev_cohort_attach(d, g->cohort); }
    } catch (const std::exception& e) {
      for (dp_ticket* t : group) { t->err = e.what(); t->state.store(DP_ERR_HIP, std::memory_order_release); }
      if (g->cohort) idle_cohorts.push_back(g->cohort);
      { std::lock_guard<std::mutex> gl(a->mu); for (Dev* d : devs) a->free_workers.push_back(d); }
      a->cv.notify_all();
      return true;
    }
    AsyncGroup* gp = g.get();
    for (size_t i = 0; i < group.size(); i++) {
      dp_ticket* t = group[i]; Dev* d = devs[i];
      fiber_spawn(sched, [t, d, gp, a] {
        int code = 1;
        const auto tb = std::chrono::steady_clock::now();
        a->queue_us += (unsigned long long)std::chrono::duration<double, std::micro>(tb - t->t_submit).count();
        // the worker's arena mark is taken OUTSIDE the try: whatever the call throws, its kernels are drained (best effort) and its arena is given back before
        // the ticket is published — the caller may free the call's input tables as soon as it sees the ticket's state
        size_t mk = 0; bool marked = false;
        try { d->bind_thread(); mk = d->mark(); marked = true; t->body(*d, *t); d->sync(); d->release(mk); marked = false; }
        catch (const DpError& e) { t->err = e.what(); code = e.code; }
        catch (const std::bad_alloc&) { t->err = "host out of memory"; code = DP_ERR_OOM; }
        catch (const std::exception& e) { t->err = e.what(); code = DP_ERR_ARG; }
        catch (...) { t->err = "unknown error"; code = DP_ERR_ARG; }
        if (code != 1) {
          try { d->abort_call(); } catch (...) {}
          try { d->sync(); } catch (...) {}
          if (marked) { try { d->release(mk); } catch (...) {} }
          // a body that had already handed out its commitment / table when a later step (the final sync) failed: the device objects go back now — the ticket of
          // a failed call holds nothing (dp_ticket_free only deletes wrappers)
          if (t->commit) { try { d->free_commit(t->commit->c); } catch (...) {} delete t->commit; t->commit = nullptr; }
          if (t->buf) { try { d->free_persistent(t->buf->b); } catch (...) {} delete t->buf; t->buf = nullptr; }
        }
        if (gp->merged) { try { hip_dev_cohort_detach(d); } catch (const std::exception& e) { if (code == 1) { t->err = e.what(); code = DP_ERR_HIP; } } }
        t->body = nullptr;
        a->body_us += (unsigned long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb).count();
        gp->live--;
        t->state.store(code, std::memory_order_release);
      });
    }
    groups.push_back(std::move(g));
    return true;
  };
  for (;;) {
    if (groups.size() < a->groups_per_thread && (groups.empty() || a->qsize.load(std::memory_order_relaxed))) { if (!admit(groups.empty())) break; }
    if (groups.empty()) continue;
    // one turn for every live fiber
    for (size_t i = 0; i < sched.fibers.size(); i++) {
      Fiber* f = sched.fibers[i].get();
      if (f->done) continue;
      sched.cur = f;
      dp_fiber_switch(&sched.main_sp, f->sp);
      sched.cur = nullptr;
    }
    // finished groups give their workers and their cohort back; finished fibers give their stacks back
    bool freed = false;
    for (size_t gi = 0; gi < groups.size();) {
      AsyncGroup* g = groups[gi].get();
      if (g->live) { gi++; continue; }
      if (g->merged) { try { hip_cohort_drain(g->cohort); } catch (...) {} idle_cohorts.push_back(g->cohort); }
      { std::lock_guard<std::mutex> gl(a->mu); for (Dev* d : g->devs) a->free_workers.push_back(d); }
      groups.erase(groups.begin() + (long)gi); freed = true;
    }
    if (freed) {
      a->cv.notify_all();
      sched.fibers.erase(std::remove_if(sched.fibers.begin(), sched.fibers.end(), [](const std::unique_ptr<Fiber>& f) { return f->done; }), sched.fibers.end());
    }
    _mm_pause();
  }
  cur = nullptr;
  for (dp::Cohort* c : idle_cohorts) hip_cohort_free(c);
}
int32_t async_submit(dp_async* a, dp_ticket* t, dp_ticket** out) {
  *out = t; t->owner = a->ctx;
  t->t_submit = std::chrono::steady_clock::now();
  // the tables of the call may have been uploaded through the owning context a moment ago: in throughput mode copies of up to 2 MB return before they ran on ITS
  // stream, and the call reads the tables on a worker's stream — nothing orders the two but this
  // (a flush that fails — device timeout, HIP error — must not leave a ticket that is neither queued nor finished: the caller holds it already and would
  // wait for ever, and dp_ticket_free refuses a running call: the ticket completes with the error instead)
  try { CtxLock lk(a->ctx); a->ctx->dev->flush_uploads(); }
  catch (const DpError& e) { t->err = e.what(); t->body = nullptr; t->state.store(e.code, std::memory_order_release); return DP_OK; }
  catch (const std::exception& e) { t->err = e.what(); t->body = nullptr; t->state.store(DP_ERR_HIP, std::memory_order_release); return DP_OK; }
  { std::lock_guard<std::mutex> g(a->mu); a->queue.push_back(t); a->qsize++; }
  a->cv.notify_all();  // (notify_one could wake a thread lingering for a DIFFERENT shape, which leaves this ticket queued while idle threads sleep)
  return DP_OK;
}
}  // namespace
int32_t dp_async_create(dp_ctx* ctx, int32_t max_in_flight, size_t worker_arena_bytes, dp_async** out) {
  return guard([&] {
    DP_REQUIRE(ctx && out && max_in_flight >= 1 && max_in_flight <= 4096, DP_ERR_ARG, "dp_async_create: 1..4096 calls in flight");
    std::unique_ptr<dp_async> a(new dp_async());
    a->ctx = ctx; a->max_in_flight = (size_t)max_in_flight; a->arena = worker_arena_bytes ? worker_arena_bytes : (size_t(512) << 20);
    CtxLock lk(ctx);
    size_t fr = 0, tot = 0; hip_mem_info(ctx->device_id, &fr, &tot);
    const size_t fit = (size_t)(0.9 * (double)fr) / (a->arena + (size_t(16) << 20));
    DP_REQUIRE(fit >= 1, DP_ERR_OOM, "dp_async_create: not one worker arena fits in the free device memory");
    a->max_in_flight = std::min(a->max_in_flight, fit);
    for (size_t i = 0; i < a->max_in_flight; i++) {
      std::unique_ptr<Dev> w(make_hip_worker(ctx->device_id, a->arena));
      try { hip_dev_pcs_share(w.get(), ctx->dev); } catch (const DpError&) {}  // (no dp_pcs_setup yet: the PCS calls of this engine will say so)
      a->free_workers.push_back(w.get()); a->workers.push_back(std::move(w));
    }
    const char* te = getenv("DP_HOST_THREADS");
    size_t nth = te ? (size_t)std::max(1, atoi(te)) : (size_t)std::max(1.0, host_cpu_budget() - 2.0);
    nth = std::min<size_t>(std::min<size_t>(nth, 22), a->max_in_flight);
    for (size_t i = 0; i < nth; i++) a->threads.emplace_back(async_thread, a.get());
    *out = a.release();
  });
}
/* Blocking seam calls that MERGE: after dp_ctx_route_to_engine(ctx, eng) the blocking forms of the seams on `ctx` — dp_pcs_commit, dp_pcs_batch_open,
 * dp_sumcheck_prove, dp_logup_prove, dp_mle_fix_high, dp_mle_eval — are a submit to `eng` plus a wait: calls of the same shape that other threads make within
 * the engine's linger window are proved in lock step with merged launches, exactly like the submit / poll forms, and the calling thread sleeps meanwhile
 * instead of spinning on a stream of its own. This is what a host written against the reference's synchronous traits gets without restructuring
 * (rust/basefold-hip, rust/sumcheck-hip-patch: rayon workers inside PCS::commit / prove_parallel): many blocked threads, few busy cores.
 * The tables may have been uploaded through `ctx` a moment ago: its pending uploads are flushed before the submit. */
static dp_async* routed_engine(dp_ctx* ctx) {
  dp_async* eng = ctx ? ctx->engine.load(std::memory_order_acquire) : nullptr;
  if (eng && eng->ctx != ctx) { CtxLock lk(ctx); ctx->dev->flush_uploads(); }  // (async_submit flushes the engine's own context)
  return eng;
}
static void routed_check(int32_t rc) { if (rc != DP_OK) throw DpError(rc, g_err); }
void RoutedTicket::wait() { routed_check(dp_wait(t)); }
dp_ctx* RoutedTicket::t_ctx() { return t->owner; }
RoutedTicket::~RoutedTicket() {
  if (!t) return;
  if (t->state.load(std::memory_order_acquire) == 0) dp_wait(t);  // (never free a running call: its body holds references to the caller's arguments)
  // results nobody took (an error on the way): the device objects go back through the engine's context
  if (t->commit) { CtxLock lk(t_ctx()); t_ctx()->dev->free_commit(t->commit->c); delete t->commit; t->commit = nullptr; }
  if (t->buf) { CtxLock lk(t_ctx()); t_ctx()->dev->free_persistent(t->buf->b); delete t->buf; t->buf = nullptr; }
  delete t;
}
int32_t dp_ctx_route_to_engine(dp_ctx* ctx, dp_async* engine) {
  return guard([&] {
    DP_REQUIRE(ctx, DP_ERR_ARG, "null context");
    DP_REQUIRE(!engine || engine->ctx->device_id == ctx->device_id, DP_ERR_ARG, "dp_ctx_route_to_engine: the engine drives another device");
    ctx->engine.store(engine, std::memory_order_release);
  });
}
int32_t dp_async_destroy(dp_async* a) {
  return guard([&] {
    if (a && getenv("DP_TIMING") && atoi(getenv("DP_TIMING")) && a->njobs.load())
      fprintf(stderr, "[dp timing] async engine: %zu calls in %zu groups (%zu merged); per call %.1f us queued before it ran, %.1f us inside\n", a->njobs.load(), a->ngroups.load(), a->nmerged.load(),
              (double)a->queue_us.load() / (double)a->njobs.load(), (double)a->body_us.load() / (double)a->njobs.load());
    delete a;
  });
}
int32_t dp_async_stats(dp_async* a, size_t* calls, size_t* groups, size_t* merged_calls, size_t* workers) {
  return guard([&] { DP_REQUIRE(a, DP_ERR_ARG, "null engine"); if (calls) *calls = a->njobs.load(); if (groups) *groups = a->ngroups.load(); if (merged_calls) *merged_calls = a->nmerged.load(); if (workers) *workers = a->workers.size(); });
}
int32_t dp_poll(dp_ticket* t) { if (!t) return DP_ERR_ARG; const int s = t->state.load(std::memory_order_acquire); if (s < 0) g_err = t->err; return s; }
int32_t dp_wait(dp_ticket* t) {
  if (!t) return DP_ERR_ARG;
  for (unsigned spin = 0;; spin++) { const int s = dp_poll(t); if (s != 0) return s < 0 ? s : DP_OK; if (spin < 200) _mm_pause(); else std::this_thread::sleep_for(std::chrono::microseconds(20)); }
}
int32_t dp_ticket_words(dp_ticket* t, int32_t which, uint64_t** words, size_t* nwords) {
  return guard([&] {
    DP_REQUIRE(t && words && nwords && (which == 0 || which == 1), DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(t->state.load(std::memory_order_acquire) == 1, DP_ERR_ARG, "dp_ticket_words: the call has not completed successfully");
    *words = copy_out(t->words[which]); *nwords = t->words[which].size();
  });
}
int32_t dp_ticket_values(dp_ticket* t, uint64_t* values, size_t nvalues) {
  return guard([&] {
    DP_REQUIRE(t && values, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(t->state.load(std::memory_order_acquire) == 1 && nvalues <= t->finals.size(), DP_ERR_ARG, "dp_ticket_values: not completed, or more values asked than the call produced");
    memcpy(values, t->finals.data(), nvalues * 8);
  });
}
int32_t dp_ticket_commit(dp_ticket* t, dp_commit** out, uint64_t root[4]) {
  return guard([&] {
    DP_REQUIRE(t && out, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(t->state.load(std::memory_order_acquire) == 1 && t->commit, DP_ERR_ARG, "dp_ticket_commit: not a completed dp_pcs_commit_submit");
    *out = t->commit; t->commit = nullptr;
    if (root) for (int k = 0; k < 4; k++) root[k] = t->root[k];
  });
}
int32_t dp_ticket_free(dp_ticket* t) {
  return guard([&] {
    if (!t) return;
    const int st = t->state.load(std::memory_order_acquire);
    DP_REQUIRE(st != 0, DP_ERR_ARG, "dp_ticket_free: the call is still running");
    // a FAILED call holds nothing: its body gives back whatever it had allocated before it threw (see the submit functions)
    DP_REQUIRE(st < 0 || (!t->commit && !t->buf), DP_ERR_ARG, "dp_ticket_free: take the commitment / table first (dp_ticket_commit, dp_ticket_buf)");
    delete t->commit; delete t->buf;
    delete t;
  });
}
int32_t dp_sumcheck_prove_submit(dp_async* a, uint32_t nv, const dp_buf* const* tables, int32_t ntables, const int32_t* term_degree, const int32_t* term_tables,
                                 const uint64_t* term_coeffs, int32_t nterms, dp_transcript* t, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && tables && term_degree && term_tables && term_coeffs && t && ticket && ntables > 0 && nterms > 0 && nv > 0 && nv < 48, DP_ERR_ARG, "bad arguments");
    auto vp = std::make_shared<DevVP>(nv);
    read_terms(*vp, tables, ntables, term_degree, term_tables, nterms, term_coeffs);
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    uint64_t h = sig_mix(1, nv); for (const DBuf& b : vp->tabs) h = sig_buf(h, b); for (auto& tm : vp->terms) { h = sig_mix(h, (uint64_t)tm.k); for (int j = 0; j < tm.k; j++) h = sig_mix(h, (uint64_t)tm.t[j]); }
    tk->sig = h;
    tk->body = [vp, t, ntables](Dev& dev, dp_ticket& me) {
      SumcheckOut so = sumcheck_prove(dev, *vp, t->t);
      Writer w; w.iop(so.proof); me.words[0] = std::move(w.w);
      for (int i = 0; i < ntables; i++) { me.finals.push_back(so.finals[i].c0); me.finals.push_back(so.finals[i].c1); }
    };
    async_submit(a, tk.release(), ticket);
  });
}
int32_t dp_logup_prove_submit(dp_async* a, const dp_buf* const* columns, int32_t ncols, int32_t cpi, const dp_buf* mult, const uint64_t cc[2], const uint64_t csc[2],
                              dp_transcript* t, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && columns && ncols > 0 && cc && csc && t && ticket && cpi > 0, DP_ERR_ARG, "bad arguments");
    auto in = std::make_shared<LogUpInputDev>(); in->is_table = mult != nullptr; in->columns_per_instance = cpi;
    for (int i = 0; i < ncols; i++) { DP_REQUIRE(columns[i], DP_ERR_ARG, "null column"); in->columns.push_back(columns[i]->b); }
    if (mult) { DP_REQUIRE(!mult->b.ext && mult->b.n == in->columns[0].n, DP_ERR_SHAPE, "multiplicities shape"); in->multiplicities = mult->b; }
    in->constant_challenge = read_point(cc, 1)[0]; in->column_separation_challenge = read_point(csc, 1)[0];
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    uint64_t h = sig_mix(2, (uint64_t)cpi); h = sig_mix(h, mult ? 1 : 0); for (const DBuf& b : in->columns) h = sig_buf(h, b);
    tk->sig = h;
    tk->body = [in, t](Dev& dev, dp_ticket& me) { LogUpProof p = logup_batch_prove(dev, *in, t->t); Writer w; w.logup(p); me.words[0] = std::move(w.w); };
    async_submit(a, tk.release(), ticket);
  });
}
/* the two table primitives a Dense layer's prover runs around its sumcheck (dense.rs:423-561): their results come back through the ticket */
int32_t dp_mle_fix_high_submit(dp_async* a, const dp_buf* m, size_t rows, size_t cols, const uint64_t* point, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && m && point && ticket && is_pow2(rows) && is_pow2(cols) && !m->b.ext && m->b.n == rows * cols, DP_ERR_SHAPE, "fix_high: bad matrix shape");
    auto p = std::make_shared<std::vector<Ext>>(read_point(point, dp_ceil_log2(rows)));
    const DBuf mb = m->b;
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    tk->sig = sig_mix(sig_mix(5, rows), cols);
    tk->body = [mb, rows, cols, p](Dev& dev, dp_ticket& me) {
      DBuf b = dev.alloc_persistent(cols, true);
      try { dev.fix_high(b, mb, rows, cols, p->data()); dev.sync(); } catch (...) { try { dev.sync(); } catch (...) {} dev.free_persistent(b); throw; }
      me.buf = new dp_buf{b};  // (only a call that succeeded hands out its table)
    };
    async_submit(a, tk.release(), ticket);
  });
}
int32_t dp_mle_eval_submit(dp_async* a, const dp_buf* f, const uint64_t* point, uint32_t k, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && f && point && ticket, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(k < 48 && f->b.n == (size_t(1) << k), DP_ERR_SHAPE, "MLE size does not match the point");
    auto p = std::make_shared<std::vector<Ext>>(read_point(point, k));
    const DBuf fb = f->b;
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    tk->sig = sig_buf(sig_mix(6, k), fb);
    tk->body = [fb, k, p](Dev& dev, dp_ticket& me) { Ext r; dev.mle_eval_batch(&fb, 1, p->data(), k, &r); me.finals.push_back(r.c0); me.finals.push_back(r.c1); };
    async_submit(a, tk.release(), ticket);
  });
}
int32_t dp_ticket_buf(dp_ticket* t, dp_buf** out) {
  return guard([&] {
    DP_REQUIRE(t && out, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(t->state.load(std::memory_order_acquire) == 1 && t->buf, DP_ERR_ARG, "dp_ticket_buf: not a completed dp_mle_fix_high_submit");
    *out = t->buf; t->buf = nullptr;
  });
}
int32_t dp_pcs_commit_submit(dp_async* a, const dp_buf* poly, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && poly && ticket, DP_ERR_ARG, "bad arguments");
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    const DBuf b = poly->b;
    tk->sig = sig_buf(3, b);
    tk->body = [b](Dev& dev, dp_ticket& me) { DevCommit c = dev.commit(b, true); for (int k = 0; k < 4; k++) me.root[k] = c.tree.root.v[k]; me.commit = new dp_commit{c}; };
    async_submit(a, tk.release(), ticket);
  });
}
/* PCS::commit(pp, &poly) as the trait has it (mpcs/src/lib.rs:126-129): the polynomial is a HOST object. One ticket uploads the evaluations and commits to
 * them: dp_ticket_buf = the device table (the commitment refers to it: free it after the commitment), dp_ticket_commit = the commitment and its root. */
int32_t dp_pcs_commit_host_submit(dp_async* a, const uint64_t* words, size_t n, int32_t is_ext, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && words && ticket && n, DP_ERR_ARG, "bad arguments");
    const size_t nw = n * (is_ext ? 2 : 1);
    for (size_t i = 0; i < nw; i++) DP_REQUIRE(words[i] < GL_P, DP_ERR_ARG, "non-canonical field element");
    auto w = std::make_shared<std::vector<uint64_t>>(words, words + nw);
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    tk->sig = sig_mix(sig_mix(7, n), is_ext ? 2 : 1);
    const bool ext = is_ext != 0;
    tk->body = [w, n, ext](Dev& dev, dp_ticket& me) {
      DBuf b = dev.alloc_persistent(n, ext);
      DevCommit c;
      try { dev.upload(b, w->data()); c = dev.commit(b, true); } catch (...) { try { dev.sync(); } catch (...) {} dev.free_persistent(b); throw; }
      for (int k = 0; k < 4; k++) me.root[k] = c.tree.root.v[k];
      me.buf = new dp_buf{b}; me.commit = new dp_commit{c};  // (only a call that succeeded hands out its table and its commitment)
    };
    async_submit(a, tk.release(), ticket);
  });
}
int32_t dp_pcs_batch_open_submit(dp_async* a, const dp_commit* const* comms, int32_t n, const uint64_t* points_flat, const uint64_t* evals, dp_transcript* t, dp_ticket** ticket) {
  return guard([&] {
    DP_REQUIRE(a && comms && n > 0 && points_flat && evals && t && ticket, DP_ERR_ARG, "bad arguments");
    std::vector<unsigned> nvs; for (int i = 0; i < n; i++) { DP_REQUIRE(comms[i], DP_ERR_ARG, "null commitment"); nvs.push_back(comms[i]->c.nv); }
    auto pts = std::make_shared<std::vector<std::vector<Ext>>>(); auto evs = std::make_shared<std::vector<Ext>>();
    read_claims(n, points_flat, evals, nvs, *pts, *evs);
    std::vector<const DevCommit*> cs; for (int i = 0; i < n; i++) cs.push_back(&comms[i]->c);
    std::unique_ptr<dp_ticket> tk(new dp_ticket());
    uint64_t h = sig_mix(4, (uint64_t)n); for (int i = 0; i < n; i++) h = sig_mix(sig_mix(h, nvs[i]), comms[i]->c.is_base ? 1 : 2);
    tk->sig = h;
    tk->body = [cs, pts, evs, t](Dev& dev, dp_ticket& me) {
      std::vector<OpenClaim> oc; for (size_t i = 0; i < cs.size(); i++) oc.push_back({cs[i], (*pts)[i], (*evs)[i]});
      BasefoldProof p = pcs_batch_open(dev, 64, oc, t->t);
      Writer w; w.basefold(p); me.words[0] = std::move(w.w);
    };
    async_submit(a, tk.release(), ticket);
  });
}
int32_t dp_pcs_batch_verify_evals(size_t max_poly_size, const uint64_t* roots, const uint32_t* num_vars, const int32_t* is_base, int32_t n_polys, const uint64_t* points_flat,
                                  const uint32_t* point_num_vars, int32_t n_points, const uint32_t* eval_poly, const uint32_t* eval_point, const uint64_t* eval_values, int32_t n_evals,
                                  const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t) {
  return guard([&] {
    DP_REQUIRE(roots && num_vars && is_base && n_polys > 0 && points_flat && point_num_vars && n_points > 0 && eval_poly && eval_point && eval_values && n_evals > 0 && proof_words && t && is_pow2(max_poly_size), DP_ERR_ARG, "bad arguments");
    std::vector<std::vector<Ext>> pts; std::vector<size_t> ep, eq; std::vector<Ext> ev;
    read_eval_lists(points_flat, point_num_vars, n_points, eval_poly, eval_point, eval_values, n_evals, n_polys, pts, ep, eq, ev);
    std::vector<Commitment> cs;
    for (int i = 0; i < n_polys; i++) { Commitment c; for (int k = 0; k < 4; k++) c.root.v[k] = roots[4 * i + k]; c.num_vars = num_vars[i]; c.is_base = is_base[i] != 0; cs.push_back(c); }
    std::vector<VerifyEval> evals;
    for (int i = 0; i < n_evals; i++) evals.push_back({ep[i], eq[i], ev[i]});
    Reader r(proof_words, proof_nwords); BasefoldProof p = r.basefold();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    VerifierParams vp; vp.full_log = dp_ceil_log2(max_poly_size);
    std::vector<MerkleJob> jobs;
    merkle_sink() = &jobs;
    try { pcs_batch_verify_evals(vp, cs, pts, evals, p, t->t); } catch (...) { merkle_sink() = nullptr; throw; }
    merkle_sink() = nullptr;
    DP_REQUIRE(merkle_jobs_ok(jobs, verify_threads()), DP_ERR_VERIFY, "merkle path does not authenticate against the root");
  });
}
/* PCS::batch_commit / simple_batch_open / simple_batch_verify (mpcs/src/basefold.rs:356-446, 777-861, 1100-1203) */
int32_t dp_pcs_batch_commit(dp_ctx* ctx, const dp_buf* const* polys, int32_t n, dp_batch_commit** out, uint64_t root[4]) {
  return guard([&] {
    DP_REQUIRE(ctx && polys && out && n >= 1 && n <= 32, DP_ERR_ARG, "bad arguments (1..32 polynomials)");
    std::vector<DBuf> ev;
    for (int i = 0; i < n; i++) { DP_REQUIRE(polys[i], DP_ERR_ARG, "null polynomial"); ev.push_back(polys[i]->b); }
    CtxLock lk(ctx);
    DevBatchCommit c = pcs_batch_commit(*ctx->dev, ev, true);
    if (root) for (int k = 0; k < 4; k++) root[k] = c.root.v[k];
    *out = new dp_batch_commit{std::move(c)};
  });
}
int32_t dp_pcs_batch_commit_free(dp_ctx* ctx, dp_batch_commit* c) {
  return guard([&] {
    if (!c) return;
    CtxLock lk(ctx);
    for (DevCommit& d : c->c.polys) {  // the evaluation tables belong to the caller's dp_bufs
      if (d.bh_evals.p == d.evals.p) d.bh_evals.p = nullptr;
      if (d.tree.leaves.p == d.evals.p) d.tree.leaves.p = nullptr;
      d.evals.p = nullptr;
      ctx->dev->free_commit(d);
    }
    if (c->c.tree.leaves.p) ctx->dev->free_persistent(c->c.tree.leaves);
    if (c->c.tree.nodes.p) ctx->dev->free_persistent(c->c.tree.nodes);
    delete c;
  });
}
int32_t dp_pcs_simple_batch_open(dp_ctx* ctx, const dp_batch_commit* comm, const uint64_t* point, uint32_t num_vars, dp_transcript* t, uint64_t** proof_words, size_t* proof_nwords) {
  return guard([&] {
    DP_REQUIRE(ctx && comm && point && proof_words && proof_nwords, DP_ERR_ARG, "bad arguments");
    DP_REQUIRE(num_vars == comm->c.nv, DP_ERR_SHAPE, "point length != the polynomials' number of variables");
    DP_REQUIRE(comm->c.trivial() || t, DP_ERR_ARG, "a non-trivial opening needs the transcript");
    CtxLock lk(ctx);
    Transcript scratch;
    BasefoldProof p = pcs_simple_batch_open(*ctx->dev, comm->c, read_point(point, num_vars), t ? t->t : scratch);
    Writer w; w.basefold(p);
    *proof_words = copy_out(w.w); *proof_nwords = w.w.size();
  });
}
int32_t dp_pcs_simple_batch_verify(size_t max_poly_size, const uint64_t root[4], uint32_t num_vars, int32_t is_base, const uint64_t* point, const uint64_t* evals, int32_t n,
                                   const uint64_t* proof_words, size_t proof_nwords, dp_transcript* t) {
  return guard([&] {
    DP_REQUIRE(root && point && evals && n >= 1 && n <= (1 << 20) && num_vars <= 64 && proof_words && is_pow2(max_poly_size), DP_ERR_ARG, "bad arguments");
    Commitment c; for (int k = 0; k < 4; k++) c.root.v[k] = root[k]; c.num_vars = num_vars; c.is_base = is_base != 0;
    Reader r(proof_words, proof_nwords); BasefoldProof p = r.basefold();
    DP_REQUIRE(r.pos == proof_nwords, DP_ERR_ARG, "proof stream: trailing words");
    DP_REQUIRE(p.is_trivial() || t, DP_ERR_ARG, "a non-trivial opening needs the transcript");
    VerifierParams vp; vp.full_log = dp_ceil_log2(max_poly_size);
    Transcript scratch;
    std::vector<MerkleJob> jobs;
    merkle_sink() = &jobs;
    try { pcs_simple_batch_verify(vp, c, read_point(point, num_vars), read_point(evals, (unsigned)n), p, t ? t->t : scratch); } catch (...) { merkle_sink() = nullptr; throw; }
    merkle_sink() = nullptr;
    DP_REQUIRE(merkle_jobs_ok(jobs, verify_threads()), DP_ERR_VERIFY, "merkle path does not authenticate against the root");
  });
}

int32_t dp_model_setup(dp_ctx* ctx, const int64_t* blob, size_t nwords, dp_model** out) {
  return guard([&] {
    DP_REQUIRE(ctx && blob && out, DP_ERR_ARG, "bad arguments");
    ModelSpec m = parse_model(blob, nwords);
    std::unique_ptr<dp_model> dm(new dp_model());
    dm->ctx = ctx; dm->zk = context_generate(*ctx->dev, m);
    *out = dm.release();
  });
}
int32_t dp_model_free(dp_model* m) { return guard([&] { delete m; }); }
int32_t dp_model_prove(dp_model* m, const int64_t* input, size_t ninput, uint64_t** proof_words, size_t* proof_nwords,
                       int64_t* output, size_t* noutput, double* prove_ms) {
  return guard([&] {
    DP_REQUIRE(m && input && proof_words && proof_nwords, DP_ERR_ARG, "bad arguments");
    std::vector<int64_t> in(input, input + ninput);
    Trace tr = run_model(m->zk->model, in);
    Transcript t = default_transcript();
    hip_dev_arena_peak_reset(m->ctx->dev);
    auto t0 = std::chrono::steady_clock::now();
    Proof p = prove(*m->zk, tr, t);
    auto t1 = std::chrono::steady_clock::now();
    m->prove_peak = std::max(m->prove_peak, hip_dev_arena_peak(m->ctx->dev));
    if (prove_ms) *prove_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    hip_dev_dump_sc_debug(m->ctx->dev);
    std::vector<u64> w = serialize_proof(p);
    *proof_words = copy_out(w); *proof_nwords = w.size();
    if (output && noutput) {
      const std::vector<int64_t> o = model_output(m->zk->model, tr);
      DP_REQUIRE(*noutput >= o.size(), DP_ERR_ARG, "output buffer too small");
      memcpy(output, o.data(), o.size() * 8); *noutput = o.size();
    }
  });
}
int32_t dp_model_prove_batch(dp_model* m, const int64_t* inputs, size_t nproofs, size_t ninput, int32_t concurrency,
                             uint64_t** proof_words, size_t* proof_nwords, int64_t* outputs, size_t noutput_cap, size_t* noutput, double* wall_ms) {
  return guard([&] {
    DP_REQUIRE(m && inputs && proof_words && proof_nwords && nproofs > 0 && concurrency > 0 && concurrency <= 1024, DP_ERR_ARG, "bad arguments");
    size_t nw = std::min<size_t>((size_t)concurrency, nproofs);
    // worker 0 is the model's own context; the others get their own stream + arena on the same GPU and share the
    // (read-only) model commitments
    // Arena of a worker: DP_WORKER_ARENA_BYTES, else 1.25 x the largest footprint a proof of this model has had (one proof
    // alone runs in latency mode, whose multi-workgroup sumchecks keep every fold level: an upper bound of what a proof in
    // flight needs) + 64 MB (at least 256 MB), else — nothing proved yet — 1.5 GB. The number of proofs in flight is cut to what fits in
    // 90 % of the free HBM instead of failing: `concurrency` is a cap, not a demand.
    const char* env = getenv("DP_WORKER_ARENA_BYTES");
    const size_t MB64 = size_t(64) << 20;
    // (round 6: 1.125 x the footprint + 16 MB in steps of 16 MB — 336 MB for Dense-4M where 1.25 x + 64 MB in steps of 64 gave 448: a proof's footprint does not depend on
    // its input, the margin only has to cover what latency mode and throughput mode allocate differently, and throughput mode allocates LESS; ~700 proofs fit in flight
    // instead of ~540, and the rate still grows with them: 1 066 / 1 116 proofs/s at 448 / 660, tools/r06/call23.sh)
    const size_t MB16 = size_t(16) << 20;
    size_t arena = env ? strtoull(env, nullptr, 10) : m->prove_peak ? std::max(4 * MB64, ((m->prove_peak + m->prove_peak / 8 + MB16 + MB16 - 1) / MB16) * MB16) : (size_t(3) << 29);
    if (m->workers.size() + 1 < nw) {
      // the cap is fixed by the first batch that needs more workers than exist (later batches must not creep into the reserve)
      if (!m->in_flight_cap) {
        size_t free_b = 0, total_b = 0; hip_mem_info(m->ctx->device_id, &free_b, &total_b);
        const size_t per = arena + (size_t(24) << 20);  // + the worker's staging and small buffers (the twiddle / coset tables are the context's, shared)
        // DP_HBM_FRACTION (default 0.9): the share of the FREE HBM this process sizes its workers against — less when several processes share one GPU
        // (bench.py with DP_FORCE_DEVICE: both ranks read the same free figure at the same moment)
        const double frac = getenv("DP_HBM_FRACTION") ? std::min(0.95, std::max(0.01, atof(getenv("DP_HBM_FRACTION")))) : 0.9;
        m->in_flight_cap = m->workers.size() + 1 + (size_t)((double)free_b * frac / (double)per);
        if (m->in_flight_cap < nw && getenv("DP_TIMING") && atoi(getenv("DP_TIMING"))) fprintf(stderr, "[dp timing] prove_batch: %zu proofs in flight asked, %zu fit in %.1f GB of free HBM (%.0f MB per worker)\n", nw, m->in_flight_cap, free_b / 1e9, per / 1048576.0);
      }
      nw = std::min(nw, m->in_flight_cap);
    }
    // the context's tables may have been rebuilt for another parameter size since this model was loaded
    auto share_pcs = [&](Dev* w) { m->ctx->dev->pcs_init(m->zk->full_log); hip_dev_pcs_share(w, m->ctx->dev); };
    for (auto& w : m->workers) share_pcs(w.get());
    while (m->workers.size() + 1 < nw) {
      std::unique_ptr<Dev> w;
      try { w.reset(make_hip_worker(m->ctx->device_id, arena)); }
      catch (const DpError& e) {
        // the device ran out of memory after all (another process took it meanwhile): go on with the workers there are — `concurrency` is a cap, not a demand
        if (e.code != DP_ERR_OOM || m->workers.empty()) throw;
        nw = m->workers.size() + 1; m->in_flight_cap = nw;
        break;
      }
      share_pcs(w.get()); m->workers.push_back(std::move(w));
    }
    m->last_in_flight = nw;
    // several proofs in flight: throughput mode on every context (see hip_dev_set_latency_mode)
    hip_dev_set_latency_mode(m->ctx->dev, nw == 1);
    for (auto& w : m->workers) hip_dev_set_latency_mode(w.get(), nw == 1);
    // Cohorts (hip_dev.hip, struct Cohort): the proofs in flight are grouped into cohorts of DP_COHORT members (default: in flight / 22, rounded up)
    // that prove in lock step — launch number i of all members of a cohort is ONE kernel launch — on one stream and one
    // host thread per cohort. DP_COHORT=0: every proof on its own stream (the round-1 scheme).
    const char* ce = getenv("DP_COHORT");
    // Default: as many cohorts as hardware queues serve without time slicing (22 of the 24), each as small as that allows — a merged
    // launch ends with its slowest member, so small cohorts stall less (batch of 8: 121 ms with cohorts of 1, 165 ms with one cohort
    // of 8; batch of 64: 262 ms with cohorts of 3, 302 ms with 12; 256 in flight: cohorts of 12; profiles/r02_batch_cohort_sweep.txt)
    // DP_STREAM_SHARE = k (default 1): k cohorts take turns on one stream (hardware queue) — cohorts of in flight / (22 k) members, and while one cohort's members
    // digest a result on the host its queue-mate's launch runs
    const size_t share = (size_t)std::max(1, getenv("DP_STREAM_SHARE") ? atoi(getenv("DP_STREAM_SHARE")) : 1);
    size_t csize = ce ? (size_t)std::max(0, atoi(ce)) : std::max<size_t>(1, (nw + 22 * share - 1) / (22 * share));
    size_t nco = (csize >= 1 && nw > 1) ? (nw + csize - 1) / csize : 0;
    while (m->cohorts.size() < nco) { const size_t c = m->cohorts.size(); m->cohorts.push_back(c % share ? hip_cohort_new_sharing(m->cohorts[c - c % share]) : hip_cohort_new()); }
    auto dev_of = [&](size_t wi) -> Dev& { return wi == 0 ? *m->ctx->dev : *m->workers[wi - 1]; };
    std::atomic<size_t> next(0);
    const bool timing = getenv("DP_TIMING") && atoi(getenv("DP_TIMING"));
    std::mutex err_mu; std::string err; int err_code = 0;
    for (size_t i = 0; i < nproofs; i++) { proof_words[i] = nullptr; proof_nwords[i] = 0; }
    for (size_t wi = 0; wi < nw && nco; wi++) hip_dev_cohort_attach(&dev_of(wi), m->cohorts[wi % nco]);
    // The host work a proof starts with — inference and the host half of the witness generation (zkml.h witness_host: lookup columns, table multiplicities;
    // 0.6 ms for Dense-4M, ~10 ms for the transformer layer) — is made AHEAD of the proofs by DP_PREP_THREADS helper threads (default 2, 0 = off): the members of
    // a cohort start a proof together on ONE host thread, and until the last of them has submitted its first launch the cohort's stream is idle.
    // pst[i]: 0 nobody has touched input i, 1 being prepared, 2 ready in prep[i], 3 the helper failed (the worker repeats it so that the error is its own).
    struct Prep { Trace tr; WitnessHost wh; };
    std::vector<std::unique_ptr<Prep>> prep(nproofs);
    std::unique_ptr<std::atomic<int>[]> pst(new std::atomic<int>[nproofs]);
    for (size_t i = 0; i < nproofs; i++) pst[i].store(0, std::memory_order_relaxed);
    auto make_prep = [&](size_t i) {
      std::unique_ptr<Prep> p(new Prep);
      p->tr = run_model(m->zk->model, std::vector<int64_t>(inputs + i * ninput, inputs + (i + 1) * ninput));
      p->wh = witness_host(*m->zk, p->tr);
      return p;
    };
    const char* pe = getenv("DP_PREP_THREADS");
    const size_t nprep = nw > 1 ? (size_t)std::max(0, pe ? atoi(pe) : 2) : 0;
    const size_t prep_ahead = getenv("DP_PREP_AHEAD") ? (size_t)std::max(1, atoi(getenv("DP_PREP_AHEAD"))) : nw;  // inputs prepared beyond the one handed out last
    std::atomic<size_t> pnext(nw);  // (the first nw proofs start at once: their workers prepare them themselves)
    std::atomic<bool> pstop(false);
    auto prep_thread = [&] {
      for (;;) {
        const size_t i = pnext.fetch_add(1);
        if (i >= nproofs) return;
        while (!pstop.load(std::memory_order_relaxed) && i >= next.load(std::memory_order_relaxed) + prep_ahead) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (pstop.load(std::memory_order_relaxed)) return;
        if (i < next.load(std::memory_order_relaxed)) continue;  // handed out meanwhile (or the batch is being abandoned)
        int e = 0;
        if (!pst[i].compare_exchange_strong(e, 1)) continue;
        try { prep[i] = make_prep(i); pst[i].store(2, std::memory_order_release); } catch (...) { pst[i].store(3, std::memory_order_release); }
      }
    };
    // Serialising a finished proof (2.1 ms for the 5.9 MB of a Dense-4M proof) and copying it out used to run on the cohort's thread: the members of a cohort
    // finish a proof together, so every pass of a cohort ended with members x 2.3 ms of host work during which its stream had nothing queued (48 ms of a 533 ms
    // pass at 21 members). DP_SER_THREADS helper threads (default 2, 0 = inline) take it over; the batch returns when they have drained their queue.
    // Measured (tools/r05/call20.sh, one box, alternating): WORSE — 483-512 against 679-694 proofs/s with two threads: the 14 proving threads spin on all the CPUs
    // the process has, the helpers get their cycles late, and the last wave's 448 proofs queue behind two threads. Off by default; the faster serialiser
    // (proof.h Writer: reserved once, bulk appends) is what shortens the stall instead.
    const size_t nser = nw > 1 ? (size_t)std::max(0, getenv("DP_SER_THREADS") ? atoi(getenv("DP_SER_THREADS")) : 0) : 0;
    std::mutex ser_mu; std::condition_variable ser_cv; std::deque<std::pair<size_t, Proof>> ser_q; bool ser_stop = false;
    auto ser_thread = [&] {
      for (;;) {
        std::pair<size_t, Proof> job;
        {
          std::unique_lock<std::mutex> lk(ser_mu);
          ser_cv.wait(lk, [&] { return ser_stop || !ser_q.empty(); });
          if (ser_q.empty()) return;
          job = std::move(ser_q.front()); ser_q.pop_front();
        }
        try { std::vector<u64> w = serialize_proof(job.second); proof_words[job.first] = copy_out(w); proof_nwords[job.first] = w.size(); }
        catch (const std::exception& e) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = DP_ERR_OOM; err = std::string("serialising a proof: ") + e.what(); } }
      }
    };
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](size_t wi) {
      Dev& dev = dev_of(wi);
      try {
        dev.bind_thread();
        for (;;) {
          size_t i = next.fetch_add(1);
          if (i >= nproofs) break;
          auto h0 = std::chrono::steady_clock::now();
          std::unique_ptr<Prep> mine;
          int e = 0;
          if (pst[i].compare_exchange_strong(e, 1)) mine = make_prep(i);
          else {
            while ((e = pst[i].load(std::memory_order_acquire)) == 1) fiber_yield();
            mine = e == 2 ? std::move(prep[i]) : make_prep(i);
          }
          Trace& tr = mine->tr;
          auto h1 = std::chrono::steady_clock::now();
          Transcript t = default_transcript();
          Proof p = prove(*m->zk, dev, tr, t, &mine->wh);
          auto h2 = std::chrono::steady_clock::now();
          if (nser) {  // several proofs in flight: the finished proof goes to the serialiser threads, this fiber to its next proof (see `ser_thread`)
            { std::lock_guard<std::mutex> g(ser_mu); ser_q.emplace_back(i, std::move(p)); }
            ser_cv.notify_one();
          } else { static thread_local std::vector<u64> ser_buf; serialize_proof_to(p, ser_buf); proof_words[i] = copy_out(ser_buf); proof_nwords[i] = ser_buf.size(); }
          auto h3 = std::chrono::steady_clock::now();
          if (timing && wi == 0) {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[dp timing] host phases of one proof: inference + witness columns (or the wait for the helper that made them) %.2f ms, prove %.2f ms (wall, shared thread), serialise %.2f ms, copy out %.2f ms\n", ms(h0, h1), ms(h1, h2), ms(h2, h3), ms(h3, std::chrono::steady_clock::now()));
          }
          if (outputs) { const std::vector<int64_t> o = model_output(m->zk->model, tr); DP_REQUIRE(o.size() <= noutput_cap, DP_ERR_ARG, "output buffer too small"); memcpy(outputs + i * noutput_cap, o.data(), o.size() * 8); if (noutput) *noutput = o.size(); }
        }
      } catch (const DpError& e) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = e.code; err = e.what(); } next = nproofs; }
      catch (const std::exception& e) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = DP_ERR_ARG; err = e.what(); } next = nproofs; }
      catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = DP_ERR_ARG; err = "unknown exception in a proof worker"; } next = nproofs; }
      // leaving the cohort releases the launches the other members have queued behind this one
      if (nco) { try { hip_dev_cohort_detach(&dev); } catch (const std::exception& e) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = DP_ERR_HIP; err = e.what(); } next = nproofs; } }
    };
    // `nw` proofs in flight on `nth` host threads: every worker is a fiber; a thread switches to its next fiber whenever the
    // current one waits for the device (fiber.h). All members of a cohort live on one thread (the cohort has no locks);
    // cohorts (or, without cohorts, workers) are dealt round robin to the threads. The thread count follows the CPUs the
    // process may really use (cgroup quota), leaving two for the HIP runtime's own threads.
    const char* te = getenv("DP_HOST_THREADS");
    size_t nth = te ? (size_t)std::max(1, atoi(te)) : (size_t)std::max(1.0, host_cpu_budget() - 2.0);
    if (!te && getenv("DP_HOST_SPONGE") && atoi(getenv("DP_HOST_SPONGE"))) {  // the sponge servers (csrc/sponge_host.h) spin too: they come out of the same CPU budget
      const int S = getenv("DP_SPONGE_THREADS") ? std::max(1, atoi(getenv("DP_SPONGE_THREADS"))) : 6;
      nth = (size_t)std::max(2.0, host_cpu_budget() - 2.0 - (double)S);
    }
    // with sleeping idle threads (fiber.h, DP_IDLE_SLEEP_US) a cohort's thread costs what its members' host work costs (~0.3 cores), not a whole core of polling:
    // one thread per cohort as long as that is at most twice the CPU budget (22 cohorts on a 16-CPU quota: 6.3 cores busy)
    if (!te && nco && nw > 1 && fiber_idle_sleep_ns() > 0 && (double)nco <= 2.0 * host_cpu_budget()) nth = std::max(nth, nco);
    nth = std::min(nth, nco ? nco : nw);
    // DP_COHORT_STAGGER_MS = d (diagnostic, default 0): cohort c starts c * d milliseconds late. The cohorts of a batch otherwise run IN PHASE — identical launch sequences
    // from a common start: every queue in one-workgroup tails, then every queue in the streaming kernels, then every queue hashing (profiles/r06_timeline_704.txt). A
    // staggered start persists (profiles/r06_timeline_704_staggered.txt) and changes the rate by nothing, in every combination tried (profiles/r06_plateau_sweeps.txt,
    // call 16 / 20 / 21 / 24): a tail that starts beside other cohorts' hash layers waits milliseconds for room.
    const double stagger_ms = getenv("DP_COHORT_STAGGER_MS") ? std::max(0.0, atof(getenv("DP_COHORT_STAGGER_MS"))) : 0.0;
    auto run_thread = [&](size_t ti) {
      if (nco && stagger_ms > 0 && nproofs > nw) std::this_thread::sleep_for(std::chrono::microseconds((long)((double)(ti % nco) * stagger_ms * 1000.0)));
      FiberSched sched;
      sched.idle_sleep = nw > 1;
      for (size_t wi = 0; wi < nw; wi++) if ((nco ? wi % nco : wi) % nth == ti) fiber_spawn(sched, [&work, wi] { work(wi); });
      fiber_run_all(sched);
      for (auto& f : sched.fibers) if (f->failed) { std::lock_guard<std::mutex> g(err_mu); if (!err_code) { err_code = DP_ERR_ARG; err = "an exception escaped a proof worker's fiber"; } }
    };
    std::vector<std::thread> th, pth, sth;
    // (threads this call spawns keep to the CPUs of the GPU's NUMA node, Dev::pin_thread; thread 0 is the caller's and keeps the affinity it came with)
    Dev* pin_dev = m->ctx->dev;
    for (size_t k = 0; k < nser; k++) sth.emplace_back([&, pin_dev] { pin_dev->pin_thread(); ser_thread(); });
    for (size_t k = 0; k < nprep && nproofs > nw; k++) pth.emplace_back([&, pin_dev] { pin_dev->pin_thread(); prep_thread(); });
    for (size_t ti = 1; ti < nth; ti++) th.emplace_back([&, pin_dev, ti] { pin_dev->pin_thread(); run_thread(ti); });
    run_thread(0);
    for (auto& t : th) t.join();
    pstop.store(true);
    for (auto& t : pth) t.join();
    { std::lock_guard<std::mutex> g(ser_mu); ser_stop = true; }
    ser_cv.notify_all();
    for (auto& t : sth) t.join();  // (they leave when the queue is empty: every proof is serialised and copied out)
    for (size_t c = 0; c < nco; c++) { try { hip_cohort_drain(m->cohorts[c]); } catch (const std::exception& e) { if (!err_code) { err_code = DP_ERR_HIP; err = e.what(); } } }
    if (nco && getenv("DP_TIMING") && atoi(getenv("DP_TIMING"))) {
      size_t f = 0, p = 0; for (size_t c = 0; c < nco; c++) { size_t a, b; hip_cohort_stats(m->cohorts[c], &a, &b); f += a; p += b; }
      fprintf(stderr, "[dp timing] %zu cohorts of <= %zu proofs: %zu merged launches for %zu proof launches\n", nco, csize, f, p);
    }
    if (getenv("DP_TIMING") && atoi(getenv("DP_TIMING"))) {
      size_t wpeak = 0; for (auto& w : m->workers) wpeak = std::max(wpeak, hip_dev_arena_peak(w.get()));
      fprintf(stderr, "[dp timing] prove_batch: %zu proofs, %zu in flight on %zu host threads, arena peak %.1f MB (context), %.1f MB (largest of the workers, arena %.1f MB each)\n", nproofs, nw, nth, hip_dev_arena_peak(m->ctx->dev) / 1048576.0, wpeak / 1048576.0, arena / 1048576.0);
    }
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hip_dev_dump_sc_debug(m->ctx->dev); hip_dev_dump_host_stats(m->ctx->dev); hip_dump_wg_times();
    for (size_t wi = 1; wi < nw && wi < 3; wi++) { hip_dev_dump_sc_debug(m->workers[wi - 1].get()); hip_dev_dump_host_stats(m->workers[wi - 1].get()); }
    hip_dev_set_latency_mode(m->ctx->dev, true);
    if (err_code) { for (size_t i = 0; i < nproofs; i++) { dp_free(proof_words[i]); proof_words[i] = nullptr; } throw DpError(err_code, err); }
  });
}
int32_t dp_model_infer_host(const int64_t* model_blob, size_t nwords, const int64_t* input, size_t ninput, int64_t* output, size_t* noutput) {
  return guard([&] {
    DP_REQUIRE(model_blob && input && output && noutput, DP_ERR_ARG, "bad arguments");
    ModelSpec m = parse_model(model_blob, nwords);
    validate_model(m);
    for (auto& l : m.layers) { prepare_fast_inference(l); if (l.kind == L_CONV) l.wfft = conv_weight_fft(l); }
    Trace tr = run_model(m, std::vector<int64_t>(input, input + ninput));
    const std::vector<int64_t> o = model_output(m, tr);
    DP_REQUIRE(*noutput >= o.size(), DP_ERR_ARG, "output buffer too small");
    memcpy(output, o.data(), o.size() * 8); *noutput = o.size();
  });
}
int32_t dp_host_poseidon2(uint64_t state[8], int32_t force_scalar, int32_t* vectorised) {
  return guard([&] {
    DP_REQUIRE(state, DP_ERR_ARG, "null state");
    if (vectorised) *vectorised = dp::p2_fast() != nullptr;
    if (force_scalar) dp::hostnc::permute_scalar(state); else dp::hostnc::permute(state);
  });
}
int32_t dp_model_in_flight(const dp_model* m, size_t* in_flight) { return guard([&] { DP_REQUIRE(m && in_flight, DP_ERR_ARG, "bad arguments"); *in_flight = m->last_in_flight; }); }
int32_t dp_model_output_len(const dp_model* m, size_t* noutput) { return guard([&] { DP_REQUIRE(m && noutput, DP_ERR_ARG, "bad arguments"); *noutput = model_output_len(m->zk->model); }); }
double dp_host_cpu_budget(void) { return host_cpu_budget(); }
int32_t dp_model_verifier_blob(const dp_model* m, uint64_t** words, size_t* nwords) {
  return guard([&] { DP_REQUIRE(m && words && nwords, DP_ERR_ARG, "bad arguments"); std::vector<u64> w = vctx_to_words(m->zk->verifier_ctx()); *words = copy_out(w); *nwords = w.size(); });
}
int32_t dp_verify(const uint64_t* vb, size_t vn, const uint64_t* pw, size_t pn, const int64_t* input, size_t ninput, const int64_t* output, size_t noutput) {
  return guard([&] {
    DP_REQUIRE(vb && pw && input && output, DP_ERR_ARG, "bad arguments");
    const bool timing = getenv("DP_TIMING") && atoi(getenv("DP_TIMING"));
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[dp timing] dp_verify: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
    VerifierContext vc = vctx_from_words(vb, vn);
    Proof p = deserialize_proof(pw, pn);
    lap("parse");
    IO io; io.input.assign(input, input + ninput); io.output.assign(output, output + noutput);
    Transcript t = default_transcript();
    // the Merkle paths are recorded while the protocol checks run and authenticated together afterwards (eight side by side on AVX-512 CPUs)
    std::vector<MerkleJob> jobs;
    merkle_sink() = &jobs;
    try { verify(vc, p, io, t); } catch (...) { merkle_sink() = nullptr; throw; }
    merkle_sink() = nullptr;
    lap("protocol checks");
    DP_REQUIRE(merkle_jobs_ok(jobs, verify_threads()), DP_ERR_VERIFY, "merkle path does not authenticate against the root");
    lap("merkle paths");
  });
}

/* zkml::verify for a batch of proofs of one model: the protocol checks run on host threads, every Merkle path of a proof is
 * authenticated on the device in one launch (ctx == NULL: on the host threads as well). results[i] = DP_OK / DP_ERR_VERIFY /
 * DP_ERR_ARG per proof; returns DP_OK when the batch was processed (whatever the verdicts). */
int32_t dp_verify_batch(dp_ctx* ctx, const uint64_t* vb, size_t vn, const uint64_t* const* proof_words, const size_t* proof_nwords, const int64_t* inputs, size_t ninput,
                        const int64_t* outputs, size_t noutput, size_t nproofs, int32_t threads, int32_t* results, double* wall_ms) {
  return guard([&] {
    DP_REQUIRE(vb && proof_words && proof_nwords && inputs && outputs && results, DP_ERR_ARG, "bad arguments");
    auto t0 = std::chrono::steady_clock::now();
    const VerifierContext vc = vctx_from_words(vb, vn);
    size_t nth = threads > 0 ? (size_t)threads : (size_t)std::max(1.0, host_cpu_budget() - 2.0);
    nth = std::max<size_t>(1, std::min(nth, nproofs));
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= nproofs) return;
        std::vector<MerkleJob> jobs;
        try {
          Proof p = deserialize_proof(proof_words[i], proof_nwords[i]);
          IO io; io.input.assign(inputs + i * ninput, inputs + (i + 1) * ninput); io.output.assign(outputs + i * noutput, outputs + (i + 1) * noutput);
          Transcript t = default_transcript();
          merkle_sink() = &jobs;
          try { verify(vc, p, io, t); } catch (...) { merkle_sink() = nullptr; throw; }
          merkle_sink() = nullptr;
          // the recorded paths: one flat pool of sibling digests + per-path leaf digest, root, leaf-pair index, offset, depth
          const size_t n = jobs.size();
          size_t pool_n = 0; for (const MerkleJob& j : jobs) pool_n += j.depth;
          std::vector<u64> leaf(4 * n), root(4 * n), x(n), off(n), depth(n), pool(4 * pool_n);
          size_t o = 0;
          for (size_t k = 0; k < n; k++) {
            const MerkleJob& j = jobs[k];
            for (int q = 0; q < 4; q++) { leaf[4 * k + q] = j.leaf.v[q]; root[4 * k + q] = j.root.v[q]; }
            x[k] = j.x; off[k] = o; depth[k] = j.depth;
            for (size_t l = 0; l < j.depth; l++) for (int q = 0; q < 4; q++) pool[4 * (o + l) + q] = j.path[l].v[q];
            o += j.depth;
          }
          bool ok;
          if (ctx) { CtxLock lk(ctx); ok = ctx->dev->merkle_paths_check(leaf.data(), root.data(), x.data(), off.data(), depth.data(), n, pool.data(), pool_n, nullptr); }
          else ok = merkle_jobs_ok(jobs);
          results[i] = ok ? DP_OK : DP_ERR_VERIFY;
        } catch (const DpError& e) { merkle_sink() = nullptr; results[i] = e.code == DP_ERR_VERIFY ? DP_ERR_VERIFY : DP_ERR_ARG; }
        catch (const std::exception&) { merkle_sink() = nullptr; results[i] = DP_ERR_ARG; }
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < nth; k++) th.emplace_back([&] { if (ctx) ctx->dev->pin_thread(); work(); });
    work();
    for (auto& t : th) t.join();
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
}

}  // extern "C"
