// ORACLE (test infrastructure only).
// Fork-join data parallelism for the multi-threaded CPU baseline ("port-mt" in bench.py's cpu_baseline): the role rayon plays in
// the reference (`par_iter().with_min_len(64)` over table halves in the sumcheck rounds, sumcheck/src/prover.rs:625-741 and
// sumcheck_macro/src/lib.rs:147-249; `par_chunks` over Merkle layers, mpcs/src/util/merkle_tree.rs:261-329; `par_iter` over
// polynomials in Basefold::batch_open, mpcs/src/basefold.rs:546-770). Field arithmetic is exact and every combined result is a
// sum, so any chunking gives the bytes of the single-threaded run — orc::set_threads(1), the default, IS the single-threaded
// oracle (no pool is created, the loops run inline).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace orc {

class ParPool {
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t, size_t)>* job_ = nullptr;
  size_t n_ = 0, chunk_ = 0;
  std::atomic<size_t> next_{0};
  size_t gen_ = 0; int running_ = 0; bool stop_ = false;
  bool in_run_ = false;  // (owner thread only) a nested par_for inside a chunk runs inline
  void work() {
    for (;;) {
      size_t lo = next_.fetch_add(chunk_);
      if (lo >= n_) break;
      (*job_)(lo, std::min(n_, lo + chunk_));
    }
  }
  void loop() {
    size_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
      }
      work();
      { std::lock_guard<std::mutex> lk(mu_); if (--running_ == 0) done_.notify_all(); }
    }
  }

 public:
  explicit ParPool(int threads) { for (int i = 1; i < threads; i++) th_.emplace_back([this] { loop(); }); }
  ~ParPool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
  int threads() const { return (int)th_.size() + 1; }
  // body(lo, hi) over [0, n) in chunks of at least `min_len`; returns when every chunk has run
  void run(size_t n, size_t min_len, const std::function<void(size_t, size_t)>& body) {
    const size_t T = (size_t)threads();
    size_t chunk = std::max<size_t>(min_len, (n + 4 * T - 1) / (4 * T));  // ~4 chunks per thread: balance without a task per element
    if (T == 1 || n <= chunk || in_run_) { body(0, n); return; }
    in_run_ = true;
    { std::lock_guard<std::mutex> lk(mu_); job_ = &body; n_ = n; chunk_ = chunk; next_.store(0); running_ = (int)th_.size(); gen_++; }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return running_ == 0; });
    job_ = nullptr; in_run_ = false;
  }
};

// the pool of the CALLING thread's proof (thread_local: the replica baseline runs one single-threaded proof per host thread, the
// "port-mt" baseline one proof on all cores — they must not share a pool)
inline ParPool*& par_pool() { static thread_local ParPool* p = nullptr; return p; }
struct ParScope {  // RAII: `threads` workers for the proofs made on this thread while the scope lives
  ParPool* saved; ParPool* mine;
  explicit ParScope(int threads) : saved(par_pool()), mine(threads > 1 ? new ParPool(threads) : nullptr) { par_pool() = mine; }
  ~ParScope() { par_pool() = saved; delete mine; }
};
constexpr size_t PAR_MIN_LEN = 64;  // rayon's with_min_len(64) of the reference
template <class F> inline void par_for(size_t n, F&& body, size_t min_len = PAR_MIN_LEN) {
  ParPool* p = par_pool();
  if (!p || n < 2 * min_len) { body(size_t(0), n); return; }
  std::function<void(size_t, size_t)> f = std::forward<F>(body);
  p->run(n, min_len, f);
}
// sum-reduction: `part(lo, hi)` returns a partial result, `add` combines (exact field addition: order does not matter)
template <class T, class Part, class Add> inline T par_reduce(size_t n, T zero, Part&& part, Add&& add, size_t min_len = PAR_MIN_LEN) {
  ParPool* p = par_pool();
  if (!p || n < 2 * min_len) return add(zero, part(size_t(0), n));
  std::mutex mu; T acc = zero;
  std::function<void(size_t, size_t)> f = [&](size_t lo, size_t hi) { T r = part(lo, hi); std::lock_guard<std::mutex> g(mu); acc = add(acc, r); };
  p->run(n, min_len, f);
  return acc;
}

}  // namespace orc
