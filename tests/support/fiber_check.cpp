// TEST HARNESS (tests/ only): the cooperative fibers of deep-prove_amd/csrc/fiber.h — interleaving, per-fiber stacks,
// exceptions thrown and caught inside a fiber, several schedulers on several threads, the CPU-budget probe.
#include "../../deep-prove_amd/csrc/fiber.h"
#include <thread>
#include <stdexcept>
#include <cstdio>
#include <cmath>
DP_FIBER_SWITCH_ASM
using namespace dp;
static int run_one_thread(int nf, int iters, std::vector<int>* order) {
  FiberSched s;
  std::vector<double> acc(nf, 0.0);
  std::vector<int> caught(nf, 0);
  for (int i = 0; i < nf; i++) fiber_spawn(s, [&, i] {
    volatile char pad[4096]; pad[0] = (char)i;  // something on the fiber's own stack
    for (int k = 0; k < iters; k++) {
      acc[i] += std::sqrt((double)(k + i));
      if (order && k < 3) order->push_back(i);
      try { if (k % 7 == 3) throw std::runtime_error("x"); } catch (const std::exception&) { caught[i]++; }
      fiber_yield();
    }
    if (pad[0] != (char)i) acc[i] = -1;
  });
  fiber_run_all(s);
  int bad = 0;
  for (int i = 0; i < nf; i++) {
    double want = 0; for (int k = 0; k < iters; k++) want += std::sqrt((double)(k + i));
    int wc = 0; for (int k = 0; k < iters; k++) wc += k % 7 == 3;
    if (acc[i] != want || caught[i] != wc) bad++;
  }
  return bad;
}
int main() {
  std::vector<int> order;
  int bad = run_one_thread(5, 1000, &order);
  // round robin: the first three rounds visit fibers 0..4 in order
  for (size_t i = 0; i < order.size(); i++) if (order[i] != (int)(i % 5)) bad++;
  std::vector<std::thread> th; std::vector<int> res(4, -1);
  for (int t = 0; t < 4; t++) th.emplace_back([&, t] { res[t] = run_one_thread(3 + t, 2000, nullptr); });
  for (auto& t : th) t.join();
  for (int r : res) bad += r;
  if (fiber_active()) bad++;
  double b = host_cpu_budget();
  printf("fibers ok=%d cpu_budget=%.1f\n", bad == 0, b);
  return bad == 0 && b >= 1.0 ? 0 : 1;
}
