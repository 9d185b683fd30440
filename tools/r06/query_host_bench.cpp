// Where do the 1.3 ms of host work before the query gather's wait go (profiles/r06_host_work_by_launch.txt: "k_query_gather .. k_download")? Host-only timing of its parts
// with the product's own transcript: 200 query-index challenges, the descriptor lists of 200 x 56 opened pairs, the layout of the stream image.
// build: g++ -O2 -std=c++17 -I deep-prove_amd/csrc -o tools/_build/query_host_bench tools/r06/query_host_bench.cpp deep-prove_amd/_obj/p2_avx512.o
#include "../../deep-prove_amd/csrc/poseidon2.h"
#include <chrono>
#include <cstdio>
int main() {
  using namespace dp;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int fast = 0; fast < 2; fast++) {
    if (fast && p2_cpu_has_avx512()) { p2_fast() = p2_permute_avx512; }
    Transcript t = default_transcript();
    auto t0 = now();
    u64 acc = 0;
    for (int rep = 0; rep < 50; rep++) for (unsigned q = 0; q < 200; q++) acc += t.get_and_append_challenge("query indices").c0;
    auto t1 = now();
    printf("%s permutation: 200 query-index challenges %.1f us (%llu)\n", fast ? "AVX-512" : "scalar", us(t0, t1) / 50, (unsigned long long)(acc & 1));
  }
  return 0;
}
