// host Poseidon2-w8 permutation rate (the transcript sponge), scalar and AVX-512:
//   g++ -O2 -std=c++17 -o /tmp/p2hb tools/p2_host_bench.cpp && /tmp/p2hb
#include "../deep-prove_amd/csrc/p2_avx512.cpp"
#include <chrono>
#include <cstdio>
int main() {
  dp::u64 s[8] = {1, 2, 3, 4, 5, 6, 7, 8}, q[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const int N = 300000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; i++) dp::hostnc::permute_scalar(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  printf("scalar  %.3f us per permutation\n", us);
  if (dp::p2_cpu_has_avx512()) {
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) dp::p2_permute_avx512(q);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("avx-512 %.3f us per permutation (%s)\n", us, s[0] == q[0] ? "same result" : "RESULTS DIFFER");
  } else printf("no AVX-512F/DQ on this CPU\n");
}
