// host Poseidon2-w8 permutation rate (the transcript sponge): g++ -O2 -std=c++17 -o /tmp/p2hb tools/p2_host_bench.cpp && /tmp/p2hb
#include "../deep-prove_amd/csrc/poseidon2.h"
#include <chrono>
#include <cstdio>
int main() {
  dp::u64 s[8] = {1,2,3,4,5,6,7,8};
  auto t0 = std::chrono::steady_clock::now(); int N = 300000;
  for (int i = 0; i < N; i++) dp::hostnc::permute(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  printf("hostnc::permute: %.3f us (%llu)\n", us, (unsigned long long)s[0]);
}
