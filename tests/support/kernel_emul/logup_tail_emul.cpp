// TEST HARNESS (tests/ only): runs the DEVICE SOURCE of k_logup_tail (cut out of deep-prove_amd/csrc/hip_dev.hip by
// extract.py) on the SIMT emulator of simt.hpp, driven by the product's own host code (csrc/logup_tail.h: descriptor,
// message layout, parser), and byte-compares the logup-GKR proof and the transcript state with the layer-by-layer path.
// usage: logup_tail_emul            (all cases)
#include "emul_dev.hpp"
#include <cstdio>
DP_FIBER_SWITCH_ASM

static uint64_t rs = 12345;
static uint64_t rnd() { rs += 0x9E3779B97F4A7C15ULL; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

static std::vector<uint64_t> prove(dp::Dev& dev, size_t n, int ncols, int cpi, bool table, dp::Ext& after) {
  using namespace dp;
  LogUpInputDev in;
  in.is_table = table; in.columns_per_instance = cpi;
  std::vector<std::vector<u64>> host((size_t)ncols, std::vector<u64>(n));
  rs = 1000 + n * 31 + ncols * 7 + (table ? 3 : 0);
  for (auto& c : host) for (auto& v : c) v = rnd() % GL_P;
  for (auto& c : host) { DBuf b = dev.alloc(n, false); dev.upload(b, c.data()); in.columns.push_back(b); }
  if (table) { std::vector<u64> m(n); for (auto& v : m) v = rnd() % 50; DBuf b = dev.alloc(n, false); dev.upload(b, m.data()); in.multiplicities = b; }
  in.constant_challenge = ex(rnd() % GL_P, rnd() % GL_P); in.column_separation_challenge = ex(rnd() % GL_P, rnd() % GL_P);
  Transcript t = default_transcript();
  t.append_field_element(rnd() % GL_P);  // leave the sponge with a partly filled input buffer, as in the middle of a proof
  LogUpProof p = logup_batch_prove(dev, in, t);
  after = t.read_challenge();
  Writer w; w.logup(p);
  return w.w;
}

int main() {
  using namespace dp;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  emul_init_constants();
  struct Case { size_t n; int ncols, cpi; bool table; unsigned threads; bool full; };
  const Case cases[] = {{4, 1, 1, false, 64, false}, {8, 2, 1, false, 64, false}, {16, 2, 2, false, 64, false}, {32, 4, 2, false, 256, false}, {16, 1, 1, true, 64, false},
                        {64, 2, 2, true, 256, false}, {8, 3, 1, false, 1024, false}, {128, 2, 1, false, 1024, false},
                        {4, 1, 1, false, 64, true}, {8, 2, 1, false, 64, true}, {16, 4, 2, false, 256, true}, {16, 1, 1, true, 64, true}, {64, 3, 3, true, 256, true},
                        {32, 6, 2, false, 1024, true}, {256, 2, 1, false, 1024, true},
                        {4096, 1, 1, true, 1024, true}, {2048, 2, 1, false, 1024, true}, {1024, 2, 2, false, 256, false}};
  int rc = 0;
  std::vector<Case> all(std::begin(cases), std::end(cases));
  {  // seeded random shapes on top of the hand-picked ones
    uint64_t saved = rs; rs = 424242;
    for (int k = 0; k < 16; k++) {
      Case c;
      c.table = rnd() % 3 == 0;
      c.n = size_t(4) << (rnd() % 7);                 // 4 .. 256 rows
      c.cpi = 1 + (int)(rnd() % 3);
      c.ncols = c.table ? c.cpi : c.cpi * (1 + (int)(rnd() % 3));  // a table proof is one instance
      c.threads = 64u << (2 * (rnd() % 3));           // 64 / 256 / 1024
      c.full = rnd() % 2 == 0;
      all.push_back(c);
    }
    rs = saved;
  }
  // every case with the dynamic LDS the host code asks for (0) and with two small ones: the layer tables then start with rounds in global memory
  // (ping-pong through bufA / bufB) and move into LDS in a later round — every source / destination pairing of a pass runs
  const unsigned lds_caps[] = {0, 64, 256};
  for (const Case& c : all) {
    CpuDev ref; Ext ref_after;
    std::vector<uint64_t> want = prove(ref, c.n, c.ncols, c.cpi, c.table, ref_after);
    for (unsigned cap : lds_caps) {
      if (cap && c.n >= 2048 && cap == 64) continue;  // (emulation time)
      EmulDev em; em.threads = c.threads; em.full = c.full; em.lds_ext = cap; Ext em_after;
      std::vector<uint64_t> got = prove(em, c.n, c.ncols, c.cpi, c.table, em_after);
      size_t first = 0; while (first < want.size() && first < got.size() && want[first] == got[first]) first++;
      bool ok = want == got && ex_eq(ref_after, em_after) && em.taken == 1;
      printf("%s n=%zu columns=%d per_instance=%d %s threads=%u lds_ext=%u: kernel taken=%zu declined=%zu words=%zu identical=%d first_diff=%zu transcript_after=%d\n", c.full ? "full" : "tail", c.n, c.ncols, c.cpi,
             c.table ? "table " : "lookup", c.threads, cap, em.taken, em.declined, want.size(), want == got, first, ex_eq(ref_after, em_after));
      if (!ok) rc = 1;
    }
  }
  return rc;
}
