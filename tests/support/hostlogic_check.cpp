// TEST HARNESS (tests/ only): runs the product's host orchestrator over the CPU test double and the oracle on the
// same synthetic model, byte-compares the canonical proof streams, and runs the product verifier on both.
// usage: hostlogic_check <width> <seed> [tamper]
#include "../../oracle/zkml.hpp"
#include "cpu_dev.hpp"
#include "../../deep-prove_amd/csrc/zkml.h"
#include <cstdio>
#include <chrono>
#include <cmath>

static uint64_t rs;
static uint64_t rnd() { rs += 0x9E3779B97F4A7C15ULL; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static int64_t rq() { return (int64_t)(rnd() % 255) - 127; }

static dp::LayerSpec dense(size_t r, size_t c) { dp::LayerSpec l; l.kind = dp::L_DENSE; l.nrows = r; l.ncols = c; l.weights.resize(r * c); for (auto& x : l.weights) x = rq(); l.bias.resize(r); for (auto& x : l.bias) x = rq(); return l; }
static dp::LayerSpec requant_for(size_t ncols, double m) {
  // Requant::from_multiplier (requant.rs:409-437) with double arithmetic (front-end, out of scope for parity)
  dp::LayerSpec l; l.kind = dp::L_REQUANT;
  double lg = std::log2(m); unsigned ip = (unsigned)std::fabs(std::trunc(lg)); double fr = lg - std::trunc(lg);
  unsigned nm = ((ip + 25 + 7) / 8) * 8; l.right_shift = ip; l.fp_scale = nm - ip;
  l.fixed_point_multiplier = (int64_t)std::llround(std::pow(2.0, fr) * (double)(1ULL << l.fp_scale));
  l.intermediate_bit_size = 2 * 7 + dp::dp_ceil_log2(ncols) + 1;
  return l;
}
static orc::Model to_orc(const dp::ModelSpec& m) {
  orc::Model o; o.input_len = m.input_len;
  for (auto& l : m.layers) { orc::Layer x; x.kind = (orc::LayerKind)l.kind; x.nrows = l.nrows; x.ncols = l.ncols; x.weights = l.weights; x.bias = l.bias; x.right_shift = l.right_shift; x.fp_scale = l.fp_scale; x.intermediate_bit_size = l.intermediate_bit_size; x.fixed_point_multiplier = l.fixed_point_multiplier; o.layers.push_back(x); }
  return o;
}
int main(int argc, char** argv) {
  size_t W = argc > 1 ? atoi(argv[1]) : 64; rs = argc > 2 ? atoll(argv[2]) : 1; int tamper = argc > 3 ? atoi(argv[3]) : 0;
  dp::ModelSpec m; m.input_len = 4;
  dp::LayerSpec relu; relu.kind = dp::L_RELU;
  m.layers.push_back(dense(W, 4)); m.layers.push_back(requant_for(4, 0.5 / 127)); m.layers.push_back(relu);
  m.layers.push_back(dense(W, W)); m.layers.push_back(requant_for(W, 1.0 / std::sqrt((double)W) / 127)); m.layers.push_back(relu);
  m.layers.push_back(dense(4, W)); m.layers.push_back(requant_for(W, 1.0 / std::sqrt((double)W) / 127)); m.layers.push_back(relu);
  std::vector<int64_t> in = {rq(), rq(), rq(), rq()};
  // oracle
  auto t0 = std::chrono::steady_clock::now();
  orc::Context octx = orc::context_generate(to_orc(m));
  orc::Transcript ot = orc::default_transcript();
  orc::Proof op = orc::prove(octx, in, ot);
  std::vector<uint64_t> ow = orc::serialize_proof(op);
  auto t1 = std::chrono::steady_clock::now();
  // product host logic over the CPU double
  dp::CpuDev dev;
  auto ctx = dp::context_generate(dev, m);
  dp::Trace tr = dp::run_model(m, in);
  dp::Transcript pt = dp::default_transcript();
  dp::Proof pp = dp::prove(*ctx, tr, pt);
  std::vector<uint64_t> pw = dp::serialize_proof(pp);
  auto t2 = std::chrono::steady_clock::now();
  bool same = ow == pw;
  size_t first = 0; while (first < ow.size() && first < pw.size() && ow[first] == pw[first]) first++;
  printf("oracle words=%zu product words=%zu identical=%d first_diff=%zu  (oracle %.0f ms, product/cpu-double %.0f ms)\n", ow.size(), pw.size(), same, first,
         std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
  // commitments roots compare
  for (auto& kv : ctx->model_comms) for (auto& pc : kv.second) { auto& oc = octx.model_comms.at(kv.first).at(pc.first); bool eq = true; for (int k = 0; k < 4; k++) eq &= oc.first.codeword_tree.root()[k] == pc.second.tree.root.v[k]; if (!eq) printf("ROOT MISMATCH node %zu %s\n", kv.first, pc.first.c_str()); }
  // verifier on the oracle's stream
  dp::VerifierContext vc = ctx->verifier_ctx();
  dp::IO io; io.input = in; io.output = tr.out.back();
  int rc = 0;
  for (int which = 0; which < 2; which++) {
    std::vector<uint64_t> w = which ? pw : ow;
    if (tamper && which == 0) w[w.size() / 2 + tamper] ^= 1;
    try { dp::Proof q = dp::deserialize_proof(w.data(), w.size()); dp::Transcript vt = dp::default_transcript(); dp::verify(vc, q, io, vt); printf("verify(%s%s): ACCEPT\n", which ? "product" : "oracle", (tamper && !which) ? ",tampered" : ""); }
    catch (const std::exception& e) { printf("verify(%s%s): REJECT: %s\n", which ? "product" : "oracle", (tamper && !which) ? ",tampered" : "", e.what()); if (!(tamper && !which)) rc = 1; }
  }
  // roundtrip of the stream
  { dp::Proof q = dp::deserialize_proof(pw.data(), pw.size()); if (dp::serialize_proof(q) != pw) { printf("stream roundtrip FAILED\n"); rc = 1; } }
  return (same ? 0 : 2) | rc;
}
