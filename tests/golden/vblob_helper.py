"""The verifier blob of a model WITHOUT a GPU (shared by the golden generators): the product's Context::generate over the CPU test
double (tests/support/cpu_dev.hpp), serialised in the layout dp_model_verifier_blob hands out (csrc/zkml.h vctx_to_words)."""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the verifier blob needs the commitments: rebuild it from the proof-independent context through the hostlogic harness
# format (see capi.cpp vctx_to_words); produced here with a tiny C++ helper to stay independent of a GPU.
src = r'''
#include "oracle/zkml.hpp"
#include "tests/support/cpu_dev.hpp"
#include "deep-prove_amd/csrc/zkml.h"
#include <cstdio>
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f) / 8; fseek(f, 0, SEEK_SET);
  std::vector<int64_t> b(n); if (fread(b.data(), 8, n, f) != (size_t)n) return 1; fclose(f);
  size_t pos = 0; dp::ModelSpec m; m.input_len = b[pos++]; size_t nl = b[pos++];
  for (size_t i = 0; i < nl; i++) { dp::LayerSpec l; l.kind = (int)b[pos++];
    if (l.kind == 0) { l.nrows = b[pos++]; l.ncols = b[pos++]; l.weights.assign(b.begin() + pos, b.begin() + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols; l.bias.assign(b.begin() + pos, b.begin() + pos + l.nrows); pos += l.nrows; }
    else if (l.kind == 1) { l.right_shift = b[pos++]; l.fp_scale = b[pos++]; l.fixed_point_multiplier = b[pos++]; l.intermediate_bit_size = b[pos++]; }
    else if (l.kind == 3) { l.kw = b[pos++]; l.kx = b[pos++]; l.real_nw = b[pos++]; l.nw = b[pos++]; for (int k = 0; k < 3; k++) l.unp_out[k] = b[pos++];
      size_t nf = l.kw * l.kx * l.real_nw * l.real_nw; l.weights.assign(b.begin() + pos, b.begin() + pos + nf); pos += nf; l.bias.assign(b.begin() + pos, b.begin() + pos + l.kw); pos += l.kw; }
    else if (l.kind == 4) { for (int k = 0; k < 3; k++) l.pin[k] = b[pos++]; }
    else if (l.kind == 9) { l.add_left = b[pos++]; l.add_right = b[pos++]; l.nrows = b[pos++]; l.ncols = b[pos++]; l.weights.assign(b.begin() + pos, b.begin() + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols; }
    else if (l.kind == 8) { l.nrows = b[pos++]; l.ncols = b[pos++]; l.weights.assign(b.begin() + pos, b.begin() + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols; }
    else if (l.kind == 7) { l.add_left = b[pos++]; l.add_right = b[pos++]; size_t cnt = b[pos++]; l.weights.assign(b.begin() + pos, b.begin() + pos + cnt); pos += cnt; }
    else if (l.kind == 6) { l.nrows = b[pos++]; l.ncols = b[pos++]; size_t fl = b[pos++], hb = fl & 1; l.mm_transpose = (fl & 2) != 0; l.weights.assign(b.begin() + pos, b.begin() + pos + l.nrows * l.ncols); pos += l.nrows * l.ncols;
      if (hb) { l.bias.assign(b.begin() + pos, b.begin() + pos + l.ncols); pos += l.ncols; } }
    m.layers.push_back(l); }
  dp::CpuDev dev; auto ctx = dp::context_generate(dev, m);
  std::vector<uint64_t> w = dp::vctx_to_words(ctx->verifier_ctx());  // the layout dp_model_verifier_blob hands out
  f = fopen(argv[2], "wb"); fwrite(w.data(), 8, w.size(), f); fclose(f); return 0; }
'''


def verifier_blob_for(blob):
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "v.cpp"), "w").write(src)
        blob.tofile(os.path.join(td, "blob.bin"))
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", ROOT, "-o", os.path.join(td, "v"), os.path.join(td, "v.cpp")])
        subprocess.check_call([os.path.join(td, "v"), os.path.join(td, "blob.bin"), os.path.join(td, "vb.bin")])
        return np.fromfile(os.path.join(td, "vb.bin"), dtype=np.uint64)


