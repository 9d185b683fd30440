"""Generates the committed golden fixtures from the oracle (run from the repo root: python tests/golden/make_golden.py).
The reference cannot run in this environment (pure Rust, un-vendored deps), so these vectors pin the ORACLE against
regressions and give the GPU parity tests fixed known answers; they are not reference outputs ("parity unpinned")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import oracle_lib  # noqa: E402
import deep_prove_amd as dpa  # noqa: E402  (models + host-only verifier blob layout)

P = 0xFFFFFFFF00000001
here = os.path.dirname(os.path.abspath(__file__))
o = oracle_lib.load()

# ---- a width-8 MLP proof
mb = dpa.models.mlp(2, 8, config=9)
blob, x = mb.blob(), mb.input()
h = o.model_setup(blob)
proof, out, _ = o.model_prove(h, x)
o.model_free(h)
# the verifier blob needs the commitments: rebuild it from the proof-independent context through the hostlogic harness
# format (see capi.cpp vctx_to_words); produced here with a tiny C++ helper to stay independent of a GPU.
import subprocess, tempfile  # noqa: E402
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


vblob = verifier_blob_for(blob)
dpa.verify(vblob, proof, x, out)
np.savez_compressed(os.path.join(here, "mlp_w8.npz"), model_blob=blob, input=x, output=out, proof=proof, verifier_blob=vblob)

# ---- a small CNN proof (conv -> requant -> relu -> maxpool -> flatten -> dense -> requant -> relu -> dense -> requant)
mb = dpa.models.cnn_tiny()
blob, x = mb.blob(), mb.input()
h = o.model_setup(blob)
proof, out, _ = o.model_prove(h, x)
o.model_free(h)
assert (out == mb.run(x)).all()
vblob = verifier_blob_for(blob)
dpa.verify(vblob, proof, x, out)
np.savez_compressed(os.path.join(here, "cnn_tiny.npz"), model_blob=blob, input=x, output=out, proof=proof, verifier_blob=vblob)

# ---- primitive vectors
rng = np.random.default_rng(2025)
perm_in = rng.integers(0, P, size=8, dtype=np.uint64)
point = rng.integers(0, P, size=(6, 2), dtype=np.uint64)
poly_base = rng.integers(0, P, size=1 << 10, dtype=np.uint64)
poly_ext = rng.integers(0, P, size=2 << 10, dtype=np.uint64)
t = o.transcript(b"test")
sp, sf = o.sumcheck_prove(10, [poly_base, poly_ext], [False, True], [((1, 0), [0, 1])], t)
np.savez_compressed(os.path.join(here, "primitives.npz"), perm_in=perm_in, perm_out=o.permute(perm_in), point=point,
                    eq_table=o.eq_table([tuple(int(v) for v in r) for r in point]), poly_base=poly_base, poly_ext=poly_ext,
                    root_base=np.array(o.pcs_commit_root(1 << 12, poly_base, False), dtype=np.uint64),
                    root_ext=np.array(o.pcs_commit_root(1 << 12, poly_ext, True), dtype=np.uint64),
                    sumcheck_proof=sp, sumcheck_finals=sf)
print("golden fixtures written:", os.listdir(here))
