"""Golden fixture of the MatMul path (run from the repo root: python tests/golden/make_seq_golden.py): a per-token MLP over a
[8][4] activation (MatMul + bias / Requant / ReLU blocks, the last MatMul without bias; layers/matrix_mul.rs with a constant right
matrix) proved by the ORACLE; like the other fixtures it pins the oracle against regressions ("parity unpinned")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from support import oracle_lib  # noqa: E402
import deep_prove_amd as dpa  # noqa: E402
from vblob_helper import verifier_blob_for  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
o = oracle_lib.load()
mb = dpa.models.seq_mlp(8, 16, config=61)
blob, x = mb.blob(), mb.input()
h = o.model_setup(blob)
proof, out, _ = o.model_prove(h, x)
o.model_free(h)
assert (out == mb.run(x)).all()
vblob = verifier_blob_for(blob)
dpa.verify(vblob, proof, x, out)
np.savez_compressed(os.path.join(here, "seq_mlp.npz"), model_blob=blob, input=x, output=out, proof=proof, verifier_blob=vblob)
print("seq_mlp.npz:", proof.size, "proof words, output", out[:8])
