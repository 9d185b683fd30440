"""Generates the committed golden fixtures from the oracle (run from the repo root: python tests/golden/make_golden.py).
The reference cannot run in this environment (pure Rust, un-vendored deps), so these vectors pin the ORACLE against
regressions and give the GPU parity tests fixed known answers; they are not reference outputs ("parity unpinned")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
from vblob_helper import verifier_blob_for  # noqa: E402  (the verifier blob without a GPU: the product's context over the CPU double)

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
