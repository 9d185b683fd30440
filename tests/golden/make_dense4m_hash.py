"""Golden for BASELINE config 2 (Dense-4M): sha256 of the oracle's canonical proof stream for the synthetic model and
input of deep_prove_amd.models.dense_4m(). Takes ~90 s of CPU; the GPU parity test compares against the committed hash
so the GPU box does not have to re-run the single-threaded oracle."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import oracle_lib  # noqa: E402
import deep_prove_amd as dpa  # noqa: E402

o = oracle_lib.load()
mb = dpa.models.dense_4m()
x = mb.input()
h = o.model_setup(mb.blob())
proof, out, ms = o.model_prove(h, x)
o.model_free(h)
rec = dict(config="dense_4m", input_index=1000, output=[int(v) for v in out], proof_words=int(proof.size),
           sha256=hashlib.sha256(proof.tobytes()).hexdigest(), oracle_prove_ms=ms)
json.dump(rec, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense4m_proof.json"), "w"), indent=1)
print(rec)
