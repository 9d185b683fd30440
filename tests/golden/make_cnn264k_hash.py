"""Golden for BASELINE config 3 (CNN-264k on CIFAR-10 shapes): sha256 of the oracle's canonical proof stream for the
synthetic model and input of deep_prove_amd.models.cnn_264k(). Takes ~40 s of CPU; the GPU parity test compares against
the committed hash so the GPU box does not have to re-run the single-threaded oracle."""
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
mb = dpa.models.cnn_264k()
x = mb.input()
assert (mb.run(x)[:10] != 0).any()
h = o.model_setup(mb.blob())
proof, out, ms = o.model_prove(h, x)
o.model_free(h)
assert (out == mb.run(x)).all()  # the FFT convolution of the oracle against a direct correlation in numpy
rec = dict(config="cnn_264k", input_index=1000, output=[int(v) for v in out], proof_words=int(proof.size),
           sha256=hashlib.sha256(proof.tobytes()).hexdigest(), oracle_prove_ms=ms)
json.dump(rec, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cnn264k_proof.json"), "w"), indent=1)
print(rec)
