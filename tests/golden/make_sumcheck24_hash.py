"""BASELINE config 5 at its stated size, pinned: the oracle's proof of the standalone sumcheck bench.py times (one product of
k = 3 base-field MLEs of 2^nv SplitMix64-derived canonical elements, coefficient one, transcript label "test" — the shape of
sumcheck/benches/devirgo_sumcheck.rs:42-101) for nv = 22 and nv = 24. Writes tests/golden/sumcheck24.json: sha256 of the proof
stream, the final evaluations, and the transcript's next challenge (the sponge state after the proof).
Run from the repo root: python tests/golden/make_sumcheck24_hash.py   (about a minute, ~3 GB of host memory for nv = 24).
These are ORACLE outputs (the reference cannot run here: "parity unpinned")."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import oracle_lib  # noqa: E402
import deep_prove_amd as dpa  # noqa: E402  (models.splitmix64: the generator bench.py uses)

P = 0xFFFFFFFF00000001
K = 3


def tables(nv, k=K):
    """the tables of bench.py:sumcheck24 — seed 0xD33B0000 ^ (5 << 32) ^ j, SURVEY 8(d)"""
    n = 1 << nv
    return [dpa.models.splitmix64(0xD33B0000 ^ (5 << 32) ^ j, n) % np.uint64(P) for j in range(k)]


def main():
    o = oracle_lib.load()
    out = {"note": "oracle outputs for BASELINE config 5 (standalone sumcheck, product of 3 base MLEs, label 'test'); generator: tests/golden/make_sumcheck24_hash.py",
           "k": K, "cases": {}}
    for nv in (22, 24):
        tabs = tables(nv)
        t = o.transcript(b"test")
        proof, finals = o.sumcheck_prove(nv, tabs, [False] * K, [((1, 0), list(range(K)))], t)
        nxt = t.read_challenge()
        out["cases"][str(nv)] = {"nv": nv, "proof_words": int(proof.size), "sha256": hashlib.sha256(proof.tobytes()).hexdigest(),
                                 "finals": [int(v) for v in finals], "next_challenge": [int(v) for v in nxt]}
        print(nv, out["cases"][str(nv)]["sha256"], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "sumcheck24.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
