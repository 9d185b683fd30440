"""The C ABI exercised by something other than ctypes: tests/support/c_consumer.c (strict C11, pthreads) drives sumcheck
(degree-4 product + a short table), four concurrent PCS commits on one context, batch_open / batch_verify, the trivial
open / verify pair and a model proof; every file it writes is compared with the oracle here."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def rand_base(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint64)


def test_c11_consumer_matches_the_oracle(oracle, tmp_path):
    import deep_prove_amd as dpa
    from test_capi import build_c_consumer
    exe = build_c_consumer()
    rng = np.random.default_rng(77)
    d = str(tmp_path)
    # seam 2 inputs: four 10-variable tables (the third one extension) and a 7-variable one
    sc = [rand_base(rng, (2 if i == 2 else 1) << 10) for i in range(4)]
    short = rand_base(rng, 1 << 7)
    for i, w in enumerate(sc):
        w.tofile(os.path.join(d, f"sc_tab{i}.bin"))
    short.tofile(os.path.join(d, "sc_short.bin"))
    # seam 1 inputs
    polys = [rand_base(rng, 1 << k) for k in (13, 12, 10, 6)]
    for i, w in enumerate(polys):
        w.tofile(os.path.join(d, f"poly{i}.bin"))
    pts = rand_base(rng, 2 * 23)
    pts.tofile(os.path.join(d, "points.bin"))
    # the model
    mb = dpa.models.mlp(2, 32, config=61)
    x = mb.input(9)
    np.ascontiguousarray(mb.blob(), dtype=np.int64).tofile(os.path.join(d, "model.bin"))
    np.ascontiguousarray(x, dtype=np.int64).tofile(os.path.join(d, "input.bin"))

    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c11 consumer: ok" in r.stdout, r.stdout + r.stderr
    rd = lambda name: np.fromfile(os.path.join(d, name), dtype=np.uint64)  # noqa: E731

    # sumcheck
    ot = oracle.transcript(b"test")
    oproof, ofinals = oracle.sumcheck_prove(10, sc + [short], [False, False, True, False, False], [((1, 0), [0, 1, 2, 3]), ((12345, 678), [4])], ot)
    assert (rd("sc_proof.bin") == oproof).all() and (rd("sc_finals.bin") == ofinals).all()
    assert tuple(int(v) for v in rd("sc_after.bin")) == ot.read_challenge()
    # commits made concurrently from four threads == the oracle's roots
    roots = rd("roots.bin")
    for i, w in enumerate(polys):
        assert [int(v) for v in roots[4 * i:4 * i + 4]] == oracle.pcs_commit_root(1 << 13, w, False), f"root of poly{i}"
    # batch opening of poly0 and poly2
    p0 = [(int(pts[2 * i]), int(pts[2 * i + 1])) for i in range(13)]
    p2 = [(int(pts[26 + 2 * i]), int(pts[26 + 2 * i + 1])) for i in range(10)]
    ev = rd("evals.bin")
    assert (ev[0:2] == oracle.mle_eval(polys[0], False, p0)).all() and (ev[2:4] == oracle.mle_eval(polys[2], False, p2)).all()
    obo = oracle.pcs_batch_open(1 << 13, [polys[0], polys[2]], [False, False], [p0, p2], [(int(ev[0]), int(ev[1])), (int(ev[2]), int(ev[3]))], oracle.transcript(b"test"))
    got = rd("bo_proof.bin")
    assert got.size == obo.size and (got == obo).all()
    # trivial opening: the stream carries the raw table
    tp = rd("triv_proof.bin")
    assert tp.size > 64 and (tp[-64:] == polys[3]).all() and int(tp[-65]) == 64  # ... {count 1, base, length 64, the 64 evaluations}
    # the 13-variable polynomial opened alone: PCS::open through the C entry == the oracle's commit phase + query phase
    oop = oracle.pcs_open(1 << 13, polys[0], False, p0, oracle.transcript(b"test"))
    got = rd("open_proof.bin")
    assert got.size == oop.size and (got == oop).all()
    # the model proof
    h = oracle.model_setup(mb.blob())
    oproof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (rd("model_proof.bin") == oproof).all()
    assert (np.fromfile(os.path.join(d, "model_out.bin"), dtype=np.int64) == oout).all()
