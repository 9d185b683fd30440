"""The RESIDENT EXECUTOR (csrc/rx.h, rx.hip; DP_RX=1): the proofs of a batch are slots of two persistent kernels, their launches
are step descriptors — every proof must equal the proof the sequential (one launch per step) path gives for the same input, word
for word, and verify. Also: a batch longer than the slots, models with the LDS-tiled commit passes, the CNN graph, and the golden
Dense-4M proof (sha256 of the oracle's stream) out of an executor batch. (Sorted last: a hang here hides nothing else.)"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(pr, vb, xs, seq, conc):
    import deep_prove_amd as dpa
    proofs, outs, _ = pr.prove_batch(xs, conc)
    assert len(proofs) == len(xs)
    for i in range(len(xs)):
        assert (outs[i] == seq[i][1]).all()
        assert proofs[i].size == seq[i][0].size and (proofs[i] == seq[i][0]).all(), f"proof {i} differs from the sequential proof"
    for i in (0, len(xs) - 1):
        dpa.verify(vb, proofs[i], xs[i], outs[i])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("width,nproofs,conc", [(64, 27, 19), (256, 12, 12), (16, 70, 64)])
def test_rx_batches_match_sequential_mlp(dev, monkeypatch, width, nproofs, conc):
    import deep_prove_amd as dpa
    monkeypatch.setenv("DP_RX", "1")
    mb = dpa.models.mlp(2, width, config=43)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(5000 + i) for i in range(nproofs)])
    monkeypatch.setenv("DP_RX", "0")
    seq = [pr.prove(x) for x in xs]
    monkeypatch.setenv("DP_RX", "1")
    _check(pr, ctx.verifier_blob(), xs, seq, conc)
    assert pr.in_flight() == min(conc, nproofs)
    _check(pr, ctx.verifier_blob(), xs[:5], seq[:5], conc)  # a second session of the same engine, fewer slots
    ctx.free()


@pytest.mark.timeout(300)
def test_rx_dense4m_golden_sha256(dev, monkeypatch):
    """the oracle's golden Dense-4M proof at a non-zero index of an executor batch (what bench.py times with DP_RX=1)"""
    import deep_prove_amd as dpa
    monkeypatch.setenv("DP_RX", "1")
    gold = json.load(open(os.path.join(HERE, "golden", "dense4m_proof.json")))
    mb = dpa.models.dense_4m()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    pr.prove(mb.input(1000))  # sizes the worker arenas
    xs = np.stack([mb.input(1000 + i) for i in range(24)])
    xs[5] = mb.input(gold["input_index"])
    proofs, outs, _ = pr.prove_batch(xs, 24)
    g = proofs[5]
    assert g.size == gold["proof_words"] and hashlib.sha256(g.tobytes()).hexdigest() == gold["sha256"]
    assert [int(v) for v in outs[5]] == gold["output"]
    verdicts, _ = dpa.verify_batch(ctx.verifier_blob(), proofs, xs, outs, dev=dev)
    assert not verdicts.any()
    ctx.free()
