"""Proofs in flight in lock-step cohorts (csrc/hip_dev.hip `struct Cohort`, dp_model_prove_batch): several cohorts, ragged
last cohort, a batch longer than the number in flight, cohort size 1 and no cohorts at all — every proof must equal the
proof the sequential path gives for the same input, and verify. (File name sorts last: these run after the parity suite.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(pr, vb, xs, seq, conc):
    import deep_prove_amd as dpa
    proofs, outs, _ = pr.prove_batch(xs, conc)
    assert len(proofs) == len(xs)
    for i in range(len(xs)):
        assert (outs[i] == seq[i][1]).all()
        assert proofs[i].size == seq[i][0].size and (proofs[i] == seq[i][0]).all(), f"proof {i} differs from the sequential proof"
    for i in (0, len(xs) - 1):
        dpa.verify(vb, proofs[i], xs[i], outs[i])


@pytest.mark.parametrize("cohort", ["8", "3", "1", "0"])
def test_cohorts_match_sequential_mlp(dev, cohort, monkeypatch):
    """27 proofs, 19 in flight: cohorts of 8+8+3 (default), 3 x 6 + 1, singletons, and DP_COHORT=0 (own stream per proof)"""
    import deep_prove_amd as dpa
    monkeypatch.setenv("DP_COHORT", cohort)
    mb = dpa.models.mlp(2, 64, config=43)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(5000 + i) for i in range(27)])
    seq = [pr.prove(x) for x in xs]
    _check(pr, ctx.verifier_blob(), xs, seq, 19)
    assert pr.in_flight() == 19
    _check(pr, ctx.verifier_blob(), xs[:5], seq[:5], 19)  # fewer proofs than the cap: 5 in flight
    assert pr.in_flight() == 5
    ctx.free()


def test_cohorts_match_sequential_cnn(dev):
    """conv / maxpool graphs issue the same launch sequence for every input, so they run in cohorts too"""
    import deep_prove_amd as dpa
    mb = dpa.models.cnn_tiny()
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(6000 + i) for i in range(11)])
    seq = [pr.prove(x) for x in xs]
    _check(pr, ctx.verifier_blob(), xs, seq, 11)
    ctx.free()


def test_in_flight_is_a_cap_and_arenas_follow_the_model(dev, monkeypatch):
    """worker arenas are sized from the footprint of the model's earlier proofs (no 1.5 GB default per worker), so a small
    model can keep hundreds of proofs in flight (the API's maximum is 1 024); `concurrency` beyond the API limit is an argument error"""
    import deep_prove_amd as dpa
    monkeypatch.delenv("DP_WORKER_ARENA_BYTES", raising=False)
    mb = dpa.models.mlp(2, 16, config=44)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    assert pr.in_flight() == 0
    xs = np.stack([mb.input(7000 + i) for i in range(64)])
    first = pr.prove(xs[0])
    proofs, outs, _ = pr.prove_batch(xs, 64)
    assert pr.in_flight() == 64
    assert (proofs[0] == first[0]).all()
    with pytest.raises(dpa.DeepProveError):
        pr.prove_batch(xs, 1025)
    ctx.free()
