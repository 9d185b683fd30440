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


def test_throughput_mode_context_seam_calls_match_oracle(oracle):
    """dp_ctx_set_throughput_mode: a plain context (no cohort, no executor) whose seam-level calls run with device-side Fiat-Shamir and the
    fused protocol kernels — sumchecks of several shapes, commitments and a batch opening with a 2^16-entry polynomial (factored eq tables,
    classic tail, commit tail) — give the oracle's streams and leave the transcript in the oracle's state; back in latency mode the same"""
    import deep_prove_amd as dpa
    P = 0xFFFFFFFF00000001
    dev = dpa.Device(0)
    pcs = dpa.Basefold(dev, 1 << 16)
    rng = np.random.default_rng(4242)
    for mode in (True, False):
        dev.set_throughput_mode(mode)
        for nv, exts, terms in [(10, [True, True, True], [((P - 1, 0), [0, 2]), ((P - 1, 0), [0, 1]), ((11, 13), [0, 1, 2])]),
                                (17, [False, False, False], [((1, 0), [0, 1, 2])]),
                                (13, [True, True, True, True, True], [((1, 0), [0, 1, 4]), ((1, 0), [0, 3, 2]), ((7, 7), [0, 2, 4])])]:
            raw = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for e in exts]
            mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, e in zip(raw, exts)]
            vp = dpa.VirtualPolynomial(nv)
            vp.tables = list(mles)
            vp.terms = [(c, ix) for c, ix in terms]
            t = dpa.Transcript(b"test")
            proof, finals = dpa.prove_parallel(dev, vp, t)
            ot = oracle.transcript(b"test")
            oproof, ofinals = oracle.sumcheck_prove(nv, raw, exts, terms, ot)
            assert proof.size == oproof.size and (proof == oproof).all() and (finals == ofinals).all(), f"throughput={mode}: sumcheck nv={nv}"
            assert t.read_challenge() == ot.read_challenge()
            for m in mles:
                m.free()
        sizes = (16, 12, 10)
        polys = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for nv in sizes]
        mles = [dpa.Mle.from_base(dev, w) for w in polys]
        comms = [pcs.commit(m) for m in mles]
        for c, w in zip(comms, polys):
            assert list(c.root) == list(oracle.pcs_commit_root(1 << 16, w, False)), f"throughput={mode}: commitment root"
        points = [[(int(a), int(b)) for a, b in rng.integers(0, P, size=(nv, 2), dtype=np.uint64)] for nv in sizes]
        evals = [m.evaluate(pt) for m, pt in zip(mles, points)]
        t = dpa.Transcript(b"test")
        proof = pcs.batch_open(comms, points, evals, t)
        ot = oracle.transcript(b"test")
        oproof = oracle.pcs_batch_open(1 << 16, polys, [False] * 3, points, evals, ot)
        assert proof.size == oproof.size and (proof == oproof).all(), f"throughput={mode}: batch_open"
        assert t.read_challenge() == ot.read_challenge()
    dev.close()
