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


@pytest.mark.timeout(300)
def test_rx_seam_level_callers_on_several_threads(oracle):
    """dp_executor_start / attach: FOUR contexts on four host threads call the seams (dp_sumcheck_prove, dp_logup_prove, dp_pcs_commit,
    dp_pcs_batch_open) concurrently as slots of the resident executor; every result equals the oracle's, as the unattached calls do"""
    import threading
    import deep_prove_amd as dpa
    P = 0xFFFFFFFF00000001
    nthreads = 4
    devs = [dpa.Device(0) for _ in range(nthreads)]
    dpa.executor_start(0, nthreads)
    errors = []

    def job(k):
        try:
            dev = devs[k]
            pcs = dpa.Basefold(dev, 1 << 14)  # (PCS::setup builds the root tables with ordinary launches: before the context becomes a slot)
            dev.executor_attach(k)
            rng = np.random.default_rng(900 + k)
            # ---- seam 2: sumchecks of several shapes (one-workgroup, streaming and mixed-degree paths)
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
                assert proof.size == oproof.size and (proof == oproof).all() and (finals == ofinals).all(), f"thread {k}: sumcheck nv={nv}"
                assert t.read_challenge() == ot.read_challenge()
                for m in mles:
                    m.free()
            # ---- seam 1: commit + batch_open of a small family of polynomials
            polys = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for nv in (14, 12, 10)]
            mles = [dpa.Mle.from_base(dev, w) for w in polys]
            comms = [pcs.commit(m) for m in mles]
            for c, w in zip(comms, polys):
                assert list(c.root) == list(oracle.pcs_commit_root(1 << 14, w, False)), f"thread {k}: commitment root"
            points = [[(int(a), int(b)) for a, b in rng.integers(0, P, size=(nv, 2), dtype=np.uint64)] for nv in (14, 12, 10)]
            evals = [m.evaluate(pt) for m, pt in zip(mles, points)]
            assert evals == [oracle.mle_eval(w, False, pt) for w, pt in zip(polys, points)], f"thread {k}: mle_eval"
            t = dpa.Transcript(b"test")
            proof = pcs.batch_open(comms, points, evals, t)
            ot = oracle.transcript(b"test")
            oproof = oracle.pcs_batch_open(1 << 14, polys, [False] * 3, points, evals, ot)
            assert proof.size == oproof.size and (proof == oproof).all(), f"thread {k}: batch_open"
            dev.executor_detach()
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {type(e).__name__}: {e}")

    th = [threading.Thread(target=job, args=(k,)) for k in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dpa.executor_stop(0)
    for d in devs:
        d.close()
    assert not errors, errors
