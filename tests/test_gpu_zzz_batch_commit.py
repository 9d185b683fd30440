"""GPU parity of dp_pcs_batch_commit / dp_pcs_simple_batch_open (run with -m gpu on the MI355X), through the C ABI against the oracle.
Its own file, sorted after every other GPU test: the one kernel behind it (k_batch_row_hash) was written after round 2's GPU budget
was spent — validated on the CPU double and on the SIMT emulator (tests/test_hostlogic.py, tests/test_kernel_emul.py), first run on
hardware by whoever runs this file."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


def rand_base(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint64)


def rand_point(rng, k):
    return [(int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))) for _ in range(k)]


@pytest.mark.parametrize("nv,ext,k,full", [(4, False, 4, 11), (6, True, 4, 11), (10, False, 1, 11), (10, False, 4, 11), (10, True, 4, 11), (9, True, 2, 9), (12, False, 9, 14), (11, True, 5, 12), (16, False, 3, 16), (3, False, 32, 11)])
def test_pcs_batch_commit_and_simple_batch_open(dev, oracle, nv, ext, k, full):
    """PCS::batch_commit / simple_batch_open / simple_batch_verify (mpcs/src/basefold.rs:356-446, 777-861, 1100-1203; the shapes of the
    reference's simple_batch_commit_open_verify_goldilocks test: base and extension, batch sizes 1 and 4, a trivial size; plus batches whose
    rows go through the sponge): the device's common tree (row hashes k_batch_row_hash + the ordinary tree) gives the oracle's root, the
    opening the oracle's stream and transcript state; the host verifier accepts it and rejects a wrong evaluation and a flipped word"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(7300 + 16 * nv + k)
    maxsize = 1 << full
    pcs = dpa.Basefold(dev, maxsize)
    raws = [rand_base(rng, (2 if ext else 1) << nv) for _ in range(k)]
    mles = [dpa.Mle.from_ext(dev, w) if ext else dpa.Mle.from_base(dev, w) for w in raws]
    c = pcs.batch_commit(mles)
    point = rand_point(rng, nv)
    evals = [m.evaluate(point) for m in mles]
    t, ot = dpa.Transcript(b"test"), oracle.transcript(b"test")
    proof = pcs.simple_batch_open(c, point, t)
    root, exp = oracle.pcs_simple_batch_open(maxsize, raws, ext, point, ot)
    assert c.root == root
    assert proof.size == exp.size and (proof == exp).all()
    assert t.read_challenge() == ot.read_challenge()
    dpa.Basefold.simple_batch_verify(maxsize, c.root, nv, not ext, point, evals, proof, dpa.Transcript(b"test"))
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.simple_batch_verify(maxsize, c.root, nv, not ext, point, evals[:-1] + [((evals[-1][0] + 1) % P, evals[-1][1])], proof, dpa.Transcript(b"test"))
    bad = proof.copy()
    bad[-5] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.simple_batch_verify(maxsize, c.root, nv, not ext, point, evals, bad, dpa.Transcript(b"test"))
    c.free()
    with pytest.raises(dpa.DeepProveError):  # different sizes cannot share a tree (basefold.rs:376-383)
        pcs.batch_commit([mles[0], dpa.Mle.from_base(dev, rand_base(rng, 1 << (nv + 1)))])


EVAL_SHAPES = [  # (polynomials as (num_vars, ext), point lengths, evaluations as (poly, point)) — tests/test_verifier_abi.py
    ([(9, False), (9, False)], [9], [(0, 0), (1, 0)]),
    ([(10, False), (10, True), (9, False), (9, False)], [10, 9], [(0, 0), (1, 0), (2, 1), (3, 1)]),
    ([(10, True)], [10, 10], [(0, 0), (0, 1)]),
    ([(13, False), (9, True), (13, True)], [13, 13, 9], [(0, 0), (2, 0), (0, 1), (1, 2), (2, 1)]),
]


@pytest.mark.parametrize("shape", range(len(EVAL_SHAPES)))
def test_pcs_batch_open_general_evaluation_lists(dev, oracle, shape):
    """PCS::batch_open / batch_verify with `evals: &[Evaluation]` (mpcs/src/basefold.rs:546-770, 964-1098; the reference's
    run_batch_commit_open_verify shapes: polynomials sharing a point, several sizes; one polynomial at two points; a mixed list): the
    device path (existing kernels only: one (f, eq) pair per evaluation, commit phase per commitment) gives the oracle's stream and
    transcript state; the host verifier accepts it and rejects a wrong value"""
    import deep_prove_amd as dpa
    polys_s, points_s, evals_s = EVAL_SHAPES[shape]
    rng = np.random.default_rng(7500 + shape)
    maxsize = 1 << 14
    pcs = dpa.Basefold(dev, maxsize)
    raws = [rand_base(rng, (2 if e else 1) << nv) for nv, e in polys_s]
    mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, (nv, e) in zip(raws, polys_s)]
    comms = [pcs.commit(m) for m in mles]
    points = [rand_point(rng, n) for n in points_s]
    evals = [(pi, qi, mles[pi].evaluate(points[qi])) for pi, qi in evals_s]
    t, ot = dpa.Transcript(b"test"), oracle.transcript(b"test")
    proof = pcs.batch_open_evals(comms, points, evals, t)
    exp = oracle.pcs_batch_open_evals(maxsize, raws, [e for _, e in polys_s], points, evals, ot)
    assert proof.size == exp.size and (proof == exp).all()
    assert t.read_challenge() == ot.read_challenge()
    args = (maxsize, [c.root for c in comms], [nv for nv, _ in polys_s], [not e for _, e in polys_s], points)
    dpa.Basefold.batch_verify_evals(*args, evals, proof, dpa.Transcript(b"test"))
    v = evals[0][2]
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify_evals(*args, [(evals[0][0], evals[0][1], ((v[0] + 1) % P, v[1]))] + evals[1:], proof, dpa.Transcript(b"test"))
