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
