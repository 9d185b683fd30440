"""Layer 1 twice (tests/support/l1_independent.py): an independent sumcheck verifier and Merkle-path check in Python big integers
accept what the oracle's provers produce — round structure, challenge schedule, final evaluations, tree layout and path order are
thereby read a second time, in a formulation that shares no code with oracle/ or csrc/ (answers "above L0 the orchestration comes from
one reading", as far as a verifier written from the protocol definition can)."""
import numpy as np
import pytest

from support import l0_independent as L
from support import l1_independent as L1

P = L.P

CASES = [
    (3, [False, False], [((1, 0), [0, 1])]),
    (5, [True, False, True], [((3, 4), [0, 1]), ((5, 9), [1, 2, 0]), ((2, 0), [2])]),
    (6, [True, True, True, True, True], [((1, 0), [0, 1, 4]), ((1, 0), [0, 3, 2]), ((7, 7), [0, 2, 4])]),  # logup layer shape
    (4, [False, True, False, True], [((1, 1), [0, 1, 2, 3])]),                                             # degree 4
]


def _parse_iop(words):
    w = [int(x) for x in words]
    p, n = 1, w[0]
    point = [(w[p + 2 * i], w[p + 2 * i + 1]) for i in range(n)]
    p += 2 * n
    rounds = []
    nr = w[p]
    p += 1
    for _ in range(nr):
        k = w[p]
        p += 1
        rounds.append([(w[p + 2 * i], w[p + 2 * i + 1]) for i in range(k)])
        p += 2 * k
    assert p == len(w)
    return point, rounds


@pytest.mark.parametrize("case", range(len(CASES)))
def test_independent_sumcheck_verifier_accepts_the_oracles_proofs(oracle, case):
    nv, exts, terms = CASES[case]
    rng = np.random.default_rng(700 + case)
    raw = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for e in exts]
    proof, finals = oracle.sumcheck_prove(nv, raw, exts, terms, oracle.transcript(b"test"))
    point, rounds = _parse_iop(proof)
    md = max(len(ix) for _, ix in terms)
    # the claimed sum by brute force over the hypercube
    total = (0, 0)
    tabs = [[(int(t[2 * k]), int(t[2 * k + 1])) for k in range(t.size // 2)] if e else [(int(v), 0) for v in t] for t, e in zip(raw, exts)]
    for b in range(1 << nv):
        for coeff, ix in terms:
            prod = (1, 0)
            for i in ix:
                prod = L.ext_mul(prod, tabs[i][b])
            total = L.ext_add(total, L.ext_mul(coeff, prod))
    chals, final_claim = L1.verify_sumcheck(total, point, rounds, nv, md, L.Transcript(b"test"))
    value, evals = L1.virtual_poly_at(raw, exts, terms, chals)
    assert value == final_claim, "the last round's claim is not the polynomial at the challenge point"
    for i in range(len(raw)):  # get_mle_final_evaluations: every table at the point, by the definition of the MLE
        if i in evals:
            assert evals[i] == (int(finals[2 * i]), int(finals[2 * i + 1]))
    # and a wrong message is refused
    rounds[1][0] = ((rounds[1][0][0] + 1) % P, rounds[1][0][1])
    with pytest.raises(AssertionError):
        L1.verify_sumcheck(total, point, rounds, nv, md, L.Transcript(b"test"))


def test_independent_merkle_paths_of_a_batch_opening(oracle):
    """every opened pair of a Basefold batch opening (oracle prover) authenticates — with the independent hash and a path check written
    from the definition — against the commitment's root (itself recomputed here from the oracle's root of the same polynomial) or the
    folded oracle's root carried in the proof; the query indices are the transcript's, squeezed by the independent sponge at the end"""
    from deep_prove_amd import wire
    rng = np.random.default_rng(42)
    shape = [(9, False), (8, True)]
    maxsize = 1 << 9
    raws = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for nv, e in shape]
    points = [[(int(a), int(b)) for a, b in rng.integers(0, P, size=(nv, 2), dtype=np.uint64)] for nv, _ in shape]
    evals = [oracle.mle_eval(w, e, p) for w, (nv, e), p in zip(raws, shape, points)]
    proof = oracle.pcs_batch_open(maxsize, raws, [e for _, e in shape], points, evals, oracle.transcript(b"test"))
    roots = [[int(v) for v in oracle.pcs_commit_root(maxsize, w, e)] for w, (_, e) in zip(raws, shape)]
    bp = wire._Reader(proof).basefold()
    assert len(bp["queries"]) == 200 and len(bp["roots"]) >= 1
    checked = 0
    for q in bp["queries"]:
        for k, cq in enumerate(q["oracle_query"]):
            words = [w for e in cq["pair"] for w in e] if cq["is_ext"] else list(cq["pair"])
            assert L1.merkle_root_from_path(words, cq["index"] >> 1, cq["path"]) == [int(v) for v in bp["roots"][k]], "oracle path"
            checked += 1
        for k, cq in enumerate(q["commitments_query"]):
            words = [w for e in cq["pair"] for w in e] if cq["is_ext"] else list(cq["pair"])
            assert cq["is_ext"] == shape[k][1]
            assert L1.merkle_root_from_path(words, cq["index"] >> 1, cq["path"]) == roots[k], "commitment path"
            checked += 1
    assert checked == 200 * (len(bp["roots"]) + len(shape))
    # a flipped word of an opened pair breaks its path
    cq = bp["queries"][0]["commitments_query"][0]
    words = list(cq["pair"])
    words[0] ^= 1
    assert L1.merkle_root_from_path(words, cq["index"] >> 1, cq["path"]) != roots[0]
