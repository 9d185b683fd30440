"""Layer 0 twice: tests/support/l0_independent.py (Python big ints, written from the Poseidon2 paper, the Grain LFSR
specification and ff_ext/src/lib.rs:167-236 — explicit matrices, per-index eq products, `%` arithmetic) against the oracle
(C++) and the product's host code (C++, literal constant table, add chains, 2^64 = 2^32 - 1 folding). Parity with the
reference stays unpinned (no reference binary, no KAT: SURVEY.md 8c); this de-correlates the bottom layer: the same
misreading would have to be made in two unrelated formulations."""
import os
import re

import numpy as np
import pytest

from support import l0_independent as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd_elems(rng, n):
    return [int(v) for v in rng.integers(0, L.P, size=n, dtype=np.uint64)]


def test_constants_regenerated_from_the_lfsr_equal_the_products_literal_table_and_the_oracles(oracle):
    src = open(os.path.join(ROOT, "deep-prove_amd", "csrc", "poseidon2.h")).read()
    body = src[src.index("POSEIDON2_RC_HOST[DP_POSEIDON2_RC_WORDS] = {"):]
    lit = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ULL", body[:body.index("};")])]
    mine = [v for r in L.RC_EXT_INITIAL for v in r] + L.RC_INTERNAL + [v for r in L.RC_EXT_TERMINAL for v in r] + L.DIAG_M1
    assert len(lit) == 94 and lit == mine
    assert [int(v) for v in oracle.rc_table()] == mine


def test_permutation_compress_hash(oracle):
    rng = np.random.default_rng(2)
    states = [[0] * 8, [L.P - 1] * 8, list(range(8)), [1 << 63] + [0] * 7] + [rnd_elems(rng, 8) for _ in range(40)]
    for s in states:
        assert [int(v) for v in oracle.permute(np.array(s, dtype=np.uint64))] == L.permute(s)
    for _ in range(20):
        x, y = rnd_elems(rng, 4), rnd_elems(rng, 4)
        assert oracle.compress(x, y) == L.compress(x, y)
    for n in (1, 3, 4, 5, 8, 9, 64, 129):
        e = rnd_elems(rng, n)
        assert oracle.hash_or_noop(e) == L.hash_or_noop(e)


def test_transcripts_three_ways(oracle):
    """random schedules of append / challenge: oracle, the product's host transcript (dp_transcript_*, no GPU needed) and the
    independent sponge agree on every challenge"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(3)
    for trial in range(6):
        label = bytes(rng.integers(97, 123, size=int(rng.integers(1, 20)), dtype=np.uint8))
        a, b, c = oracle.transcript(label), dpa.Transcript(label), L.Transcript(label)
        for _ in range(40):
            op = int(rng.integers(0, 3))
            if op == 0:
                e = rnd_elems(rng, int(rng.integers(1, 7)))
                a.append_field_elements(e); b.append_field_elements(e); c.append_field_elements(e)
            elif op == 1:
                m = bytes(rng.integers(0, 256, size=int(rng.integers(1, 30)), dtype=np.uint8))
                a.append_message(m); b.append_message(m); c.append_message(m)
            else:
                lab = [b"Internal round", b"sumcheck round", b"commit round", b"query indices", b"batch coeffs"][int(rng.integers(0, 5))]
                x, y, z = a.get_and_append_challenge(lab), b.get_and_append_challenge(lab), c.get_and_append_challenge(lab)
                assert tuple(x) == tuple(y) == tuple(z)
        assert tuple(a.read_challenge()) == tuple(b.read_challenge()) == tuple(c.read_challenge())


def test_extension_arithmetic_through_eq_tables_and_evaluations(oracle):
    rng = np.random.default_rng(4)
    for k in (1, 3, 6):
        pt = [tuple(rnd_elems(rng, 2)) for _ in range(k)]
        mine = L.eq_table(pt)
        got = oracle.eq_table(pt)
        assert [(int(got[2 * i]), int(got[2 * i + 1])) for i in range(1 << k)] == mine
        vals = rnd_elems(rng, 2 << k)
        ext = [(vals[2 * i], vals[2 * i + 1]) for i in range(1 << k)]
        assert oracle.mle_eval(np.array(vals, dtype=np.uint64), True, pt) == L.mle_eval(ext, pt)
        base = rnd_elems(rng, 1 << k)
        assert oracle.mle_eval(np.array(base, dtype=np.uint64), False, pt) == L.mle_eval([(v, 0) for v in base], pt)
    for _ in range(20):
        a, b = tuple(rnd_elems(rng, 2)), tuple(rnd_elems(rng, 2))
        assert L.ext_mul(L.ext_mul(a, b), L.ext_inv(b)) == a


def test_whole_proof_transcript_replayed_through_the_independent_sponge(oracle):
    """every element the oracle absorbs and every challenge it draws while proving a (small) model, in order, replayed through
    the independent permutation + sponge: all challenges equal. Covers ~10^4 sponge operations of a real protocol run."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_w8.npz"))
    h = oracle.model_setup(g["model_blob"])
    oracle.trace_begin()
    proof, out, _ = oracle.model_prove(h, g["input"])
    tr = oracle.trace_take()
    oracle.model_free(h)
    assert (proof == g["proof"]).all()
    ops = tr.reshape(-1, 2)
    assert ops.shape[0] > 1500 and int((ops[:, 0] == 1).sum()) > 300
    d = L.Duplex()
    for kind, val in ops:
        if int(kind) == 0:
            d.observe(int(val))
        else:
            assert d.sample() == int(val)
