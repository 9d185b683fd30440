"""GPU parity tests (run with -m gpu on the MI355X): every device op reached through the C ABI is compared bit-exactly
with the oracle on the same seeded inputs, and against the committed golden vectors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_base(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint64)


def rand_point(rng, k):
    return [(int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))) for _ in range(k)]


def test_native_library_is_loaded(dev):
    import deep_prove_amd as dpa
    assert os.path.exists(dpa.LIB_PATH)
    assert "gfx950" in dev.name or "hip:" in dev.name, dev.name


@pytest.mark.parametrize("k", [1, 4, 11, 16])
def test_eq_table(dev, oracle, k):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(k)
    pt = rand_point(rng, k)
    got = dpa.build_eq_x_r(dev, pt).to_numpy()
    assert (got == oracle.eq_table(pt)).all()


@pytest.mark.parametrize("nv,ext", [(1, False), (3, True), (10, False), (15, True), (18, False)])
def test_mle_evaluate(dev, oracle, nv, ext):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(100 + nv)
    w = rand_base(rng, (2 if ext else 1) << nv)
    m = dpa.Mle.from_ext(dev, w) if ext else dpa.Mle.from_base(dev, w)
    pt = rand_point(rng, nv)
    assert m.evaluate(pt) == oracle.mle_eval(w, ext, pt)
    assert (m.to_numpy() == w).all()


def test_fieldizer_and_fix_high_kat(dev, oracle):
    import deep_prove_amd as dpa
    # Fieldizer: negative -> p - |v|
    v = np.array([-1, 0, 1, -127, 127, -(1 << 40), (1 << 40)], dtype=np.int64)
    got = dpa.Mle.from_i64(dev, np.concatenate([v, [0]])).to_numpy()
    assert got.tolist()[:7] == [P - 1, 0, 1, P - 127, 127, P - (1 << 40), 1 << 40]
    # the reference's own KAT (multilinear_extensions/src/test.rs:46-82) through the device path
    m = dpa.Mle.from_base(dev, np.array([13, 97, 11, 101, 7, 103, 5, 107], dtype=np.uint64))
    r2 = m.fix_high_variables(4, 2, [(3, 0), (5, 0)]).to_numpy()
    assert r2.tolist() == [P - 23, 0, 139, 0]
    r1 = m.fix_high_variables(2, 4, [(5, 0)]).to_numpy()
    assert r1.tolist() == [P - 17, 0, 127, 0, P - 19, 0, 131, 0]


@pytest.mark.parametrize("rows,cols", [(2, 2), (4, 1024), (1024, 4), (256, 512), (1024, 1024)])
def test_fix_high_matches_oracle(dev, oracle, rows, cols):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(rows * 7 + cols)
    w = rng.integers(-127, 128, size=rows * cols, dtype=np.int64)
    m = dpa.Mle.from_i64(dev, w)
    pt = rand_point(rng, rows.bit_length() - 1)
    got = m.fix_high_variables(rows, cols, pt).to_numpy()
    assert (got == oracle.fix_high(m.to_numpy(), rows, cols, pt)).all()


SC_CASES = [
    # (nv, [is_ext per table], [(coeff, [table idx])...])
    (1, [False, False], [((1, 0), [0, 1])]),
    (2, [True], [((3, 4), [0])]),
    (7, [False, True, True], [((1, 0), [0, 1]), ((5, 9), [1, 2, 0]), ((2, 0), [2])]),
    (12, [False, False, False], [((1, 0), [0, 1, 2])]),
    (13, [True, True, True, True, True], [((1, 0), [0, 1, 4]), ((1, 0), [0, 3, 2]), ((7, 7), [0, 2, 4])]),  # logup layer shape
    (10, [True, True, True], [((P - 1, 0), [0, 2]), ((P - 1, 0), [0, 1]), ((11, 13), [0, 1, 2])]),         # initial lookup layer shape
    (16, [False, True], [((1, 0), [0, 1])]),
    # large single products: the fused fold+sum streaming kernel (K3'), then the persistent kernel for the tail
    (17, [False, False, False], [((1, 0), [0, 1, 2])]),
    (16, [True, True], [((3, 5), [0, 1])]),
    (18, [True], [((1, 0), [0])]),
    (19, [False, False], [((1, 0), [1, 0])]),
    # single products with coefficient one above the multi-workgroup range: the fused kernel runs several rounds knowing the
    # round's claimed sum and skips the t = 1 products (base and extension inputs, degrees 3, 2, 1)
    (21, [False, False, False], [((1, 0), [0, 1, 2])]),
    (20, [True, True], [((1, 0), [0, 1])]),
    (20, [True], [((1, 0), [0])]),
    (20, [False, False, False], [((1, 1), [2, 0, 1])]),  # same shape, coefficient != 1: no skipping
    # round 2: the whole range the reference dispatches (sumcheck/src/prover.rs:706-713): products of 4 and 5 tables
    # (pooling.rs:430 needs 5), alone, mixed with lower degrees (extrapolated), small and streaming sizes
    (9, [True, True, True, True, True], [((1, 0), [0, 1, 2, 3, 4])]),
    (11, [False, True, False, True, True, False], [((3, 1), [0, 1, 2, 3]), ((1, 0), [4, 5]), ((P - 2, 7), [5, 4, 3, 2, 1]), ((9, 0), [0])]),
    (17, [False, False, True, True], [((1, 0), [0, 1, 2, 3])]),
    (14, [True, False, False, True, True], [((2, 3), [0, 1, 2, 3, 4]), ((1, 0), [3, 4])]),
    # round 6: single products of base tables of 2^20 entries and more take the two-round grid (k_sc_terms2 / k_sc_fused2: rounds 1 and 2 from one pass, both folds
    # in one more) — cases 11 and 14 above are its degree-3 forms with and without the claim; degree 2 (tables in swapped order) and degree 1 here
    (20, [False, False], [((1, 0), [1, 0])]),
    (20, [False], [((1, 0), [0])]),
    (20, [False, False], [((6, 2), [0, 1])]),
]
# tables with FEWER variables than the polynomial (sumcheck_macro/src/lib.rs:236-247): (nv, [(table nv, is_ext)], terms)
SC_MIXED = [
    (6, [(6, False), (4, True), (4, False), (1, False)], [((1, 0), [0]), ((5, 6), [1, 2]), ((7, 0), [3])]),
    (12, [(12, True), (12, False), (9, True), (9, True), (9, False), (3, False)], [((1, 0), [0, 1]), ((3, 3), [2, 3, 4]), ((1, 2), [5]), ((8, 0), [4, 2])]),
    (15, [(15, False), (15, False), (15, True), (14, True), (14, False), (14, False), (14, True), (14, True)], [((1, 0), [0, 1, 2]), ((2, 9), [3, 4, 5, 6, 7])]),
]


@pytest.mark.parametrize("case", range(len(SC_CASES)))
def test_sumcheck_prove_parallel(dev, oracle, case):
    import deep_prove_amd as dpa
    nv, exts, terms = SC_CASES[case]
    rng = np.random.default_rng(1000 + case)
    raw = [rand_base(rng, (2 if e else 1) << nv) for e in exts]
    mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, e in zip(raw, exts)]
    vp = dpa.VirtualPolynomial(nv)
    vp.tables = list(mles)  # keep the oracle's table order
    vp.terms = [(c, ix) for c, ix in terms]
    t = dpa.Transcript(b"test")
    proof, finals = dpa.prove_parallel(dev, vp, t)
    ot = oracle.transcript(b"test")
    oproof, ofinals = oracle.sumcheck_prove(nv, raw, exts, terms, ot)
    assert proof.size == oproof.size and (proof == oproof).all()
    assert (finals == ofinals).all()
    assert t.read_challenge() == ot.read_challenge()  # transcripts end in the same state


@pytest.mark.parametrize("case", range(len(SC_MIXED)))
def test_sumcheck_tables_with_fewer_variables(dev, oracle, case):
    """a table of k < num_vars variables is constant in the missing ones: the round sums carry the 2^(missing) factor, after k
    folds it is a constant factor, its final evaluation is f(r_1..r_k) — through dp_sumcheck_prove, bit-identical to the oracle"""
    import deep_prove_amd as dpa
    nv, shapes, terms = SC_MIXED[case]
    rng = np.random.default_rng(4000 + case)
    raw = [rand_base(rng, (2 if e else 1) << k) for k, e in shapes]
    exts = [e for _, e in shapes]
    mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, e in zip(raw, exts)]
    vp = dpa.VirtualPolynomial(nv)
    vp.tables = list(mles)
    vp.terms = [(c, ix) for c, ix in terms]
    t = dpa.Transcript(b"test")
    proof, finals = dpa.prove_parallel(dev, vp, t)
    ot = oracle.transcript(b"test")
    oproof, ofinals = oracle.sumcheck_prove(nv, raw, exts, terms, ot)
    assert proof.size == oproof.size and (proof == oproof).all()
    assert (finals == ofinals).all()
    assert t.read_challenge() == ot.read_challenge()
    # shape violations are refused, not mis-proved: a constant table, tables of one product with different lengths
    bad = dpa.VirtualPolynomial(nv)
    bad.tables = [mles[0], mles[1]]
    bad.terms = [((1, 0), [0, 1])] if shapes[0][0] != shapes[1][0] else [((1, 0), [0, 0, 0, 0, 0, 0])]
    with pytest.raises(dpa.DeepProveError):
        dpa.prove_parallel(dev, bad, dpa.Transcript(b"test"))


def test_sumcheck_golden_vector(dev):
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "primitives.npz"))
    a, b = dpa.Mle.from_base(dev, g["poly_base"]), dpa.Mle.from_ext(dev, g["poly_ext"])
    vp = dpa.VirtualPolynomial(10)
    vp.add_mle_list([a, b])
    proof, finals = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
    assert (proof == g["sumcheck_proof"]).all() and (finals == g["sumcheck_finals"]).all()
    pt = [tuple(int(x) for x in r) for r in g["point"]]
    assert (dpa.build_eq_x_r(dev, pt).to_numpy() == g["eq_table"]).all()


@pytest.mark.parametrize("nv,ncols,cpi,table", [(2, 2, 2, False), (6, 1, 1, False), (10, 2, 2, False), (10, 5, 1, False), (8, 2, 2, True), (8, 1, 1, True), (14, 2, 2, True)])
def test_logup_batch_prove(dev, oracle, nv, ncols, cpi, table):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(nv * 31 + ncols)
    n = 1 << nv
    cols = [rng.integers(0, 1 << 20, size=n, dtype=np.uint64) for _ in range(ncols)]
    mult = rng.integers(0, 1000, size=n, dtype=np.uint64) if table else None
    cc, csc = rand_point(rng, 1)[0], rand_point(rng, 1)[0]
    t = dpa.Transcript(b"test")
    got = dpa.logup_batch_prove(dev, [dpa.Mle.from_base(dev, c) for c in cols], cpi, cc, csc, t,
                                dpa.Mle.from_base(dev, mult) if table else None)
    ot = oracle.transcript(b"test")
    exp = oracle.logup_prove(cols, cpi, cc, csc, ot, mult)
    assert got.size == exp.size and (got == exp).all()
    assert t.read_challenge() == ot.read_challenge()


@pytest.mark.parametrize("nv,ext", [(1, False), (2, True), (5, False), (7, True), (8, False), (9, True), (12, False), (13, True), (16, False)])
def test_pcs_commit_root(dev, oracle, nv, ext):
    """Basefold::commit: Merkle root over the bit-reversed RS codeword (trivial commitments for <= 7 variables)"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(500 + nv)
    w = rand_base(rng, (2 if ext else 1) << nv)
    pcs = dpa.Basefold(dev, 1 << 16)
    c = pcs.commit(dpa.Mle.from_ext(dev, w) if ext else dpa.Mle.from_base(dev, w))
    assert c.root == oracle.pcs_commit_root(1 << 16, w, ext)


@pytest.mark.parametrize("nv,ext,full", [(15, False, 15), (17, True, 18), (20, False, 20), (19, True, 20)])
def test_pcs_commit_root_large_polynomials(dev, oracle, nv, ext, full):
    """commits above 2^14 coefficients take the LDS-tiled butterfly passes (k_butterfly_pass: contiguous tile for the low
    stages, strided 128-byte-segment tiles for the high ones; Moebius and the 2n-point DIT NTT, base and extension) — roots
    against the oracle, including the coset dependence on the parameter size; the per-stage path (DP_NTT_STAGEWISE) gave the
    same roots in round 1"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(900 + nv)
    w = rand_base(rng, (2 if ext else 1) << nv)
    pcs = dpa.Basefold(dev, 1 << full)
    c = pcs.commit(dpa.Mle.from_ext(dev, w) if ext else dpa.Mle.from_base(dev, w))
    assert c.root == oracle.pcs_commit_root(1 << full, w, ext)


def test_pcs_commit_golden_and_context_dependence(dev, oracle):
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "primitives.npz"))
    pcs = dpa.Basefold(dev, 1 << 12)
    assert pcs.commit(dpa.Mle.from_base(dev, g["poly_base"])).root == [int(x) for x in g["root_base"]]
    assert pcs.commit(dpa.Mle.from_ext(dev, g["poly_ext"])).root == [int(x) for x in g["root_ext"]]
    # the coset shift depends on the parameter size (rs.rs:494-499): a different max size gives a different commitment
    pcs2 = dpa.Basefold(dev, 1 << 14)
    r2 = pcs2.commit(dpa.Mle.from_base(dev, g["poly_base"])).root
    assert r2 != [int(x) for x in g["root_base"]] and r2 == oracle.pcs_commit_root(1 << 14, g["poly_base"], False)


def test_pcs_too_large_polynomial_is_rejected(dev):
    import deep_prove_amd as dpa
    pcs = dpa.Basefold(dev, 1 << 10)
    with pytest.raises(dpa.DeepProveError):
        pcs.commit(dpa.Mle.from_base(dev, np.zeros(1 << 11, dtype=np.uint64)))


@pytest.mark.parametrize("nv,ext,full", [(5, False, 12), (7, True, 12), (8, False, 12), (9, True, 9), (12, False, 14), (13, True, 14), (16, False, 16), (18, False, 19)])
def test_pcs_open_and_verify_single_polynomial(dev, oracle, nv, ext, full):
    """PCS::open / PCS::verify of one polynomial (mpcs/src/basefold.rs:466-544, 863-962): the device's commit phase (sumcheck rounds
    interleaved with FRI folds and Merkle trees, the fused commit tail included) and query phase give the oracle's stream and
    transcript state, base and extension, trivial and not, parameters equal to and larger than the polynomial"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(7000 + nv)
    maxsize = 1 << full
    pcs = dpa.Basefold(dev, maxsize)
    w = rand_base(rng, (2 if ext else 1) << nv)
    m = dpa.Mle.from_ext(dev, w) if ext else dpa.Mle.from_base(dev, w)
    c = pcs.commit(m)
    point = rand_point(rng, nv)
    ev = m.evaluate(point)
    t, ot = dpa.Transcript(b"test"), oracle.transcript(b"test")
    proof = pcs.open(c, point, ev, t)
    exp = oracle.pcs_open(maxsize, w, ext, point, ot)
    assert proof.size == exp.size and (proof == exp).all()
    assert t.read_challenge() == ot.read_challenge()
    dpa.Basefold.verify(maxsize, c.root, nv, not ext, point, ev, proof, dpa.Transcript(b"test"))
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.verify(maxsize, c.root, nv, not ext, point, ((ev[0] + 1) % P, ev[1]), proof, dpa.Transcript(b"test"))
    bad = proof.copy()
    bad[-5] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.verify(maxsize, c.root, nv, not ext, point, ev, bad, dpa.Transcript(b"test"))


@pytest.mark.parametrize("shape", [[(10, False)], [(12, False), (10, False), (12, True), (9, False)], [(8, False), (8, True), (14, False), (11, True), (14, False)]])
def test_pcs_batch_open_and_verify(dev, oracle, shape):
    import deep_prove_amd as dpa
    rng = np.random.default_rng(len(shape) * 17 + shape[0][0])
    maxsize = 1 << 14
    pcs = dpa.Basefold(dev, maxsize)
    raws = [rand_base(rng, (2 if e else 1) << nv) for nv, e in shape]
    mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, (nv, e) in zip(raws, shape)]
    comms = [pcs.commit(m) for m in mles]
    points = [rand_point(rng, nv) for nv, _ in shape]
    evals = [m.evaluate(p) for m, p in zip(mles, points)]
    t = dpa.Transcript(b"test")
    proof = pcs.batch_open(comms, points, evals, t)
    ot = oracle.transcript(b"test")
    exp = oracle.pcs_batch_open(maxsize, raws, [e for _, e in shape], points, evals, ot)
    assert proof.size == exp.size and (proof == exp).all()
    assert t.read_challenge() == ot.read_challenge()
    roots = [c.root for c in comms]
    args = (maxsize, roots, [nv for nv, _ in shape], [not e for _, e in shape], points, evals)
    dpa.Basefold.batch_verify(*args, proof, dpa.Transcript(b"test"))
    bad = proof.copy()
    bad[7] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(*args, bad, dpa.Transcript(b"test"))
    wrong = list(evals)
    wrong[0] = ((evals[0][0] + 1) % P, evals[0][1])
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize, roots, [nv for nv, _ in shape], [not e for _, e in shape], points, wrong, proof, dpa.Transcript(b"test"))


@pytest.mark.parametrize("case", [1, 2])
def test_device_sumcheck_accepted_by_the_independent_verifier(dev, case):
    """the DEVICE prover's proof through tests/support/l1_independent.py (a sumcheck verifier written from the protocol definition on
    the independent Poseidon2 / transcript): no oracle code and no product code decides this acceptance"""
    import deep_prove_amd as dpa
    from support import l0_independent as L
    from support import l1_independent as L1
    from test_l1_independent import CASES, _parse_iop
    nv, exts, terms = CASES[case]
    rng = np.random.default_rng(700 + case)
    raw = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for e in exts]
    mles = [dpa.Mle.from_ext(dev, w) if e else dpa.Mle.from_base(dev, w) for w, e in zip(raw, exts)]
    vp = dpa.VirtualPolynomial(nv)
    vp.tables = list(mles)
    vp.terms = [(c, ix) for c, ix in terms]
    proof, finals = dpa.prove_parallel(dev, vp, dpa.Transcript(b"test"))
    for m in mles:
        m.free()
    point, rounds = _parse_iop(proof)
    tabs = [[(int(t[2 * k]), int(t[2 * k + 1])) for k in range(t.size // 2)] if e else [(int(v), 0) for v in t] for t, e in zip(raw, exts)]
    total = (0, 0)
    for b in range(1 << nv):
        for coeff, ix in terms:
            prod = (1, 0)
            for i in ix:
                prod = L.ext_mul(prod, tabs[i][b])
            total = L.ext_add(total, L.ext_mul(coeff, prod))
    chals, final_claim = L1.verify_sumcheck(total, point, rounds, nv, max(len(ix) for _, ix in terms), L.Transcript(b"test"))
    value, evals = L1.virtual_poly_at(raw, exts, terms, chals)
    assert value == final_claim
    for i, ev in evals.items():
        assert ev == (int(finals[2 * i]), int(finals[2 * i + 1]))
