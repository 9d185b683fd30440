"""The verifier side of the sumcheck / logup-GKR seams through the C ABI (dp_sumcheck_verify, dp_logup_verify: host only,
no device) — ports of the reference's own tests with the ORACLE as the prover and the PRODUCT as the verifier, so the two
independent implementations must agree on transcript, message layout and every check:
  sumcheck/src/test.rs:23-56   random VirtualPolynomial, nv = 1 and 12: verify, then sub-claim == the polynomial at the point
  zkml/src/lookup/logup_gkr/mod.rs:26-94   two random columns, n = 5..: verify, fractional sums, column claims."""
import os

import numpy as np
import pytest

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ops():
    from deep_prove_amd.sharded import e_add, e_mul
    return e_add, e_mul


def _rand_ext(rng):
    return (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))


def _random_vp(rng, nv, degree_range, num_products):
    """VirtualPolynomial::random (virtual_poly.rs:230-262): products of fresh random MLEs with random coefficients, and the sum
    over the hypercube"""
    e_add, e_mul = _ops()
    tables, terms, total = [], [], (0, 0)
    for _ in range(num_products):
        k = int(rng.integers(degree_range[0], degree_range[1]))
        idx = []
        for _ in range(k):
            idx.append(len(tables))
            tables.append(rng.integers(0, P, size=1 << nv, dtype=np.uint64))
        coeff = _rand_ext(rng)
        s = 0
        for b in range(1 << nv):
            prod = 1
            for i in idx:
                prod = prod * int(tables[i][b]) % P
            s = (s + prod) % P
        total = e_add(total, e_mul(coeff, (s, 0)))
        terms.append((coeff, idx))
    return tables, terms, total


@pytest.mark.parametrize("nv,degree_range,num_products", [(1, (2, 4), 3), (12, (2, 4), 3), (5, (1, 2), 2), (7, (3, 4), 1)])
def test_sumcheck_verifier_port_of_reference_test(oracle, nv, degree_range, num_products):
    import deep_prove_amd as dpa
    e_add, e_mul = _ops()
    rng = np.random.default_rng(1000 + nv)
    tables, terms, asserted_sum = _random_vp(rng, nv, degree_range, num_products)
    max_degree = max(len(ix) for _, ix in terms)
    proof, finals = oracle.sumcheck_prove(nv, tables, [False] * len(tables), terms, oracle.transcript(b"test"))
    t = dpa.Transcript(b"test")
    point, expected = dpa.verify_sumcheck(asserted_sum, proof, nv, max_degree, t)
    # the sub-claim is the virtual polynomial at the point (test.rs:38-49) ...
    value = (0, 0)
    for coeff, idx in terms:
        prod = (1, 0)
        for i in idx:
            prod = e_mul(prod, oracle.mle_eval(tables[i], False, point))
        value = e_add(value, e_mul(coeff, prod))
    assert value == expected, "wrong subclaim"
    # ... the proof's point is the verifier's point (test.rs:50-55), the final evaluations are the tables at that point
    assert int(proof[0]) == nv and [tuple(int(x) for x in proof[1 + 2 * i:3 + 2 * i]) for i in range(nv)] == point
    for i, tab in enumerate(tables):
        assert oracle.mle_eval(tab, False, point) == (int(finals[2 * i]), int(finals[2 * i + 1]))
    # both transcripts are in the same state afterwards
    ot = oracle.transcript(b"test")
    oracle.sumcheck_prove(nv, tables, [False] * len(tables), terms, ot)
    assert ot.read_challenge() == t.read_challenge()
    # a wrong sum, a tampered round message and a truncated stream are rejected
    with pytest.raises(dpa.DeepProveError):
        dpa.verify_sumcheck(e_add(asserted_sum, (1, 0)), proof, nv, max_degree, dpa.Transcript(b"test"))
    bad = proof.copy()
    bad[1 + 2 * nv + 2] ^= np.uint64(1)  # evaluation at 0 of the first round (the very last evaluation of a proof only moves the sub-claim)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify_sumcheck(asserted_sum, bad, nv, max_degree, dpa.Transcript(b"test"))
    with pytest.raises(dpa.DeepProveError):
        dpa.verify_sumcheck(asserted_sum, proof[:-3], nv, max_degree, dpa.Transcript(b"test"))


@pytest.mark.parametrize("n", [5, 6, 8, 11])
def test_logup_verifier_port_of_reference_test(oracle, n):
    import deep_prove_amd as dpa
    e_add, e_mul = _ops()
    rng = np.random.default_rng(2000 + n)
    cols = [rng.integers(0, P, size=1 << n, dtype=np.uint64) for _ in range(2)]
    cc, csc = _rand_ext(rng), _rand_ext(rng)
    proof = oracle.logup_prove(cols, 1, cc, csc, oracle.transcript())  # LogUpInput::new_lookup(columns, .., 1): two instances
    t = dpa.Transcript()
    nums, dens, claims = dpa.verify_logup(proof, 2, cc, csc, t)
    for col, num, den in zip(cols, nums, dens):
        # sum of the fractions -1 / (c + v) (mod.rs:62-83); Fraction addition keeps (n1 d2 + n2 d1, d1 d2) unnormalised
        N, D = (0, 0), (1, 0)
        for v in col:
            d = e_add(cc, (int(v), 0))
            N, D = e_add(e_mul(N, d), ((P - D[0]) % P, (P - D[1]) % P)), e_mul(D, d)
        assert (N, D) == (num, den)
    assert len(claims) == 2
    for (point, ev), col in zip(claims, cols):
        assert len(point) == n and oracle.mle_eval(col, False, point) == ev
    ot = oracle.transcript()
    oracle.logup_prove(cols, 1, cc, csc, ot)
    assert ot.read_challenge() == t.read_challenge()
    with pytest.raises(dpa.DeepProveError):  # other challenges than the prover's
        dpa.verify_logup(proof, 2, csc, cc, dpa.Transcript())
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify_logup(bad, 2, cc, csc, dpa.Transcript())


def test_logup_table_proof_is_accepted(oracle):
    """LogUpInput::Table: one instance, the multiplicity column as numerators (the table side of a lookup argument)"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(77)
    n = 8
    table = np.arange(1 << n, dtype=np.uint64)
    mult = rng.integers(0, 50, size=1 << n, dtype=np.uint64)
    cc, csc = _rand_ext(rng), _rand_ext(rng)
    proof = oracle.logup_prove([table], 1, cc, csc, oracle.transcript(), multiplicities=mult)
    nums, dens, claims = dpa.verify_logup(proof, 1, cc, csc, dpa.Transcript())
    e_add, e_mul = _ops()
    N, D = (0, 0), (1, 0)
    for v, m in zip(table, mult):
        d = e_add(cc, (int(v), 0))
        N, D = e_add(e_mul(N, d), e_mul(D, (int(m), 0))), e_mul(D, d)
    assert (N, D) == (nums[0], dens[0])
    assert len(claims) == 2  # multiplicities, then the table column


@pytest.mark.parametrize("shape", [[(8, False)], [(9, True)], [(10, False), (9, True), (10, True)],
                                   [(8, False), (8, True), (12, False), (11, True), (12, False)]])
def test_basefold_batch_verify_port_of_reference_round_trips(oracle, shape):
    """mpcs commit -> batch_open -> batch_verify round trips (mpcs/src/lib.rs:467-727, basefold.rs:1239-1331: base and
    extension polynomials, single / batch / multi-size batch): the oracle commits and opens, the product's host verifier
    (dp_pcs_batch_verify) checks — and rejects a tampered proof, a wrong evaluation, a wrong root and a wrong point"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(31 * len(shape) + shape[0][0])
    maxsize = 1 << 12
    raws = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for nv, e in shape]
    points = [[_rand_ext(rng) for _ in range(nv)] for nv, _ in shape]
    evals = [oracle.mle_eval(w, e, p) for w, (nv, e), p in zip(raws, shape, points)]
    roots = [oracle.pcs_commit_root(maxsize, w, e) for w, (_, e) in zip(raws, shape)]
    ot = oracle.transcript(b"test")
    proof = oracle.pcs_batch_open(maxsize, raws, [e for _, e in shape], points, evals, ot)
    nvs, is_base = [nv for nv, _ in shape], [not e for _, e in shape]
    t = dpa.Transcript(b"test")
    dpa.Basefold.batch_verify(maxsize, roots, nvs, is_base, points, evals, proof, t)
    assert t.read_challenge() == ot.read_challenge()
    bad = proof.copy()
    bad[7] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize, roots, nvs, is_base, points, evals, bad, dpa.Transcript(b"test"))
    wrong = list(evals)
    wrong[-1] = ((evals[-1][0] + 1) % P, evals[-1][1])
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize, roots, nvs, is_base, points, wrong, proof, dpa.Transcript(b"test"))
    roots2 = [list(r) for r in roots]
    roots2[0][2] = (roots2[0][2] + 1) % P
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize, roots2, nvs, is_base, points, evals, proof, dpa.Transcript(b"test"))
    points2 = [list(p) for p in points]
    points2[0][0] = ((points2[0][0][0] + 1) % P, points2[0][0][1])
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize, roots, nvs, is_base, points2, evals, proof, dpa.Transcript(b"test"))
    # the commitment depends on the parameter size (coset shift, rs.rs:494-499): a verifier with other parameters rejects
    with pytest.raises(dpa.DeepProveError):
        dpa.Basefold.batch_verify(maxsize * 4, roots, nvs, is_base, points, evals, proof, dpa.Transcript(b"test"))


@pytest.mark.parametrize("nv,ext", [(6, False), (7, True), (8, False), (10, True), (12, False)])
def test_basefold_single_verify_port_of_reference_round_trip(oracle, nv, ext):
    """mpcs commit -> open -> verify of one polynomial (mpcs/src/lib.rs:467-540 run_commit_open_verify, base and extension; <= 7
    variables: the trivial proof): the oracle commits and opens, dp_pcs_verify checks — same transcript state afterwards — and
    rejects a tampered proof, a wrong evaluation, a wrong root, a wrong point and other parameters"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(4100 + nv)
    maxsize = 1 << 13
    w = rng.integers(0, P, size=(2 if ext else 1) << nv, dtype=np.uint64)
    point = [_rand_ext(rng) for _ in range(nv)]
    ev = oracle.mle_eval(w, ext, point)
    root = oracle.pcs_commit_root(maxsize, w, ext)
    ot = oracle.transcript(b"test")
    proof = oracle.pcs_open(maxsize, w, ext, point, ot)
    t = dpa.Transcript(b"test")
    dpa.Basefold.verify(maxsize, root, nv, not ext, point, ev, proof, t)
    assert t.read_challenge() == ot.read_challenge()
    def rejected(**kw):
        a = dict(max_poly_size=maxsize, root=root, num_vars=nv, is_base=not ext, point=point, eval_=ev, proof_words=proof, transcript=dpa.Transcript(b"test"))
        a.update(kw)
        with pytest.raises(dpa.DeepProveError):
            dpa.Basefold.verify(**a)
    rejected(eval_=((ev[0] + 1) % P, ev[1]))
    rejected(root=[root[0], root[1] ^ 1, root[2], root[3]])
    rejected(point=[((point[0][0] + 1) % P, point[0][1])] + point[1:])
    bad = proof.copy(); bad[-5] ^= np.uint64(1)  # trivial: an evaluation; otherwise a sibling digest of the last query's path
    rejected(proof_words=bad)
    rejected(proof_words=proof[:-2])
    if nv > 7:
        rejected(max_poly_size=maxsize * 2)  # the coset shift of the code depends on the parameters (rs.rs:494-499)
        bad = proof.copy(); bad[4] ^= np.uint64(1)  # the first commit-phase message
        rejected(proof_words=bad)
        rejected(is_base=ext)


EVAL_SHAPES = [  # (polynomials as (num_vars, ext), point lengths, evaluations as (poly, point))
    ([(9, False), (9, False)], [9], [(0, 0), (1, 0)]),                                        # the reference's run_batch_commit_open_verify: one point, two polynomials
    ([(10, False), (10, True), (9, False), (9, False)], [10, 9], [(0, 0), (1, 0), (2, 1), (3, 1)]),  # ..._multiple_sizes
    ([(10, True)], [10, 10], [(0, 0), (0, 1)]),                                               # one polynomial at two points
    ([(11, False), (9, True), (11, True)], [11, 11, 9], [(0, 0), (2, 0), (0, 1), (1, 2), (2, 1)]),
]


@pytest.mark.parametrize("shape", range(len(EVAL_SHAPES)))
def test_basefold_batch_verify_general_evaluation_lists(oracle, shape):
    """mpcs batch_open / batch_verify with `evals: &[Evaluation]` (mpcs/src/lib.rs:508-700 run_batch_commit_open_verify{,_multiple_sizes}: polynomials
    that share a point; plus a polynomial opened at two points and a mixed list): the oracle opens, dp_pcs_batch_verify_evals accepts with the
    prover's transcript state and rejects a wrong value, swapped indices, a missing evaluation and a flipped proof word"""
    import deep_prove_amd as dpa
    polys_s, points_s, evals_s = EVAL_SHAPES[shape]
    rng = np.random.default_rng(6100 + shape)
    maxsize = 1 << 12
    polys = [rng.integers(0, P, size=(2 if e else 1) << nv, dtype=np.uint64) for nv, e in polys_s]
    points = [[_rand_ext(rng) for _ in range(n)] for n in points_s]
    evals = [(pi, qi, oracle.mle_eval(polys[pi], polys_s[pi][1], points[qi])) for pi, qi in evals_s]
    roots = [oracle.pcs_commit_root(maxsize, w, e) for w, (_, e) in zip(polys, polys_s)]
    ot = oracle.transcript(b"test")
    proof = oracle.pcs_batch_open_evals(maxsize, polys, [e for _, e in polys_s], points, evals, ot)
    args = dict(max_poly_size=maxsize, roots=roots, num_vars=[nv for nv, _ in polys_s], is_base=[not e for _, e in polys_s], points=points, evals=evals, proof_words=proof)
    t = dpa.Transcript(b"test")
    dpa.Basefold.batch_verify_evals(transcript=t, **args)
    assert t.read_challenge() == ot.read_challenge()
    def rejected(**kw):
        a = dict(args); a.update(kw)
        with pytest.raises(dpa.DeepProveError):
            dpa.Basefold.batch_verify_evals(transcript=dpa.Transcript(b"test"), **a)
    v = evals[-1][2]
    rejected(evals=evals[:-1] + [(evals[-1][0], evals[-1][1], ((v[0] + 1) % P, v[1]))])
    rejected(evals=evals[:-1])
    rejected(evals=[evals[1], evals[0]] + evals[2:])
    rejected(roots=[[roots[0][0] ^ 1] + roots[0][1:]] + roots[1:])
    for at in (7, len(proof) // 2, len(proof) - 5):
        bad = proof.copy(); bad[at] ^= np.uint64(1)
        rejected(proof_words=bad)


@pytest.mark.parametrize("nv,ext,k", [(4, False, 4), (6, True, 4), (9, False, 1), (9, False, 4), (9, True, 4), (10, False, 7), (8, True, 3)])
def test_basefold_simple_batch_verify_port_of_reference_round_trip(oracle, nv, ext, k):
    """mpcs batch_commit -> simple_batch_open -> simple_batch_verify (mpcs/src/basefold.rs:1254-1297 simple_batch_commit_open_verify_goldilocks:
    base and extension, batch sizes 1 and 4, the trivial size; plus a batch whose rows need the sponge): the oracle commits and opens,
    dp_pcs_simple_batch_verify checks — same transcript state afterwards — and rejects tampered proofs, evaluations, roots, points"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(5200 + 16 * nv + k)
    maxsize = 1 << 12
    polys = [rng.integers(0, P, size=(2 if ext else 1) << nv, dtype=np.uint64) for _ in range(k)]
    point = [_rand_ext(rng) for _ in range(nv)]
    evals = [oracle.mle_eval(w, ext, point) for w in polys]
    ot = oracle.transcript(b"test")
    root, proof = oracle.pcs_simple_batch_open(maxsize, polys, ext, point, ot)
    if k == 1:
        assert root == oracle.pcs_commit_root(maxsize, polys[0], ext)  # one polynomial: the ordinary tree (merkle_tree.rs:273-285)
    t = dpa.Transcript(b"test")
    dpa.Basefold.simple_batch_verify(maxsize, root, nv, not ext, point, evals, proof, t)
    assert t.read_challenge() == ot.read_challenge()
    def rejected(**kw):
        a = dict(max_poly_size=maxsize, root=root, num_vars=nv, is_base=not ext, point=point, evals=evals, proof_words=proof, transcript=dpa.Transcript(b"test"))
        a.update(kw)
        with pytest.raises(dpa.DeepProveError):
            dpa.Basefold.simple_batch_verify(**a)
    rejected(evals=evals[:-1] + [((evals[-1][0] + 1) % P, evals[-1][1])])
    if k > 1:
        rejected(evals=evals[1:] + evals[:1])
        rejected(evals=evals[:-1])  # a polynomial less than committed
    rejected(root=[root[0], root[1] ^ 1, root[2], root[3]])
    rejected(point=[((point[0][0] + 1) % P, point[0][1])] + point[1:])
    rejected(is_base=ext)
    rejected(proof_words=proof[:-2])
    for at in (len(proof) - 5, len(proof) // 2, 4):
        bad = proof.copy(); bad[at] ^= np.uint64(1)
        rejected(proof_words=bad)
    if nv > 7:
        rejected(max_poly_size=maxsize * 2)


def test_simple_batch_verifier_rejects_mutated_streams_without_crashing(oracle):
    """1 200 mutated simple-batch openings (overwritten / incremented words, truncations, cut-out runs, single-bit words in the header): every one
    is rejected with a status, none crashes the library"""
    import deep_prove_amd as dpa
    rng = np.random.default_rng(77)
    for nv, ext, k in ((9, False, 3), (5, True, 3), (8, True, 5)):
        polys = [rng.integers(0, P, size=(2 if ext else 1) << nv, dtype=np.uint64) for _ in range(k)]
        pt = [_rand_ext(rng) for _ in range(nv)]
        evals = [oracle.mle_eval(w, ext, pt) for w in polys]
        root, proof = oracle.pcs_simple_batch_open(1 << 11, polys, ext, pt, oracle.transcript(b"t"))
        dpa.Basefold.simple_batch_verify(1 << 11, root, nv, not ext, pt, evals, proof, dpa.Transcript(b"t"))
        for it in range(400):
            p, m = proof.copy(), it % 5
            if m == 0:
                p[rng.integers(0, p.size)] = np.uint64(rng.integers(0, 1 << 63))
            elif m == 1:
                p = p[:rng.integers(0, p.size)]
            elif m == 2:
                i = rng.integers(0, p.size); p[i] = np.uint64((int(p[i]) + 1) & ((1 << 64) - 1))
            elif m == 3:
                i = rng.integers(0, p.size - 8); p = np.concatenate([p[:i], p[i + rng.integers(1, 8):]])
            else:
                p[rng.integers(0, min(64, p.size))] = np.uint64(1 << int(rng.integers(0, 64)))
            if p.size == proof.size and (p == proof).all():
                continue
            with pytest.raises(dpa.DeepProveError):
                dpa.Basefold.simple_batch_verify(1 << 11, root, nv, not ext, pt, evals, p, dpa.Transcript(b"t"))


def test_batch_verifier_host_only_accepts_and_rejects_per_proof():
    """dp_verify_batch without a device (ctx NULL): protocol checks on host threads with the Merkle paths deferred and then
    authenticated on the same threads — a verdict per proof: the golden proofs are accepted; a flipped word inside a layer
    proof, inside a Merkle path of the batch opening, a wrong output and a truncated stream are each rejected on their own"""
    import deep_prove_amd as dpa
    for name, inner in (("mlp_w8.npz", 300), ("cnn_tiny.npz", 900)):
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        p = g["proof"]
        bad_inner = p.copy(); bad_inner[inner] ^= np.uint64(1)
        bad_path = p.copy(); bad_path[-40] ^= np.uint64(1)
        proofs = [p, bad_inner, p, bad_path, p[:1000], p]
        xs = np.stack([g["input"]] * len(proofs))
        ys = np.stack([g["output"]] * len(proofs))
        ys[5, 0] += 1
        res, ms = dpa.verify_batch(g["verifier_blob"], proofs, xs, ys, threads=3)
        got = [int(v) for v in res]
        # (a flipped word is a rejection when it stays a canonical field element / consistent length, a malformed stream otherwise)
        assert got[0] == 0 and got[2] == 0 and got[1] in (-5, -1) and got[3] in (-5, -1) and got[4] == -1 and got[5] == -5, (name, res)
        if name == "cnn_tiny.npz":
            assert got[3] == -5  # this one sits in a sibling digest of the batch opening's last Merkle path
        assert ms > 0


def test_mutated_proof_streams_never_verify_and_never_crash():
    """dp_verify on 1 500 mutated golden proofs (MLP, CNN, MatMul models): flipped bits, out-of-range words, truncations, cut and duplicated
    ranges. Every word of a stream is bound: the only accepted streams are those a mutation left identical to the proof (booleans and
    narrowed values have one encoding; logup output claims sit at the verifier's point; sumcheck points equal the challenges; trivially
    opened commitments describe their table; opened pairs carry their LEFT index — tools/flip_sweep.py is the exhaustive counterpart,
    profiles/r02_flip_sweep.txt its result)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "support", "fuzz_proof.py"), "5", "1500"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "accepted-but-different 0" in r.stdout, r.stdout[-800:] + r.stderr[-1500:]


def test_single_bit_flips_of_every_word_in_two_windows_are_rejected():
    """the exhaustive sweep of tools/flip_sweep.py on two windows of the golden MLP proof: words 0..700 (layer proofs: every kind of field a
    step holds, among them the `num_vars` / `is_base` words of trivially opened witness commitments, which the reference never reads) and
    words 4100..4700 (the first queries of the batch opening: pair values, pair indices, whose lowest bit the reference ignores, path
    digests): no flipped stream verifies"""
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_w8.npz"))
    p0 = g["proof"]
    dpa.verify(g["verifier_blob"], p0, g["input"], g["output"])
    accepted = []
    for lo, hi in ((0, 700), (4100, 4700)):
        for i in range(lo, hi):
            p = p0.copy(); p[i] ^= np.uint64(1)
            try:
                dpa.verify(g["verifier_blob"], p, g["input"], g["output"]); accepted.append(i)
            except dpa.DeepProveError:
                pass
    assert accepted == []
