"""TEST INFRASTRUCTURE — layer 3 a second time: the VERIFIER of Basefold's batch opening (mpcs/src/basefold.rs:964-1098 `batch_verify`,
basefold/query_phase.rs:213-289 + 1116-1230 the query checks, sum_check/classic.rs:287-314 + classic/coeff.rs:41-62 the coefficient-form
sumcheck, basefold/encoding/rs.rs:412-456 the folding coefficients of the Reed-Solomon code, util/merkle_tree.rs:422-447 path
authentication) and of the trivial openings (basefold.rs:873-894), in Python on the independent field / Poseidon2 / transcript of
l0_independent.py. The Reed-Solomon objects are used by their DEFINITION, not through the reference's FFT tables: a codeword of a
2^k-coefficient message is the polynomial's values on the coset gamma^(2^(full-k)) * H, |H| = 2^(k+1), listed in bit-reversed order;
positions 2i, 2i+1 of such a list hold the values at x and -x; folding by r maps them to the value at x^2 of
f_even + r f_odd, i.e. the line through (x, v) and (-x, v') evaluated at r."""
from . import l0_independent as L
from . import l1_independent as L1
from .l2_independent import e, fe, add, sub, mul, eq_xy_eval, ONE, ZERO, challenge

P = L.P
BASECODE_LOG = 7   # messages of 2^7 coefficients are sent in the clear (trivial_num_vars, basefold.rs:1204-1206)
RATE_LOG = 1
QUERIES = 200      # RSCodeDefaultSpec::get_number_queries (rs.rs:204-206)
GAMMA = 7          # the multiplicative generator of the Goldilocks field
W32 = pow(GAMMA, (P - 1) >> 32, P)  # generator of the 2^32 subgroup: two_adic_generator(32)


def two_adic(bits):
    return pow(W32, 1 << (32 - bits), P)


def bitrev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


def smul(x, s):
    """extension element times a base-field scalar"""
    return (x[0] * s % P, x[1] * s % P)


def build_eq(point):
    return L.eq_table(point)


def horner(coeffs, x):
    acc = ZERO
    for c in reversed(coeffs):
        acc = add(mul(acc, x), c)
    return acc


def fold_point(full_log, level, pair_index):
    """the field point x whose values sit at positions 2 * pair_index (x) and 2 * pair_index + 1 (-x) of a bit-reversed codeword of
    2^(level+1) entries (rs.rs:412-433): coset shift gamma^(2^(full + rate - level - 1)), root of order 2^(level+1)"""
    shift = pow(GAMMA, 1 << (full_log + RATE_LOG - level - 1), P)
    return shift * pow(two_adic(level + 1), bitrev(pair_index, level), P) % P


def merkle_check(pair_words, index, path, root):
    """util/merkle_tree.rs:422-447: the lowest bit of the index is ignored (either leaf of the pair)"""
    assert [int(v) for v in L1.merkle_root_from_path(pair_words, index >> 1, path)] == [int(v) for v in root], "merkle path does not authenticate"


def cq_words(q):
    return [q["pair"][0][0], q["pair"][0][1], q["pair"][1][0], q["pair"][1][1]] if q["is_ext"] else [q["pair"][0], q["pair"][1]]


def cq_ext(q):
    return (e(q["pair"][0]), e(q["pair"][1])) if q["is_ext"] else (fe(q["pair"][0]), fe(q["pair"][1]))


def trivial_verify(comm, point, ev, proof):
    """Basefold::verify on a trivial proof: the evaluations themselves; their Merkle root must be the commitment"""
    assert len(proof["trivial_proof"]) == 1
    t = proof["trivial_proof"][0]
    vals = [e((t["w"][2 * i], t["w"][2 * i + 1])) for i in range(len(t["w"]) // 2)] if t["is_ext"] else [fe(v) for v in t["w"]]
    assert len(vals) == 1 << len(point)
    words = [[int(w) for w in (t["w"][4 * i:4 * i + 4] if t["is_ext"] else t["w"][2 * i:2 * i + 2])] for i in range(len(vals) // 2)]
    layer = [L.hash_or_noop(w) for w in words]
    while len(layer) > 1:
        layer = [L.compress(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    assert [int(v) for v in layer[0]] == [int(v) for v in comm["root"]], "trivial opening: Merkle root mismatch"
    assert L.mle_eval(vals, point) == ev, "trivial opening: wrong evaluation"


def batch_verify(full_log, comms, points, evals, proof, tr, check_every=1):
    """comms[i] = {'root', 'num_vars'}; claim i: comms[i] at points[i] is evals[i] (Evaluation::new(i, i, eval), commit/context.rs:541-553).
    check_every > 1: the Merkle paths and fold checks of every check_every-th query only (all indices are still compared with the
    transcript's) — for tests that run this verifier many times"""
    assert comms and len(comms) == len(points) == len(evals)
    num_vars = max(len(p) for p in points)
    num_rounds = num_vars - BASECODE_LOG
    for c, p in zip(comms, points):
        assert len(p) == c["num_vars"] >= BASECODE_LOG
    bsl = (len(evals) - 1).bit_length()
    t = [challenge(tr, b"batch coeffs") for _ in range(bsl)]
    eq_xt = build_eq(t)
    target = ZERO
    for i, ev in enumerate(evals):
        target = add(target, mul(smul(ev, (1 << (num_vars - len(points[i]))) % P), eq_xt[i]))
    # the coefficient-form sumcheck (sum_check/classic.rs:287-314)
    rounds = [[e(x) for x in m] for m in proof["sumcheck_proof"]]
    assert len(rounds) == num_vars
    verify_point, s = [], target
    for m in rounds:
        assert len(m) == 3
        tr.append_field_elements([w for x in m for w in x])
        verify_point.append(challenge(tr, b"sumcheck round"))
    for m, c in zip(rounds, verify_point):
        assert s == add(m[0], add(add(m[0], m[1]), m[2])), "batch opening: sumcheck round inconsistent"
        s = horner(m, c)
    new_target = s
    coeffs = [mul(eq_xy_eval(verify_point[:len(p)], p), eq_xt[i]) for i, p in enumerate(points)]
    # the commit phase as the transcript saw it
    msgs = [[e(x) for x in m] for m in proof["sumcheck_messages"]]
    roots = proof["roots"]
    assert len(msgs) == num_rounds and len(roots) == num_rounds - 1
    fold = []
    for i in range(num_rounds):
        tr.append_field_elements([w for x in msgs[i] for w in x])
        fold.append(challenge(tr, b"commit round"))
        if i < num_rounds - 1:
            tr.append_field_elements([int(w) for w in roots[i]])
    final_message = [e(x) for x in proof["final_message"]]
    assert len(final_message) == 1 << BASECODE_LOG
    tr.append_field_elements([w for x in final_message for w in x])
    indices = [challenge(tr, b"query indices")[0] % (1 << (num_vars + RATE_LOG)) for _ in range(QUERIES)]
    # final codeword by definition: coefficients = Moebius transform of the bit-reversed message, evaluated on the coset of the last level
    k = BASECODE_LOG
    msg = [final_message[bitrev(i, k)] for i in range(1 << k)]
    for i in range(k):  # evaluations on the hypercube -> monomial coefficients (hypercube.rs:16-37)
        half = 1 << i
        for base in range(0, 1 << k, 2 * half):
            for j in range(base + half, base + 2 * half):
                msg[j] = sub(msg[j], msg[j - half])
    shift_small = pow(GAMMA, 1 << (full_log - k), P)
    w_small = two_adic(k + RATE_LOG)

    def final_codeword_at(pos):  # position in the bit-reversed codeword of 2^(k+1) entries
        x = shift_small * pow(w_small, bitrev(pos, k + RATE_LOG), P) % P
        return horner(msg, (x, 0))

    assert len(proof["queries"]) == QUERIES
    for qi, (q, index) in enumerate(zip(proof["queries"], indices)):
        assert q["index"] == index, "query index is not the transcript's"
        if qi % check_every:
            continue
        oq, cqs = q["oracle_query"], q["commitments_query"]
        assert len(oq) == num_rounds - 1 and len(cqs) == len(comms)
        for o, root in zip(oq, roots):
            merkle_check(cq_words(o), o["index"], o["path"], root)
        for c, comm in zip(cqs, comms):
            merkle_check(cq_words(c), c["index"], c["path"], comm["root"])
        left, right = ZERO, ZERO
        right_index = index | 1
        left_index = right_index - 1
        for i in range(num_rounds):
            for ci, comm in enumerate(comms):
                if comm["num_vars"] == num_vars - i:
                    assert cqs[ci]["index"] >> 1 == left_index >> 1
                    l, r = cq_ext(cqs[ci])
                    left, right = add(left, mul(l, coeffs[ci])), add(right, mul(r, coeffs[ci]))
            x0 = fold_point(full_log, num_vars + RATE_LOG - i - 1, left_index >> 1)
            # the line through (x0, left), (-x0, right) at the folding challenge
            w = L.inv((P - 2 * x0) % P)
            res = add(left, mul(mul(sub(fold[i], (x0, 0)), sub(right, left)), (w, 0)))
            next_index = right_index >> 1
            if i < num_rounds - 1:
                right_index = next_index | 1
                left_index = right_index - 1
                left, right = cq_ext(oq[i])
                assert oq[i]["index"] >> 1 == left_index >> 1, "oracle query index"
                expect = left if next_index & 1 == 0 else right
            else:
                for ci, comm in enumerate(comms):
                    if comm["num_vars"] == num_vars - i - 1:
                        assert cqs[ci]["index"] >> 1 == next_index >> 1
                        l, r = cq_ext(cqs[ci])
                        res = add(res, mul(l if next_index & 1 == 0 else r, coeffs[ci]))
                expect = final_codeword_at(next_index)
            assert res == expect, f"query {index}: folding check failed at round {i}"
    # the sumcheck part of the commit phase (query_phase.rs:266-287)
    d2 = lambda m, x: add(m[0], add(mul(x, m[1]), mul(mul(x, x), m[2])))
    z1 = lambda m: add(add(m[0], m[0]), add(m[1], m[2]))
    assert new_target == z1(msgs[0]), "commit phase: first message does not continue the batch sumcheck"
    for i in range(num_rounds - 1):
        assert d2(msgs[i], fold[i]) == z1(msgs[i + 1])
    rev = list(reversed(fold))
    coeff = eq_xy_eval(verify_point[num_vars - num_rounds:], rev)
    eq = [mul(x, coeff) for x in build_eq(verify_point[:num_vars - num_rounds])]
    inner = ZERO
    for a, b in zip(final_message, eq):
        inner = add(inner, mul(a, b))
    assert d2(msgs[-1], fold[-1]) == inner, "commit phase: final message does not match the last sumcheck message"
