"""The identity behind the two-round grid of large base-table sumchecks (deep-prove_amd/csrc/kernels.inc k_sc_terms2 / k_sc_fused2, hip_dev.hip grid2_coeffs),
restated in plain Python integers and checked against the round-by-round definition of prove_parallel (sumcheck/src/prover.rs:498-585: pairs (2 i, 2 i + 1), message s(t) for
t = 0..K, fold with the challenge): sixteen sums over every quad's bilinear extension, taken at the points {0, 1, oo, -1} of each of the first two variables ("oo" = the leading
coefficient), determine the messages of rounds 1 and 2, and one pass folds both variables. Exact field arithmetic, so the device path's proof bytes are the reference's — the GPU
tests (tests/test_gpu_primitives.py SC_CASES, tests/test_gpu_sharded.py config-5 goldens) check that against the oracle; this file checks the algebra without a GPU."""
import random

import pytest

P = 2**64 - 2**32 + 1
HALF = pow(2, P - 2, P)


def round_message(tabs, K):
    m = len(tabs[0]) // 2
    out = []
    for t in range(K + 1):
        s = 0
        for i in range(m):
            pr = 1
            for tb in tabs:
                pr = pr * ((tb[2 * i] + t * (tb[2 * i + 1] - tb[2 * i])) % P) % P
            s = (s + pr) % P
        out.append(s)
    return out


def fold(tabs, r):
    return [[(tb[2 * i] + r * (tb[2 * i + 1] - tb[2 * i])) % P for i in range(len(tb) // 2)] for tb in tabs]


def grid_sums(tabs, K):
    """what k_sc_terms2 accumulates: A[p1][p2], points (0, 1, oo, -1) of x1 and x2"""
    A = [[0] * 4 for _ in range(4)]
    for q in range(len(tabs[0]) // 4):
        pts = []
        for tb in tabs:
            a0, a1, a2, a3 = tb[4 * q:4 * q + 4]
            d0, d1 = (a1 - a0) % P, (a3 - a2) % P
            pts.append([(a0, a2), (a1, a3), (d0, d1), ((a0 - d0) % P, (a2 - d1) % P)])
        for p1 in range(4):
            pr = [1, 1, 1, 1]
            for j in range(K):
                c0, c1 = pts[j][p1]
                e = (c1 - c0) % P
                for k, v in enumerate((c0, c1, e, (c0 - e) % P)):
                    pr[k] = pr[k] * v % P
            for p2 in range(4):
                A[p1][p2] = (A[p1][p2] + pr[p2]) % P
    return A


def coeffs(K, v):
    """grid2_coeffs: the polynomial of degree K from its values at 0 and 1, its leading coefficient and (K = 3) its value at -1"""
    c = [v[0], 0, 0, 0]
    if K == 1:
        c[1] = (v[1] - v[0]) % P
    elif K == 2:
        c[2] = v[2]
        c[1] = (v[1] - v[0] - v[2]) % P
    else:
        s, d = (v[1] + v[3]) * HALF % P, (v[1] - v[3]) * HALF % P
        c[3], c[2], c[1] = v[2], (s - v[0]) % P, (d - v[2]) % P
    return c


def ev(c, x):
    return (((c[3] * x + c[2]) * x + c[1]) * x + c[0]) % P


@pytest.mark.parametrize("K", [1, 2, 3])
@pytest.mark.parametrize("n", [4, 64])
def test_two_rounds_from_one_grid_and_both_folds_in_one_pass(K, n):
    rng = random.Random(100 * K + n)
    tabs = [[rng.randrange(P) for _ in range(n)] for _ in range(K)]
    if n == 64:
        tabs[0][5] = P - 1  # (edge values ride along)
        tabs[-1][6] = 0
    s1 = round_message(tabs, K)
    r1 = rng.randrange(P)
    t1 = fold(tabs, r1)
    s2 = round_message(t1, K)
    r2 = rng.randrange(P)
    t2 = fold(t1, r2)
    A = grid_sums(tabs, K)
    c0, c1 = coeffs(K, [A[i][0] for i in range(4)]), coeffs(K, [A[i][1] for i in range(4)])
    assert [(ev(c0, t) + ev(c1, t)) % P for t in range(K + 1)] == s1  # round 1: s1(t) = Q(t, 0) + Q(t, 1)
    g = coeffs(K, [ev(coeffs(K, [A[i][p2] for i in range(4)]), r1) for p2 in range(4)])
    assert [ev(g, t) for t in range(K + 1)] == s2                      # round 2: s2(t) = Q(r1, t)
    both = []
    for tb in tabs:                                                    # k_sc_fused2: f = e0 + r2 (e1 - e0), e = a + r1 (b - a)
        o = []
        for q in range(n // 4):
            a0, a1, a2, a3 = tb[4 * q:4 * q + 4]
            e0, e1 = (a0 + r1 * (a1 - a0)) % P, (a2 + r1 * (a3 - a2)) % P
            o.append((e0 + r2 * (e1 - e0)) % P)
        both.append(o)
    assert both == t2
