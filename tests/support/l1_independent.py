"""TEST INFRASTRUCTURE — layer 1 a second time: the sumcheck VERIFIER and the Merkle-path check, written from the protocol
definitions (Lund-Fortnow-Karloff-Nisan / Thaler's sumcheck: p_j(0) + p_j(1) = claim_j, claim_{j+1} = p_j(r_j) by Lagrange
interpolation over 0..d, final claim = the polynomial at the point; a Merkle path: fold the leaf digest with its siblings, the index
bit choosing the side) on top of tests/support/l0_independent.py (Python big integers, its own Poseidon2 / sponge / transcript).
Nothing here is derived from oracle/*.hpp or csrc/*: a misreading of the round structure, the challenge schedule or the tree layout in
the oracle would have to be repeated in this unrelated formulation to go unnoticed. What IS taken from the reference's conventions
(because they are conventions, not mathematics): the transcript schedule of IOPProverState::prove_parallel (num_vars and max_degree as
8-byte little-endian messages, the round message as 2 (d + 1) base elements, the label "Internal round": sumcheck/src/prover.rs:498-585),
and the leaf pairing of mpcs' MerkleTree (a leaf pair hashes to one digest: util/merkle_tree.rs:261-329, hash_or_noop of the pair's words)."""
from . import l0_independent as L

P = L.P


def ext(v):
    return (int(v[0]) % P, int(v[1]) % P)


def lagrange_at(ys, x):
    """the polynomial of degree < len(ys) through (i, ys[i]) evaluated at the extension point x"""
    n = len(ys)
    acc = (0, 0)
    for i in range(n):
        num, den = (1, 0), 1
        for j in range(n):
            if j == i:
                continue
            num = L.ext_mul(num, L.ext_sub(x, (j, 0)))
            den = den * ((i - j) % P) % P
        acc = L.ext_add(acc, L.ext_mul(L.ext_mul(ys[i], num), (L.inv(den), 0)))
    return acc


def verify_sumcheck(claimed_sum, point, rounds, num_vars, max_degree, transcript):
    """returns (challenges, final claim) or raises AssertionError; `rounds[j]` = evaluations of the round polynomial at 0..max_degree"""
    assert len(rounds) == num_vars and len(point) == num_vars
    transcript.append_message(int(num_vars).to_bytes(8, "little"))
    transcript.append_message(int(max_degree).to_bytes(8, "little"))
    claim, chals = claimed_sum, []
    for j in range(num_vars):
        msg = [ext(e) for e in rounds[j]]
        assert len(msg) == max_degree + 1, "round message length"
        assert L.ext_add(msg[0], msg[1]) == claim, f"round {j}: p(0) + p(1) != claim"
        transcript.append_field_elements([w for e in msg for w in e])
        r = transcript.get_and_append_challenge(b"Internal round")
        assert r == ext(point[j]), f"round {j}: the proof's point is not the transcript's challenge"
        chals.append(r)
        claim = lagrange_at(msg, r)
    return chals, claim


def virtual_poly_at(tables, is_ext, terms, point):
    """sum_i coeff_i * prod_j MLE(table_j)(point) by the definition of the multilinear extension (tables may have fewer variables)"""
    total = (0, 0)
    cache = {}
    for coeff, idx in terms:
        prod = (1, 0)
        for t in idx:
            if t not in cache:
                raw = [int(v) for v in tables[t]]
                vals = [(raw[2 * k], raw[2 * k + 1]) for k in range(len(raw) // 2)] if is_ext[t] else [(v, 0) for v in raw]
                k = (len(vals)).bit_length() - 1
                cache[t] = L.mle_eval(vals, point[:k])
            prod = L.ext_mul(prod, cache[t])
        total = L.ext_add(total, L.ext_mul(ext(coeff), prod))
    return total, cache


def merkle_root_from_path(pair_words, pair_index, path):
    """`pair_words`: the words of an opened leaf pair (2 base or 4 extension words), `pair_index`: index of the pair among the pairs,
    `path`: sibling digests from the pair's level up (no leaf sibling, no root)"""
    h = L.hash_or_noop([int(w) for w in pair_words])
    x = pair_index
    for sib in path:
        sib = [int(v) for v in sib]
        h = L.compress(sib, h) if x & 1 else L.compress(h, sib)
        x >>= 1
    return h
