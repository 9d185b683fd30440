"""TEST INFRASTRUCTURE — layer 2 a second time: the IOP part of zkml's VERIFIER for Dense / Requant / ReLU chains, written in Python from the
reference's verifier sources (zkml/src/iop/verifier.rs:72-318, layers/dense.rs:576-643, layers/requant.rs:692-817 + 499-529,
layers/activation.rs:459-517, lookup/logup_gkr/verifier.rs:16-210, commit/same_poly.rs:157-183, lookup/context.rs:323-410 + 758-781,
commit/mod.rs:45-53) on top of the independent field / Poseidon2 / transcript of l0_independent.py and the sumcheck verifier of
l1_independent.py. Nothing here is derived from oracle/*.hpp or csrc/*: it is a second READING of the reference's verifier — transcript
schedule, challenge labels, claim chaining, the recombination formulas of every layer, the table column closed forms, the final logup
fraction sum. What it does not do is open commitments: instead of the batch opening it RETURNS every claim (which polynomial, at which
point, which value), and the test checks each one against the polynomial itself (the weights, the witness columns recomputed from the
inference, the multiplicities) by direct multilinear evaluation — stronger than an opening, and independent of Basefold.
Input: the canonical proof stream parsed by deep_prove_amd.wire.parse_stream (a pure parser) and the model description."""
from . import l0_independent as L
from . import l1_independent as L1

P = L.P
BIT_LEN = 8
Q_MIN, Q_MAX = -127, 127
ONE, ZERO = (1, 0), (0, 0)


def e(v):
    return (int(v[0]) % P, int(v[1]) % P)


def fe(x):
    """a (possibly negative) integer as an extension element"""
    return (int(x) % P, 0)


def add(a, b):
    return L.ext_add(a, b)


def sub(a, b):
    return L.ext_sub(a, b)


def mul(a, b):
    return L.ext_mul(a, b)


def identity_eval(r1, r2):
    """commit/mod.rs:45-53"""
    acc = ONE
    for a, b in zip(r1, r2):
        acc = mul(acc, add(mul(a, b), mul(sub(ONE, a), sub(ONE, b))))
    return acc


def eq_xy_eval(x, y):
    """mpcs/src/sum_check.rs:126-135"""
    assert len(x) == len(y) and x
    acc = ONE
    for a, b in zip(x, y):
        ab = mul(a, b)
        acc = mul(acc, sub(sub(add(add(ab, ab), ONE), a), b))
    return acc


def append_ext(tr, x):
    tr.append_field_elements([x[0], x[1]])


def challenge(tr, label):
    return tr.get_and_append_challenge(label)


def read_challenges(tr, n):
    """zkml/src/lib.rs:146-150 outside cfg(test): n times read_challenge"""
    return [tr.read_challenge() for _ in range(n)]


def verify_logup(proof, num_instances, constant_challenge, column_separation_challenge, tr):
    """lookup/logup_gkr/verifier.rs:16-170; returns (output claims, numerators, denominators)"""
    tr.append_field_elements([num_instances % P])
    outs = [[e(x) for x in row] for row in proof["circuit_outputs"]]
    for row in outs:
        assert len(row) == 4
        for x in row:
            append_ext(tr, x)
    nums = [add(mul(r[0], r[3]), mul(r[1], r[2])) for r in outs]
    dens = [mul(r[2], r[3]) for r in outs]
    batching = challenge(tr, b"initial_batching")
    alpha = challenge(tr, b"initial_alpha")
    lam = challenge(tr, b"initial_lambda")
    claim, comb = ZERO, ONE
    for r in outs:
        term = add(add(mul(batching, sub(r[1], r[0])), r[0]), mul(lam, add(mul(batching, sub(r[3], r[2])), r[2])))
        claim = add(claim, mul(comb, term))
        comb = mul(comb, alpha)
    point = [batching]
    assert len(proof["sumcheck_proofs"]) == len(proof["round_evaluations"])
    for i, (sc, revals) in enumerate(zip(proof["sumcheck_proofs"], proof["round_evaluations"])):
        append_ext(tr, claim)
        sc_point = [e(x) for x in sc["point"]]
        eq_eval = identity_eval(point, sc_point)
        chals, expected = L1.verify_sumcheck(claim, sc_point, sc["proofs"], i + 1, 3, tr)
        batching = challenge(tr, b"logup_batching")
        next_alpha = challenge(tr, b"logup_alpha")
        next_lambda = challenge(tr, b"logup_lambda")
        ev = [e(x) for x in revals]
        assert len(ev) % num_instances == 0
        per = len(ev) // num_instances
        nxt, ncomb, sc_claim, pa = ZERO, ONE, ZERO, ONE
        if per == 4:
            for k in range(0, len(ev), 4):
                c = ev[k:k + 4]
                nxt = add(nxt, mul(ncomb, add(add(mul(batching, sub(c[2], c[0])), c[0]), mul(next_lambda, add(mul(batching, sub(c[1], c[3])), c[3])))))
                inner = add(add(mul(c[0], c[1]), mul(c[2], c[3])), mul(lam, mul(c[3], c[1])))
                sc_claim = add(sc_claim, mul(pa, mul(eq_eval, inner)))
                ncomb = mul(ncomb, next_alpha)
                pa = mul(pa, alpha)
        else:
            assert per == 2
            for k in range(0, len(ev), 2):
                c = ev[k:k + 2]
                nxt = add(nxt, mul(ncomb, add(mul(batching, sub(c[0], c[1])), c[1])))
                inner = add(sub(sub(ZERO, c[1]), c[0]), mul(lam, mul(c[0], c[1])))
                sc_claim = add(sc_claim, mul(mul(pa, eq_eval), inner))
                ncomb = mul(ncomb, next_alpha)
                pa = mul(pa, alpha)
        assert sc_claim == expected, f"logup layer {i}: the round evaluations do not recombine to the sumcheck's final claim"
        claim = nxt
        alpha, lam = next_alpha, next_lambda
        point = chals + [batching]
    claims = [{"point": [e(x) for x in c["point"]], "eval": e(c["eval"])} for c in proof["output_claims"]]
    if not proof["is_table"]:
        per = len(claims) // num_instances
        acc, comb = ZERO, ONE
        for k in range(0, len(claims), per):
            chunk_eval, csc = constant_challenge, ONE
            for cl in claims[k:k + per]:
                chunk_eval = add(chunk_eval, mul(cl["eval"], csc))
                csc = mul(csc, column_separation_challenge)
            acc = add(acc, mul(chunk_eval, comb))
            comb = mul(comb, alpha)
        final = acc
    else:
        cols, csc = constant_challenge, ONE
        for cl in claims[1:]:
            cols = add(cols, mul(cl["eval"], csc))
            csc = mul(csc, column_separation_challenge)
        final = add(claims[0]["eval"], mul(lam, cols))
    assert final == claim, "logup: the output claims do not recombine to the last layer's claim"
    # (the reference takes the claims' points as the proof gives them; an honest proof has them at the last sumcheck point + challenge)
    for cl in claims:
        assert cl["point"] == point, "logup: an output claim is not at the final point of the circuit"
    return claims, nums, dens


def same_poly_verify(claims, proof, num_vars, tr):
    """commit/same_poly.rs:157-183"""
    for c in claims:
        assert len(c["point"]) == num_vars
    a = read_challenges(tr, len(claims))
    y = ZERO
    for c, ai in zip(claims, a):
        y = add(y, mul(c["eval"], ai))
    sc_point = [e(x) for x in proof["sumcheck"]["point"]]
    chals, expected = L1.verify_sumcheck(y, sc_point, proof["sumcheck"]["proofs"], num_vars, 2, tr)
    evals = [e(x) for x in proof["evals"]]
    computed = ZERO
    for c, ai in zip(claims, a):
        computed = add(computed, mul(ai, identity_eval(c["point"], sc_point)))
    assert computed == evals[0], "same_poly: beta evaluation"
    assert mul(evals[0], evals[1]) == expected, "same_poly: final evaluations"
    return {"point": sc_point, "eval": evals[1]}


def table_column_evals(kind, size, point):
    """TableType::evaluate_table_columns (lookup/context.rs:323-410); kind: 'relu' | 'range' | 'clamping'"""
    idx = ZERO
    for k, p in enumerate(point):
        idx = add(idx, mul(p, fe(1 << k)))
    if kind == "range":
        assert len(point) == BIT_LEN
        return [idx]
    if kind == "relu":
        assert len(point) == BIT_LEN
        second = ZERO
        for k, p in enumerate(point[:-1]):
            second = add(second, mul(mul(p, fe(1 << k)), point[-1]))
        return [sub(idx, fe(1 << (BIT_LEN - 1))), second]
    if kind == "gelu":  # the input column min .. max - 1 with min = -2^(size - 1); the output column is a commitment (context.rs:364-378)
        assert len(point) == size[1]
        return [sub(idx, fe(1 << (size[1] - 1)))]
    if kind == "softmax":  # the input column; the output column is a commitment (context.rs:409-423)
        assert len(point) == size[1]
        return [idx]
    if kind == "error":  # nothing but the committed column (:424)
        return []
    if kind == "zero":  # (:425-443)
        assert len(point) == size
        o = ONE
        for p in point:
            o = mul(o, sub(ONE, p))
        return [idx, o]
    if kind == "inv_sqrt":  # only the input column; the output column is a commitment (context.rs:445-462)
        assert len(point) == 2 * (BIT_LEN - 1) + 1
        return [sub(idx, fe(1 << (2 * (BIT_LEN - 1))))]
    assert kind == "clamping" and len(point) == size
    mx = 1 << (size - 1)
    col = [fe(min(max(i, Q_MIN), Q_MAX)) for i in range(-mx, mx)]
    return [sub(idx, fe(mx)), L.mle_eval(col, point)]


def table_order_key(t):
    """derive(Ord) of lookup/context.rs:52-72: Relu < GELU < Range < Clamping(n) < ..."""
    return ({"relu": 0, "gelu": 1, "range": 2, "clamping": 3, "softmax": 4, "error": 5, "zero": 6, "inv_sqrt": 7}[t[0]], t[1])


def verify_chain(layers, model_roots, tree, x, y, label=b"m2vec"):
    """Verifier::verify (iop/verifier.rs:72-318) for a chain of Dense / Requant / ReLU nodes. `layers`: dicts with kind in {'dense',
    'requant', 'relu'} and the parameters of the node; model_roots: {node: [(poly id, root words)]} in BTreeMap order; x / y: the padded
    input and output vectors (integers). Returns the claims the commitment verifier would be handed: ('model', node, poly id, point, eval),
    ('witness', node, k, (root, num_vars), point, eval) with k the position among the node's committed columns, ('multiplicity', table, (root,
    num_vars), point, eval) — in the order the reference's commitment verifier receives them — and the transcript, ready for the opening."""
    tr = L.Transcript(label)
    for node in sorted(model_roots):
        for _, root in sorted(model_roots[node]):
            tr.append_field_elements([int(w) for w in root])
    tables = set()
    for l in layers:
        if l["kind"] == "requant":
            tables.add(("range", 0))
            tables.add(("clamping", l["clamping_size"]))
        elif l["kind"] == "relu":
            tables.add(("relu", 0))
    tables = sorted(tables, key=table_order_key)
    chmap = {}
    if tables:
        constant = challenge(tr, b"table_constant")
        for t in tables:
            chmap[t] = {"relu": lambda: challenge(tr, b"Relu"), "range": lambda: ONE, "clamping": lambda: challenge(tr, b"Clamping")}[t[0]]()
    steps = {node: (kind, lp) for node, kind, lp in tree["steps"]}
    nums, dens = [], []

    def fractions(lg):
        for r in lg["circuit_outputs"]:
            r = [e(v) for v in r]
            nums.append(add(mul(r[0], r[3]), mul(r[1], r[2])))
            dens.append(mul(r[2], r[3]))

    for node in range(len(layers)):  # forward order: the lookup data of every step
        kind, lp = steps[node]
        if layers[node]["kind"] == "relu":
            fractions(lp["lookup"])
        elif layers[node]["kind"] == "requant":
            fractions(lp["clamping_lookup"])
            fractions(lp["shifted_lookup"])
    for tp in tree["table_proofs"]:
        fractions(tp["lookup"])
    # output claim
    n_out = len(y)
    r = read_challenges(tr, n_out.bit_length() - 1)
    cur = {"point": r, "eval": L.mle_eval([fe(v) for v in y], r)}
    out = []
    for node in range(len(layers) - 1, -1, -1):
        l = layers[node]
        kind, lp = steps[node]
        if l["kind"] == "dense":
            bias_eval = e(lp["bias_eval"])
            sc_point = [e(v) for v in lp["sumcheck"]["point"]]
            nv = (l["ncols"]).bit_length() - 1
            chals, expected = L1.verify_sumcheck(sub(cur["eval"], bias_eval), sc_point, lp["sumcheck"]["proofs"], nv, 2, tr)
            ic = [e(v) for v in lp["individual_claims"]]
            # add_common_claims: BTreeMap order of the poly ids ("DenseBias" < "DenseWeight")
            out.append(("model", node, "DenseBias", cur["point"], bias_eval))
            out.append(("model", node, "DenseWeight", sc_point + cur["point"], ic[0]))
            assert mul(ic[0], ic[1]) == expected, f"dense {node}: sumcheck claim failed"
            cur = {"point": sc_point, "eval": ic[1]}
        elif l["kind"] == "requant":
            ct = ("clamping", l["clamping_size"])
            shift = l["fp_scale"] + l["right_shift"]
            inst = shift // BIT_LEN
            cclaims, _, _ = verify_logup(lp["clamping_lookup"], 1, constant, chmap[ct], tr)
            sclaims, _, _ = verify_logup(lp["shifted_lookup"], inst, constant, ONE, tr)
            b = challenge(tr, b"requant_batching")
            cpt, spt = cclaims[0]["point"], sclaims[0]["point"]
            init, ch = ZERO, ONE
            for v in [cur["eval"], cclaims[1]["eval"], cclaims[0]["eval"]] + [c["eval"] for c in sclaims]:
                init = add(init, mul(ch, v))
                ch = mul(ch, b)
            acc_pt = [e(v) for v in lp["io_accumulation"]["point"]]
            chals, expected = L1.verify_sumcheck(init, acc_pt, lp["io_accumulation"]["proofs"], len(cpt), 2, tr)
            ae = [e(v) for v in lp["accumulation_evals"]]
            lb, cb, sb = eq_xy_eval(cur["point"], acc_pt), eq_xy_eval(cpt, acc_pt), eq_xy_eval(spt, acc_pt)
            calc = mul(add(lb, mul(b, cb)), ae[1])
            comb = mul(b, b)
            calc = add(calc, mul(mul(comb, cb), ae[0]))
            comb = mul(comb, b)
            for v in ae[2:]:
                calc = add(calc, mul(mul(v, sb), comb))
                comb = mul(comb, b)
            assert calc == expected, f"requant {node}: accumulation evaluations do not recombine"
            # recombine_claims (requant.rs:499-529)
            full, pw = mul(fe(1 << shift), ae[0]), ONE
            for v in ae[2:]:
                full = add(full, mul(v, pw))
                pw = mul(pw, fe(1 << BIT_LEN))
            nxt = mul(sub(full, fe(1 << (shift - 1))), L.ext_inv(fe(l["fixed_point_multiplier"])))
            assert len(lp["commitments"]) == len(ae)
            for q, (v, c) in enumerate(zip(ae, lp["commitments"])):
                out.append(("witness", node, q, (tuple(c["root"]), c["num_vars"]), acc_pt, v))
            cur = {"point": acc_pt, "eval": nxt}
        else:
            assert l["kind"] == "relu"
            claims, _, _ = verify_logup(lp["lookup"], 1, constant, chmap[("relu", 0)], tr)
            nv = len(cur["point"])
            new_out = same_poly_verify([cur] + claims[1:], lp["io_accumulation"], nv, tr)
            assert len(lp["commits"]) == 2
            out.append(("witness", node, 0, (tuple(lp["commits"][0]["root"]), lp["commits"][0]["num_vars"]), claims[0]["point"], claims[0]["eval"]))
            out.append(("witness", node, 1, (tuple(lp["commits"][1]["root"]), lp["commits"][1]["num_vars"]), new_out["point"], new_out["eval"]))
            cur = claims[0]
    # table proofs, in the order of the lookup context
    assert len(tree["table_proofs"]) == len(tables)
    for tp, t in zip(tree["table_proofs"], tables):
        claims, _, _ = verify_logup(tp["lookup"], 1, constant, chmap[t], tr)
        out.append(("multiplicity", t, (tuple(tp["multiplicity_commit"]["root"]), tp["multiplicity_commit"]["num_vars"]), claims[0]["point"], claims[0]["eval"]))
        expect = table_column_evals(t[0], t[1], claims[0]["point"])
        assert len(expect) == len(claims) - 1
        for cl, ex in zip(claims[1:], expect):
            assert cl["eval"] == ex, f"table {t}: claimed column evaluation is wrong"
    # the input claim
    assert len(cur["point"]) == len(x).bit_length() - 1 and L.mle_eval([fe(v) for v in x], cur["point"]) == cur["eval"], "input claim is incorrect"
    # the global logup check (iop/verifier.rs:273-291)
    fn, fd = ZERO, ONE
    for nu, de in zip(nums, dens):
        fn, fd = add(mul(fn, de), mul(nu, fd)), mul(fd, de)
    assert fn == ZERO, "final logup numerator is not zero"
    assert fd != ZERO, "final logup denominator is zero"
    return out, tr


# ---------------------------------------------------------------------------------------------------------------- graphs
def backward_order(nodes, outputs):
    """NodeIterator<_, false> (model/iterator.rs:152-185): again and again the smallest unvisited id all of whose readers have been visited"""
    readers = {}
    for nid, n in enumerate(nodes):
        for pos, (src, slot) in enumerate(n["inputs"]):
            readers.setdefault((src, slot), []).append((nid, pos))
    for k, (src, slot) in enumerate(outputs):
        readers.setdefault((src, slot), []).append((-1, k))
    order, done = [], set()
    while len(order) < len(nodes):
        for nid, n in enumerate(nodes):
            if nid in done:
                continue
            if all(r[0] < 0 or r[0] in done for j in range(n["n_out"]) for r in readers.get((nid, j), [])):
                order.append(nid)
                done.add(nid)
                break
        else:
            raise AssertionError("cycle")
    return order, readers


def verify_graph(nodes, outputs, model_roots, tree, input_tensors, output_tensors, label=b"m2vec"):
    """Verifier::verify (iop/verifier.rs:72-318) for a GRAPH of QKV / ConcatMatMul / MatMul / Add / Requant nodes. nodes[i]: kind, inputs
    [(node or -1, slot)], n_out and the parameters of the kind; outputs: [(node, slot)]; input_tensors / output_tensors: lists of integer
    vectors. Returns (claims for the commitment verifier, transcript). The claims on the model's input tensors are checked here (step 6)."""
    tr = L.Transcript(label)
    for node in sorted(model_roots):
        for _, root in sorted(model_roots[node]):
            tr.append_field_elements([int(w) for w in root])
    tables = set()
    for n in nodes:
        if n["kind"] == "requant":
            tables.add(("range", 0))
            tables.add(("clamping", n["clamping_size"]))
        elif n["kind"] == "relu":
            tables.add(("relu", 0))
        elif n["kind"] == "gelu":  # TableType::GELU(GELUQuantData {multiplier, min, max}) (activation.rs:163-167): (multiplier, log2 of the table length)
            tables.add(("gelu", (n["multiplier"], 8 + (n["multiplier"] - 1).bit_length())))
        elif n["kind"] == "maxpool":
            tables.add(("range", 0))
        elif n["kind"] == "layernorm":
            tables.add(("range", 0))
            tables.add(("inv_sqrt", (n["eps_bits"], n["range_check_bits"])))
        elif n["kind"] in ("softmax", "mha"):  # (an Mha node brings the tables of its softmax, Mha::step_info, mha.rs:432-503)
            tables.add(("range", 0))
            tables.add(("softmax", (n["temp_bits"], n["table_size"], n["bkm"])))
            tables.add(("error", n["allowable_error"]))
            if n["zero_vars"]:
                tables.add(("zero", n["zero_vars"]))
    tables = sorted(tables, key=table_order_key)
    chmap, constant = {}, None
    if tables:
        constant = challenge(tr, b"table_constant")
        for t in tables:
            chmap[t] = ONE if t[0] in ("range", "error") else challenge(tr, {"relu": b"Relu", "gelu": b"GELU", "clamping": b"Clamping", "inv_sqrt": b"InverseSQRT", "softmax": b"Softmax", "zero": b"Zero"}[t[0]])
    steps = {node: (kind, lp) for node, kind, lp in tree["steps"]}
    nums, dens = [], []

    def fractions(lg):
        for r in lg["circuit_outputs"]:
            r = [e(v) for v in r]
            nums.append(add(mul(r[0], r[3]), mul(r[1], r[2])))
            dens.append(mul(r[2], r[3]))

    for nid, n in enumerate(nodes):
        if n["kind"] == "requant":
            fractions(steps[nid][1]["clamping_lookup"])
            fractions(steps[nid][1]["shifted_lookup"])
        elif n["kind"] in ("relu", "gelu", "maxpool"):
            fractions(steps[nid][1]["lookup"])
        elif n["kind"] in ("layernorm", "softmax"):
            for lg in steps[nid][1]["logup_proofs"]:
                fractions(lg)
        elif n["kind"] == "mha":  # MhaProof::get_lookup_data (mha.rs:130-134): the softmax's
            for lg in steps[nid][1]["softmax_proof"]["logup_proofs"]:
                fractions(lg)
    for tp in tree["table_proofs"]:
        fractions(tp["lookup"])
    out_claims = []
    for (src, slot), y in zip(outputs, output_tensors):
        r = read_challenges(tr, len(y).bit_length() - 1)
        out_claims.append({"point": r, "eval": L.mle_eval([fe(v) for v in y], r)})
    order, readers = backward_order(nodes, outputs)
    made, out = {}, []
    # MhaCtx::verify (layers/transformer/mha.rs:792-893): final_mul.verify on the node's output claim -> (claim on the probabilities, claim on V);
    # softmax.verify on the first; qk.verify on the softmax's claim -> (claims on Q, K); the node hands on [Q, K, V]. Walked as three steps through
    # the concat_matmul / softmax branches below, each with the sub-layer Mha::new builds (mha.rs:147-186) and its part of the MhaProof.
    vsteps, mha_hold, last_vstep = [], None, None
    for nid in order:
        vsteps += [(nid, 1), (nid, 2), (nid, 3)] if nodes[nid]["kind"] == "mha" else [(nid, 0)]

    def settle(step):
        nonlocal mha_hold
        if step is not None and step[1] == 1:
            mha_hold = made[step[0]]
            assert len(mha_hold) == 2
        if step is not None and step[1] == 3:
            made[step[0]] = made[step[0]] + [mha_hold[1]]
            assert len(made[step[0]]) == 3

    for nid, part in vsteps + [(None, 0)]:
        settle(last_vstep)
        last_vstep = (nid, part)
        if nid is None:
            break
        n = nodes[nid]
        if part:
            S_, H_, D_ = n["shape"]
            if part == 1:
                n = dict(kind="concat_matmul", a_shape=(H_, S_, S_), b_shape=(S_, H_, D_), left=(0, 2, 1), right=(1, 0, 2), perm=(1, 0, 2), n_out=1)
            elif part == 2:
                n = dict(n, kind="softmax", shape=(H_, S_, S_))
            else:
                n = dict(kind="concat_matmul", a_shape=(S_, H_, D_), b_shape=(S_, H_, D_), left=(1, 2, 0), right=(1, 2, 0), perm=None, n_out=1)
        got = []
        if part == 2:
            got = [mha_hold[0]]
        elif part == 3:
            got = [made[nid][0]]
        else:
            for j in range(nodes[nid]["n_out"]):
                (rd,) = readers[(nid, j)]  # exactly one reader per tensor (provable/mod.rs:243-248)
                got.append(out_claims[rd[1]] if rd[0] < 0 else made[rd[0]][rd[1]])
        cur = got[0]
        if n["kind"] == "reshape":
            made[nid] = [cur]
            continue
        kind, lp = steps[nid]
        if part:
            assert kind == 16
            lp = lp[("final_mul_proof", "softmax_proof", "qk_proof")[part - 1]]
        if n["kind"] == "conv":  # layers/convolution.rs:1143-1383 (+ hadamard.rs:128-156, verify_fft_delegation :1090-1141)
            from .l3_independent import two_adic
            kw, kx, rnw, nw = n["kw"], n["kx"], n["real_nw"], n["nw"]
            fs = nw * nw
            lfs, lkw, lkx = fs.bit_length() - 1, kw.bit_length() - 1, kx.bit_length() - 1
            l2n = lfs + 1
            ve = lambda v: [e(t) for t in v]

            def sc(claim, proof, nv, deg):  # IOPVerifierState::verify: rounds and challenges are checked, the caller decides about the final claim
                pt = ve(proof["point"])
                _, final = L1.verify_sumcheck(claim, pt, proof["proofs"], nv, deg, tr)
                return pt, final

            def pow_two_omegas(m, is_fft):  # (:1442-1454)
                rou = two_adic(m)
                if is_fft:
                    rou = L.inv(rou)
                pows = [rou]
                for _ in range(1, m - 1):
                    pows.append(pows[-1] * pows[-1] % P)
                return pows

            def phi_eval(r, rand1, rand2, exps, first):  # (:1456-1476)
                ev = ONE
                for i, ri in enumerate(r):
                    ev = mul(ev, add(sub(ONE, ri), mul(ri, fe(exps[len(exps) - len(r) + i]))))
                if first:
                    return mul(sub(ONE, rand2), add(sub(ONE, rand1), mul(rand1, ev)))
                return add(sub(ONE, rand1), mul(mul(sub(ONE, add(rand2, rand2)), rand1), ev))

            oc, oh, ow = n["unp_out"]
            clr = [fe(1 if (c < oc and yy < oh and xx < ow) else 0) for c in range(kw) for yy in range(nw) for xx in range(nw)]  # new_clearing_tensor (:1508-1529)
            hp, hfinal = sc(cur["eval"], lp["clearing_proof"]["sumcheck"], lfs + lkw, 3)
            v1, v2 = ve(lp["clearing_proof"]["individual_claim"])
            assert L.mle_eval(clr, hp) == v2, "Hadamard verification failed for v2 eval"
            assert mul(mul(identity_eval(cur["point"], hp), v1), v2) == hfinal, "Hadamard verification failed for product eval"
            last = {"point": hp, "eval": v1}
            bias_claim = e(lp["bias_claim"])
            ifft_pt, _ = sc(sub(last["eval"], bias_claim), lp["ifft_proof"], l2n, 2)
            ifft_claims, dclaims = ve(lp["ifft_claims"]), [ve(c) for c in lp["ifft_delegation_claims"]]
            it = len(lp["ifft_delegation_proof"])
            assert it == lfs and len(dclaims) == it
            claim, exps, prev = ifft_claims[1], pow_two_omegas(it + 1, True), ifft_pt
            for i in range(it):
                dpt, _ = sc(claim, lp["ifft_delegation_proof"][i], lfs - i, 3)
                assert identity_eval(dpt, prev) == dclaims[i][0], f"identity evaluation, ifft delegation {i}"
                assert phi_eval(dpt, sub(ONE, last["point"][i]), prev[-1], exps, False) == dclaims[i][1], f"phi, ifft delegation {i}"
                prev, claim = dpt, dclaims[i][2]
            scale = fe(L.inv((1 << (it + 1)) % P))
            assert claim == add(mul(scale, prev[0]), mul(scale, sub(ONE, prev[0]))), "final iFFT delegation step"
            had_clams = ve(lp["hadamard_clams"])
            had_pt, _ = sc(ifft_claims[0], lp["hadamard_proof"], lkx + lfs + 1, 3)
            assert had_clams[2] == identity_eval(ifft_pt, had_pt), "Error in Beta evaluation"

            def fft_delegation(claim, proofs, claims, prev):
                m = len(proofs)
                exps = pow_two_omegas(m + 1, False)
                claims = [ve(c) for c in claims]
                for i in range(m):
                    dpt, _ = sc(claim, proofs[i], lfs - i, 3)
                    assert identity_eval(dpt, prev) == claims[i][0], f"identity evaluation, fft delegation {i}"
                    assert phi_eval(dpt, had_pt[i], prev[-1], exps, i == 0) == claims[i][1], f"phi, fft delegation {i}"
                    claim, prev = claims[i][2], dpt
                assert claim == add(sub(mul(sub(ONE, add(had_pt[m], had_pt[m])), prev[0]), prev[0]), ONE), "final FFT delegation step"

            fft_pt, _ = sc(had_clams[1], lp["fft_proof"], l2n, 2)
            fft_claims = ve(lp["fft_claims"])
            fft_delegation(fft_claims[1], lp["fft_delegation_proof"], lp["fft_delegation_claims"], fft_pt)
            fftw_pt, _ = sc(had_clams[0], lp["fft_proof_weights"], l2n, 2)
            fftw_claims = ve(lp["fft_weight_claims"])
            fft_delegation(fftw_claims[1], lp["fft_delegation_proof_weights"], lp["fft_delegation_weights_claims"], fftw_pt)
            wpt = list(fftw_pt)
            v = L.ext_inv(sub(ONE, wpt.pop()))
            pe = ve(lp["partial_evals"])
            yw = ZERO
            lgnw2 = 2 * (nw.bit_length() - 1)
            for i in range(rnw):
                for j in range(rnw):
                    bits = [fe(((i * nw + j) >> b) & 1) for b in range(lgnw2)]
                    yw = add(yw, mul(pe[i * rnw + j], identity_eval(bits, wpt)))
            assert mul(fftw_claims[0], v) == yw, "Error in padded_fft evaluation claim"
            wrand = read_challenges(tr, (rnw * rnw).bit_length() - 1)
            point = had_pt + last["point"][lfs:]
            out.append(("model", nid, "ConvBias", last["point"][it:], bias_claim))
            out.append(("model", nid, "ConvFilter", wrand + point[(2 * nw * nw).bit_length() - 1:], L.mle_eval(pe, wrand)))
            ipt = list(fft_pt)
            v = L.ext_inv(sub(ONE, ipt.pop()))
            ipt = [sub(ONE, t) for t in ipt]
            made[nid] = [{"point": ipt + had_pt[(2 * fs).bit_length() - 1:], "eval": mul(fft_claims[0], v)}]
        elif n["kind"] == "maxpool":  # layers/pooling.rs:525-647
            claims, _, _ = verify_logup(lp["lookup"], 4, constant, ONE, tr)
            c_, h_, w_ = n["pin"]
            nv = (c_ * (h_ // 2) * (w_ // 2)).bit_length() - 1
            b = challenge(tr, b"batch_pooling")
            init, comb = ZERO, b
            for cl in claims:
                init = add(init, mul(cl["eval"], comb))
                comb = mul(comb, b)
            init = add(init, mul(comb, cur["eval"]))
            zp = [e(v) for v in lp["sumcheck"]["point"]]
            chals, expected = L1.verify_sumcheck(init, zp, lp["sumcheck"]["proofs"], nv, 5, tr)
            beta_eval, last_beta = eq_xy_eval(claims[0]["point"], zp), eq_xy_eval(cur["point"], zp)
            ze = [e(v) for v in lp["zerocheck_evals"]]
            ks = len(ze) - 1
            prod_c, sum_c, bc = beta_eval, ZERO, b
            for v in ze[:ks]:
                prod_c, sum_c, bc = mul(prod_c, v), add(sum_c, mul(bc, v)), mul(bc, b)
            out_eval = ze[ks]
            assert add(add(prod_c, mul(sum_c, beta_eval)), mul(mul(out_eval, last_beta), bc)) == expected, "pooling zerocheck claim"
            assert len(lp["commitments"]) == len(ze)
            for q, (v, c) in enumerate(zip(ze, lp["commitments"])):
                out.append(("witness", nid, q, (tuple(c["root"]), c["num_vars"]), zp, v))
            r1 = challenge(tr, b"input_batching")  # `[challenge; 2]`: ONE challenge, used twice (:611-614)
            r2 = r1
            om1, om2 = sub(ONE, r1), sub(ONE, r2)
            mults = [mul(om1, om2), mul(om1, r2), mul(r1, om2), mul(r1, r2)]
            gap = lp["variable_gap"]
            zin = ZERO
            for v, m_ in zip(ze[:ks], mults):
                zin = add(zin, mul(sub(out_eval, v), m_))
            made[nid] = [{"point": [r1] + zp[:gap] + [r2] + zp[gap:], "eval": zin}]
        elif n["kind"] == "softmax":  # layers/transformer/softmax.rs:1274-1586
            zc, zv, ts, bkm = n["zero_chunks"], n["zero_vars"], n["table_size"], n["bkm"]
            ve = lambda v: [e(t) for t in v]
            lgs = lp["logup_proofs"]
            assert len(lgs) == (4 if zc else 3)
            exp_c, _, _ = verify_logup(lgs[0], 1, constant, chmap[("softmax", (n["temp_bits"], ts, bkm))], tr)
            rng_c, _, _ = verify_logup(lgs[1], 2, constant, ONE, tr)
            err_c, _, _ = verify_logup(lgs[2], 1, constant, ONE, tr)
            zero_c = verify_logup(lgs[3], zc, constant, chmap[("zero", zv)], tr)[0] if zc else []
            assert len(exp_c) == 2 and len(rng_c) == 2 and len(err_c) == 1 and len(zero_c) == 2 * zc
            alpha = challenge(tr, b"batching_challenge")
            total, bc = ZERO, ONE
            for cl in exp_c + rng_c + zero_c + err_c:
                total, bc = add(total, mul(bc, cl["eval"])), mul(bc, alpha)
            total = add(total, mul(bc, cur["eval"]))
            nv = len(exp_c[0]["point"])
            extra = nv - len(err_c[0]["point"])
            C_, R_, K_ = n["shape"]
            assert extra == K_.bit_length() - 1 and len(cur["point"]) == nv
            two_inv, two_mult = L.ext_inv(fe(2)), fe(1 << extra)
            acc_pt = ve(lp["accumulation_proof"]["point"])
            _, expected = L1.verify_sumcheck(total, acc_pt, lp["accumulation_proof"]["proofs"], nv, zc + 2 if zc else 2, tr)
            last_beta, exp_beta, range_beta = eq_xy_eval(cur["point"], acc_pt), eq_xy_eval(exp_c[0]["point"], acc_pt), eq_xy_eval(rng_c[0]["point"], acc_pt)
            error_beta = eq_xy_eval([two_inv] * extra + err_c[0]["point"], acc_pt)
            ev = ve(lp["evaluations"])
            assert len(ev) == 5 + 2 * zc and len(lp["commitments"]) == len(ev)
            calc = mul(exp_beta, add(ev[0], mul(ev[1], alpha)))
            bc = mul(alpha, alpha)
            for k in (2, 3):
                calc, bc = add(calc, mul(mul(range_beta, ev[k]), bc)), mul(bc, alpha)
            out_eval = ev[1]
            if zc:
                zbeta = eq_xy_eval(zero_c[0]["point"], acc_pt)
                for k in range(5, len(ev)):
                    calc, bc = add(calc, mul(mul(zbeta, ev[k]), bc)), mul(bc, alpha)
                for k in range(6, len(ev), 2):
                    out_eval = mul(out_eval, ev[k])
            calc = add(calc, mul(mul(bc, out_eval), add(mul(error_beta, two_mult), mul(alpha, last_beta))))
            assert calc == expected, "softmax: accumulation evaluations do not recombine"
            mask_in = add(add(mul(ev[0], fe(1 << 16)), mul(ev[3], fe(1 << 8))), ev[2])
            m_ = fe(1 << (16 + ts))
            for k in range(5, len(ev), 2):
                mask_in, m_ = add(mask_in, mul(ev[k], m_)), mul(m_, fe(1 << zv))
            mpt = ve(lp["mask_proof"]["point"])
            _, mexp = L1.verify_sumcheck(sub(ZERO, mask_in), mpt, lp["mask_proof"]["proofs"], nv, 3, tr)
            eqv = eq_xy_eval(mpt, acc_pt)
            cv, rv = K_.bit_length() - 1, R_.bit_length() - 1
            tril = ONE  # eval_zeroifier_mle (mha.rs:894-901)
            for c_, r_ in zip(mpt[:cv], mpt[cv:cv + rv]):
                tril = add(mul(tril, add(sub(sub(ONE, c_), r_), mul(fe(2), mul(c_, r_)))), mul(sub(ONE, c_), r_))
            neg_inf = fe((-(((bkm >> 16) + 1) << 16)) % P)
            shifted = mul(sub(mexp, mul(eqv, mul(neg_inf, sub(ONE, tril)))), L.ext_inv(mul(eqv, tril)))
            for q, (v, c) in enumerate(zip(ev, lp["commitments"])):
                out.append(("witness", nid, q, (tuple(c["root"]), c["num_vars"]), mpt[extra:] if q == 4 else acc_pt, v))
            made[nid] = [{"point": mpt, "eval": mul(sub(shifted, ev[4]), L.ext_inv(fe(n["scalar"])))}]
        elif n["kind"] == "layernorm":  # layers/transformer/layernorm.rs:1230-1505
            rcb, nrc = n["range_check_bits"], (n["range_check_bits"] - 1) // BIT_LEN + 1
            ve = lambda v: [e(t) for t in v]
            assert len(lp["logup_proofs"]) == 2
            ic, _, _ = verify_logup(lp["logup_proofs"][0], 1, constant, chmap[("inv_sqrt", (n["eps_bits"], rcb))], tr)
            rc, _, _ = verify_logup(lp["logup_proofs"][1], nrc, constant, ONE, tr)
            assert len(ic) == 2 and len(rc) == nrc
            bc = [challenge(tr, b"batching") for _ in range((2 + nrc - 1).bit_length())]
            rlc = L.eq_table(bc)
            acc_init = ZERO
            for cl, ch in zip(ic + rc, rlc):
                acc_init = add(acc_init, mul(cl["eval"], ch))
            nv = len(ic[0]["point"])
            acc_pt = ve(lp["accumulation_proof"]["point"])
            _, acc_expected = L1.verify_sumcheck(acc_init, acc_pt, lp["accumulation_proof"]["proofs"], nv, 2, tr)
            eq_sqrt, eq_range = eq_xy_eval(ic[0]["point"], acc_pt), eq_xy_eval(rc[0]["point"], acc_pt)
            acc_evals, evaluations = ve(lp["acc_evals"]), ve(lp["evaluations"])
            assert len(acc_evals) == 2 + nrc and len(evaluations) == 4 + nrc and len(lp["commitments"]) == 2 + nrc
            calc = ZERO
            for k, (v, ch) in enumerate(zip(acc_evals, rlc)):
                calc = add(calc, mul(mul(v, eq_sqrt if k < 2 else eq_range), ch))
            assert calc == acc_expected, "layernorm: accumulation evaluations do not recombine"
            c1, c2 = challenge(tr, b"batching"), challenge(tr, b"batching")
            first, second, third = mul(sub(ONE, c1), sub(ONE, c2)), mul(c1, sub(ONE, c2)), mul(sub(ONE, c1), c2)
            partial, pw = mul(acc_evals[0], fe(1 << rcb)), ONE
            for v in acc_evals[2:2 + nrc - 1]:
                partial, pw = add(partial, mul(v, pw)), mul(pw, fe(1 << BIT_LEN))
            top_inv = L.ext_inv(fe(1 << n["top_chunk_scalar_log"]))
            io_init = add(add(mul(first, add(partial, mul(mul(acc_evals[-1], top_inv), pw))), mul(second, cur["eval"])), mul(acc_evals[1], third))
            sdv = (n["dim_size"] - 1).bit_length()
            io_pt = ve(lp["io_proof"]["point"])
            _, io_expected = L1.verify_sumcheck(io_init, io_pt, lp["io_proof"]["proofs"], len(cur["point"]), 4, tr)
            input_io, mean_io, inv_ev = evaluations[-2], evaluations[-1], evaluations[1]
            gamma_eval, beta_eval = e(lp["gamma_eval"]), e(lp["beta_eval"])
            n_f, two_inv, two_mul, mult_f = fe(n["dim_size"]), L.ext_inv(fe(2)), fe(1 << sdv), fe(n["multiplier"])
            full_point = [two_inv] * sdv + acc_pt
            input_eq, last_eq = eq_xy_eval(full_point, io_pt), eq_xy_eval(cur["point"], io_pt)
            p1 = mul(mul(mul(first, mult_f), input_eq), sub(mul(mul(n_f, two_mul), mul(input_io, input_io)), mul(mean_io, mean_io)))
            p2 = mul(mul(second, last_eq), add(mul(mul(inv_ev, gamma_eval), sub(mul(n_f, input_io), mean_io)), beta_eval))
            p3 = mul(mul(third, input_eq), inv_ev)
            assert add(add(p1, p2), p3) == io_expected, "layernorm: io evaluations do not recombine"
            ich = challenge(tr, b"batching")
            in_pt = ve(lp["input_proof"]["point"])
            _, in_expected = L1.verify_sumcheck(add(input_io, mul(ich, sub(mean_io, input_io))), in_pt, lp["input_proof"]["proofs"], len(io_pt), 2, tr)
            eq_io, eq_sum = eq_xy_eval(io_pt, in_pt), eq_xy_eval([two_inv] * sdv + io_pt[sdv:], in_pt)
            non_input = add(eq_io, mul(ich, sub(mul(two_mul, eq_sum), eq_io)))
            for q, (v, c) in enumerate(zip(evaluations[:2 + nrc], lp["commitments"])):
                out.append(("witness", nid, q, (tuple(c["root"]), c["num_vars"]), io_pt[sdv:] if q == 1 else acc_pt, v))
            out.append(("model", nid, "LayerNormBeta", io_pt[:sdv], beta_eval))
            out.append(("model", nid, "LayerNormGamma", io_pt[:sdv], gamma_eval))
            made[nid] = [{"point": in_pt, "eval": mul(in_expected, L.ext_inv(non_input))}]
        elif n["kind"] == "dense":  # layers/dense.rs:576-643
            bias_eval = e(lp["bias_eval"])
            sc_point = [e(v) for v in lp["sumcheck"]["point"]]
            chals, expected = L1.verify_sumcheck(sub(cur["eval"], bias_eval), sc_point, lp["sumcheck"]["proofs"], (n["ncols"]).bit_length() - 1, 2, tr)
            ic = [e(v) for v in lp["individual_claims"]]
            out.append(("model", nid, "DenseBias", cur["point"], bias_eval))
            out.append(("model", nid, "DenseWeight", sc_point + cur["point"], ic[0]))
            assert mul(ic[0], ic[1]) == expected, f"dense {nid}: sumcheck claim failed"
            made[nid] = [{"point": sc_point, "eval": ic[1]}]
        elif n["kind"] == "embeddings":  # layers/transformer/embeddings.rs:473-528
            nvc = (n["ncols"]).bit_length() - 1
            cols, rows = cur["point"][:nvc], cur["point"][nvc:]  # split_output_point (:95-106)
            sc_point = [e(v) for v in lp["sumcheck"]["point"]]
            chals, expected = L1.verify_sumcheck(cur["eval"], sc_point, lp["sumcheck"]["proofs"], (n["nrows"]).bit_length() - 1, 2, tr)
            ic = [e(v) for v in lp["individual_claims"]]
            out.append(("model", nid, "EmbeddingMat", cols + sc_point, ic[1]))
            assert mul(ic[0], ic[1]) == expected, "embeddings: sumcheck claim failed"
            made[nid] = [{"point": sc_point + rows, "eval": ic[0], "one_hot_of": n["nrows"]}]
        elif n["kind"] in ("add_const", "positional"):  # layers/add.rs:586-625 with a static operand; transformer/positional.rs:480-583
            le, re_ = e(lp["left_eval"]), e(lp["right_eval"])
            assert add(mul(le, fe(n["left"])), mul(re_, fe(n["right"]))) == cur["eval"], "Add layer verification failed"
            if n["kind"] == "add_const":
                out.append(("model", nid, "255", cur["point"], re_))
            else:
                diff = n["table_vars"] - len(cur["point"])
                assert diff >= 0 and len(lp["sub_matrix_evals"]) == diff
                # sample_random_coordinates (:80-96): both claims enter the transcript, then `diff` plain challenges
                tr.append_field_elements([w for x in cur["point"] for w in x])
                append_ext(tr, cur["eval"])
                tr.append_field_elements([w for x in cur["point"] for w in x])
                append_ext(tr, re_)
                extra = read_challenges(tr, diff)
                point = cur["point"] + extra
                val = re_
                for sm, c in zip([e(v) for v in lp["sub_matrix_evals"]], extra):  # compute_positional_matrix_claim (:106-124)
                    val = add(mul(val, sub(ONE, c)), mul(sm, c))
                out.append(("model", nid, "PositionalMatrix", point, val))
            made[nid] = [{"point": cur["point"], "eval": le}]
        elif n["kind"] == "add2":  # layers/add.rs:586-625, no operand
            le, re_ = e(lp["left_eval"]), e(lp["right_eval"])
            assert add(mul(le, fe(n["left"])), mul(re_, fe(n["right"]))) == cur["eval"], "Add layer verification failed"
            made[nid] = [{"point": cur["point"], "eval": le}, {"point": cur["point"], "eval": re_}]
        elif n["kind"] in ("matmul2", "matmul"):  # layers/matrix_mul.rs:1048-1139
            nvc = (n["ncols"]).bit_length() - 1
            cols, rows = cur["point"][:nvc], cur["point"][nvc:]  # split_claim: the low coordinates address the columns (:339-356)
            ev = cur["eval"]
            if n["kind"] == "matmul" and n.get("bias") is not None:
                be = e(lp["bias_eval"])
                out.append(("model", nid, "MatMulBias", cols, be))
                ev = sub(ev, be)
            else:
                assert lp["bias_eval"] is None
            sc_point = [e(v) for v in lp["sumcheck"]["point"]]
            chals, expected = L1.verify_sumcheck(ev, sc_point, lp["sumcheck"]["proofs"], (n["nrows"]).bit_length() - 1, 2, tr)
            ic = [e(v) for v in lp["individual_claims"]]
            assert mul(ic[0], ic[1]) == expected, "matmul: sumcheck claim failed"
            p_left = sc_point + rows
            p_right = (sc_point + cols) if n.get("transpose_b") else (cols + sc_point)  # full_points (:364-383)
            if n["kind"] == "matmul":
                out.append(("model", nid, "MatMulWeight", p_right, ic[1]))
                made[nid] = [{"point": p_left, "eval": ic[0]}]
            else:
                made[nid] = [{"point": p_left, "eval": ic[0]}, {"point": p_right, "eval": ic[1]}]
        elif n["kind"] == "concat_matmul":  # layers/concat_matmul.rs:801-892
            a_shape, b_shape, left, right, perm = n["a_shape"], n["b_shape"], n["left"], n["right"], n["perm"]
            C_, R_, M_, N_ = a_shape[left[0]], a_shape[left[2]], a_shape[left[1]], b_shape[right[2]]
            res_shape = [C_, R_, N_]
            oshape = res_shape if perm is None else [res_shape[p] for p in perm]
            sc_point = [e(v) for v in lp["sumcheck_proof"]["point"]]
            chals, expected = L1.verify_sumcheck(cur["eval"], sc_point, lp["sumcheck_proof"]["proofs"], (C_ * M_).bit_length() - 1, 3, tr)
            # split_output_claim_point (:295-343): the last axis of the output owns the lowest coordinates
            parts, hi = [], len(cur["point"])
            for d in range(3):
                nv = oshape[d].bit_length() - 1
                parts.append(cur["point"][hi - nv:hi])
                hi -= nv
            assert hi == 0
            where = [0, 1, 2]
            if perm is not None:
                for i, src_dim in enumerate(perm):
                    where[src_dim] = i
            p_concat, p_row, p_col = parts[where[0]], parts[where[1]], parts[where[2]]
            nvm = M_.bit_length() - 1
            s_mm, s_concat = sc_point[:nvm], sc_point[nvm:]  # split_sumcheck_point (:256-281)
            ic = [e(v) for v in lp["individual_claims"]]
            assert identity_eval(s_concat, p_concat) == ic[0], "concat matmul: beta evaluation"
            assert mul(mul(ic[0], ic[1]), ic[2]) == expected, "concat matmul: sumcheck claim failed"

            def build(dims, po):  # build_point_for_input (:116-132): sub-points sorted by axis, last axis first
                by = {dims[0]: s_concat, dims[1]: s_mm, dims[2]: po}
                return [v for d in (2, 1, 0) for v in by[d]]

            made[nid] = [{"point": build(left, p_row), "eval": ic[1]}, {"point": build(right, p_col), "eval": ic[2]}]
        elif n["kind"] == "qkv":  # layers/transformer/qkv.rs:680-810
            assert len(got) == 3
            nvc = (n["ncols"]).bit_length() - 1
            pre = [e(v) for v in lp["pre_bias_evals"]]
            cols = [g["point"][:nvc] for g in got]
            rows = [g["point"][nvc:] for g in got]
            for g, pv in zip(got, pre):
                tr.append_field_elements([w for x in g["point"] for w in x])
                append_ext(tr, pv)
            coeff = [ONE, tr.read_challenge(), tr.read_challenge()]
            batched = ZERO
            for pv, c in zip(pre, coeff):
                batched = add(batched, mul(pv, c))
            sc_point = [e(v) for v in lp["sumcheck"]["point"]]
            chals, expected = L1.verify_sumcheck(batched, sc_point, lp["sumcheck"]["proofs"], (n["nrows"]).bit_length() - 1, 2, tr)
            ic = [e(v) for v in lp["individual_claims"]]
            virt = ZERO
            for w in range(3):
                virt = add(virt, mul(mul(ic[2 * w], ic[2 * w + 1]), coeff[w]))
            names_w, names_b = ["WeightQ", "WeightK", "WeightV"], ["BiasQ", "BiasK", "BiasV"]
            commons = {}
            for w in range(3):
                commons[names_w[w]] = (cols[w] + sc_point, ic[2 * w + 1])
                commons[names_b[w]] = (cols[w], sub(got[w]["eval"], pre[w]))
            for pid in sorted(commons):  # add_common_claims walks the node's BTreeMap of polynomials
                out.append(("model", nid, pid, commons[pid][0], commons[pid][1]))
            assert virt == expected, "qkv: sumcheck claim failed"
            in_claims = [{"point": sc_point + rows[w], "eval": ic[2 * w]} for w in range(3)]
            in_len = n["seq"] * n["nrows"]
            made[nid] = [same_poly_verify(in_claims, lp["aggregation_proof"], in_len.bit_length() - 1, tr)]
        elif n["kind"] in ("relu", "gelu"):  # ActivationCtx::verify_activation (layers/activation.rs:459-517)
            t = ("relu", 0) if n["kind"] == "relu" else ("gelu", (n["multiplier"], 8 + (n["multiplier"] - 1).bit_length()))
            claims, _, _ = verify_logup(lp["lookup"], 1, constant, chmap[t], tr)
            new_out = same_poly_verify([cur] + claims[1:], lp["io_accumulation"], len(cur["point"]), tr)
            # (:495-505) the commitment verifier gets `verifier_claims.claims().iter().take(1)` — the lookup's OWN claim on the first column, for a GELU the
            # column of input * multiplier — and the accumulated output claim
            out.append(("witness", nid, 0, (tuple(lp["commits"][0]["root"]), lp["commits"][0]["num_vars"]), claims[0]["point"], claims[0]["eval"]))
            out.append(("witness", nid, 1, (tuple(lp["commits"][1]["root"]), lp["commits"][1]["num_vars"]), new_out["point"], new_out["eval"]))
            # (:507-515) what goes on to the previous node: the claim itself for a Relu, the claim times 1 / multiplier for a GELU
            made[nid] = [claims[0] if n["kind"] == "relu" else {"point": claims[0]["point"], "eval": mul(claims[0]["eval"], L.ext_inv(fe(n["multiplier"])))}]
        else:
            assert n["kind"] == "requant"
            made[nid], o2 = [None], []
            ct = ("clamping", n["clamping_size"])
            shift = n["fp_scale"] + n["right_shift"]
            inst = shift // BIT_LEN
            cclaims, _, _ = verify_logup(lp["clamping_lookup"], 1, constant, chmap[ct], tr)
            sclaims, _, _ = verify_logup(lp["shifted_lookup"], inst, constant, ONE, tr)
            b = challenge(tr, b"requant_batching")
            cpt, spt = cclaims[0]["point"], sclaims[0]["point"]
            init, ch = ZERO, ONE
            for v in [cur["eval"], cclaims[1]["eval"], cclaims[0]["eval"]] + [c["eval"] for c in sclaims]:
                init = add(init, mul(ch, v))
                ch = mul(ch, b)
            acc_pt = [e(v) for v in lp["io_accumulation"]["point"]]
            chals, expected = L1.verify_sumcheck(init, acc_pt, lp["io_accumulation"]["proofs"], len(cpt), 2, tr)
            ae = [e(v) for v in lp["accumulation_evals"]]
            lb, cb, sb = eq_xy_eval(cur["point"], acc_pt), eq_xy_eval(cpt, acc_pt), eq_xy_eval(spt, acc_pt)
            calc = mul(add(lb, mul(b, cb)), ae[1])
            comb = mul(b, b)
            calc = add(calc, mul(mul(comb, cb), ae[0]))
            comb = mul(comb, b)
            for v in ae[2:]:
                calc = add(calc, mul(mul(v, sb), comb))
                comb = mul(comb, b)
            assert calc == expected, f"requant {nid}: accumulation evaluations do not recombine"
            full, pw = mul(fe(1 << shift), ae[0]), ONE
            for v in ae[2:]:
                full = add(full, mul(v, pw))
                pw = mul(pw, fe(1 << BIT_LEN))
            nxt = mul(sub(full, fe(1 << (shift - 1))), L.ext_inv(fe(n["fixed_point_multiplier"])))
            for q, (v, c) in enumerate(zip(ae, lp["commitments"])):
                out.append(("witness", nid, q, (tuple(c["root"]), c["num_vars"]), acc_pt, v))
            made[nid] = [{"point": acc_pt, "eval": nxt}]
    assert len(tree["table_proofs"]) == len(tables)
    for tp, t in zip(tree["table_proofs"], tables):
        claims, _, _ = verify_logup(tp["lookup"], 1, constant, chmap[t], tr)
        out.append(("multiplicity", t, (tuple(tp["multiplicity_commit"]["root"]), tp["multiplicity_commit"]["num_vars"]), claims[0]["point"], claims[0]["eval"]))
        expect = table_column_evals(t[0], t[1], claims[0]["point"])
        if t[0] in ("inv_sqrt", "softmax", "error", "gelu"):  # table_claims (lookup/context.rs:548-563): the claim on the committed output column goes to the opening
            out.append(("table", t, claims[-1]["point"], claims[-1]["eval"]))
            claims = claims[:-1]
        assert len(expect) == len(claims) - 1
        for cl, ex in zip(claims[1:], expect):
            assert cl["eval"] == ex, f"table {t}: claimed column evaluation is wrong"
    # the claims on the model's input tensors (iop/verifier.rs:237-262)
    for q, x in enumerate(input_tensors):
        (rd,) = readers[(-1, q)]
        c = made[rd[0]][rd[1]]
        if "one_hot_of" in c:  # Embeddings::verify_input_claim (embeddings.rs:530-571): the claim is on the one-hot encoding of the tokens
            vnv = (c["one_hot_of"]).bit_length() - 1
            assert len(c["point"]) == vnv + len(x).bit_length() - 1
            r1, r2 = c["point"][:vnv], c["point"][vnv:]
            total = ZERO
            for token, beta in zip(x, L.eq_table(r2)):
                bits = [fe((token >> b) & 1) for b in range(vnv)]
                total = add(total, mul(beta, identity_eval(r1, bits)))
            assert total == c["eval"], "one hot encoding claim is incorrect"
            continue
        assert len(c["point"]) == len(x).bit_length() - 1 and L.mle_eval([fe(v) for v in x], c["point"]) == c["eval"], f"input claim {q} is incorrect"
    fn, fd = ZERO, ONE
    for nu, de in zip(nums, dens):
        fn, fd = add(mul(fn, de), mul(nu, fd)), mul(fd, de)
    assert fn == ZERO and fd != ZERO, "final logup fraction"
    return out, tr
