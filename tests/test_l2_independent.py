"""The IOP part of the verifier a SECOND time (tests/support/l2_independent.py: Python, from the reference's verifier sources, on the
independent field / Poseidon2 / transcript of l0_independent.py): it follows the oracle's proof of an MLP through the whole transcript —
model commitments, table challenges, every layer's sumcheck / logup-GKR / accumulation proof, the table proofs, the input claim, the final
fraction sum — and every claim it would hand to the commitment verifier is then checked against the polynomial itself: the weights and
biases, the witness columns recomputed here from a numpy inference (Requant: clamping input / output and the range-check chunks; ReLU:
input / output), the table multiplicities. A misreading of the protocol that the oracle shares with the product's own verifier would have
to be repeated in this unrelated formulation to go unnoticed."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def _layers_and_witness(mb, x):
    """the model as the independent verifier wants it, and every committed witness column from a plain numpy forward pass"""
    from deep_prove_amd import models as M
    layers, cols, cur = [], {}, np.asarray(x, dtype=np.int64)
    lookups = {"relu": [], "range": [], "clamping": {}}
    for node, l in enumerate(mb.layers):
        if l["kind"] == M.L_DENSE:
            layers.append(dict(kind="dense", nrows=l["nrows"], ncols=l["ncols"]))
            cur = l["weights"] @ cur + l["bias"]
        elif l["kind"] == M.L_REQUANT:
            shift = l["fp_scale"] + l["right_shift"]
            cs = l["intermediate_bit_size"] + int(l["fixed_point_multiplier"] - 1).bit_length() - shift
            layers.append(dict(kind="requant", clamping_size=cs, **{k: l[k] for k in ("fp_scale", "right_shift", "fixed_point_multiplier")}))
            tmp = cur * l["fixed_point_multiplier"] + (1 << (shift - 1))
            cin = tmp >> shift
            cout = np.clip(cin, -127, 127)
            masked = tmp & ((1 << shift) - 1)
            chunks = [(masked >> (8 * j)) & 255 for j in range(shift // 8)]
            cols[node] = [cin, cout] + chunks
            lookups["clamping"].setdefault(cs, []).extend(int(v) for v in cin)
            for c in chunks:
                lookups["range"].extend(int(v) for v in c)
            cur = cout
        else:
            assert l["kind"] == M.L_RELU
            layers.append(dict(kind="relu"))
            out = np.maximum(cur, 0)
            cols[node] = [cur, out]
            lookups["relu"].extend(int(v) for v in cur)
            cur = out
    return layers, cols, lookups, cur


@pytest.mark.parametrize("width,config", [(8, 41), (16, 42)])
def test_independent_iop_verifier_accepts_the_oracles_proof_and_every_claim_is_true(oracle, width, config):
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l0_independent as L, l2_independent as V
    mb = dpa.models.mlp(2, width, config=config)
    x = mb.input()
    layers, cols, lookups, y = _layers_and_witness(mb, x)
    h = oracle.model_setup(mb.blob())
    proof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (oout == y).all()
    tree = wire.parse_stream(proof)
    # model commitments (their roots enter the transcript first): the PCS parameters follow the largest committed polynomial
    sizes = [mb.input_len] + [t for l in layers if l["kind"] == "requant" for t in (256, 1 << l["clamping_size"])] + [256]
    for l in mb.layers:
        if l["kind"] == 0:
            sizes += [l["weights"].size, l["bias"].size]
    for node, l in enumerate(mb.layers):
        if l["kind"] in (1, 2):
            sizes.append(cols[node][0].size)
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {}
    for node, l in enumerate(mb.layers):
        if l["kind"] == 0:
            roots[node] = [("DenseBias", oracle.pcs_commit_root(max_poly, to_words(l["bias"]), False)), ("DenseWeight", oracle.pcs_commit_root(max_poly, to_words(l["weights"]), False))]
    claims, tr = V.verify_chain(layers, roots, tree, [int(v) for v in x], [int(v) for v in y])
    # ---- every claim against the polynomial itself
    fe = lambda v: (int(v) % P, 0)
    n_model = n_wit = n_mult = 0
    for c in claims:
        if c[0] == "model":
            _, node, pid, point, ev = c
            poly = mb.layers[node]["weights"].reshape(-1) if pid == "DenseWeight" else mb.layers[node]["bias"]
            assert L.mle_eval([fe(v) for v in poly], point) == ev, f"model claim {node} {pid}"
            n_model += 1
        elif c[0] == "witness":
            _, node, k, comm, point, ev = c
            assert L.mle_eval([fe(v) for v in cols[node][k]], point) == ev, f"witness claim of node {node}, column {k}"
            n_wit += 1
        else:
            _, table, comm, point, ev = c
            if table[0] == "relu":
                lo, hi, data = -128, 128, lookups["relu"]
            elif table[0] == "range":
                lo, hi, data = 0, 256, lookups["range"]
            else:
                lo, hi, data = -(1 << (table[1] - 1)), 1 << (table[1] - 1), lookups["clamping"][table[1]]
            mult = [0] * (hi - lo)
            for v in data:
                mult[v - lo] += 1
            assert L.mle_eval([fe(v) for v in mult], point) == ev, f"multiplicity claim of table {table}"
            n_mult += 1
    assert n_model == 2 * sum(l["kind"] == "dense" for l in layers) and n_wit >= 4 and n_mult >= 3
    # ---- and the openings, by the independent Basefold verifier (l3_independent.py), on the same transcript
    from support import l3_independent as V3
    root_of = {(node, pid): r for node, lst in roots.items() for pid, r in lst}
    uniform = []
    for c in claims:
        if c[0] == "model":
            poly = mb.layers[c[1]]["weights"] if c[2] == "DenseWeight" else mb.layers[c[1]]["bias"]
            uniform.append(({"root": root_of[(c[1], c[2])], "num_vars": int(poly.size).bit_length() - 1}, c[3], c[4]))
        elif c[0] == "witness":
            uniform.append(({"root": list(c[3][0]), "num_vars": c[3][1]}, c[4], c[5]))
        else:
            uniform.append(({"root": list(c[2][0]), "num_vars": c[2][1]}, c[3], c[4]))
    trivial = [u for u in uniform if len(u[1]) <= V3.BASECODE_LOG]
    batch = [u for u in uniform if len(u[1]) > V3.BASECODE_LOG]
    assert len(trivial) == len(tree["trivial_proofs"])
    for (comm, point, ev), tp in zip(trivial, tree["trivial_proofs"]):
        V3.trivial_verify(comm, point, ev, tp)
    assert len(batch) >= 3 and len(trivial) >= 4  # (weights and multiplicity polynomials are opened by Basefold, the short columns trivially)
    V3.batch_verify(max_poly.bit_length() - 1, [u[0] for u in batch], [u[1] for u in batch], [u[2] for u in batch], tree["batch_proof"], tr)


def test_independent_iop_verifier_rejects_tampered_proofs(oracle):
    """flipped words in the layer / table proofs: the independent verifier refuses every one of them"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l2_independent as V
    mb = dpa.models.mlp(1, 8, config=43)
    x = mb.input()
    layers, cols, lookups, y = _layers_and_witness(mb, x)
    h = oracle.model_setup(mb.blob())
    proof, _, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    sizes = [mb.input_len, 256] + [1 << l["clamping_size"] for l in layers if l["kind"] == "requant"] + [l["weights"].size for l in mb.layers if l["kind"] == 0]
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {node: [("DenseBias", oracle.pcs_commit_root(max_poly, to_words(l["bias"]), False)), ("DenseWeight", oracle.pcs_commit_root(max_poly, to_words(l["weights"]), False))]
             for node, l in enumerate(mb.layers) if l["kind"] == 0}
    V.verify_chain(layers, roots, wire.parse_stream(proof), [int(v) for v in x], [int(v) for v in y])
    tree0 = wire.parse_stream(proof)

    def without_commitments(tree):
        steps = [(n, k, {f: v for f, v in lp.items() if f not in ("commitments", "commits")}) for n, k, lp in tree["steps"]]
        return steps, [tp["lookup"] for tp in tree["table_proofs"]]

    n_iop = proof.size - _opening_words(tree0)
    rejected = through = 0
    for at in list(range(2, 160, 3)) + list(range(160, n_iop, max(1, n_iop // 80))):
        bad = proof.copy()
        bad[at] ^= np.uint64(1)
        try:
            t1 = wire.parse_stream(bad)
            V.verify_chain(layers, roots, t1, [int(v) for v in x], [int(v) for v in y])
        except (AssertionError, ValueError, IndexError, KeyError, ZeroDivisionError, OverflowError, MemoryError):
            rejected += 1
            continue
        # accepted: the flipped word must belong to a commitment (root / num_vars / is_base), which only the opening consumes
        assert without_commitments(t1) == without_commitments(tree0), f"word {at}: a flipped protocol word was accepted"
        through += 1
    assert rejected > 100 and through < rejected // 5, (rejected, through)


def _opening_words(tree):
    """words of the batch opening + trivial openings at the end of the canonical stream"""
    def bf_words(p):
        w = 1 + sum(1 + 2 * len(m) for m in p["sumcheck_messages"]) + 1 + 4 * len(p["roots"]) + 1 + 2 * len(p["final_message"]) + 1
        for q in p["queries"]:
            w += 3
            for c in q["oracle_query"] + q["commitments_query"]:
                w += 1 + (4 if c["is_ext"] else 2) + 1 + 1 + 4 * len(c["path"])
        w += 1 + sum(1 + 2 * len(m) for m in p["sumcheck_proof"]) + 1
        for t in p["trivial_proof"]:
            w += 2 + len(t["w"])
        return w
    return bf_words(tree["batch_proof"]) + 1 + sum(bf_words(t) for t in tree["trivial_proofs"])


def test_independent_basefold_verifier_rejects_tampered_openings(oracle):
    """single-word flips inside the batch opening and the trivial openings (sumcheck coefficients, commit-phase messages, roots, the final
    message, query indices, opened pairs, path digests): the independent verifier (l2 + l3) refuses every one"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l2_independent as V, l3_independent as V3
    mb = dpa.models.mlp(1, 16, config=45)
    x = mb.input()
    layers, cols, lookups, y = _layers_and_witness(mb, x)
    h = oracle.model_setup(mb.blob())
    proof, _, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    sizes = [mb.input_len, 256] + [1 << l["clamping_size"] for l in layers if l["kind"] == "requant"] + [l["weights"].size for l in mb.layers if l["kind"] == 0]
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {node: [("DenseBias", oracle.pcs_commit_root(max_poly, to_words(l["bias"]), False)), ("DenseWeight", oracle.pcs_commit_root(max_poly, to_words(l["weights"]), False))]
             for node, l in enumerate(mb.layers) if l["kind"] == 0}
    root_of = {(node, pid): r for node, lst in roots.items() for pid, r in lst}

    def full_verify(words):
        tree = wire.parse_stream(words)
        claims, tr = V.verify_chain(layers, roots, tree, [int(v) for v in x], [int(v) for v in y])
        uniform = []
        for c in claims:
            if c[0] == "model":
                poly = mb.layers[c[1]]["weights"] if c[2] == "DenseWeight" else mb.layers[c[1]]["bias"]
                uniform.append(({"root": root_of[(c[1], c[2])], "num_vars": int(poly.size).bit_length() - 1}, c[3], c[4]))
            else:
                comm = c[3] if c[0] == "witness" else c[2]
                uniform.append(({"root": list(comm[0]), "num_vars": comm[1]}, c[-2], c[-1]))
        trivial = [u for u in uniform if len(u[1]) <= V3.BASECODE_LOG]
        batch = [u for u in uniform if len(u[1]) > V3.BASECODE_LOG]
        assert len(trivial) == len(tree["trivial_proofs"])
        for (comm, point, ev), tp in zip(trivial, tree["trivial_proofs"]):
            V3.trivial_verify(comm, point, ev, tp)
        V3.batch_verify(max_poly.bit_length() - 1, [u[0] for u in batch], [u[1] for u in batch], [u[2] for u in batch], tree["batch_proof"], tr)

    full_verify(proof)
    n_open = _opening_words(wire.parse_stream(proof))
    start = proof.size - n_open
    rng = np.random.default_rng(7)
    # the head of the opening (commit-phase messages, roots, final message), somewhere in the queries, the classic sumcheck near the end, a trivial opening
    spots = [start + 3, start + 40, start + 300] + [int(v) for v in rng.integers(start + 400, proof.size - 600, size=3)] + [proof.size - 5]
    for at in spots:
        bad = proof.copy()
        bad[at] ^= np.uint64(1)
        with pytest.raises((AssertionError, ValueError, IndexError, KeyError, ZeroDivisionError, OverflowError, MemoryError)):
            full_verify(bad)


def _graph_description(g, x):
    """models.GraphBuilder -> the node list of l2_independent.verify_graph, the per-node witness columns of the Requant nodes (numpy), the
    model's input tensors and output tensors"""
    from deep_prove_amd import models as M
    offs = np.concatenate([[0], np.cumsum(g.input_lens)])
    x = np.asarray(x, dtype=np.int64)
    vals, nodes, cols, clamp_lookups, range_lookups, relu_lookups = {}, [], {}, {}, [], []

    def get(e):
        return x[offs[e[1]]:offs[e[1] + 1]] if e[0] < 0 else vals[e]

    for i, (l, edges) in enumerate(g.nodes):
        a = get(edges[0])
        k = l["kind"]
        d = dict(inputs=[tuple(e) for e in edges], n_out=1)
        if k == M.L_QKV:
            d.update(kind="qkv", n_out=3, nrows=l["nrows"], ncols=l["ncols"], seq=a.size // l["nrows"])
            xs = a.reshape(-1, l["nrows"])
            for w in range(3):
                vals[(i, w)] = (xs @ l["weights"][w] + l["bias"][w]).reshape(-1)
        elif k == M.L_MATMUL2:
            d.update(kind="matmul2", nrows=l["nrows"], ncols=l["ncols"], transpose_b=l["transpose_b"])
            b = get(edges[1])
            bm = b.reshape(l["ncols"], l["nrows"]).T if l["transpose_b"] else b.reshape(l["nrows"], l["ncols"])
            vals[(i, 0)] = (a.reshape(-1, l["nrows"]) @ bm).reshape(-1)
        elif k == M.L_ADD2:
            d.update(kind="add2", left=l["left"], right=l["right"])
            vals[(i, 0)] = l["left"] * a + l["right"] * get(edges[1])
        elif k == M.L_CONCAT_MATMUL:
            d.update(kind="concat_matmul", a_shape=l["a_shape"], b_shape=l["b_shape"], left=l["left"], right=l["right"], perm=l["perm"])
            b = get(edges[1])
            ta = a.reshape(l["a_shape"]).transpose(l["left"][0], l["left"][2], l["left"][1])
            tb = b.reshape(l["b_shape"]).transpose(l["right"][0], l["right"][1], l["right"][2])
            r = np.einsum("crm,cmn->crn", ta, tb)
            vals[(i, 0)] = (r if l["perm"] is None else r.transpose(l["perm"])).reshape(-1)
        elif k == M.L_MATMUL:
            d.update(kind="matmul", nrows=l["nrows"], ncols=l["ncols"], bias=l["bias"], transpose_b=False)
            y = a.reshape(-1, l["nrows"]) @ l["weights"]
            vals[(i, 0)] = (y + l["bias"] if l["bias"] is not None else y).reshape(-1)
        elif k == M.L_REQUANT:
            shift = l["fp_scale"] + l["right_shift"]
            cs = l["intermediate_bit_size"] + int(l["fixed_point_multiplier"] - 1).bit_length() - shift
            d.update(kind="requant", clamping_size=cs, **{q: l[q] for q in ("fp_scale", "right_shift", "fixed_point_multiplier")})
            tmp = a * l["fixed_point_multiplier"] + (1 << (shift - 1))
            cin = tmp >> shift
            cout = np.clip(cin, -127, 127)
            masked = tmp & ((1 << shift) - 1)
            chunks = [(masked >> (8 * j)) & 255 for j in range(shift // 8)]
            cols[i] = [cin, cout] + chunks
            clamp_lookups.setdefault(cs, []).extend(int(v) for v in cin)
            for c in chunks:
                range_lookups.extend(int(v) for v in c)
            vals[(i, 0)] = cout
        elif k == M.L_RELU:
            d.update(kind="relu")
            cols[i] = [a, np.maximum(a, 0)]
            relu_lookups.extend(int(v) for v in a)
            vals[(i, 0)] = np.maximum(a, 0)
        else:
            raise AssertionError("unexpected node kind")
        nodes.append(d)
    outs = g.outputs if g.outputs is not None else [(len(g.nodes) - 1, 0)]
    inputs = [[int(v) for v in x[offs[q]:offs[q + 1]]] for q in range(len(g.input_lens))]
    return nodes, [tuple(o) for o in outs], cols, (clamp_lookups, range_lookups, relu_lookups), inputs, [[int(v) for v in vals[tuple(o)]] for o in outs], vals


@pytest.mark.parametrize("name,kw", [("attention_block", dict(seq=8, emb=16, heads=2, head_dim=8, config=61)), ("matmul_pair", dict(seq=8, k=32, n=16, config=63, transpose_b=True)),
                                     ("qkv_two_outputs", dict(seq=4, k=16, n=16, config=64))])
def test_independent_verifier_on_graph_models(oracle, name, kw):
    """the graph layers a second time (l2_independent.verify_graph: QKV, ConcatMatMul, MatMul with two inputs or a constant matrix, Add of two
    inputs, Requant; the backward node order, one claim per output tensor, the claims on every input tensor) plus the openings (l3): the
    oracle's proofs of models.attention_block / matmul_pair / qkv_two_outputs are accepted, and every claim on a model polynomial or a
    witness column is true for the polynomial itself"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l0_independent as L, l2_independent as V, l3_independent as V3
    g = getattr(dpa.models, name)(**kw)
    x = g.input()
    nodes, outs, cols, (clamp_lookups, range_lookups, relu_lookups), inputs, outputs, vals = _graph_description(g, x)
    assert (np.concatenate([np.asarray(o) for o in outputs]) == g.run(x)).all()
    h = oracle.model_setup(g.blob())
    proof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (oout == g.run(x)).all()
    tree = wire.parse_stream(proof)
    # model polynomials and the size of the PCS parameters
    polys = {}
    for i, (l, _) in enumerate(g.nodes):
        if l["kind"] == 13:
            for w, (wn, bn) in enumerate((("WeightQ", "BiasQ"), ("WeightK", "BiasK"), ("WeightV", "BiasV"))):
                polys[(i, wn)] = l["weights"][w].reshape(-1)
                polys[(i, bn)] = l["bias"][w].reshape(-1)
        elif l["kind"] == 6:
            polys[(i, "MatMulWeight")] = l["weights"].reshape(-1)
            if l["bias"] is not None:
                polys[(i, "MatMulBias")] = l["bias"].reshape(-1)
    sizes = list(g.input_lens) + [p.size for p in polys.values()] + [c[0].size for c in cols.values()]
    for n in nodes:
        if n["kind"] == "requant":
            sizes += [256, 1 << n["clamping_size"]]
        elif n["kind"] == "relu":
            sizes.append(256)
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {}
    for (i, pid), poly in polys.items():
        roots.setdefault(i, []).append((pid, oracle.pcs_commit_root(max_poly, to_words(poly), False)))
    claims, tr = V.verify_graph(nodes, outs, roots, tree, inputs, outputs)
    fe = lambda v: (int(v) % P, 0)
    root_of = {(node, pid): r for node, lst in roots.items() for pid, r in lst}
    uniform = []
    for c in claims:
        if c[0] == "model":
            assert L.mle_eval([fe(v) for v in polys[(c[1], c[2])]], c[3]) == c[4], f"model claim {c[1]} {c[2]}"
            uniform.append(({"root": root_of[(c[1], c[2])], "num_vars": int(polys[(c[1], c[2])].size).bit_length() - 1}, c[3], c[4]))
        elif c[0] == "witness":
            assert L.mle_eval([fe(v) for v in cols[c[1]][c[2]]], c[4]) == c[5], f"witness claim of node {c[1]}, column {c[2]}"
            uniform.append(({"root": list(c[3][0]), "num_vars": c[3][1]}, c[4], c[5]))
        else:
            t = c[1]
            lo, hi, data = (0, 256, range_lookups) if t[0] == "range" else (-128, 128, relu_lookups) if t[0] == "relu" else (-(1 << (t[1] - 1)), 1 << (t[1] - 1), clamp_lookups[t[1]])
            mult = [0] * (hi - lo)
            for v in data:
                mult[v - lo] += 1
            assert L.mle_eval([fe(v) for v in mult], c[3]) == c[4], f"multiplicity claim of table {t}"
            uniform.append(({"root": list(c[2][0]), "num_vars": c[2][1]}, c[3], c[4]))
    trivial = [u for u in uniform if len(u[1]) <= V3.BASECODE_LOG]
    batch = [u for u in uniform if len(u[1]) > V3.BASECODE_LOG]
    assert len(trivial) == len(tree["trivial_proofs"])
    for (comm, point, ev), tp in zip(trivial, tree["trivial_proofs"]):
        V3.trivial_verify(comm, point, ev, tp)
    if batch:
        V3.batch_verify(max_poly.bit_length() - 1, [u[0] for u in batch], [u[1] for u in batch], [u[2] for u in batch], tree["batch_proof"], tr, check_every=5)
    else:
        assert not tree["batch_proof"]["queries"]


def _chain_description(mb, x):
    """models.ModelBuilder (a chain) -> the node list of l2_independent.verify_graph, model polynomials, witness columns"""
    from deep_prove_amd import models as M
    cur = np.asarray(x, dtype=np.int64)
    nodes, cols, polys = [], {}, {}
    lk = {"clamp": {}, "range": [], "relu": []}
    for i, l in enumerate(mb.layers):
        d = dict(inputs=[(i - 1, 0)], n_out=1)
        k = l["kind"]
        if k == M.L_DENSE:
            d.update(kind="dense", nrows=l["nrows"], ncols=l["ncols"])
            polys[(i, "DenseWeight")], polys[(i, "DenseBias")] = l["weights"].reshape(-1), l["bias"]
            cur = l["weights"] @ cur + l["bias"]
        elif k == M.L_MATMUL:
            d.update(kind="matmul", nrows=l["nrows"], ncols=l["ncols"], bias=l["bias"], transpose_b=l["transpose_b"])
            polys[(i, "MatMulWeight")] = l["weights"].reshape(-1)
            if l["bias"] is not None:
                polys[(i, "MatMulBias")] = l["bias"]
            y = cur.reshape(-1, l["nrows"]) @ (l["weights"].T if l["transpose_b"] else l["weights"])
            cur = (y + l["bias"] if l["bias"] is not None else y).reshape(-1)
        elif k == M.L_EMBED:
            d.update(kind="embeddings", nrows=l["nrows"], ncols=l["ncols"])
            polys[(i, "EmbeddingMat")] = l["table"].reshape(-1)
            cur = l["table"][cur].reshape(-1)
        elif k == M.L_POSITIONAL:
            d.update(kind="positional", left=l["left"], right=l["right"], table_vars=int(l["table"].size).bit_length() - 1)
            polys[(i, "PositionalMatrix")] = l["table"].reshape(-1)
            cur = l["left"] * cur + l["right"] * l["table"].reshape(-1)[:cur.size]
        elif k == M.L_ADD:
            d.update(kind="add_const", left=l["left"], right=l["right"])
            polys[(i, "255")] = l["operand"]
            cur = l["left"] * cur + l["right"] * l["operand"]
        elif k == M.L_REQUANT:
            shift = l["fp_scale"] + l["right_shift"]
            cs = l["intermediate_bit_size"] + int(l["fixed_point_multiplier"] - 1).bit_length() - shift
            d.update(kind="requant", clamping_size=cs, **{q: l[q] for q in ("fp_scale", "right_shift", "fixed_point_multiplier")})
            tmp = cur * l["fixed_point_multiplier"] + (1 << (shift - 1))
            cin = tmp >> shift
            cout = np.clip(cin, -127, 127)
            masked = tmp & ((1 << shift) - 1)
            chunks = [(masked >> (8 * j)) & 255 for j in range(shift // 8)]
            cols[i] = [cin, cout] + chunks
            lk["clamp"].setdefault(cs, []).extend(int(v) for v in cin)
            for c in chunks:
                lk["range"].extend(int(v) for v in c)
            cur = cout
        elif k == M.L_CONV:
            d.update(kind="conv", **{q: l[q] for q in ("kw", "kx", "real_nw", "nw", "unp_out")})
            polys[(i, "ConvFilter")], polys[(i, "ConvBias")] = l["filter"].reshape(-1), l["bias"]
            kk, nw = l["kernel"], l["nw"]
            xs = cur.reshape(l["kx"], nw, nw)
            oc, oh, ow = l["unp_out"]
            o = np.zeros((l["kw"], nw, nw), dtype=np.int64)
            for a in range(kk):
                for b in range(kk):
                    o[:, :oh, :ow] += np.einsum("oc,cyx->oyx", l["filter"][:, :, a, b], xs[:, a:a + oh, b:b + ow])
            o += l["bias"][:, None, None]
            o[oc:], o[:, oh:], o[:, :, ow:] = 0, 0, 0
            cur = o.reshape(-1)
        elif k == M.L_MAXPOOL:
            d.update(kind="maxpool", pin=l["pin"])
            c, h, w_ = l["pin"]
            t = cur.reshape(c, h // 2, 2, w_ // 2, 2)
            o = t.max(axis=(2, 4))
            # the four committed difference polynomials (pooling.rs:660-735): (dy, dx) = (0,0), (1,0), (0,1), (1,1), then the output
            cols[i] = [(o - t[:, :, dy, :, dx]).reshape(-1) for dy, dx in ((0, 0), (1, 0), (0, 1), (1, 1))] + [o.reshape(-1)]
            for q in cols[i][:4]:
                lk["range"].extend(int(v) for v in q)
            cur = o.reshape(-1)
        elif k == M.L_FLATTEN:
            d.update(kind="reshape")
        elif k == M.L_SOFTMAX:
            d.update(kind="softmax", **{q: l[q] for q in ("shape", "scalar", "temp_bits", "table_size", "bkm", "zero_chunks", "zero_vars", "allowable_error")})
            t = {}
            o = M.softmax_apply(l, cur, trace=t)
            A = lambda v: np.asarray(v, dtype=np.int64)
            cols[i] = [A(t["exp_in"]), A(t["exp_out"]), A(t["low"]), A(t["high"]), A(t["shift"])] + [A(c) for z in range(l["zero_chunks"]) for c in (t["zero_in"][z], t["zero_out"][z])]
            lk["range"].extend(t["low"] + t["high"])
            lk.setdefault("softmax", {}).setdefault((l["temp_bits"], l["table_size"], l["bkm"]), []).extend(t["exp_in"])
            lk.setdefault("error", {}).setdefault(l["allowable_error"], []).extend(int(v) for v in o.reshape(-1, l["shape"][2]).sum(axis=1))
            for z in range(l["zero_chunks"]):
                lk.setdefault("zero", {}).setdefault(l["zero_vars"], []).extend(t["zero_in"][z])
            cur = o
        elif k == M.L_LAYERNORM:
            d.update(kind="layernorm", **{q: l[q] for q in ("dim_size", "multiplier", "eps_bits", "range_check_bits", "top_chunk_scalar_log")})
            polys[(i, "LayerNormGamma")], polys[(i, "LayerNormBeta")] = l["gamma"], l["beta"]
            o, (lin, inv, rc) = M.layernorm_apply(l, cur)
            nrc = (l["range_check_bits"] - 1) // 8 + 1
            chunks = [((rc >> (8 * j)) & 255) * ((1 << l["top_chunk_scalar_log"]) if j == nrc - 1 else 1) for j in range(nrc)]
            cols[i] = [lin, inv] + chunks
            lk.setdefault("inv_sqrt", {}).setdefault((l["eps_bits"], l["range_check_bits"]), []).extend(int(v) for v in lin)
            for c in chunks:
                lk["range"].extend(int(v) for v in c)
            cur = o
        elif k == M.L_GELU:
            d.update(kind="gelu", multiplier=l["multiplier"])
            o = M.gelu_apply(l, cur)
            cols[i] = [cur * l["multiplier"], o]  # (activation.rs:268-276: the committed first column is the SCALED input)
            lk.setdefault("gelu", {}).setdefault((l["multiplier"], 8 + (l["multiplier"] - 1).bit_length()), []).extend(int(v) * l["multiplier"] for v in cur)
            cur = o
        else:
            assert k == M.L_RELU
            d.update(kind="relu")
            cols[i] = [cur, np.maximum(cur, 0)]
            lk["relu"].extend(int(v) for v in cur)
            cur = np.maximum(cur, 0)
        nodes.append(d)
    return nodes, cols, polys, lk, cur


@pytest.mark.parametrize("name,args,kw", [("token_mlp", (8, 20, 16), dict(config=73, max_positions=30)), ("token_mlp", (8, 20, 16), dict(config=74)),
                                           ("seq_mlp", (8, 16), dict(config=75, transpose_last=True, positional=True)), ("cnn_tiny", (), dict(config=76)),
                                           ("layernorm_mlp", (8, 12, 16), dict(config=83)), ("softmax_only", (2, 8), dict(config=84)),
                                           ("softmax_only", (2, 8), dict(config=85, in_scale=8.0 / 127.0))])
def test_independent_verifier_on_token_and_sequence_models(oracle, name, args, kw):
    """Embeddings (tokens in: the input claim is a claim on the one-hot encoding), Positional::Learned with a table longer than the sequence
    (the slice claim lifted to the table), Add with a static operand, MatMul with a constant matrix (plain and TransposeB), Requant, ReLU —
    the oracle's proofs through the independent verifier (l2 + l3), every model / witness / multiplicity claim checked on the polynomial"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l0_independent as L, l2_independent as V, l3_independent as V3
    mb = getattr(dpa.models, name)(*args, **kw)
    x = mb.input()
    nodes, cols, polys, lk, y = _chain_description(mb, x)
    assert (y == mb.run(x)).all()
    h = oracle.model_setup(mb.blob())
    proof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (oout == y).all()
    tree = wire.parse_stream(proof)
    sizes = [mb.input_len] + [p.size for p in polys.values()] + [c[0].size for c in cols.values()]
    for n in nodes:
        if n["kind"] == "requant":
            sizes += [256, 1 << n["clamping_size"]]
        elif n["kind"] in ("relu", "maxpool"):
            sizes.append(256)
        elif n["kind"] == "gelu":
            sizes.append(1 << (8 + (n["multiplier"] - 1).bit_length()))
        elif n["kind"] == "layernorm":
            sizes += [256, 1 << 15]
        elif n["kind"] == "softmax":
            sizes += [256, 1 << n["table_size"]]
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {}
    for (i, pid), poly in polys.items():
        roots.setdefault(i, []).append((pid, oracle.pcs_commit_root(max_poly, to_words(poly), False)))
    claims, tr = V.verify_graph(nodes, [(len(nodes) - 1, 0)], roots, tree, [[int(v) for v in x]], [[int(v) for v in y]])
    _check_claims_and_openings(oracle, claims, tr, tree, roots, polys, cols, lk, max_poly)


def _gelu_through_the_independent_verifier(oracle, mb, files_lookup_claim):
    import deep_prove_amd as dpa  # noqa: F401
    from deep_prove_amd import wire
    from support import l2_independent as V
    x = mb.input()
    nodes, cols, polys, lk, y = _chain_description(mb, x)
    assert (y == mb.run(x)).all()
    oracle.set_gelu_files_lookup_claim(files_lookup_claim)
    try:
        h = oracle.model_setup(mb.blob())
        proof, oout, _ = oracle.model_prove(h, x)
        oracle.model_free(h)
    finally:
        oracle.set_gelu_files_lookup_claim(False)
    assert (oout == y).all()
    tree = wire.parse_stream(proof)
    sizes = [mb.input_len] + [p.size for p in polys.values()] + [c[0].size for c in cols.values()]
    for n in nodes:
        if n["kind"] == "requant":
            sizes += [256, 1 << n["clamping_size"]]
        elif n["kind"] == "gelu":
            sizes.append(1 << (8 + (n["multiplier"] - 1).bit_length()))
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {}
    for (i, pid), poly in polys.items():
        roots.setdefault(i, []).append((pid, oracle.pcs_commit_root(max_poly, to_words(poly), False)))
    claims, tr = V.verify_graph(nodes, [(len(nodes) - 1, 0)], roots, tree, [[int(v) for v in x]], [[int(v) for v in y]])
    _check_claims_and_openings(oracle, claims, tr, tree, roots, polys, cols, lk, max_poly)


def test_independent_verifier_on_gelu_models(oracle):
    """Activation::Gelu a second time (l2_independent.verify_graph, from layers/activation.rs:459-517 and lookup/context.rs:364-378, 466, 548-563): the
    table (multiplier, size) in derive(Ord) order with the label "GELU", its input column evaluated by the verifier and its output column opened against a
    commitment made HERE from the numpy table (models.gelu_table_output), the lookup's claim on the scaled column filed with the first witness commitment,
    the claim / multiplier handed on. gelu_only(32) — the reference's own test shape, columns opened by showing them — is accepted with the oracle TO THE
    LETTER of the reference's prover; gelu_mlp(256) — a real batch opening (l3) — is accepted when the prover files the claim this verifier files."""
    import deep_prove_amd as dpa
    _gelu_through_the_independent_verifier(oracle, dpa.models.gelu_only(32, config=111), files_lookup_claim=False)
    _gelu_through_the_independent_verifier(oracle, dpa.models.gelu_mlp(256, config=112), files_lookup_claim=True)


def test_independent_verifier_refuses_the_references_gelu_prover_beyond_trivial_openings(oracle):
    """... and refuses gelu_mlp(256) proved to the letter of the reference (prove_step files the lookup claim DIVIDED by the multiplier with the commitment
    of the scaled column, activation.rs:405-430): every IOP check passes — the independent verifier gets as far as the openings — and the batch opening
    (l3_independent.batch_verify), which starts from the claims the VERIFIER holds, fails"""
    import deep_prove_amd as dpa
    with pytest.raises(AssertionError, match="batch opening: sumcheck round inconsistent"):
        _gelu_through_the_independent_verifier(oracle, dpa.models.gelu_mlp(256, config=112), files_lookup_claim=False)


def _check_claims_and_openings(oracle, claims, tr, tree, roots, polys, cols, lk, max_poly):
    """every claim the IOP verifier ends with — on a model polynomial, a witness column, the committed column of a table, a multiplicity
    polynomial — is true for the polynomial itself, and the openings (l3: trivial + batch) are accepted"""
    from support import l0_independent as L, l3_independent as V3
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    fe = lambda v: (int(v) % P, 0)
    root_of = {(node, pid): r for node, lst in roots.items() for pid, r in lst}
    uniform = []
    for c in claims:
        if c[0] == "model":
            assert L.mle_eval([fe(v) for v in polys[(c[1], c[2])]], c[3]) == c[4], f"model claim {c[1]} {c[2]}"
            uniform.append(({"root": root_of[(c[1], c[2])], "num_vars": int(polys[(c[1], c[2])].size).bit_length() - 1}, c[3], c[4]))
        elif c[0] == "witness":
            assert L.mle_eval([fe(v) for v in cols[c[1]][c[2]]], c[4]) == c[5], f"witness claim of node {c[1]}, column {c[2]}"
            uniform.append(({"root": list(c[3][0]), "num_vars": c[3][1]}, c[4], c[5]))
        elif c[0] == "table":  # the committed column of a table (inverse square root, exponential, error): the claim is true, and it is opened against a commitment made HERE
            from deep_prove_amd import models as M
            if c[1][0] == "gelu":
                column = np.asarray([M.gelu_table_output(j) for j in range(-(1 << (c[1][1][1] - 1)), 1 << (c[1][1][1] - 1))], dtype=np.int64)
            elif c[1][0] == "softmax":
                column = np.asarray([M.softmax_table_output(dict(temp_bits=c[1][1][0], bkm=c[1][1][2]), j) for j in range(1 << c[1][1][1])], dtype=np.int64)
            elif c[1][0] == "error":
                nt = 1 << (2 * c[1][1] - 1).bit_length()
                column = np.asarray((list(range(4096 - c[1][1], 4096 + c[1][1] + 1))[:nt] + [0] * nt)[:nt], dtype=np.int64)
            else:
                column = M.inv_sqrt_table_output(c[1][1][0], c[1][1][1], np.arange(-(1 << 14), 1 << 14))
            assert L.mle_eval([fe(v) for v in column], c[2]) == c[3], f"claim on the column of table {c[1]}"
            uniform.append(({"root": oracle.pcs_commit_root(max_poly, to_words(column), False), "num_vars": int(column.size).bit_length() - 1}, c[2], c[3]))
        else:
            t = c[1]
            if t[0] == "gelu":
                lo, hi, data = -(1 << (t[1][1] - 1)), 1 << (t[1][1] - 1), lk["gelu"][t[1]]
            elif t[0] == "softmax":
                lo, hi, data = 0, 1 << t[1][1], lk["softmax"][t[1]]
            elif t[0] == "zero":
                lo, hi, data = 0, 1 << t[1], lk["zero"][t[1]]
            elif t[0] == "error":
                lo, hi, data = 4096 - t[1], 4096 - t[1] + (1 << (2 * t[1] - 1).bit_length()), lk["error"][t[1]]
            else:
                lo, hi, data = (0, 256, lk["range"]) if t[0] == "range" else (-128, 128, lk["relu"]) if t[0] == "relu" else (-(1 << 14), 1 << 14, lk["inv_sqrt"][t[1]]) if t[0] == "inv_sqrt" else (-(1 << (t[1] - 1)), 1 << (t[1] - 1), lk["clamp"][t[1]])
            mult = [0] * (hi - lo)
            for v in data:
                mult[v - lo] += 1
            assert L.mle_eval([fe(v) for v in mult], c[3]) == c[4], f"multiplicity claim of table {t}"
            uniform.append(({"root": list(c[2][0]), "num_vars": c[2][1]}, c[3], c[4]))
    trivial = [u for u in uniform if len(u[1]) <= V3.BASECODE_LOG]
    batch = [u for u in uniform if len(u[1]) > V3.BASECODE_LOG]
    assert len(trivial) == len(tree["trivial_proofs"])
    for (comm, point, ev), tp in zip(trivial, tree["trivial_proofs"]):
        V3.trivial_verify(comm, point, ev, tp)
    V3.batch_verify(max_poly.bit_length() - 1, [u[0] for u in batch], [u[1] for u in batch], [u[2] for u in batch], tree["batch_proof"], tr, check_every=5)


def test_independent_verifier_on_the_mha_node(oracle):
    """Mha as ONE node (layers/transformer/mha.rs:792-893) a second time: l2_independent.verify_graph walks it as final_mul.verify ->
    softmax.verify -> qk.verify with the sub-layers Mha::new builds, one MhaProof, the claims on Q, K, V checked against the three input tensors;
    the softmax's witness columns, its tables' committed columns and multiplicities are checked on the polynomials; then the openings (l3)"""
    import deep_prove_amd as dpa
    from deep_prove_amd import models as M, wire
    from support import l2_independent as V
    S, H, D = 8, 2, 4
    g = dpa.models.GraphBuilder([S * H * D] * 3, config=99)
    g.mha((-1, 0), (-1, 1), (-1, 2), S, H, D, 1.0 / 256.0, 127 * 127 * D)
    x = g.input(amplitude=15)
    y = g.run(x)
    l = g.nodes[0][0]
    qh, kh = (x[q * S * H * D:(q + 1) * S * H * D].reshape(S, H, D).transpose(1, 0, 2) for q in (0, 1))
    t = {}
    probs = M.softmax_apply(dict(l, shape=(H, S, S)), np.einsum("hsd,htd->hst", qh, kh).reshape(-1), trace=t)
    A = lambda v: np.asarray(v, dtype=np.int64)
    cols = {0: [A(t["exp_in"]), A(t["exp_out"]), A(t["low"]), A(t["high"]), A(t["shift"])] + [A(c) for z in range(l["zero_chunks"]) for c in (t["zero_in"][z], t["zero_out"][z])]}
    lk = {"range": t["low"] + t["high"], "relu": [], "clamp": {}, "softmax": {(l["temp_bits"], l["table_size"], l["bkm"]): list(t["exp_in"])},
          "error": {l["allowable_error"]: [int(v) for v in probs.reshape(-1, S).sum(axis=1)]}, "zero": {l["zero_vars"]: [v for z in range(l["zero_chunks"]) for v in t["zero_in"][z]]}}
    nodes = [dict(kind="mha", inputs=[(-1, 0), (-1, 1), (-1, 2)], n_out=1, **{q: l[q] for q in ("shape", "scalar", "temp_bits", "table_size", "bkm", "zero_chunks", "zero_vars", "allowable_error")})]
    h = oracle.model_setup(g.blob())
    proof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (oout == y).all()
    tree = wire.parse_stream(proof)
    assert [k for _, k, _ in tree["steps"]] == [16]
    max_poly = 1 << (max(S * H * D, H * S * S, 256, 1 << l["table_size"]) - 1).bit_length()
    ins = [[int(v) for v in x[q * S * H * D:(q + 1) * S * H * D]] for q in range(3)]
    claims, tr = V.verify_graph(nodes, [(0, 0)], {}, tree, ins, [[int(v) for v in y]])
    _check_claims_and_openings(oracle, claims, tr, tree, {}, {}, cols, lk, max_poly)
    # a flipped word in each of the three sub-proofs (final_mul's sumcheck, a softmax evaluation far into the proof, qk's claims at its end)
    n_steps = sum(1 for _ in tree["steps"])
    assert n_steps == 1
    for at in (8, 40):
        bad = proof.copy()
        bad[at] ^= np.uint64(1)
        with pytest.raises((AssertionError, ValueError, IndexError, KeyError, ZeroDivisionError, OverflowError)):
            V.verify_graph(nodes, [(0, 0)], {}, wire.parse_stream(bad), ins, [[int(v) for v in y]])


def test_independent_verifier_rejects_tampered_cnn_proofs(oracle):
    """flipped words anywhere in the IOP part of a CNN proof (Convolution: the clearing Hadamard product, the iFFT / FFT sumchecks and their
    delegation chains, the Hadamard sumcheck, partial evaluations; Pooling: logup, zerocheck, evaluations): refused, unless the word belongs
    to a commitment, which only the opening consumes"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l2_independent as V
    mb = dpa.models.cnn_tiny(config=77)
    x = mb.input()
    nodes, cols, polys, lk, y = _chain_description(mb, x)
    h = oracle.model_setup(mb.blob())
    proof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (oout == y).all()
    sizes = [mb.input_len, 256] + [p.size for p in polys.values()] + [c[0].size for c in cols.values()] + [1 << n["clamping_size"] for n in nodes if n["kind"] == "requant"]
    max_poly = 1 << (max(sizes) - 1).bit_length()
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    roots = {}
    for (i, pid), poly in polys.items():
        roots.setdefault(i, []).append((pid, oracle.pcs_commit_root(max_poly, to_words(poly), False)))
    run = lambda tree: V.verify_graph(nodes, [(len(nodes) - 1, 0)], roots, tree, [[int(v) for v in x]], [[int(v) for v in y]])
    tree0 = wire.parse_stream(proof)
    run(tree0)

    def without_commitments(tree):
        return [(n, k, {f: v for f, v in lp.items() if f not in ("commitments", "commits")}) for n, k, lp in tree["steps"]], [tp["lookup"] for tp in tree["table_proofs"]]

    n_iop = proof.size - _opening_words(tree0)
    rejected = through = 0
    for at in range(2, n_iop, max(1, n_iop // 90)):
        bad = proof.copy()
        bad[at] ^= np.uint64(1)
        try:
            t1 = wire.parse_stream(bad)
            run(t1)
        except (AssertionError, ValueError, IndexError, KeyError, ZeroDivisionError, OverflowError, MemoryError):
            rejected += 1
            continue
        assert without_commitments(t1) == without_commitments(tree0), f"word {at}: a flipped protocol word was accepted"
        through += 1
    assert rejected > 60 and through < rejected // 5, (rejected, through)
