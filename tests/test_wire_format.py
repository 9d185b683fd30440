"""deep_prove_amd/wire.py: the reference's `rmp_serde::to_vec_named(&Proof)` wire format rebuilt from the canonical stream.
Unverifiable against the reference here (no Rust toolchain, no reference-serialised proof exists: SURVEY.md 8f2); what these
tests pin is that the bytes are well-formed MessagePack an independent decoder (the `msgpack` package) reads into the structure
the reference's serde derives describe (names and order from the cited files), that encode -> decode reproduces the stream up to
the one thing the reference's own format drops, and that the recalled conventions are switches, not assumptions baked in."""
import os

import msgpack
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["mlp_w8.npz", "cnn_tiny.npz"])
def test_named_messagepack_structure_and_round_trip(name):
    from deep_prove_amd import wire
    p = np.load(os.path.join(ROOT, "tests", "golden", name))["proof"]
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)  # an independent MessagePack decoder accepts every byte
    assert list(m) == ["steps", "table_proofs", "commit"]                       # zkml/src/iop/mod.rs:26-31
    assert list(m["commit"]) == ["batch_proof", "trivial_proofs"]               # commit/context.rs:230-231
    bp = m["commit"]["batch_proof"]
    assert list(bp) == ["sumcheck_messages", "roots", "final_message", "query_result_with_merkle_path", "sumcheck_proof", "trivial_proof"]  # structure.rs:339-344
    (variant, q), = bp["query_result_with_merkle_path"].items()
    assert variant == "Batched" and len(q["inner"]) == 200                      # 200 queries, each (index, BatchedSingleQueryResultWithMerklePath)
    idx, one = q["inner"][0]
    assert isinstance(idx, int) and list(one) == ["oracle_query", "commitments_query"]
    cq = one["commitments_query"]["inner"][0]
    assert list(cq) == ["query", "merkle_path"] and list(cq["query"]) == ["codepoints", "index"] and list(cq["merkle_path"]) == ["inner", "_phantom"]
    assert cq["merkle_path"]["_phantom"] == [] and list(bp["sumcheck_proof"]) == ["rounds", "phantom"]
    assert list(bp["sumcheck_proof"]["rounds"][0]) == ["Ext"] and len(bp["sumcheck_proof"]["rounds"][0]["Ext"]) == 3  # Coefficients(FieldType::Ext)
    for node, lp in m["steps"].items():
        (kind, body), = lp.items()
        assert kind in ("Dense", "Requant", "Activation", "Convolution", "Pooling")
        if kind == "Dense":
            assert list(body) == ["sumcheck", "bias_eval", "individual_claims"]                                   # dense.rs:57-68
            assert list(body["sumcheck"]) == ["point", "proofs"] and list(body["sumcheck"]["proofs"][0]) == ["evaluations"]
            assert list(body["bias_eval"]) == ["value"] and list(body["bias_eval"]["value"][0]) == ["value"]    # Ext2 {value: [Goldilocks {value}; 2]}
        if kind == "Requant":
            assert list(body) == ["io_accumulation", "accumulation_evals", "clamping_lookup", "shifted_lookup", "commitments"]  # requant.rs:84-99
            assert list(body["commitments"][0]) == ["root", "num_vars", "is_base", "num_polys"] and len(body["commitments"][0]["root"]) == 4
            assert body["clamping_lookup"]["proof_type"] == "Lookup"
        if kind == "Convolution":
            assert list(body)[:4] == ["fft_proof", "fft_proof_weights", "fft_delegation_proof", "fft_delegation_proof_weights"] and list(body)[-1] == "clearing_proof"
    assert all(t["lookup"]["proof_type"] == "Table" for t in m["table_proofs"])
    assert list(m["steps"]) == sorted(m["steps"])  # HashMap order is arbitrary in the reference; ascending NodeId here
    # trivial openings: BasefoldProof::trivial -> Single(empty), no sumcheck proof, and FieldType::Base travels WITHOUT its data
    for t in m["commit"]["trivial_proofs"]:
        assert t["query_result_with_merkle_path"] == {"Single": {"inner": []}} and t["sumcheck_proof"] is None and t["trivial_proof"] == ["Base"]
    back = wire.from_rmp(data)
    assert np.array_equal(back, wire.strip_skipped(p))
    assert back.size < p.size and wire.stream_equal_modulo_skipped(back, p)    # (the skipped tables are the only difference)
    assert wire.to_rmp(back) == data                                           # encode(decode(x)) == x


def test_recalled_conventions_are_switches():
    from deep_prove_amd import wire
    p = np.load(os.path.join(ROOT, "tests", "golden", "mlp_w8.npz"))["proof"]

    class Bare(wire.Conventions):
        field_as_map = False
        ext_as_map = False
        phantom_is_empty_array = False
    a, b = wire.to_rmp(p), wire.to_rmp(p, Bare)
    assert len(b) < 0.7 * len(a)
    m = msgpack.unpackb(b, raw=False, strict_map_key=False)
    assert isinstance(m["steps"][0]["Dense"]["bias_eval"], list) and m["commit"]["batch_proof"]["sumcheck_proof"]["phantom"] is None
    assert np.array_equal(wire.from_rmp(b, Bare), wire.from_rmp(a))


def test_integers_use_the_smallest_encoding():
    from deep_prove_amd import wire
    for v, n in ((0, 1), (127, 1), (128, 2), (255, 2), (256, 3), (65535, 3), (65536, 5), (2**32 - 1, 5), (2**32, 9), (2**64 - 2**32, 9)):
        out = []
        wire._pack(v, out, wire.Conventions)
        assert len(b"".join(out)) == n and msgpack.unpackb(b"".join(out)) == v
        assert b"".join(out) == msgpack.packb(v)


def test_matmul_step_in_the_wire_format():
    """LayerProof::MatMul(MatMulProof {sumcheck, individual_claims, bias_eval: Option<E>}) (layers/mod.rs:135, matrix_mul.rs:153-161):
    Some / None for a layer with / without bias, round trip through the named MessagePack encoding"""
    from deep_prove_amd import wire
    p = np.load(os.path.join(ROOT, "tests", "golden", "seq_mlp.npz"))["proof"]
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)
    mm = [body for lp in m["steps"].values() for kind, body in lp.items() if kind == "MatMul"]
    assert len(mm) == 3 and all(list(b) == ["sumcheck", "individual_claims", "bias_eval"] for b in mm)
    assert sum(b["bias_eval"] is None for b in mm) == 1 and all(len(b["individual_claims"]) == 2 for b in mm)
    back = wire.from_rmp(data)
    assert wire.stream_equal_modulo_skipped(back, p) and wire.to_rmp(back) == data


def test_add_step_in_the_wire_format(oracle):
    """LayerProof::Add(AddProof {left_eval, right_eval}) (layers/add.rs:59-63): a model with a positional Add proved by the oracle,
    encoded, decoded by an independent MessagePack reader, and back"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    mb = dpa.models.seq_mlp(8, 16, config=62, positional=True)
    x = mb.input()
    h = oracle.model_setup(mb.blob())
    p, out, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (out == mb.run(x)).all()
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)
    adds = [body for lp in m["steps"].values() for kind, body in lp.items() if kind == "Add"]
    assert len(adds) == 1 and list(adds[0]) == ["left_eval", "right_eval"]
    back = wire.from_rmp(data)
    assert wire.stream_equal_modulo_skipped(back, p) and wire.to_rmp(back) == data


def test_standalone_pcs_proofs_all_three_query_variants(oracle):
    """ProofQueriesResultWithMerklePath::{Single, Batched, SimpleBatched} (mpcs/src/basefold/structure.rs:292-300) for proofs that leave the PCS
    entry points on their own: names and nesting as the serde derives give them (query_phase.rs:556-566, 1328-1383, 1419-1440, 1545-1551),
    read back by an independent MessagePack decoder, and encode -> decode reproduces the stream"""
    from deep_prove_amd import wire
    P = 0xFFFFFFFF00000001
    rng = np.random.default_rng(12)
    re = lambda: (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    for nv, ext, k in ((9, False, 3), (8, True, 2), (9, False, 1), (5, True, 3)):
        polys = [rng.integers(0, P, size=(2 if ext else 1) << nv, dtype=np.uint64) for _ in range(k)]
        pt = [re() for _ in range(nv)]
        _, proof = oracle.pcs_simple_batch_open(1 << 10, polys, ext, pt, oracle.transcript(b"t"))
        data = wire.pcs_proof_to_rmp(proof, simple_batch=True)
        m = msgpack.unpackb(data, raw=False, strict_map_key=False)
        assert list(m) == ["sumcheck_messages", "roots", "final_message", "query_result_with_merkle_path", "sumcheck_proof", "trivial_proof"]
        (variant, q), = m["query_result_with_merkle_path"].items()
        if nv <= 7:  # BasefoldProof::trivial: no queries, the k tables travel
            assert variant == "Single" and q["inner"] == [] and len(m["trivial_proof"]) == k and m["sumcheck_proof"] is None
        else:
            assert variant == "SimpleBatched" and len(q["inner"]) == 200 and m["sumcheck_proof"] is None
            idx, one = q["inner"][0]
            assert list(one) == ["oracle_query", "commitment_query"] and list(one["commitment_query"]) == ["query", "merkle_path"]
            assert list(one["commitment_query"]["query"]) == ["leaves", "index"]
            (ft, pairs), = one["commitment_query"]["query"]["leaves"].items()
            assert ft == ("Ext" if ext else "Base") and len(pairs) == k and all(len(pr) == 2 for pr in pairs)
            assert len(one["commitment_query"]["merkle_path"]["inner"]) == nv + 1 - 1  # log2(codeword) - 1 digests: the row pair's sibling is not sent
        back = wire.pcs_proof_from_rmp(data)
        assert back.size == proof.size and (back == proof).all()
    # Single (PCS::open) and Batched (PCS::batch_open over an Evaluation list)
    w = rng.integers(0, P, size=1 << 9, dtype=np.uint64)
    pt = [re() for _ in range(9)]
    single = oracle.pcs_open(1 << 10, w, False, pt, oracle.transcript(b"t"))
    m = msgpack.unpackb(wire.pcs_proof_to_rmp(single), raw=False, strict_map_key=False)
    assert list(m["query_result_with_merkle_path"]) == ["Single"] and list(m["query_result_with_merkle_path"]["Single"]["inner"][0][1]) == ["oracle_query", "commitment_query"]
    assert (wire.pcs_proof_from_rmp(wire.pcs_proof_to_rmp(single)) == single).all()
    w2 = rng.integers(0, P, size=1 << 9, dtype=np.uint64)
    evals = [(0, 0, oracle.mle_eval(w, False, pt)), (1, 0, oracle.mle_eval(w2, False, pt))]
    batched = oracle.pcs_batch_open_evals(1 << 10, [w, w2], [False, False], [pt], evals, oracle.transcript(b"t"))
    m = msgpack.unpackb(wire.pcs_proof_to_rmp(batched), raw=False, strict_map_key=False)
    assert list(m["query_result_with_merkle_path"]) == ["Batched"] and len(m["sumcheck_proof"]["rounds"]) == 9
    assert (wire.pcs_proof_from_rmp(wire.pcs_proof_to_rmp(batched)) == batched).all()


def test_graph_steps_in_the_wire_format(oracle):
    """LayerProof::QKV(QKVProof {sumcheck, aggregation_proof, pre_bias_evals, individual_claims: [(E, E); 3]}) (transformer/qkv.rs:63-83),
    LayerProof::ConcatMatMul(ConcatMatMulProof {sumcheck_proof, individual_claims}) (concat_matmul.rs:365-373), and the two-input MatMul / Add
    in the variants they share with the constant forms: an attention block proved by the oracle, encoded, read by an independent MessagePack
    decoder, and back (MatMul / Add of two inputs come back as the reference's single variants: kinds 10 / 11 -> 6 / 7)"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    g = dpa.models.attention_block(8, 16, 2, 8, config=61)
    x = g.input()
    h = oracle.model_setup(g.blob())
    p, out, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)
    by_kind = {}
    for lp in m["steps"].values():
        for kind, body in lp.items():
            by_kind.setdefault(kind, []).append(body)
    assert len(by_kind["QKV"]) == 1 and list(by_kind["QKV"][0]) == ["sumcheck", "aggregation_proof", "pre_bias_evals", "individual_claims"]
    q = by_kind["QKV"][0]
    assert len(q["individual_claims"]) == 3 and all(len(pair) == 2 for pair in q["individual_claims"]) and len(q["pre_bias_evals"]) == 3
    assert list(q["aggregation_proof"]) == ["sumcheck", "evals"]
    assert len(by_kind["ConcatMatMul"]) == 2 and all(list(b) == ["sumcheck_proof", "individual_claims"] and len(b["individual_claims"]) == 3 for b in by_kind["ConcatMatMul"])
    assert len(by_kind["Add"]) == 1 and len(by_kind["MatMul"]) == 1 and len(by_kind["Requant"]) == 6
    back = wire.from_rmp(data)
    assert wire.to_rmp(back) == data
    norm = p.copy()  # the canonical stream with kind 11 (Add of two inputs) renamed to 7, as it comes back
    tree = wire.parse_stream(p)
    assert sorted(k for _, k, _ in tree["steps"]).count(11) == 1
    assert wire.parse_stream(back)["steps"][-1][1] == 7


def test_mha_step_in_the_wire_format(oracle):
    """LayerProof::Mha(MhaProof {final_mul_proof: ConcatMatMulProof, softmax_proof: SoftmaxProof, qk_proof: ConcatMatMulProof})
    (layers/transformer/mha.rs:122-128): a block with the Mha layer as one node proved by the oracle, encoded, read by an independent MessagePack
    decoder (field names and order), and back"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    g = dpa.models.mha_block(8, 16, 2, 8, config=97)
    x = g.input()
    h = oracle.model_setup(g.blob())
    p, out, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (out == g.run(x)).all()
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)
    mha = [body for lp in m["steps"].values() for kind, body in lp.items() if kind == "Mha"]
    assert len(mha) == 1 and list(mha[0]) == ["final_mul_proof", "softmax_proof", "qk_proof"]
    assert list(mha[0]["final_mul_proof"]) == ["sumcheck_proof", "individual_claims"] and list(mha[0]["qk_proof"]) == ["sumcheck_proof", "individual_claims"]
    assert list(mha[0]["softmax_proof"]) == ["logup_proofs", "commitments", "accumulation_proof", "mask_proof", "evaluations"]
    assert len(mha[0]["softmax_proof"]["logup_proofs"]) in (3, 4) and len(mha[0]["final_mul_proof"]["individual_claims"]) == 3
    assert not any(kind in ("ConcatMatMul", "Softmax") for lp in m["steps"].values() for kind in lp)
    back = wire.from_rmp(data)
    assert wire.to_rmp(back) == data
    assert [k for _, k, _ in wire.parse_stream(back)["steps"]].count(16) == 1


def test_gelu_step_in_the_wire_format(oracle):
    """LayerProof::Activation(ActivationProof {io_accumulation, lookup, commits}) (layers/activation.rs:60-76) carries a GELU exactly as it carries a Relu:
    the canonical stream's kind 17 becomes the variant "Activation", and comes back as kind 2 (the wire format cannot tell them apart, as for MatMul / Add of
    two inputs); the MessagePack bytes survive the round trip"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    mb = dpa.models.gelu_mlp(64, config=114)
    x = mb.input()
    oracle.set_gelu_files_lookup_claim(True)
    try:
        h = oracle.model_setup(mb.blob())
        p, out, _ = oracle.model_prove(h, x)
        oracle.model_free(h)
    finally:
        oracle.set_gelu_files_lookup_claim(False)
    assert (out == mb.run(x)).all()
    assert [k for _, k, _ in wire.parse_stream(p)["steps"]].count(17) == 1
    data = wire.to_rmp(p)
    m = msgpack.unpackb(data, raw=False, strict_map_key=False)
    act = [body for lp in m["steps"].values() for kind, body in lp.items() if kind == "Activation"]
    assert len(act) == 1 and list(act[0]) == ["io_accumulation", "lookup", "commits"] and len(act[0]["commits"]) == 2
    back = wire.from_rmp(data)
    assert wire.to_rmp(back) == data
    kinds, kinds_back = [k for _, k, _ in wire.parse_stream(p)["steps"]], [k for _, k, _ in wire.parse_stream(back)["steps"]]
    assert kinds_back == [2 if k == 17 else k for k in kinds] and kinds_back.count(2) == 1  # (base-field trivial openings come back empty, as for every model: not compared word for word)

