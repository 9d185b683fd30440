"""GPU parity of the whole proving path (zkml::Prover::prove) through the C ABI: proof bytes identical to the oracle,
verifier acceptance, and the BASELINE.json configs."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prove_both(dev, oracle, mb, x):
    import deep_prove_amd as dpa
    blob = mb.blob()
    ctx = dpa.Context.generate(dev, blob)
    prover = dpa.Prover(ctx)
    proof, out = prover.prove(x)
    h = oracle.model_setup(blob)
    oproof, oout, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    return ctx, proof, out, oproof, oout


@pytest.mark.parametrize("num_dense,width", [(1, 8), (2, 16), (2, 64), (3, 256)])
def test_mlp_proof_bytes_identical_to_oracle(dev, oracle, num_dense, width):
    import deep_prove_amd as dpa
    mb = dpa.models.mlp(num_dense, width, config=20 + width)
    x = mb.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, mb, x)
    assert (out == oout).all() and (out == mb.run(x)).all()
    assert proof.size == oproof.size, (proof.size, oproof.size)
    diff = np.nonzero(proof != oproof)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {proof.size}"
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    ctx.free()


def test_golden_mlp_w8(dev):
    """committed fixture tests/golden/mlp_w8.npz (made by tests/golden/make_golden.py)"""
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_w8.npz"))
    ctx = dpa.Context.generate(dev, g["model_blob"])
    proof, out = dpa.Prover(ctx).prove(g["input"])
    assert (out == g["output"]).all()
    assert proof.size == g["proof"].size and (proof == g["proof"]).all()
    assert (ctx.verifier_blob() == g["verifier_blob"]).all()
    ctx.free()


def test_config1_dense128_plumbing(dev, oracle):
    """BASELINE config 1: single Dense 128 -> 128, no lookups"""
    import deep_prove_amd as dpa
    mb = dpa.models.dense_128()
    x = mb.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, mb, x)
    assert (out == oout).all() and (proof.size == oproof.size) and (proof == oproof).all()
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    ctx.free()


def test_proofs_are_deterministic_and_inputs_independent(dev):
    """two proofs of the same input are byte identical; different inputs both verify against one context (replica mode)"""
    import deep_prove_amd as dpa
    mb = dpa.models.mlp(2, 32, config=33)
    ctx = dpa.Context.generate(dev, mb.blob())
    vb = ctx.verifier_blob()
    pr = dpa.Prover(ctx)
    x0, x1 = mb.input(1000), mb.input(1001)
    p0, o0 = pr.prove(x0)
    p0b, _ = pr.prove(x0)
    p1, o1 = pr.prove(x1)
    assert (p0 == p0b).all()
    dpa.verify(vb, p0, x0, o0)
    dpa.verify(vb, p1, x1, o1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, p0, x1, o0)  # proof bound to its input
    bad = p1.copy()
    bad[bad.size // 2] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, bad, x1, o1)
    ctx.free()


def test_config2_dense4m_full_size(dev):
    """BASELINE config 2: Dense-4M (5 x 1024 MLP, 4.2 M parameters). Bit-exact vs the oracle through the committed
    sha256 of the oracle's proof stream (tests/golden/dense4m_proof.json) + verifier acceptance at full size."""
    import deep_prove_amd as dpa
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dense4m_proof.json")))
    mb = dpa.models.dense_4m()
    x = mb.input(gold["input_index"])
    ctx = dpa.Context.generate(dev, mb.blob())
    proof, out = dpa.Prover(ctx).prove(x)
    assert [int(v) for v in out] == gold["output"]
    assert proof.size == gold["proof_words"]
    assert hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"]
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    ctx.free()


def test_concurrent_batch_matches_sequential(dev):
    """dp_model_prove_batch: several proofs in flight on one GPU (own stream / arena each, shared model commitments)
    give exactly the proofs the sequential path gives"""
    import deep_prove_amd as dpa
    mb = dpa.models.mlp(2, 64, config=41)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(2000 + i) for i in range(6)])
    seq = [pr.prove(x) for x in xs]
    proofs, outs, _ = pr.prove_batch(xs, 3)
    vb = ctx.verifier_blob()
    for i in range(len(xs)):
        assert (outs[i] == seq[i][1]).all()
        assert proofs[i].size == seq[i][0].size and (proofs[i] == seq[i][0]).all()
        dpa.verify(vb, proofs[i], xs[i], outs[i])
    ctx.free()


def test_cnn_tiny_proof_bytes_identical_to_oracle(dev, oracle):
    """conv (zkCNN FFT protocol) -> requant -> relu -> maxpool (degree-5 zero check) -> flatten -> dense ...: the
    device proof stream equals the oracle's and the host verifier accepts it"""
    import deep_prove_amd as dpa
    mb = dpa.models.cnn_tiny()
    x = mb.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, mb, x)
    assert (out == oout).all() and (out == mb.run(x)).all()
    assert proof.size == oproof.size, (proof.size, oproof.size)
    diff = np.nonzero(proof != oproof)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {proof.size}"
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    bad = proof.copy()
    bad[900] ^= np.uint64(1)  # inside the convolution proof
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(ctx.verifier_blob(), bad, x, out)
    ctx.free()


def test_golden_cnn_tiny(dev):
    """committed fixture tests/golden/cnn_tiny.npz (made by tests/golden/make_golden.py)"""
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "cnn_tiny.npz"))
    ctx = dpa.Context.generate(dev, g["model_blob"])
    proof, out = dpa.Prover(ctx).prove(g["input"])
    assert (out == g["output"]).all()
    assert proof.size == g["proof"].size and (proof == g["proof"]).all()
    assert (ctx.verifier_blob() == g["verifier_blob"]).all()
    ctx.free()


def test_config3_cnn264k_full_size(dev):
    """BASELINE config 3: CNN-264k on CIFAR-10 shapes (conv 3->12, conv 12->33, fc 825->247->173->10). Bit-exact vs the
    oracle through the committed sha256 of the oracle's proof stream (tests/golden/cnn264k_proof.json) + verifier
    acceptance at full size; several proofs in flight give the same bytes."""
    import deep_prove_amd as dpa
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cnn264k_proof.json")))
    mb = dpa.models.cnn_264k()
    x = mb.input(gold["input_index"])
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    proof, out = pr.prove(x)
    assert [int(v) for v in out] == gold["output"] and (out == mb.run(x)).all()
    assert proof.size == gold["proof_words"]
    assert hashlib.sha256(proof.tobytes()).hexdigest() == gold["sha256"]
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    xs = np.stack([mb.input(3000 + i) for i in range(4)])
    proofs, outs, _ = pr.prove_batch(xs, 4)
    for i in range(len(xs)):
        assert (outs[i] == mb.run(xs[i])).all()
        dpa.verify(ctx.verifier_blob(), proofs[i], xs[i], outs[i])
    ctx.free()


def test_batch_verifier_with_device_side_merkle_paths(dev):
    """dp_verify_batch with a device: every Merkle path of a proof authenticated in one launch (k_merkle_paths), protocol checks
    on host threads; the verdicts equal the host verifier's on accepted and on tampered proofs"""
    import deep_prove_amd as dpa
    mb = dpa.models.mlp(2, 64, config=47)
    ctx = dpa.Context.generate(dev, mb.blob())
    pr = dpa.Prover(ctx)
    xs = np.stack([mb.input(8000 + i) for i in range(12)])
    proofs, outs, _ = pr.prove_batch(xs, 12)
    vb = ctx.verifier_blob()
    res, ms = dpa.verify_batch(vb, proofs, xs, outs, dev=dev)
    assert not res.any()
    tampered = [p for p in proofs]
    tampered[3] = proofs[3].copy(); tampered[3][-40] ^= np.uint64(1)          # a sibling digest of the last Merkle path
    tampered[7] = proofs[7].copy(); tampered[7][proofs[7].size // 2] ^= np.uint64(1)
    tampered[9] = proofs[9][:5000]
    wrong = outs.copy(); wrong[11, 0] += 1
    res, _ = dpa.verify_batch(vb, tampered, xs, wrong, dev=dev, threads=4)
    expect = [0] * 12
    expect[3] = -5; expect[7] = -5; expect[9] = -1; expect[11] = -5
    got = [int(v) for v in res]
    assert got[3] == -5 and got[9] == -1 and got[11] == -5 and got[7] in (-5, -1) and [g for i, g in enumerate(got) if i not in (3, 7, 9, 11)] == [0] * 8, got
    for i in (3, 11):  # the single-proof host verifier agrees
        with pytest.raises(dpa.DeepProveError):
            dpa.verify(vb, tampered[i], xs[i], wrong[i])
    ctx.free()


@pytest.mark.parametrize("seq,width,transpose,positional", [(8, 16, False, False), (16, 64, True, False), (64, 256, False, True), (32, 128, True, True)])
def test_matmul_model_proof_bytes_identical_to_oracle(dev, oracle, seq, width, transpose, positional):
    """MatMul with a constant right matrix over a [seq][features] activation (layers/matrix_mul.rs; k_fix_low on the weights,
    fix_high on the activation, the degree-2 sumcheck): proof stream == the oracle's, the verifier accepts, numpy inference agrees"""
    import deep_prove_amd as dpa
    # Config::TransposeB on the last MatMul; a positional table added to the input (Add with a static operand, layers/add.rs)
    mb = dpa.models.seq_mlp(seq, width, config=60 + seq, transpose_last=transpose, positional=positional)
    x = mb.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, mb, x)
    assert (out == oout).all() and (out == mb.run(x)).all()
    assert proof.size == oproof.size, (proof.size, oproof.size)
    diff = np.nonzero(proof != oproof)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {proof.size}"
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    bad = proof.copy()
    bad[40] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(ctx.verifier_blob(), bad, x, out)
    # throughput mode (cohorts, device-side Fiat-Shamir): every proof of a batch equals the sequential proof of its input
    xs = np.stack([mb.input(500 + i) for i in range(8)])
    proofs, outs, _ = dpa.Prover(ctx).prove_batch(xs, 8)
    single, sout = dpa.Prover(ctx).prove(xs[5])
    assert proofs[5].size == single.size and (proofs[5] == single).all() and (outs[5] == sout).all()
    ctx.free()


def test_golden_seq_mlp(dev):
    """committed fixture tests/golden/seq_mlp.npz (made by tests/golden/make_seq_golden.py)"""
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "seq_mlp.npz"))
    ctx = dpa.Context.generate(dev, g["model_blob"])
    proof, out = dpa.Prover(ctx).prove(g["input"])
    assert (out == g["output"]).all()
    assert proof.size == g["proof"].size and (proof == g["proof"]).all()
    assert (ctx.verifier_blob() == g["verifier_blob"]).all()
    ctx.free()


@pytest.mark.parametrize("seq,vocab,width,max_positions", [(16, 50, 32, 0), (64, 1000, 256, 0), (16, 50, 32, 100), (32, 300, 128, 32)])
def test_token_model_proof_bytes_identical_to_oracle(dev, oracle, seq, vocab, width, max_positions):
    """tokens -> Embeddings -> + positional table -> MatMul blocks (models.token_mlp; layers/transformer/embeddings.rs, layers/add.rs,
    layers/matrix_mul.rs): proof stream == the oracle's, verifier accepts (one-hot input claim included), batch proofs == sequential"""
    import deep_prove_amd as dpa
    mb = dpa.models.token_mlp(seq, vocab, width, config=70 + seq, max_positions=max_positions)  # (> 0: Positional::Learned instead of the static Add)
    x = mb.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, mb, x)
    assert (out == oout).all() and (out == mb.run(x)).all()
    assert proof.size == oproof.size and (proof == oproof).all()
    dpa.verify(ctx.verifier_blob(), proof, x, out)
    other = x.copy(); other[1] = (other[1] + 1) % vocab
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(ctx.verifier_blob(), proof, other, out)
    xs = np.stack([mb.input(900 + i) for i in range(8)])
    proofs, outs, _ = dpa.Prover(ctx).prove_batch(xs, 8)
    single, sout = dpa.Prover(ctx).prove(xs[6])
    assert proofs[6].size == single.size and (proofs[6] == single).all() and (outs[6] == sout).all()
    ctx.free()


def _graph_cases():
    with open(os.path.join(ROOT, "tests", "golden", "graph_models.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", range(11))
def test_graph_model_proof_bytes_identical_to_oracle_and_golden(dev, oracle, case):
    """Models that are GRAPHS (layers/provable/mod.rs:195-565): QKV (three outputs, one batched sumcheck + same_poly), ConcatMatMul (per-head
    products, degree-3 sumcheck), MatMul and Add of two inputs, several input / output tensors; cases 5 / 6: LayerNorm (two lookups, one of
    them into the inverse-square-root table whose output column is a commitment of the context, a degree-4 sumcheck); 7 / 8: Softmax (four
    lookups — exponential, range, error, zero tables — the causal mask, row shifts computed in f32); 9 / 10: the attention half of a pre-LN
    transformer block, thirteen nodes from LayerNorm to the residual Add (models.transformer_block). The device proof equals the oracle's and the
    committed sha256 (tests/golden/graph_models.json); the verifier accepts it and refuses a flipped word and a wrong input tensor; proofs of
    a batch (cohorts, device-side Fiat-Shamir) equal the sequential ones."""
    import deep_prove_amd as dpa
    c = _graph_cases()[case]
    g = getattr(dpa.models, c["model"])(**c["args"])
    x = g.input()
    ctx, proof, out, oproof, oout = prove_both(dev, oracle, g, x)
    assert (out == oout).all() and (out == g.run(x)).all()
    assert proof.size == oproof.size, (proof.size, oproof.size)
    diff = np.nonzero(proof != oproof)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {proof.size}"
    assert hashlib.sha256(proof.tobytes()).hexdigest() == c["proof_sha256"]
    vb = ctx.verifier_blob()
    dpa.verify(vb, proof, x, out)
    for at in (3, 40, 90):
        bad = proof.copy(); bad[at] ^= np.uint64(1)
        with pytest.raises(dpa.DeepProveError):
            dpa.verify(vb, bad, x, out)
    other = x.copy(); other[-1] += 1   # the LAST input tensor: its own claim is checked against it
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, proof, other, out)
    xs = np.stack([g.input(300 + i) for i in range(6)])
    proofs, outs, _ = dpa.Prover(ctx).prove_batch(xs, 6)
    single, sout = dpa.Prover(ctx).prove(xs[4])
    assert proofs[4].size == single.size and (proofs[4] == single).all() and (outs[4] == sout).all()
    assert (outs[4] == g.run(xs[4])).all()
    ctx.free()


def test_device_proofs_pass_the_independent_iop_verifier(dev, oracle):
    """tests/support/l2_independent.py — the IOP part of the reference's verifier written a second time in Python (own field, Poseidon2,
    transcript, sumcheck and logup verifiers) — follows DEVICE proofs (one sequential, one out of a throughput-mode batch) through the
    whole transcript, and every claim it would hand to the commitment verifier is true for the polynomial itself"""
    import deep_prove_amd as dpa
    from deep_prove_amd import wire
    from support import l0_independent as L, l2_independent as V
    sys_path_tests = os.path.join(ROOT, "tests")
    import sys
    if sys_path_tests not in sys.path:
        sys.path.insert(0, sys_path_tests)
    import test_l2_independent as T
    P = T.P
    mb = dpa.models.mlp(2, 16, config=44)
    ctx = dpa.Context.generate(dev, mb.blob())
    xs = np.stack([mb.input(700 + i) for i in range(6)])
    proofs, outs, _ = dpa.Prover(ctx).prove_batch(xs, 6)
    single, sout = dpa.Prover(ctx).prove(xs[0])
    to_words = lambda v: np.asarray([int(t) % P for t in np.asarray(v).reshape(-1)], dtype=np.uint64)
    for proof, x, out in ((single, xs[0], sout), (proofs[3], xs[3], outs[3])):
        layers, cols, lookups, y = T._layers_and_witness(mb, x)
        assert (out == y).all()
        sizes = [mb.input_len, 256] + [1 << l["clamping_size"] for l in layers if l["kind"] == "requant"] + [l["weights"].size for l in mb.layers if l["kind"] == 0] + [c[0].size for c in cols.values()]
        max_poly = 1 << (max(sizes) - 1).bit_length()
        roots = {node: [("DenseBias", oracle.pcs_commit_root(max_poly, to_words(l["bias"]), False)), ("DenseWeight", oracle.pcs_commit_root(max_poly, to_words(l["weights"]), False))]
                 for node, l in enumerate(mb.layers) if l["kind"] == 0}
        claims, _ = V.verify_chain(layers, roots, wire.parse_stream(proof), [int(v) for v in x], [int(v) for v in y])
        fe = lambda v: (int(v) % P, 0)
        for c in claims:
            if c[0] == "model":
                poly = mb.layers[c[1]]["weights"].reshape(-1) if c[2] == "DenseWeight" else mb.layers[c[1]]["bias"]
                assert L.mle_eval([fe(v) for v in poly], c[3]) == c[4]
            elif c[0] == "witness":
                assert L.mle_eval([fe(v) for v in cols[c[1]][c[2]]], c[4]) == c[5]
    ctx.free()
