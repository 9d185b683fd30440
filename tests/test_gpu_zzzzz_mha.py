"""The reference's Mha layer as ONE node (layers/transformer/mha.rs:633-724: final_mul, softmax, qk under one node id, one MhaProof) on the device:
golden cases 11 / 12 of tests/golden/graph_models.json (models.mha_block: LayerNorm -> QKV -> Mha -> projection -> residual) and 13
(models.transformer_layer: a whole pre-LN layer, attention and feed-forward half, as one graph of 19 nodes). Same checks as the
other graph cases (tests/test_gpu_model.py): bytes equal to the oracle's and to the committed sha256, verifier verdicts, batch = sequential.
(Its own file, sorted last: the node kind has not run on hardware before this round's end.)"""
import pytest

from test_gpu_model import test_graph_model_proof_bytes_identical_to_oracle_and_golden as _graph_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [11, 12, 13])
def test_mha_block_proof_bytes_identical_to_oracle_and_golden(dev, oracle, case):
    _graph_case(dev, oracle, case)


def test_transformer_layer_at_the_benched_size_has_the_oracles_sha256(dev):
    """Golden case 14: the whole transformer layer at the size bench.py times (64 tokens x 256 features, 4 heads of 64, ffn 1024, config 66; 1.29 M proof
    words). The oracle needs ~16 s for it, so the GPU box compares against the committed sha256 of the oracle's stream (tests/golden/graph_models.json,
    made by make_graph_golden.py): the latency-mode proof, and the same input proved inside a batch (cohorts, device-side Fiat-Shamir, fused tails) must
    both have it; the verifier accepts the proof and refuses a flipped word."""
    import hashlib
    import json
    import os
    import numpy as np
    import deep_prove_amd as dpa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "graph_models.json")) as f:
        c = json.load(f)[14]
    assert c["model"] == "transformer_layer" and c.get("gpu_only") and c["args"]["seq"] == 64 and c["args"]["emb"] == 256
    g = dpa.models.transformer_layer(**c["args"])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    x = g.input()
    assert sha(g.blob()) == c["blob_sha256"] and sha(x) == c["input_sha256"]
    ctx = dpa.Context.generate(dev, g.blob())
    pr = dpa.Prover(ctx)
    proof, out = pr.prove(x)
    assert sha(out) == c["output_sha256"] and (out == g.run(x)).all()
    assert proof.size == c["proof_words"] and sha(proof) == c["proof_sha256"]
    vb = ctx.verifier_blob()
    dpa.verify(vb, proof, x, out)
    bad = proof.copy(); bad[proof.size // 2] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, bad, x, out)
    xs = np.stack([g.input(500 + i) for i in range(8)])
    xs[5] = x
    proofs, outs, _ = pr.prove_batch(xs, 8)
    assert proofs[5].size == c["proof_words"] and sha(proofs[5]) == c["proof_sha256"] and sha(outs[5]) == c["output_sha256"]
    v, _ = dpa.verify_batch(vb, proofs, xs, outs, dev=dev)
    assert not v.any()
    ctx.free()
