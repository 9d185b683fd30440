"""Activation::Gelu on the device (zkml/src/layers/activation.rs:238-318, 385-517; the GELU table of lookup/context.rs:163-182, 364-378, 495-503):
golden cases 15 - 17 of tests/golden/graph_models.json. Case 15 is the reference's own proving test (one GELU over a small tensor), where the
oracle follows the reference's prover to the letter; in 16 / 17 the committed column has a real opening and the oracle files the claim the
reference's VERIFIER checks (tests/test_hostlogic.py::test_gelu_to_the_letter_of_the_reference_only_verifies_when_the_column_is_shown)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    with open(os.path.join(ROOT, "tests", "golden", "graph_models.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", [15, 16, 17])
def test_gelu_model_proof_bytes_identical_to_oracle_and_golden(dev, oracle, case):
    """the device proof (latency mode) equals the oracle's word for word and has the committed sha256; the verifier accepts it, refuses a flipped word
    and a wrong input; the same input proved inside a batch (cohorts, device-side Fiat-Shamir, fused tails) gives the same words"""
    import deep_prove_amd as dpa
    c = _cases()[case]
    g = getattr(dpa.models, c["model"])(**c["args"])
    x, blob = g.input(), g.blob()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert sha(blob) == c["blob_sha256"] and sha(x) == c["input_sha256"]
    ctx = dpa.Context.generate(dev, blob)
    pr = dpa.Prover(ctx)
    proof, out = pr.prove(x)
    oracle.set_gelu_files_lookup_claim(c.get("oracle_gelu_claim") != "reference")
    try:
        h = oracle.model_setup(blob)
        oproof, oout, _ = oracle.model_prove(h, x)
        oracle.model_free(h)
    finally:
        oracle.set_gelu_files_lookup_claim(False)
    assert (out == oout).all() and (out == g.run(x)).all() and sha(out) == c["output_sha256"]
    assert proof.size == oproof.size == c["proof_words"]
    diff = np.nonzero(proof != oproof)[0]
    assert diff.size == 0, f"first differing word {diff[:5]} of {proof.size}"
    assert sha(proof) == c["proof_sha256"]
    vb = ctx.verifier_blob()
    dpa.verify(vb, proof, x, out)
    for at in (3, 40, 90, proof.size // 2):
        bad = proof.copy(); bad[at] ^= np.uint64(1)
        with pytest.raises(dpa.DeepProveError):
            dpa.verify(vb, bad, x, out)
    other = x.copy(); other[-1] += 1
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, proof, other, out)
    xs = np.stack([g.input(300 + i) for i in range(6)])
    xs[2] = x
    proofs, outs, _ = pr.prove_batch(xs, 6)
    assert proofs[2].size == proof.size and (proofs[2] == proof).all() and (outs[2] == out).all()
    single, sout = pr.prove(xs[4])
    assert proofs[4].size == single.size and (proofs[4] == single).all() and (outs[4] == sout).all() and (outs[4] == g.run(xs[4])).all()
    v, _ = dpa.verify_batch(vb, proofs, xs, outs, dev=dev)
    assert not v.any()
    ctx.free()


def test_gelu_reference_letter_switch_on_the_device(dev, oracle, monkeypatch):
    """DP_GELU_REFERENCE_LETTER=1 (csrc/zkml.h prove_relu): the library emits the reference PROVER's bytes — the descaled claim filed with the scaled column's
    commitment (activation.rs:419-430). Case 16 (a 2^8-entry column): word for word the oracle's letter-mode stream, in latency mode and inside a batch; the verifier
    refuses it (as it would the reference's own proof), and accepts the default-mode proof of the same input."""
    import deep_prove_amd as dpa
    c = _cases()[16]
    g = getattr(dpa.models, c["model"])(**c["args"])
    x, blob = g.input(), g.blob()
    ctx = dpa.Context.generate(dev, blob)
    pr = dpa.Prover(ctx)
    default_proof, out = pr.prove(x)
    monkeypatch.setenv("DP_GELU_REFERENCE_LETTER", "1")
    proof, out2 = pr.prove(x)
    xs = np.stack([x] + [g.input(300 + i) for i in range(3)])
    proofs, _, _ = pr.prove_batch(xs, 4)
    monkeypatch.delenv("DP_GELU_REFERENCE_LETTER")
    oracle.set_gelu_files_lookup_claim(False)  # (the oracle's default: to the letter of the reference's prover)
    h = oracle.model_setup(blob)
    oproof, _, _ = oracle.model_prove(h, x)
    oracle.model_free(h)
    assert (out == out2).all() and proof.size == oproof.size and (proof == oproof).all() and (proofs[0] == oproof).all()
    assert proof.size == default_proof.size and (proof != default_proof).any()
    vb = ctx.verifier_blob()
    dpa.verify(vb, default_proof, x, out)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(vb, proof, x, out)
