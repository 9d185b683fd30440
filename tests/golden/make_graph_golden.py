"""Golden fixture of the GRAPH models (run from the repo root: python tests/golden/make_graph_golden.py): QKV, ConcatMatMul, MatMul and Add of
two inputs (layers/transformer/qkv.rs, layers/concat_matmul.rs, layers/matrix_mul.rs, layers/add.rs; models with several input and output
tensors) proved by the ORACLE. Like the other fixtures it pins the oracle against regressions ("parity unpinned"): sha256 of the model blob,
the input, the output (checked against the numpy inference of models.GraphBuilder.run) and the canonical proof stream."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from support import oracle_lib  # noqa: E402
import deep_prove_amd as dpa  # noqa: E402

CASES = [
    ("attention_block", dict(seq=8, emb=16, heads=2, head_dim=8, config=61)),
    ("attention_block", dict(seq=16, emb=64, heads=4, head_dim=16, config=65)),
    ("matmul_pair", dict(seq=4, k=8, n=16, config=62)),
    ("matmul_pair", dict(seq=8, k=32, n=16, config=63, transpose_b=True)),
    ("qkv_two_outputs", dict(seq=4, k=16, n=16, config=64)),
    # LayerNorm (layers/transformer/layernorm.rs) -> Linear -> ReLU -> Linear; the second with N = 12 of a padded dimension of 16
    ("layernorm_mlp", dict(seq=8, features=16, width=16, config=81)),
    ("layernorm_mlp", dict(seq=16, features=12, width=32, config=82)),
    # Softmax (layers/transformer/softmax.rs) over [heads][n][n] scores under the causal mask; the second with a zero table
    ("softmax_only", dict(heads=2, n=8, config=91)),
    ("softmax_only", dict(heads=4, n=16, config=93, in_scale=3.0 / 127.0)),
    # the attention half of a pre-LN transformer block: LayerNorm, QKV, per-head scores, Softmax, probabilities x V, projection, residual
    ("transformer_block", dict(seq=8, emb=16, heads=2, head_dim=8, config=95)),
    ("transformer_block", dict(seq=16, emb=32, heads=4, head_dim=8, config=96)),
    # the same half block with the reference's Mha layer as ONE node (layers/transformer/mha.rs): Q K^T, the softmax straight on the products, x V
    ("mha_block", dict(seq=8, emb=16, heads=2, head_dim=8, config=97)),
    ("mha_block", dict(seq=16, emb=32, heads=4, head_dim=8, config=98)),
    # one whole pre-LN transformer layer as one graph of 19 nodes: attention half (with the Mha node) and feed-forward half, two LayerNorms, three inputs
    ("transformer_layer", dict(seq=8, emb=16, heads=2, head_dim=8, ffn=32, config=101)),
    # the same layer at the size bench.py TIMES (section `transformer_layer`: 64 tokens x 256 features, 4 heads of 64, ffn 1024, config 66): ~16 s of oracle time,
    # 1.29 M proof words. `gpu_only`: the CPU suites (test_oracle, test_hostlogic) skip it; the GPU suite and the bench compare against this sha256.
    ("transformer_layer", dict(seq=64, emb=256, heads=4, head_dim=64, ffn=1024, config=66)),
    # Activation::Gelu (layers/activation.rs:559-671). 15: the reference's own proving test (:686-697), one GELU over 32 entries: committed columns that small are
    # opened by showing them, and the oracle runs TO THE LETTER of the reference (GELU_LITERAL below). 16 / 17: columns with a real opening — the reference's prover
    # files a claim its verifier does not check there (:419-430 against :495-505), the oracle files the verifier's (oracle_lib.set_gelu_files_lookup_claim)
    ("gelu_only", dict(n=32, config=111)),
    ("gelu_mlp", dict(width=256, config=112)),
    ("transformer_layer", dict(seq=8, emb=16, heads=2, head_dim=8, ffn=32, config=101, gelu=True)),
]
GPU_ONLY = {14}
GELU_LITERAL = {15}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build(name, kw):
    return getattr(dpa.models, name)(**kw)


if __name__ == "__main__":
    o = oracle_lib.load()
    out = []
    for ci, (name, kw) in enumerate(CASES):
        g = build(name, kw)
        blob, x = g.blob(), g.input()
        o.set_gelu_files_lookup_claim(ci not in GELU_LITERAL)
        h = o.model_setup(blob)
        proof, y, _ = o.model_prove(h, x)
        o.model_free(h)
        assert y.size == g.run(x).size and (y == g.run(x)).all(), name
        out.append(dict(model=name, args=kw, blob_sha256=sha(blob), input_sha256=sha(x), output_sha256=sha(y), proof_sha256=sha(proof), proof_words=int(proof.size), **({"gpu_only": True} if ci in GPU_ONLY else {}), **({"oracle_gelu_claim": "reference"} if ci in GELU_LITERAL else {})))
        print(name, kw, proof.size, "proof words")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_models.json"), "w") as f:
        json.dump(out, f, indent=1)
