"""dp_model_infer_host: the quantised inference the library runs before every proof (Model::run semantics; int16 weights with 32-bit
accumulators and vector-ISA clones of the inner loops where exact) against the numpy inference of deep_prove_amd/models.py — at the sizes
of the measured workloads, where the fast paths are taken, and on inputs that force the 64-bit fallback."""
import numpy as np
import pytest


@pytest.mark.parametrize("name,args,kw", [("mlp", (2, 64), dict(config=31)), ("dense_4m", (), {}), ("cnn_tiny", (), {}), ("seq_mlp", (16, 64), dict(config=63, transpose_last=True, positional=True)),
                                          ("seq_1m", (), {}), ("token_mlp", (32, 300, 128), dict(config=72, max_positions=50)),
                                          # the lookup-table semantics the oracle and the product share as TEXT (softmax / inverse-square-root tables, the FFT convolution):
                                          # models.py restates them in numpy from the reference (softmax.rs:153-345,455-566; layernorm.rs; convolution.rs) — a third reading
                                          ("cnn_264k", (), {}), ("transformer_layer", (16, 64, 4, 16, 128), dict(config=65)), ("transformer_layer", (64, 256, 4, 64, 1024), dict(config=66)),
                                          ("attention_block", (16, 64, 4, 16), dict(config=64)),
                                          ("gelu_mlp", (256,), dict(config=112)), ("transformer_layer", (16, 64, 4, 16, 128), dict(config=65, gelu=True))])
def test_library_inference_equals_numpy(name, args, kw):
    import deep_prove_amd as dpa
    mb = getattr(dpa.models, name)(*args, **kw)
    for idx in (1000, 1001):
        x = mb.input(idx)
        assert (dpa.infer_host(mb.blob(), x) == mb.run(x)).all()


def test_gelu_table_on_every_input_of_the_quantised_range():
    """the library's inference of a GELU over all of -128 .. 127 equals models.gelu_apply (f32 after every operation, the C library's tanhf) for the
    multipliers of four input scales — a table of 2^13, 2^14 and 2^20 rows among them"""
    import deep_prove_amd as dpa
    x = np.arange(256, dtype=np.int64) - 128
    for scale in (1.0 / 128.0, 1.4 / 128.0, 1.0 / 4096.0 * 3, 1.0):
        mb = dpa.models.gelu_only(256, config=113, in_scale=scale)
        m = mb.layers[0]["multiplier"]
        xs = x if -128 * m >= -(1 << (7 + (m - 1).bit_length())) else np.clip(x, -127, 127)
        assert (dpa.infer_host(mb.blob(), xs) == mb.run(xs)).all(), scale


def test_large_inputs_take_the_64_bit_path():
    """activations beyond int16 (no Requant in front): the int16 shortcut must not be taken, the results stay exact"""
    import deep_prove_amd as dpa
    mb = dpa.models.ModelBuilder((8, 4), 5)
    mb.matmul(16, requant=False).matmul(8, bias=False, requant=False)  # the second MatMul sees un-requantised sums
    x = mb.input() * 3000
    assert np.abs(x).max() > 32767
    assert (dpa.infer_host(mb.blob(), x) == mb.run(x)).all()


def test_mutated_model_blobs_are_rejected_or_run_but_never_crash():
    """parse_model / validate_model / run_model behind dp_model_infer_host on 2 000 mutated blobs of four model families (flipped bits, huge
    and negative dimensions, truncations): every call returns (a result or DP_ERR_*), the process survives"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "support", "fuzz_blob.py"), "11", "2000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz done" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
    rejected = int(r.stdout.split("rejected")[1].split()[0])
    assert rejected > 500
