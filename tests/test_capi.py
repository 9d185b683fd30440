"""The C-ABI library loads and exports every symbol declared in include/deep_prove_hip.h; the host-only entry points
(verify, transcript) work without a GPU; golden fixtures pin the oracle and the verifier."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "deep_prove_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    import deep_prove_amd as dpa
    lib = ctypes.CDLL(dpa.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"missing symbol {s}"
    assert sorted(dpa._lib.SIGNATURES) == syms  # the Python binding covers the whole header


def test_no_cpu_fallback_without_device():
    import deep_prove_amd as dpa
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(dpa.DeepProveError) as e:
        dpa.Device(0)
    assert e.value.code == -6  # DP_ERR_NODEVICE


def test_golden_proof_pins_oracle_and_verifier(oracle):
    """tests/golden/mlp_w8.npz was produced by tests/golden/make_golden.py (oracle, width-8 MLP)."""
    import deep_prove_amd as dpa
    g = np.load(os.path.join(ROOT, "tests", "golden", "mlp_w8.npz"))
    h = oracle.model_setup(g["model_blob"])
    proof, out, _ = oracle.model_prove(h, g["input"])
    oracle.model_free(h)
    assert (out == g["output"]).all()
    assert proof.size == g["proof"].size and (proof == g["proof"]).all()  # oracle regression pin (bit exact)
    dpa.verify(g["verifier_blob"], g["proof"], g["input"], g["output"])    # product verifier accepts the golden proof
    bad = g["proof"].copy()
    bad[bad.size // 3] ^= np.uint64(1)
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(g["verifier_blob"], bad, g["input"], g["output"])
    wrong_out = g["output"].copy()
    wrong_out[0] += 1
    with pytest.raises(dpa.DeepProveError):
        dpa.verify(g["verifier_blob"], g["proof"], g["input"], wrong_out)


def test_golden_primitive_vectors(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "primitives.npz"))
    assert (oracle.permute(g["perm_in"]) == g["perm_out"]).all()
    pt = [tuple(int(x) for x in r) for r in g["point"]]
    assert (oracle.eq_table(pt) == g["eq_table"]).all()
    assert oracle.pcs_commit_root(1 << 12, g["poly_base"], False) == [int(x) for x in g["root_base"]]
    assert oracle.pcs_commit_root(1 << 12, g["poly_ext"], True) == [int(x) for x in g["root_ext"]]
    t = oracle.transcript(b"test")
    p, f = oracle.sumcheck_prove(10, [g["poly_base"], g["poly_ext"]], [False, True], [((1, 0), [0, 1])], t)
    assert (p == g["sumcheck_proof"]).all() and (f == g["sumcheck_finals"]).all()
