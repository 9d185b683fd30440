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


def build_c_consumer():
    """tests/support/c_consumer.c with the flags a strict C11 host would use; links libdeepprove_hip.so"""
    import subprocess
    import deep_prove_amd as dpa
    out = os.path.join(ROOT, "tests", "support", "_build", "c_consumer")
    src = os.path.join(ROOT, "tests", "support", "c_consumer.c")
    deps = [src, os.path.join(ROOT, "include", "deep_prove_hip.h"), dpa.LIB_PATH]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-o", out, src, "-L", os.path.dirname(dpa.LIB_PATH),
                               "-ldeepprove_hip", "-lpthread", "-Wl,-rpath," + os.path.dirname(dpa.LIB_PATH)])
    return out


def test_header_is_c11_and_a_c_program_links_every_entry_point():
    """the boundary is a C ABI, not a Python one: include/deep_prove_hip.h compiles as strict C11 (-pedantic -Werror), a C
    program that references every declared entry point links against the library, and its list is the header's list"""
    import subprocess
    exe = build_c_consumer()
    r = subprocess.run([exe, "--symbols"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    src = open(os.path.join(ROOT, "tests", "support", "c_consumer.c")).read()
    listed = sorted(set(re.findall(r"\(fn_t\)(dp_[a-z0-9_]+)", src)))
    assert listed == declared_symbols()
    assert f"{len(listed)} of {len(listed)} entry points resolved" in r.stdout
