"""The host transcript's Poseidon2-w8 permutation has two implementations in the library: scalar (csrc/poseidon2.h hostnc) and AVX-512
(csrc/p2_avx512.cpp: the eight state words in the eight lanes of one register). They must agree word for word — with the oracle's and the
independent big-int permutation too — on random, boundary and non-canonical states; everything host-side (transcripts, verifier, the
sponge service of DP_HOST_SPONGE) goes through whichever the CPU selects."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _permute(lib, st, force_scalar):
    s = np.array(st, dtype=np.uint64)
    v = C.c_int32(0)
    rc = lib.dp_host_poseidon2(s.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if force_scalar else 0, C.byref(v))
    assert rc == 0
    return s, bool(v.value)


def test_vectorised_and_scalar_permutations_agree_with_the_oracle(oracle):
    from deep_prove_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    states = [rng.integers(0, P, size=8, dtype=np.uint64) for _ in range(300)]
    states += [np.zeros(8, dtype=np.uint64), np.full(8, P - 1, dtype=np.uint64), np.array([0, 1, P - 1, 1 << 32, (1 << 32) - 1, P - (1 << 32), 2, 3], dtype=np.uint64)]
    vec = None
    for st in states:
        a, vec = _permute(lib, st, False)
        b, _ = _permute(lib, st, True)
        assert (a == b).all()
        assert (a == oracle.permute(st)).all()
    print("vectorised host permutation in use:", vec)


def test_golden_proof_verifies_with_either_permutation():
    """the host verifier (Merkle paths = Poseidon2 compressions, transcript) accepts the golden proof with the scalar code forced
    (DP_NO_AVX512=1, a fresh process) and with the CPU's default"""
    code = ("import numpy as np, os, sys; sys.path.insert(0, %r); import deep_prove_amd as dpa; from deep_prove_amd import _lib; import ctypes as C;"
            "g = np.load(os.path.join(%r, 'tests', 'golden', 'mlp_w8.npz')); dpa.verify(g['verifier_blob'], g['proof'], g['input'], g['output']);"
            "v = C.c_int32(0); s = np.zeros(8, dtype=np.uint64); _lib.load().dp_host_poseidon2(s.ctypes.data_as(C.POINTER(C.c_uint64)), 0, C.byref(v)); print('ok', v.value)") % (ROOT, ROOT)
    outs = []
    for env in ({"DP_NO_AVX512": "1"}, {}):
        e = dict(os.environ); e.pop("DP_NO_AVX512", None); e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
        outs.append(r.stdout.split()[1])
    assert outs[0] == "0"  # the scalar code when asked for
