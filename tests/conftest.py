import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (test infrastructure). Built on demand with its own Makefile."""
    from support import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hostlogic_bin():
    """tests/support/hostlogic_check: product host orchestrator over the CPU test double vs the oracle"""
    out = os.path.join(ROOT, "tests", "support", "_build", "hostlogic_check")
    src = os.path.join(ROOT, "tests", "support", "hostlogic_check.cpp")
    deps = [src, os.path.join(ROOT, "tests", "support", "cpu_dev.hpp")]
    deps += [os.path.join(ROOT, "deep-prove_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "deep-prove_amd", "csrc")) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith(".hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out, src])
    return out


@pytest.fixture(scope="session")
def dev():
    """the MI355X device context; GPU tests fail loudly (not skip) when the HIP path is unavailable"""
    import deep_prove_amd as dpa
    d = dpa.Device(0)
    yield d
    d.close()
