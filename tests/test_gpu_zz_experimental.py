"""Parity of the EXPERIMENTAL code paths on hardware (fused protocol kernels with device-side Fiat-Shamir, staging-ring uploads,
fused Merkle layers: DESIGN.md §6 / §9). They are off by default and were validated on the CPU SIMT emulator only
(tests/test_kernel_emul.py), so this file is opt-in: DP_TEST_EXPERIMENTAL=1 python -m pytest tests/test_gpu_zz_experimental.py -m gpu
Every configuration runs in its own process (tools/knob_sweep.py --one): a kernel that faults cannot take the suite down; the
check is that proof 0 of a batch (experimental path: batches run in throughput mode) equals the sequential proof of the same
input (latency mode: validated path) and that sampled proofs verify."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("DP_TEST_EXPERIMENTAL"), reason="opt-in: DP_TEST_EXPERIMENTAL=1")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = [{"DP_DEVICE_LOGUP": "1"}, {"DP_DEVICE_LOGUP": "2"}, {"DP_DEVICE_CLASSIC": "1"}, {"DP_DEVICE_DENSE": "1"}, {"DP_DEVICE_EQSUM": "1"}, {"DP_DEVICE_COMMIT": "1"},
         {"DP_ASYNC_UPLOAD": "1"}, {"DP_MERKLE_FUSE": "4"}, {"DP_TAIL_MAX": "2048"}, {"DP_COHORT_XCD": "1"},
         {"DP_DEVICE_LOGUP": "2", "DP_DEVICE_CLASSIC": "1", "DP_DEVICE_DENSE": "1", "DP_DEVICE_EQSUM": "1", "DP_DEVICE_COMMIT": "1", "DP_ASYNC_UPLOAD": "1", "DP_MERKLE_FUSE": "4", "DP_TAIL_MAX": "2048"}]


@pytest.mark.parametrize("workload,conc", [("dense_4m", 16), ("cnn_264k", 8)])
@pytest.mark.parametrize("flags", FLAGS, ids=lambda f: "+".join(f"{k[3:].lower()}={v}" for k, v in f.items()))
def test_experimental_path_matches_the_validated_path(workload, conc, flags):
    env = dict(os.environ)
    env.update(flags)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "knob_sweep.py"), "--one", workload, str(conc)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and line.startswith("{"), (r.stdout + r.stderr)[-2000:]
    rec = json.loads(line)
    assert rec["batch0_equals_single"] and rec["verified"] == 2, rec
