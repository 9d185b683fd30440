#!/bin/bash
out=${1:-gpurun_out/r02_call21}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_model.py tests/test_gpu_fused.py -m gpu -q -x -k "commit or config2 or config3 or oracles_proof or batch_open" --durations=6 > "$out/pytest.log" 2>&1; tail -14 "$out/pytest.log" | cut -c1-200
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
for env in ("0", "1"):
    pass
mb = dpa.models.dense_4m()
for rep in range(3):
    t0 = time.perf_counter(); ctx = dpa.Context.generate(dev, mb.blob()); dt = time.perf_counter() - t0; ctx.free()
    print(f"Context::generate dense_4m (5 weight commits of 2^20 + biases): {1000 * dt:.1f} ms")
pcs = dpa.Basefold(dev, 1 << 20)
w = np.random.default_rng(1).integers(0, dpa.P, size=1 << 20, dtype=np.uint64)
m = dpa.Mle.from_base(dev, w)
for rep in range(3):
    t0 = time.perf_counter(); c = pcs.commit(m); print(f"commit 2^20 base: {1000 * (time.perf_counter() - t0):.2f} ms")
PY
DP_NTT_STAGEWISE=1 python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import deep_prove_amd as dpa
dev = dpa.Device(0)
pcs = dpa.Basefold(dev, 1 << 20)
w = np.random.default_rng(1).integers(0, dpa.P, size=1 << 20, dtype=np.uint64)
m = dpa.Mle.from_base(dev, w)
for rep in range(3):
    t0 = time.perf_counter(); c = pcs.commit(m); print(f"DP_NTT_STAGEWISE=1 commit 2^20 base: {1000 * (time.perf_counter() - t0):.2f} ms")
PY
