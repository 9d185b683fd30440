#!/bin/bash
# parity of the sumcheck paths + 2^24 sumcheck timing (two settings) + a quick torch-free bench line
out=${1:-gpurun_out/r02_check}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_model.py tests/test_gpu_sharded.py -m gpu -q -x -k "sumcheck or sharded or config1 or mlp or batch_verifier" > "$out/pytest.log" 2>&1; grep -E "passed|failed" "$out/pytest.log" | tail -1
SC24_PROFILE=1 timeout 100 python tools/sumcheck24_only.py 6 2>&1 | grep -v "^W2\|^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" | cut -c1-200
echo "--- DP_PERSIST_GLOBAL_MID=1"; DP_PERSIST_GLOBAL_MID=1 timeout 100 python tools/sumcheck24_only.py 4 2>&1 | grep " ms " | cut -c1-100
DP_BENCH_NO_TORCH=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cnn > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; python -c "
import json
r=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('value',r['value'],'ms/step',r['ms_per_step'],'steps min/med/max',r['step_ms_min_median_max'],'golden',r['config']['golden_sha256_ok'],'sc24',r['sumcheck24']['wall_ms'],r['sumcheck24']['roofline']['frac'], 'peak',r['roofline']['peak'],'job_frac',r['roofline']['job_frac'])"
