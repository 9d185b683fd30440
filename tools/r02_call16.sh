#!/bin/bash
out=${1:-gpurun_out/r02_call16}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py -m gpu -q -x -k "sumcheck or sharded" > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log" | cut -c1-300
SC24_PROFILE=1 timeout 120 python tools/sumcheck24_only.py 6 > "$out/sc24.log" 2>&1; tail -12 "$out/sc24.log" | cut -c1-250
