#!/bin/bash
# r06 call 48 (run twice: the adopted fused grids; then k_sc_terms2 on the points {0, 1, oo, -1}): parity (sumcheck cases, config-5 goldens, ticket stress, sharded forms, model proofs) and timings
o=gpurun_out/r06_call48${TAG}; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.txt 2>&1; grep -E "passed|failed|error" $o/pytest.txt | tail -3
for nv in 24 26 22; do SC24_PROFILE=1 timeout -s KILL 200 python tools/sumcheck24_only.py 8 $nv > $o/sc$nv.txt 2>&1; echo "== 2^$nv"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $o/sc$nv.txt | tail -10; done
