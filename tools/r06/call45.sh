#!/bin/bash
# r06 call 45: k_sc_terms2 chip-sized with prefetch and lazy extrapolation sums: parity (config-5 goldens, sumcheck cases) and the 2^24 / 2^26 timings
o=gpurun_out/r06_call45; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py -m gpu -x -q > $o/pytest.txt 2>&1; grep -E "passed|failed|error" $o/pytest.txt | tail -3
for k in 1 0 1; do SC24_PROFILE=1 DP_SC_GRID2=$k timeout -s KILL 200 python tools/sumcheck24_only.py 8 > $o/sc24_grid$k.txt 2>&1; echo "== 2^24 DP_SC_GRID2=$k"; tail -12 $o/sc24_grid$k.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"; done
for k in 1; do SC24_PROFILE=1 DP_SC_GRID2=$k timeout -s KILL 200 python tools/sumcheck24_only.py 5 26 > $o/sc26_grid$k.txt 2>&1; echo "== 2^26 DP_SC_GRID2=$k"; tail -9 $o/sc26_grid$k.txt; done
