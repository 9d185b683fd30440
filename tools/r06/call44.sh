#!/bin/bash
# r06 call 44: the two-round grid of large base-table sumchecks (k_sc_terms2 / k_sc_fused2, DP_SC_GRID2): parity (sumcheck cases vs the oracle, config-5 goldens, ticket stress, the
# sharded forms) and the 2^24 / 2^26 sumcheck with and without it
o=gpurun_out/r06_call44; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py -m gpu -x -q > $o/pytest.txt 2>&1; tail -5 $o/pytest.txt
for k in 1 0 1 0; do SC24_PROFILE=1 DP_SC_GRID2=$k timeout -s KILL 200 python tools/sumcheck24_only.py 8 > $o/sc24_grid$k.txt 2>&1; echo "== 2^24 DP_SC_GRID2=$k"; tail -14 $o/sc24_grid$k.txt; done
for k in 1 0; do SC24_PROFILE=1 DP_SC_GRID2=$k timeout -s KILL 200 python tools/sumcheck24_only.py 5 26 > $o/sc26_grid$k.txt 2>&1; echo "== 2^26 DP_SC_GRID2=$k"; tail -11 $o/sc26_grid$k.txt; done
