#!/bin/bash
# r06 call 46: the CHAIN of two-round grids (k_sc_fused2g: both folds + the grid of the next two rounds; DP_SC_GRID2_CHAIN): parity and the 2^24 / 2^26 timings, three forms
o=gpurun_out/r06_call46; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py -m gpu -x -q > $o/pytest.txt 2>&1; grep -E "passed|failed|error|Error" $o/pytest.txt | tail -5
run() { tag=$1; nv=$2; reps=$3; shift 3; env SC24_PROFILE=1 "$@" timeout -s KILL 200 python tools/sumcheck24_only.py $reps $nv > $o/$tag.txt 2>&1; echo "== $tag"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $o/$tag.txt | tail -11; }
run sc24_chain_a 24 8 X=1
run sc24_grid_only 24 8 DP_SC_GRID2_CHAIN=0
run sc24_round_by_round 24 8 DP_SC_GRID2=0
run sc24_chain_b 24 8 X=1
run sc26_chain 26 5 X=1
run sc26_round_by_round 26 5 DP_SC_GRID2=0
