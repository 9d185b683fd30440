#!/bin/bash
# staged activations (DP_STAGE_ACTIVATIONS) and several Merkle layers per launch (DP_MERKLE_FUSE) at 448 in flight: one process per point, 5 waves
o=${1:-gpurun_out/r03_sweep4}; mkdir -p "$o"; export TMPDIR=/tmp
run() { local tag=$1 conc=$2; shift 2; env "$@" timeout -s KILL 240 python tools/rx_probe.py dense $conc 5 0 > "$o/$tag.log" 2>&1; echo "$tag: $(tail -1 $o/$tag.log)"; }
timeout -s KILL 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "mlp or dense4m or golden_mlp or deterministic or concurrent" > "$o/tests.log" 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"
run stage1_a 448 DP_X=0
run stage0_a 448 DP_STAGE_ACTIVATIONS=0
run fuse8_a 448 DP_MERKLE_FUSE=8
run stage1_b 448 DP_X=0
run stage0_b 448 DP_STAGE_ACTIVATIONS=0
run fuse8_b 448 DP_MERKLE_FUSE=8
run fuse4 448 DP_MERKLE_FUSE=4
run stage1_c 448 DP_X=0
