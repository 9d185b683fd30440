#!/bin/bash
# Merkle: parents up to which a layer uses the 8-lanes-per-node kernel (DP_MERKLE_LP_MAX, default 4096) at 448 in flight
o=${1:-gpurun_out/r03_sweep5}; mkdir -p "$o"; export TMPDIR=/tmp
run() { local tag=$1 conc=$2; shift 2; env "$@" timeout -s KILL 240 python tools/rx_probe.py dense $conc 5 0 > "$o/$tag.log" 2>&1; echo "$tag: $(tail -1 $o/$tag.log)"; }
run lp4096_a 448 DP_X=0
run lp512_a 448 DP_MERKLE_LP_MAX=512
run lp0_a 448 DP_MERKLE_LP_MAX=0
run lp4096_b 448 DP_X=0
run lp512_b 448 DP_MERKLE_LP_MAX=512
run lp0_b 448 DP_MERKLE_LP_MAX=0
run lp16384 448 DP_MERKLE_LP_MAX=16384
