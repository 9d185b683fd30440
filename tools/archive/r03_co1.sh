#!/bin/bash
# cohort path (the default) after round 3's shared improvements (k_download, k_axpy_many): proofs in flight / arena sweep
o=${1:-gpurun_out/r03_co1}; mkdir -p "$o"; export TMPDIR=/tmp
run() { name=$1; conc=$2; waves=$3; shift; shift; shift; env "$@" timeout -s KILL 300 python tools/rx_probe.py dense $conc $waves 0 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run co_320 320 5 DP_X=0
run co_384 384 4 DP_X=0
run co_416 416 4 DP_X=0
run co_448_a480 448 4 DP_WORKER_ARENA_BYTES=503316480
run co_480_a480 480 4 DP_WORKER_ARENA_BYTES=503316480
run co_384_c12 384 4 DP_COHORT=12
