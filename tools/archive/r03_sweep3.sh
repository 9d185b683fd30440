#!/bin/bash
# cohort size and proofs in flight around the new default (448 in flight, cohorts of 21): one process per point, 5 waves
o=${1:-gpurun_out/r03_sweep3}; mkdir -p "$o"; export TMPDIR=/tmp
run() { local tag=$1 conc=$2; shift 2; env "$@" timeout -s KILL 240 python tools/rx_probe.py dense $conc 5 0 > "$o/$tag.log" 2>&1; echo "$tag: $(tail -1 $o/$tag.log)"; }
run c21_448_a 448 DP_X=0
run c14_448 448 DP_COHORT=14
run c32_448 448 DP_COHORT=32
run c64_448 448 DP_COHORT=64
run c21_448_b 448 DP_X=0
run c24_512 512 DP_X=0
run c25_544 544 DP_X=0
run c32_512 512 DP_COHORT=32
run c21_448_c 448 DP_X=0
