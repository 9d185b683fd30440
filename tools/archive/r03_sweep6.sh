#!/bin/bash
# cohorts started out of phase (DP_COHORT_STAGGER_MS: cohort c of n starts c/n of that late): Dense-4M, 448 in flight, 8 waves per batch
o=${1:-gpurun_out/r03_sweep6}; mkdir -p "$o"; export TMPDIR=/tmp
run() { local tag=$1; shift; env "$@" timeout -s KILL 300 python tools/rx_probe.py dense 448 8 0 > "$o/$tag.log" 2>&1; echo "$tag: $(tail -1 $o/$tag.log)"; }
run st0_a DP_COHORT_STAGGER_MS=0
run st400_a DP_COHORT_STAGGER_MS=400
run st800_a DP_COHORT_STAGGER_MS=800
run st0_b DP_COHORT_STAGGER_MS=0
run st400_b DP_COHORT_STAGGER_MS=400
run st800_b DP_COHORT_STAGGER_MS=800
run st200 DP_COHORT_STAGGER_MS=200
