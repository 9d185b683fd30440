#!/bin/bash
# r06 call 16: cohorts started out of phase in G groups d ms apart (DP_COHORT_GROUPS / DP_COHORT_STAGGER_MS), Dense-4M 448 in flight, 6 and 12 waves per batch
o=gpurun_out/r06_call16; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
run base1 dense_4m 448 6 X=1
run g2_250 dense_4m 448 6 DP_COHORT_GROUPS=2 DP_COHORT_STAGGER_MS=250
run g3_160 dense_4m 448 6 DP_COHORT_GROUPS=3 DP_COHORT_STAGGER_MS=160
run g22_22 dense_4m 448 6 DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=22
run base2 dense_4m 448 6 X=1
run g2_120 dense_4m 448 6 DP_COHORT_GROUPS=2 DP_COHORT_STAGGER_MS=120
run g4_60 dense_4m 448 6 DP_COHORT_GROUPS=4 DP_COHORT_STAGGER_MS=60
run g22_10 dense_4m 448 6 DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=10
run base12 dense_4m 448 12 X=1
run g3_160_12 dense_4m 448 12 DP_COHORT_GROUPS=3 DP_COHORT_STAGGER_MS=160
run g22_22_12 dense_4m 448 12 DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=22
