#!/bin/bash
# r06 call 14: (a) host work of a member between two device waits by the launches it issues there (DP_TIMING=3, [dp host-work]); (b) DP_HEAVY_GATE: at most L cohorts
# inside the batch opening at a time (the cohorts of a batch run in phase otherwise: all hashing, then all in one-workgroup tails), Dense-4M 448 in flight, 6 waves per batch
o=gpurun_out/r06_call14; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=3 timeout -s KILL 200 python tools/archive/conc_hoststats.py 448 > $o/hostwork_448.txt 2>&1; grep "host-work" $o/hostwork_448.txt | head -60 > $o/hostwork_top.txt; grep "proofs/s" $o/hostwork_448.txt
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
run base1 dense_4m 448 6 X=1
run gate4 dense_4m 448 6 DP_HEAVY_GATE=4
run gate6 dense_4m 448 6 DP_HEAVY_GATE=6
run gate8 dense_4m 448 6 DP_HEAVY_GATE=8
run base2 dense_4m 448 6 X=1
run gate11 dense_4m 448 6 DP_HEAVY_GATE=11
run gate14 dense_4m 448 6 DP_HEAVY_GATE=14
run gate3 dense_4m 448 6 DP_HEAVY_GATE=3
run base3 dense_4m 448 6 X=1
run gate6_cnn cnn_264k 448 4 DP_HEAVY_GATE=6
run gate11_cnn cnn_264k 448 4 DP_HEAVY_GATE=11
run base_cnn cnn_264k 448 4 X=1
DP_HEAVY_GATE=6 DP_TIMING=1 timeout -s KILL 200 python tools/archive/conc_hoststats.py 448 2>&1 | grep -E "gate|proofs/s|cohort:" | tail -8
