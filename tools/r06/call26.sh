#!/bin/bash
# r06 call 26: the members of a cohort on k host threads (DP_COHORT_THREADS; the cohort's launch sequence under a lock): the host phase of a cohort step k wide
o=gpurun_out/r06_call26; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run t1 dense_4m 704 8 X=1
run t2 dense_4m 704 8 DP_COHORT_THREADS=2
run t3 dense_4m 704 8 DP_COHORT_THREADS=3
run t1_b dense_4m 704 8 X=1
run t2_b dense_4m 704 8 DP_COHORT_THREADS=2
run t4 dense_4m 704 8 DP_COHORT_THREADS=4
run t2_448 dense_4m 448 12 DP_COHORT_THREADS=2
run t1_448 dense_4m 448 12 X=1
run t2_cnn cnn_264k 704 4 DP_COHORT_THREADS=2
run t1_cnn cnn_264k 704 4 X=1
run t2_tf transformer_layer 320 3 DP_COHORT_THREADS=2
run t1_tf transformer_layer 320 3 X=1
DP_COHORT_THREADS=2 DP_TIMING=1 timeout -s KILL 200 python tools/archive/conc_hoststats.py 704 2>&1 | grep -E "proofs/s|cohort:" | tail -4 | cut -c1-250
