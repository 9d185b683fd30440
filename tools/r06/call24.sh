#!/bin/bash
# r06 call 24: re-tune around the new defaults (wave priority 2 + DP_WIDE_WG_CAP 256): hash-layer caps, wide caps, stagger, cohort count at 660 in flight
o=gpurun_out/r06_call24; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
A=DP_WORKER_ARENA_BYTES=318767104
run base1 dense_4m 448 12 X=1
run m512 dense_4m 448 12 DP_MERKLE_WG_CAP=512
run m1024 dense_4m 448 12 DP_MERKLE_WG_CAP=1024
run m2048 dense_4m 448 12 DP_MERKLE_WG_CAP=2048
run w128 dense_4m 448 12 DP_WIDE_WG_CAP=128
run w192 dense_4m 448 12 DP_WIDE_WG_CAP=192
run base2 dense_4m 448 12 X=1
run w384 dense_4m 448 12 DP_WIDE_WG_CAP=384
run w512 dense_4m 448 12 DP_WIDE_WG_CAP=512
run st22 dense_4m 448 12 DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=20
run lp2048 dense_4m 448 12 DP_LP_MAX_TP=2048
run lp128 dense_4m 448 12 DP_LP_MAX_TP=128
run base3 dense_4m 448 12 X=1
run c660 dense_4m 660 12 $A
run c660_co24 dense_4m 660 12 $A DP_COHORT=28 GPU_MAX_HW_QUEUES=26
run c660_co20 dense_4m 660 12 $A DP_COHORT=33
run c704 dense_4m 704 12 $A
