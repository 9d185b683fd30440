#!/bin/bash
# r06 call 13: (a) DP_SER_THREADS again now that the cohort threads sleep when idle (round 5 measured it with 14 spinning threads: the helpers starved);
# (b) cohort size (DP_COHORT) at 448 / 660 in flight with the idle-sleeping threads
o=gpurun_out/r06_call13; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
run base1 dense_4m 448 X=1
run ser2 dense_4m 448 DP_SER_THREADS=2
run ser4 dense_4m 448 DP_SER_THREADS=4
run base2 dense_4m 448 X=1
run ser3 dense_4m 448 DP_SER_THREADS=3
run ser6 dense_4m 448 DP_SER_THREADS=6
run co16 dense_4m 448 DP_COHORT=16
run co28 dense_4m 448 DP_COHORT=28
run co32_ser3 dense_4m 448 DP_COHORT=32 DP_SER_THREADS=3
run base3 dense_4m 448 X=1
run ser3_660 dense_4m 660 DP_SER_THREADS=3 DP_WORKER_ARENA_BYTES=318767104
run ser3_cnn cnn_264k 448 DP_SER_THREADS=3
run base_cnn cnn_264k 448 X=1
