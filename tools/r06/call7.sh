#!/bin/bash
# r06 call 7: idle sleep + one thread per cohort as the candidate default; more proofs in flight with tighter worker arenas; the prio2 build on top
o=gpurun_out/r06_call7; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
S="DP_IDLE_SLEEP_US=20 DP_HOST_THREADS=22"
run base1 dense_4m 448 X=1
run sleep22 dense_4m 448 $S
run sleep22_560 dense_4m 560 $S DP_WORKER_ARENA_BYTES=335544320
run sleep22_660 dense_4m 660 $S DP_WORKER_ARENA_BYTES=318767104
run sleep22_p2 dense_4m 448 $S DP_LIB_VARIANT=prio2
run sleep22_p2_560 dense_4m 560 $S DP_LIB_VARIANT=prio2 DP_WORKER_ARENA_BYTES=335544320
run base2 dense_4m 448 X=1
run sleep22_cnn cnn_264k 448 $S
run base_cnn cnn_264k 448 X=1
run sleep10_22 dense_4m 448 DP_IDLE_SLEEP_US=10 DP_HOST_THREADS=22
run sleep40_22 dense_4m 448 DP_IDLE_SLEEP_US=40 DP_HOST_THREADS=22
