#!/bin/bash
# r06 call 22: DP_WIDE_WG_CAP=256 is the default now. (a) the prio2 build (hash waves at priority 0, every other kernel at 2) on top; (b) more proofs in flight with tighter
# worker arenas; (c) CNN-264k and the transformer layer with and without the cap
o=gpurun_out/r06_call22; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
A=DP_WORKER_ARENA_BYTES=318767104
GPU_MAX_HW_QUEUES=24 timeout -s KILL 300 tools/_build/prioprobe 300 2000 > $o/prioprobe.txt 2>&1; cat $o/prioprobe.txt | cut -c1-230
run w256 dense_4m 448 12 X=1
run nocap dense_4m 448 12 DP_WIDE_WG_CAP=0
run p2_w256 dense_4m 448 12 DP_LIB_VARIANT=prio2
run w256_b dense_4m 448 12 X=1
run p2_w256_b dense_4m 448 12 DP_LIB_VARIANT=prio2
run w256_560 dense_4m 560 12 $A
run w256_660 dense_4m 660 12 $A
run w256_740 dense_4m 740 12 $A
run p2_w256_660 dense_4m 660 12 $A DP_LIB_VARIANT=prio2
run cnn_w256 cnn_264k 448 6 X=1
run cnn_nocap cnn_264k 448 6 DP_WIDE_WG_CAP=0
run cnn_p2 cnn_264k 448 6 DP_LIB_VARIANT=prio2
run tf_w256 transformer_layer 320 3 X=1
run tf_nocap transformer_layer 320 3 DP_WIDE_WG_CAP=0
