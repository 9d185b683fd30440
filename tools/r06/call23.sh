#!/bin/bash
# r06 call 23: release = DP_BASE_PRIO 2 + DP_WIDE_WG_CAP 256 now. DP_HASH_STREAMS = N: the wide hash layers of the cohorts on N streams of the LOWEST hardware-queue priority
# (tools/r06/prioprobe.hip: a chain of small kernels beside 22 low-priority queues of chip-filling grids runs at its idle-chip speed, 2-25 us per link against 4-7 ms)
o=gpurun_out/r06_call23; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
A=DP_WORKER_ARENA_BYTES=318767104
run base1 dense_4m 448 12 X=1
run hs1 dense_4m 448 12 DP_HASH_STREAMS=1
run hs2 dense_4m 448 12 DP_HASH_STREAMS=2
run hs4 dense_4m 448 12 DP_HASH_STREAMS=4
run hs8 dense_4m 448 12 DP_HASH_STREAMS=8
run base2 dense_4m 448 12 X=1
run hs22 dense_4m 448 12 DP_HASH_STREAMS=22
run hs2_min15 dense_4m 448 12 DP_HASH_STREAMS=2 DP_HASH_STREAM_MIN_N=32768
run hs4_660 dense_4m 660 12 DP_HASH_STREAMS=4 $A
run base_660 dense_4m 660 12 $A
run hs4_cnn cnn_264k 448 6 DP_HASH_STREAMS=4
run base_cnn cnn_264k 448 6 X=1
run hs4_tf transformer_layer 320 3 DP_HASH_STREAMS=4
run base_tf transformer_layer 320 3 X=1
