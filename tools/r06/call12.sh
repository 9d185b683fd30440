#!/bin/bash
# r06 call 12: cap the STREAMING kernels only (few fat workgroups instead of tens of thousands of short ones), Merkle layers uncapped
o=gpurun_out/r06_call12; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
run base1 dense_4m 448 X=1
run w128_mU dense_4m 448 DP_WIDE_WG_CAP=128 DP_MERKLE_WG_CAP=86016
run w256_mU dense_4m 448 DP_WIDE_WG_CAP=256 DP_MERKLE_WG_CAP=86016
run w512_mU dense_4m 448 DP_WIDE_WG_CAP=512 DP_MERKLE_WG_CAP=86016
run base2 dense_4m 448 X=1
run w256_mU_b dense_4m 448 DP_WIDE_WG_CAP=256 DP_MERKLE_WG_CAP=86016
run w256_mU_660 dense_4m 660 DP_WIDE_WG_CAP=256 DP_MERKLE_WG_CAP=86016 DP_WORKER_ARENA_BYTES=318767104
run base_660 dense_4m 660 DP_WORKER_ARENA_BYTES=318767104
run w256_mU_cnn cnn_264k 448 DP_WIDE_WG_CAP=256 DP_MERKLE_WG_CAP=86016
run base_cnn cnn_264k 448 X=1
