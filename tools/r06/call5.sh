#!/bin/bash
# r06 call 5: everything placeable — small caps for the streaming kernels AND the Merkle layers together
o=gpurun_out/r06_call5; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-120)"; }
run base1 dense_4m 448 X=1
run w64_m128 dense_4m 448 DP_WIDE_WG_CAP=64 DP_MERKLE_WG_CAP=128
run w64_m192 dense_4m 448 DP_WIDE_WG_CAP=64 DP_MERKLE_WG_CAP=192
run w96_m192 dense_4m 448 DP_WIDE_WG_CAP=96 DP_MERKLE_WG_CAP=192
run w42_m128 dense_4m 448 DP_WIDE_WG_CAP=42 DP_MERKLE_WG_CAP=128
run w128_m256 dense_4m 448 DP_WIDE_WG_CAP=128 DP_MERKLE_WG_CAP=256
run base2 dense_4m 448 X=1
run p2_w128_m256 dense_4m 448 DP_LIB_VARIANT=prio2 DP_MERKLE_WG_CAP=256
