#!/bin/bash
# r06 call 3: DP_WIDE_WG_CAP — every grid-stride launch of a cohort capped so that it can be placed at once (tools/r06/qprobe.hip: a grid that cannot be placed
# holds its queue's pipe and every queue behind it), Dense-4M at 448 in flight, alternating on one box
o=gpurun_out/r06_call3; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run base1 dense_4m 448 X=1
run w64 dense_4m 448 DP_WIDE_WG_CAP=64
run w128 dense_4m 448 DP_WIDE_WG_CAP=128
run w256 dense_4m 448 DP_WIDE_WG_CAP=256
run w512 dense_4m 448 DP_WIDE_WG_CAP=512
run w1024 dense_4m 448 DP_WIDE_WG_CAP=1024
run base2 dense_4m 448 X=1
run w128_m512 dense_4m 448 DP_WIDE_WG_CAP=128 DP_MERKLE_WG_CAP=512
run w64_m256 dense_4m 448 DP_WIDE_WG_CAP=64 DP_MERKLE_WG_CAP=256
