#!/bin/bash
# r06 call 21: around call 20's best point (cohorts out of phase + every grid-stride launch AND the hash layers capped at 256 workgroups per merged launch: 1 001 against 936-944)
o=gpurun_out/r06_call21; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
S="DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=22"
run base1 dense_4m 448 12 X=1
run st_c256_w256 dense_4m 448 12 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run c256_w256 dense_4m 448 12 DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run st_w256 dense_4m 448 12 $S DP_WIDE_WG_CAP=256
run st_c384_w384 dense_4m 448 12 $S DP_MERKLE_WG_CAP=384 DP_WIDE_WG_CAP=384
run st_c512_w256 dense_4m 448 12 $S DP_MERKLE_WG_CAP=512 DP_WIDE_WG_CAP=256
run base2 dense_4m 448 12 X=1
run st_c256_w192 dense_4m 448 12 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=192
run st_c192_w256 dense_4m 448 12 $S DP_MERKLE_WG_CAP=192 DP_WIDE_WG_CAP=256
run st_c256_w256_b dense_4m 448 12 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run st_c256_w256_6 dense_4m 448 6 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run g4_c256_w256_6 dense_4m 448 6 DP_COHORT_GROUPS=4 DP_COHORT_STAGGER_MS=60 DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run st_c256_w256_660 dense_4m 660 12 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256 DP_WORKER_ARENA_BYTES=318767104
run base3 dense_4m 448 12 X=1
