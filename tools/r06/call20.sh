#!/bin/bash
# r06 call 20: cohorts out of phase (22 groups, 22 ms apart) AND the merged hash launches capped so that the ~8 cohorts that hash at the same time leave wave slots to
# the chains of the others; 12 waves per batch (the stagger costs at the two ends of a batch)
o=gpurun_out/r06_call20; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
S="DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=22"
run base1 dense_4m 448 12 X=1
run st_c128 dense_4m 448 12 $S DP_MERKLE_WG_CAP=128
run st_c192 dense_4m 448 12 $S DP_MERKLE_WG_CAP=192
run st_c256 dense_4m 448 12 $S DP_MERKLE_WG_CAP=256
run st_c384 dense_4m 448 12 $S DP_MERKLE_WG_CAP=384
run base2 dense_4m 448 12 X=1
run st_c256_w256 dense_4m 448 12 $S DP_MERKLE_WG_CAP=256 DP_WIDE_WG_CAP=256
run st_c128_w128 dense_4m 448 12 $S DP_MERKLE_WG_CAP=128 DP_WIDE_WG_CAP=128
run st_c192_660 dense_4m 660 12 $S DP_MERKLE_WG_CAP=192 DP_WORKER_ARENA_BYTES=318767104
run base_660 dense_4m 660 12 DP_WORKER_ARENA_BYTES=318767104
