#!/bin/bash
# r06 call 34: hash-layer grids small enough that the hash launches of ALL cohorts are placeable at once (22 x cap <= ~1 800 workgroup slots at 7 hash waves per SIMD):
# DP_MERKLE_WG_CAP below the 256 of the general cap (call 24 only swept it upwards), in phase and staggered
o=gpurun_out/r06_call34; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-110)"; }
ST="DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=25"
run base1 dense_4m 704 8 X=1
run m64 dense_4m 704 8 DP_MERKLE_WG_CAP=64
run m128 dense_4m 704 8 DP_MERKLE_WG_CAP=128
run m32 dense_4m 704 8 DP_MERKLE_WG_CAP=32
run m192 dense_4m 704 8 DP_MERKLE_WG_CAP=192
run base2 dense_4m 704 8 X=1
run m64_st dense_4m 704 8 DP_MERKLE_WG_CAP=64 $ST
run m128_st dense_4m 704 8 DP_MERKLE_WG_CAP=128 $ST
run m96 dense_4m 704 8 DP_MERKLE_WG_CAP=96
run m64_w128 dense_4m 704 8 DP_MERKLE_WG_CAP=64 DP_WIDE_WG_CAP=128
run base3 dense_4m 704 8 X=1
