#!/bin/bash
# r06 call 39: the transcript's sponge served by the host (DP_HOST_SPONGE=1, sponge_host.h) in the cohort regime at 704 in flight: a single proof's chain of fused tails is 49 ms
# with it against 65 ms with the wave sponge (call 38) — does the batch rate see any of it?
o=gpurun_out/r06_call39; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-110)"; }
run base1 dense_4m 704 8 X=1
run hs1 dense_4m 704 8 DP_HOST_SPONGE=1
run hs2 dense_4m 704 8 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=2
run hs4 dense_4m 704 8 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=4
run base2 dense_4m 704 8 X=1
run hs8 dense_4m 704 8 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=8
run b64 dense_4m 64 8 X=1
run b64_hs2 dense_4m 64 8 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=2
run b64_hs4 dense_4m 64 8 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=4
