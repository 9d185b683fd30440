#!/bin/bash
# r06 call 28: DP_HASH_STREAMS = N with a CU mask — the wide hash layers of all cohorts on N streams that may use c of the 256 CUs (the others never hold a hash wave: a
# protocol tail, which needs 154-200 VGPRs on all four SIMDs of one CU, always finds room there), hash grids uncapped on those streams
o=gpurun_out/r06_call28; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run base1 dense_4m 704 8 X=1
run hs2_192 dense_4m 704 8 DP_HASH_STREAMS=2 DP_HASH_STREAM_CUS=192
run hs2_224 dense_4m 704 8 DP_HASH_STREAMS=2 DP_HASH_STREAM_CUS=224
run hs2_160 dense_4m 704 8 DP_HASH_STREAMS=2 DP_HASH_STREAM_CUS=160
run hs1_192 dense_4m 704 8 DP_HASH_STREAMS=1 DP_HASH_STREAM_CUS=192
run hs2_256 dense_4m 704 8 DP_HASH_STREAMS=2 DP_HASH_STREAM_CUS=256
run base2 dense_4m 704 8 X=1
run hs4_192 dense_4m 704 8 DP_HASH_STREAMS=4 DP_HASH_STREAM_CUS=192 GPU_MAX_HW_QUEUES=26
run hs2_192_st dense_4m 704 8 DP_HASH_STREAMS=2 DP_HASH_STREAM_CUS=192 DP_COHORT_STAGGER_MS=28
