#!/bin/bash
# r06 call 19: DP_HASH_GATE — at most L cohorts build a large Merkle tree at a time — with the merged hash launches capped (DP_MERKLE_WG_CAP) so that the hashing of those
# inside leaves wave slots to the latency-bound chains of the others (call 18: without the hash layers the batch runs at 1 330-1 380 proofs/s, with them at 900-930)
o=gpurun_out/r06_call19; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
run base1 dense_4m 448 6 X=1
run h1 dense_4m 448 6 DP_HASH_GATE=1
run h2 dense_4m 448 6 DP_HASH_GATE=2
run h4 dense_4m 448 6 DP_HASH_GATE=4
run h1_c1024 dense_4m 448 6 DP_HASH_GATE=1 DP_MERKLE_WG_CAP=1024
run h1_c768 dense_4m 448 6 DP_HASH_GATE=1 DP_MERKLE_WG_CAP=768
run base2 dense_4m 448 6 X=1
run h2_c512 dense_4m 448 6 DP_HASH_GATE=2 DP_MERKLE_WG_CAP=512
run h2_c1024 dense_4m 448 6 DP_HASH_GATE=2 DP_MERKLE_WG_CAP=1024
run h3_c512 dense_4m 448 6 DP_HASH_GATE=3 DP_MERKLE_WG_CAP=512
run h1_c1536 dense_4m 448 6 DP_HASH_GATE=1 DP_MERKLE_WG_CAP=1536
run h8 dense_4m 448 6 DP_HASH_GATE=8
run base3 dense_4m 448 6 X=1
DP_HASH_GATE=1 DP_MERKLE_WG_CAP=1024 DP_TIMING=1 timeout -s KILL 200 python tools/archive/conc_hoststats.py 448 2>&1 | grep -E "gate|proofs/s|cohort:" | tail -6
