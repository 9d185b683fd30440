#!/bin/bash
# r06 call 18: what bounds the batch rate? The diagnostic build whose one-node-per-lane Merkle layers xor instead of hashing (72 % of the VALU instructions of a proof gone,
# same launches, same memory traffic; proofs invalid) against the release build, alternating on one box
o=gpurun_out/r06_call18; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
run rel1 dense_4m 448 6 X=1
run skip1 dense_4m 448 6 DP_LIB_VARIANT=skiphash
run rel2 dense_4m 448 6 X=1
run skip2 dense_4m 448 6 DP_LIB_VARIANT=skiphash
run skip_660 dense_4m 660 6 DP_LIB_VARIANT=skiphash DP_WORKER_ARENA_BYTES=318767104
run rel_660 dense_4m 660 6 DP_WORKER_ARENA_BYTES=318767104
run skip_224 dense_4m 224 6 DP_LIB_VARIANT=skiphash
run rel_224 dense_4m 224 6 X=1
