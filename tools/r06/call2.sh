#!/bin/bash
# r06 call 2: wave priority of every kernel that is not a bulk Poseidon2 layer (build variant prio2: -DDP_BASE_PRIO=2, kernels.inc kf_prologue / dp_hash_prio)
# against the release build (priority 0 everywhere except the one-workgroup tails), alternating on one box
o=gpurun_out/r06_call2; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; shift 3; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py $wl $n 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run rel1 dense_4m 448 X=1
run prio2_1 dense_4m 448 DP_LIB_VARIANT=prio2
run rel2 dense_4m 448 X=1
run prio2_2 dense_4m 448 DP_LIB_VARIANT=prio2
run prio2_cap512 dense_4m 448 DP_LIB_VARIANT=prio2 DP_MERKLE_WG_CAP=512
run rel_cnn cnn_264k 448 X=1
run prio2_cnn cnn_264k 448 DP_LIB_VARIANT=prio2
