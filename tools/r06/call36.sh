#!/bin/bash
# r06 call 36: LDS per k_logup_tail member (DP_LOGUP_LDS_KB: 64 = two members per CU): do the 704 tail workgroups of the in-phase stretch queue for CU slots?
o=gpurun_out/r06_call36; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-110)"; }
run base1 dense_4m 704 8 X=1
run l32 dense_4m 704 8 DP_LOGUP_LDS_KB=32
run l16 dense_4m 704 8 DP_LOGUP_LDS_KB=16
run l80 dense_4m 704 8 DP_LOGUP_LDS_KB=80
run base2 dense_4m 704 8 X=1
run l24 dense_4m 704 8 DP_LOGUP_LDS_KB=24
run l8 dense_4m 704 8 DP_LOGUP_LDS_KB=8
run cnn_base cnn_264k 674 4 X=1
run cnn_l32 cnn_264k 674 4 DP_LOGUP_LDS_KB=32
