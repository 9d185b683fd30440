#!/bin/bash
# r06 call 42: three tail workgroups per CU. The KF_PRIO forms of the one-workgroup tails compiled for three waves per SIMD (-DDP_TAIL_WPE=3: <= 168 VGPRs; k_logup_tail has 169,
# k_dense_tail 172, k_commit_tail 171) together with DP_LOGUP_LDS_KB <= 40 (three k_logup_tail workgroups of one CU inside its 160 KB): in the tails stretch of an in-phase batch
# 704 tail workgroups want a slot, two per CU (512 slots) is what 169-200 VGPRs and 64 KB + the message leave. Each knob alone was flat (call 33 had WPE=3 only together with h168, call 36 the LDS sizes).
o=gpurun_out/r06_call42; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run rel_a dense_4m 704 8 X=1
run wpe3_a dense_4m 704 8 DP_LIB_VARIANT=wpe3
run wpe3_l32_a dense_4m 704 8 DP_LIB_VARIANT=wpe3 DP_LOGUP_LDS_KB=32
run rel_l32 dense_4m 704 8 DP_LOGUP_LDS_KB=32
run wpe3_l16 dense_4m 704 8 DP_LIB_VARIANT=wpe3 DP_LOGUP_LDS_KB=16
run rel_b dense_4m 704 8 X=1
run wpe3_l32_b dense_4m 704 8 DP_LIB_VARIANT=wpe3 DP_LOGUP_LDS_KB=32
run wpe3_b dense_4m 704 8 DP_LIB_VARIANT=wpe3
run cnn_rel cnn_264k 674 4 X=1
run cnn_wpe3_l32 cnn_264k 674 4 DP_LIB_VARIANT=wpe3 DP_LOGUP_LDS_KB=32
run tf_rel transformer_layer 320 3 X=1
run tf_wpe3_l32 transformer_layer 320 3 DP_LIB_VARIANT=wpe3 DP_LOGUP_LDS_KB=32
