#!/bin/bash
# r06 call 41: DP_MED_THREADS — the LDS-resident kernels of a medium commit (k_med_prepare, k_med_ntt_local) as 256 / 512-thread workgroups in throughput mode instead of 1024
o=gpurun_out/r06_call41; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run cnn_1024_a cnn_264k 674 4 X=1
run cnn_256_a cnn_264k 674 4 DP_MED_THREADS=256
run cnn_512_a cnn_264k 674 4 DP_MED_THREADS=512
run cnn_1024_b cnn_264k 674 4 X=1
run cnn_256_b cnn_264k 674 4 DP_MED_THREADS=256
run cnn_128 cnn_264k 674 4 DP_MED_THREADS=128
run tf_1024 transformer_layer 320 3 X=1
run tf_256 transformer_layer 320 3 DP_MED_THREADS=256
