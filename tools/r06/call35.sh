#!/bin/bash
# r06 call 35: the query section written by the device (k_query_section) against the host form (DP_QUERY_SECTION_HOST=1): parity (GPU model / cohort / batch-commit / transformer
# tests: goldens), Dense-4M and CNN-264k rate and single-proof latency, alternating on one box
o=gpurun_out/r06_call35; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_cohorts.py tests/test_gpu_zzz_batch_commit.py tests/test_gpu_zzzzz_mha.py tests/test_gpu_c_consumer.py -m gpu -x -q > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
H=DP_QUERY_SECTION_HOST=1
run dev_a dense_4m 704 8 X=1
run host_a dense_4m 704 8 $H
run dev_b dense_4m 704 8 X=1
run host_b dense_4m 704 8 $H
run cnn_dev cnn_264k 674 4 X=1
run cnn_host cnn_264k 674 4 $H
run tf_dev transformer_layer 320 3 X=1
run tf_host transformer_layer 320 3 $H
run b64_dev dense_4m 64 8 X=1
run b64_host dense_4m 64 8 $H
