#!/bin/bash
# r05 call 35: lookups of >= 2048 rows on a 512-thread throughput-mode form of k_logup_tail (DP_LOGUP_WIDE_N=2048) against the 256-thread form for all (=0):
# parity (fused kernels, models, cohorts), then the three workloads alternating on one box
o=gpurun_out/r05_call35; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_zz_cohorts.py tests/test_gpu_zzzzz_mha.py tests/test_gpu_zzzzzz_gelu.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.txt | cut -c1-200
for rep in 1 2; do
  for wl in "cnn_264k 448 3" "transformer_layer 320 2" "dense_4m 448 3"; do
    set -- $wl
    timeout -s KILL 300 python tools/r04/ab_batch.py $1 $2 $3 > $o/${1}_wide_$rep.txt 2>&1; echo "$1 wide $rep: $(tail -1 $o/${1}_wide_$rep.txt | cut -c1-130)"
    DP_LOGUP_WIDE_N=0 timeout -s KILL 300 python tools/r04/ab_batch.py $1 $2 $3 > $o/${1}_256_$rep.txt 2>&1; echo "$1 DP_LOGUP_WIDE_N=0 $rep: $(tail -1 $o/${1}_256_$rep.txt | cut -c1-130)"
  done
done
