#!/bin/bash
# r05 call 11: argument-pack rings and the descriptor ring mapped non-coherent (cacheable on the GPU) against coherent (DP_HOST_NC=0), alternating on one box
o=gpurun_out/r05_call11; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  for nc in 1 0; do
    DP_HOST_NC=$nc timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_nc${nc}_$rep.txt 2>&1; echo "DP_HOST_NC=$nc $rep rc=$? $(tail -1 $o/ab_nc${nc}_$rep.txt | cut -c1-200)"
  done
done
for nc in 1 0; do
  DP_HOST_NC=$nc timeout -s KILL 200 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/ab_cnn_nc${nc}.txt 2>&1; echo "DP_HOST_NC=$nc rc=$? $(tail -1 $o/ab_cnn_nc${nc}.txt | cut -c1-200)"
done
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_cohorts.py tests/test_gpu_fused.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt
