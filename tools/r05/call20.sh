#!/bin/bash
# r05 call 20: finished proofs serialised and copied out by helper threads (DP_SER_THREADS=2, the default) against inline on the cohort's thread (0), alternating
o=gpurun_out/r05_call20; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  for k in 2 0; do
    DP_SER_THREADS=$k timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_ser${k}_$rep.txt 2>&1; echo "DP_SER_THREADS=$k $rep rc=$? $(tail -1 $o/ab_ser${k}_$rep.txt | cut -c1-200)"
  done
done
for k in 2 0; do
  DP_SER_THREADS=$k timeout -s KILL 200 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/ab_cnn_ser$k.txt 2>&1; echo "cnn DP_SER_THREADS=$k: $(tail -1 $o/ab_cnn_ser$k.txt | cut -c1-160)"
  DP_SER_THREADS=$k timeout -s KILL 200 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/ab_tl_ser$k.txt 2>&1; echo "tl DP_SER_THREADS=$k: $(tail -1 $o/ab_tl_ser$k.txt | cut -c1-160)"
done
timeout -s KILL 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_cohorts.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt
