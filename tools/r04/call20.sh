#!/bin/bash
# r04 call 20: the short polynomials of a batch opening summed among their own length first (k_axpy_classes, DP_AXPY_CLASSES=1 default): parity tests, then A/B against DP_AXPY_CLASSES=0
o=gpurun_out/r04_call20; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_primitives.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.txt | cut -c1-200
for rep in 1 2; do
  for v in 1 0; do
    DP_AXPY_CLASSES=$v timeout -s KILL 120 python tools/r04/ab_batch.py dense_4m 448 3 2>> $o/ab.err | sed "s/^/DP_AXPY_CLASSES=$v /" >> $o/ab.txt
  done
done
for v in 1 0; do
  DP_AXPY_CLASSES=$v timeout -s KILL 120 python tools/r04/ab_batch.py transformer_layer 320 3 2>> $o/ab.err | sed "s/^/DP_AXPY_CLASSES=$v /" >> $o/ab.txt
  DP_AXPY_CLASSES=$v timeout -s KILL 120 python tools/r04/ab_batch.py cnn_264k 448 3 2>> $o/ab.err | sed "s/^/DP_AXPY_CLASSES=$v /" >> $o/ab.txt
done
cat $o/ab.txt | cut -c1-260; tail -5 $o/ab.err | cut -c1-300
