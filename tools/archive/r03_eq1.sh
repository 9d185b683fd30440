#!/bin/bash
# factored eq tables in the batch-opening sumcheck (DP_CLASSIC_EQ_SPLIT): parity, then throughput / footprint A/B on the cohort path
o=${1:-gpurun_out/r03_eq1}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_cohorts.py -m gpu -x -q > "$o/tests_model.log" 2>&1; echo "tests model rc=$? $(tail -1 $o/tests_model.log)"
timeout -s KILL 600 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "batch_open or pcs or independent" > "$o/tests_prim.log" 2>&1; echo "tests prim rc=$? $(tail -1 $o/tests_prim.log)"
for split in 0 1; do
  DP_CLASSIC_EQ_SPLIT=$split DP_TIMING=1 timeout -s KILL 200 python tools/rx_probe.py dense 32 1 0 2>&1 | grep -E "arena peak|proofs/s" | tail -3 > "$o/peak_$split.log"; echo "split $split: $(tr '\n' ' ' < $o/peak_$split.log)"
done
for rep in 1 2; do
  for split in 0 1; do
    DP_CLASSIC_EQ_SPLIT=$split timeout -s KILL 200 python tools/rx_probe.py dense 256 6 0 > "$o/ab_${split}_$rep.log" 2>&1; echo "split $split rep $rep: $(tail -1 $o/ab_${split}_$rep.log)"
  done
done
for rep in 1 2; do
  timeout -s KILL 200 python tools/rx_probe.py dense 448 5 0 > "$o/c448_$rep.log" 2>&1; echo "448 in flight rep $rep: $(tail -1 $o/c448_$rep.log)"
done
