#!/bin/bash
# r05 call 7: k_logup_tail with its message assembled in LDS and sent in one burst at the end (no PCIe-acknowledged store inside the rounds): member timing
# (entry -> exit) against the round-4 kernel, parity of the fused kernels, the A/B probe
o=gpurun_out/r05_call7; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q > $o/pytest_fused.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest_fused.txt
for rep in 1 2; do
for v in wgtimes wgtimesv1; do
  DP_LIB_VARIANT=$v DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/${v}_$rep.txt 2>&1
  echo "$v $rep rc=$?"; grep -E "wg-times|proofs/s" $o/${v}_$rep.txt | tail -2 | cut -c1-460
done
done
for rep in 1 2; do
  for v in release v1; do
    if [ $v = release ]; then unset DP_LIB_VARIANT; else export DP_LIB_VARIANT=$v; fi
    timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_${v}_$rep.txt 2>&1; echo "$v $rep rc=$? $(tail -1 $o/ab_${v}_$rep.txt | cut -c1-200)"
  done
done
