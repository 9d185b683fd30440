#!/bin/bash
# r06 call 56: the pre-launch VARIANT library (tools/r06/experiments/prelaunch_fused_rounds.patch) through the whole GPU suite, and single Dense-4M / CNN-264k proofs (latency mode) with it against the release
o=gpurun_out/r06_call56; mkdir -p $o; export TMPDIR=/tmp
DP_LIB_VARIANT=prelaunch timeout -s KILL 900 python -m pytest tests -m gpu -q > $o/pytest_variant.txt 2>&1; echo "variant GPU suite rc=$?"; grep -E "passed|failed|error" $o/pytest_variant.txt | tail -3
for i in 1 2 3; do
  for v in prelaunch release; do
    if [ $v = prelaunch ]; then e="DP_LIB_VARIANT=prelaunch"; else e="X=1"; fi
    env $e timeout -s KILL 200 python tools/proof_only.py dense_4m 12 2>&1 | grep "prove wall" | awk '{print $4}' | sort -n | tr '\n' ' ' > $o/d4m_${v}_$i.txt; echo "dense_4m $v $i: $(cat $o/d4m_${v}_$i.txt)"
  done
done
for v in prelaunch release; do
  if [ $v = prelaunch ]; then e="DP_LIB_VARIANT=prelaunch"; else e="X=1"; fi
  env $e timeout -s KILL 200 python tools/proof_only.py cnn_264k 8 2>&1 | grep "prove wall" | awk '{print $4}' | sort -n | tr '\n' ' ' > $o/cnn_${v}.txt; echo "cnn_264k $v: $(cat $o/cnn_${v}.txt)"
done
