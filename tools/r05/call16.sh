#!/bin/bash
# r05 call 16: lookup-table columns resident per context, uploads above 2 MB through the asynchronous staging ring: the whole GPU suite, then the A/B probe on the three workloads
o=gpurun_out/r05_call16; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -3
for wl in "dense_4m 448 3" "cnn_264k 448 2" "transformer_layer 320 2" "dense_4m 448 3"; do
  set -- $wl
  timeout -s KILL 300 python tools/r04/ab_batch.py $1 $2 $3 > $o/ab_$1_$$.txt 2>&1; echo "$1 rc=$? $(tail -1 $o/ab_$1_$$.txt | cut -c1-220)"
done
