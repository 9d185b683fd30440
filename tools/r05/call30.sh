#!/bin/bash
# r05 call 30: host threads kept on the GPU's NUMA node by the library (DP_NUMA_PIN): single-proof latency without taskset, throughput A/B
o=gpurun_out/r05_call30; mkdir -p $o; export TMPDIR=/tmp
run() { echo "$1: $(grep -E 'proof [3-5]' $2 | sed 's/.*library //' | tr '\n' ' ')"; }
timeout -s KILL 100 python tools/archive/latency_probe.py > $o/warm.txt 2>&1
for rep in 1 2; do
  DP_TIMING=1 timeout -s KILL 100 python tools/archive/latency_probe.py > $o/pin_$rep.txt 2>&1; run "pinned by the library $rep" $o/pin_$rep.txt; grep -m1 "NUMA" $o/pin_$rep.txt | cut -c1-160
  DP_NUMA_PIN=0 timeout -s KILL 100 python tools/archive/latency_probe.py > $o/nopin_$rep.txt 2>&1; run "DP_NUMA_PIN=0 $rep" $o/nopin_$rep.txt
done
timeout -s KILL 100 python tools/archive/latency_probe.py cnn_264k > $o/cnn.txt 2>&1; run "cnn pinned" $o/cnn.txt
for rep in 1 2; do
  timeout -s KILL 300 python tools/r04/ab_batch.py dense_4m 448 3 > $o/tp_pin_$rep.txt 2>&1; echo "throughput pinned $rep: $(tail -1 $o/tp_pin_$rep.txt | cut -c1-160)"
  DP_NUMA_PIN=0 timeout -s KILL 300 python tools/r04/ab_batch.py dense_4m 448 3 > $o/tp_nopin_$rep.txt 2>&1; echo "throughput DP_NUMA_PIN=0 $rep: $(tail -1 $o/tp_nopin_$rep.txt | cut -c1-160)"
done
timeout -s KILL 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_zz_cohorts.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt
