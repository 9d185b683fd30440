#!/bin/bash
# r04 call 14: after the knob merge (DP_FUSED_OFF) and prune: GPU suite; transformer layer with the big lookups sent through chip-wide kernels (DP_LOGUP_TAIL_MAX_N)
o=gpurun_out/r04_call14; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/pytest_gpu.txt | tail -2
for n in 65536 16384 4096 1024; do
  DP_LOGUP_TAIL_MAX_N=$n GRAPH_MODEL=transformer_layer GRAPH_NO_ORACLE=1 timeout -s KILL 300 python tools/graph_probe.py 64 256 4 64 192 > $o/tl_maxn_$n.txt 2>&1
  echo "DP_LOGUP_TAIL_MAX_N=$n: $(grep -E 'single proof' $o/tl_maxn_$n.txt | cut -c1-200)"
done
GRAPH_MODEL=transformer_layer GRAPH_NO_ORACLE=1 timeout -s KILL 300 python tools/graph_probe.py 64 256 4 64 320 > $o/tl_320.txt 2>&1; echo "320 in flight: $(grep -E 'single proof' $o/tl_320.txt | cut -c1-200)"
