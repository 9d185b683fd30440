#!/bin/bash
# round 3: DP_LOGUP_TAIL_MAX_N sweep on the transformer layer at 64 x 256, 64 in flight: where do long lookups stop paying for the one-workgroup tail?
o=gpurun_out/r03_lt1; mkdir -p "$o"; export TMPDIR=/tmp GRAPH_MODEL=transformer_layer
DP_LOGUP_TAIL_MAX_N=4096 timeout -s KILL 60 python tools/graph_probe.py 64 256 4 64 64 > "$o/max_4096.txt" 2>&1; echo "4096 (with oracle parity) rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/max_4096.txt"
for m in 16384 1024; do
  GRAPH_NO_ORACLE=1 DP_LOGUP_TAIL_MAX_N=$m timeout -s KILL 40 python tools/graph_probe.py 64 256 4 64 64 > "$o/max_$m.txt" 2>&1; echo "$m rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/max_$m.txt"
done
