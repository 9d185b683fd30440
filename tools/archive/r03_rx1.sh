#!/bin/bash
# resident executor, first contact with hardware (every step under a hard timeout: a protocol bug shows as a hang)
o=${1:-gpurun_out/r03_rx1}; mkdir -p "$o"; export TMPDIR=/tmp
export DP_TIMING=0; export DP_RX_STATS=$o/rx_stats.jsonl
timeout -s KILL 150 python tools/rx_probe.py small 8 > "$o/small8.log" 2>&1; echo "small8 rc=$?" | tee -a "$o/summary.txt"
tail -5 "$o/small8.log"
if grep -q RX_PARITY_OK "$o/small8.log"; then
  timeout -s KILL 150 python tools/rx_probe.py small 48 > "$o/small48.log" 2>&1; echo "small48 rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/small48.log"
  timeout -s KILL 300 python tools/rx_probe.py dense 64 2 1,0 > "$o/dense64.log" 2>&1; echo "dense64 rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/dense64.log"
  timeout -s KILL 400 python tools/rx_probe.py dense 256 3 1,0 > "$o/dense256.log" 2>&1; echo "dense256 rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/dense256.log"
fi
