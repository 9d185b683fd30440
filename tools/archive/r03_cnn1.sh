#!/bin/bash
o=${1:-gpurun_out/r03_cnn1}; mkdir -p "$o"; export TMPDIR=/tmp
DP_RX_STATS=$o/rx_stats_cnn.jsonl DP_RX_TRACE=8 DP_RX_TRACE_FILE=$o/rx_trace_cnn.txt timeout -s KILL 400 python tools/cnn_stretch.py 256 > "$o/cnn.log" 2>&1; echo "cnn rc=$?"; tail -5 "$o/cnn.log"
DP_TIMING=1 DP_LAUNCH_NAMES=1 timeout -s KILL 300 python tools/cnn_stretch.py 16 > "$o/cnn_timing.log" 2>&1; grep "dp launches\|dp timing\] device context" "$o/cnn_timing.log" | head -70
