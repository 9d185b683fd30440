#!/bin/bash
# resident executor, tuning round: urgent ring + step trace + grid-cap sweep (Dense-4M, 256 in flight)
o=${1:-gpurun_out/r03_rx4}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_default.log" 2>&1; echo "probe_default rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/probe_default.log"
grep -q RX_PARITY_OK "$o/probe_default.log" || exit 0
run() { name=$1; shift; env DP_RX_STATS=$o/stats_$name.jsonl DP_RX_TRACE=8 DP_RX_TRACE_FILE=$o/trace_$name.txt "$@" timeout -s KILL 300 python tools/rx_probe.py dense 256 3 1 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run cap256 DP_X=0
run cap1024 DP_RX_GRID_CAP=1024
run cap64 DP_RX_GRID_CAP=64
run stream3 DP_RX_STREAM_PER_CU=3
run threads8 DP_HOST_THREADS=8
timeout -s KILL 300 python tools/rx_probe.py dense 256 3 0 > "$o/cohorts.log" 2>&1; tail -1 "$o/cohorts.log"
