#!/bin/bash
# resident executor, after the worker-placement fix (BIG first, one per CU): parity, alive counts, throughput vs grid cap
o=${1:-gpurun_out/r03_rx5}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_default.log" 2>&1; echo "probe_default rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/probe_default.log"
grep -q RX_PARITY_OK "$o/probe_default.log" || exit 0
run() { name=$1; waves=$2; shift; shift; env DP_RX_STATS=$o/stats_$name.jsonl DP_RX_TRACE=8 DP_RX_TRACE_FILE=$o/trace_$name.txt "$@" timeout -s KILL 300 python tools/rx_probe.py dense 256 $waves 1 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run cap64 3 DP_RX_GRID_CAP=64
run cap32 3 DP_RX_GRID_CAP=32
run cap128 3 DP_RX_GRID_CAP=128
run cap64_w6 6 DP_RX_GRID_CAP=64
timeout -s KILL 300 python tools/rx_probe.py dense 256 3 0 > "$o/cohorts.log" 2>&1; tail -1 "$o/cohorts.log"
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_rx5/stats_*.jsonl")):
    d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], d["alive_per_xcd_stream_big"], d["busy_frac"])
P
