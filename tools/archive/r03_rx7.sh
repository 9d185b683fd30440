#!/bin/bash
o=${1:-gpurun_out/r03_rx7}; mkdir -p "$o"; export TMPDIR=/tmp
run() { name=$1; conc=$2; waves=$3; shift; shift; shift; env DP_RX_STATS=$o/stats_$name.jsonl "$@" timeout -s KILL 300 python tools/rx_probe.py dense $conc $waves 1 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run cap8_w6 256 6 DP_RX_GRID_CAP=8
run cap4_w6 256 6 DP_RX_GRID_CAP=4
run cap12_w6 256 6 DP_RX_GRID_CAP=12
run cap8_320 320 5 DP_RX_GRID_CAP=8
run cap8_192 192 8 DP_RX_GRID_CAP=8
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_rx7/stats_*.jsonl")):
    d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], d["busy_frac"], d["session_ms"], round(sum(b["total_ms"] for b in d["bodies"])/1536,1))
P
