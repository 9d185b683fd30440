#!/bin/bash
# resident executor: staggered starts + fat classic tiles + default cap 32 — sweep
o=${1:-gpurun_out/r03_rx6}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_default.log" 2>&1; echo "probe_default rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/probe_default.log"
grep -q RX_PARITY_OK "$o/probe_default.log" || exit 0
run() { name=$1; conc=$2; waves=$3; shift; shift; shift; env DP_RX_STATS=$o/stats_$name.jsonl DP_RX_TRACE=8 DP_RX_TRACE_FILE=$o/trace_$name.txt "$@" timeout -s KILL 300 python tools/rx_probe.py dense $conc $waves 1 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run default_w3 256 3 DP_X=0
run default_w6 256 6 DP_X=0
run nostagger_w6 256 6 DP_RX_STAGGER_MS=0
run stagger800_w6 256 6 DP_RX_STAGGER_MS=800
run cap16_w6 256 6 DP_RX_GRID_CAP=16
run stream3_w6 256 6 DP_RX_STREAM_PER_CU=3
run inflight384_w4 384 4 DP_X=0
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_rx6/stats_*.jsonl")):
    d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], d["busy_frac"], d["session_ms"])
P
