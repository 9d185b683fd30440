#!/bin/bash
o=${1:-gpurun_out/r03_rx8}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_gpu_zzzz_rx.py -m gpu -q -x > "$o/rx_tests.log" 2>&1; echo "rx tests rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/rx_tests.log"
run() { name=$1; conc=$2; waves=$3; mode=$4; shift; shift; shift; shift; env DP_RX_STATS=$o/stats_$name.jsonl "$@" timeout -s KILL 300 python tools/rx_probe.py dense $conc $waves $mode > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/$name.log"; }
run rx_256 256 6 1 DP_X=0
run rx_320 320 5 1 DP_X=0
run rx_384 384 4 1 DP_X=0
run rx_416 416 4 1 DP_X=0
run co_256 256 6 0 DP_X=0
run co_384 384 4 0 DP_X=0
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_rx8/stats_rx*.jsonl")):
    d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], d["busy_frac"], d["session_ms"])
P
