#!/bin/bash
# resident executor after the download fix: parity (probe2), then Dense-4M throughput against cohorts with the per-body accounting
o=${1:-gpurun_out/r03_rx3}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_default.log" 2>&1; echo "probe_default rc=$?" | tee -a "$o/summary.txt"; tail -11 "$o/probe_default.log"
if grep -q RX_PARITY_OK "$o/probe_default.log"; then
  env DP_DEVICE_LOGUP=0 DP_DEVICE_CLASSIC=0 DP_DEVICE_DENSE=0 DP_DEVICE_EQSUM=0 DP_DEVICE_COMMIT=0 timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_nofused.log" 2>&1; echo "probe_nofused rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/probe_nofused.log"
  timeout -s KILL 200 python tools/rx_probe2.py 256 6 > "$o/probe_w256.log" 2>&1; echo "probe_w256 rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/probe_w256.log"
  DP_RX_STATS=$o/rx_stats_dense64.jsonl timeout -s KILL 300 python tools/rx_probe.py dense 64 3 1,0 > "$o/dense64.log" 2>&1; echo "dense64 rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/dense64.log"
  DP_RX_STATS=$o/rx_stats_dense256.jsonl timeout -s KILL 400 python tools/rx_probe.py dense 256 4 1,0 > "$o/dense256.log" 2>&1; echo "dense256 rc=$?" | tee -a "$o/summary.txt"; tail -3 "$o/dense256.log"
else
  for k in DP_DEVICE_LOGUP DP_DEVICE_DENSE DP_DEVICE_EQSUM DP_DEVICE_CLASSIC DP_DEVICE_COMMIT; do
    env $k=0 timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/probe_$k.log" 2>&1; echo "$k=0 rc=$?" | tee -a "$o/summary.txt"; tail -10 "$o/probe_$k.log" | head -9
  done
fi
