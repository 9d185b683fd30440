#!/bin/bash
# round 3: the whole transformer layer (models.transformer_layer: LayerNorm, QKV, the Mha node, projection, feed-forward half) on the device at two
# shapes: parity, single-proof latency, batch throughput; the per-layer timing of one proof; rocprofv3 kernel statistics of the larger shape
o=gpurun_out/r03_tl1; mkdir -p "$o"; export TMPDIR=/tmp GRAPH_MODEL=transformer_layer
timeout -s KILL 90 python tools/graph_probe.py 16 64 4 16 128 > "$o/probe_16x64.txt" 2>&1; echo "probe 16x64 rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/probe_16x64.txt"
timeout -s KILL 150 python tools/graph_probe.py 64 256 4 64 64 > "$o/probe_64x256.txt" 2>&1; echo "probe 64x256 rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/probe_64x256.txt"
DP_TIMING=1 timeout -s KILL 60 python - > "$o/timing_64x256.txt" 2>&1 <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import deep_prove_amd as dpa
g = dpa.models.transformer_layer(64, 256, 4, 64, 1024, config=66)
dev = dpa.Device(0); ctx = dpa.Context.generate(dev, g.blob()); pr = dpa.Prover(ctx)
pr.prove(g.input()); print("---- second proof ----", file=sys.stderr, flush=True); pr.prove(g.input(7))
P
echo "timing rc=$?" | tee -a "$o/summary.txt"; grep -c "dp timing" "$o/timing_64x256.txt"
cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -o tl -- python "$GRAFT_REPO_ROOT/tools/graph_probe.py" 64 256 4 64 32 > "$GRAFT_REPO_ROOT/$o/rocprof.log" 2>&1; echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$o/summary.txt"
cd "$GRAFT_REPO_ROOT"; f=$(find "$o/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" > "$o/kernel_stats_top.csv"; find "$o/prof" -type f ! -name "*stats*" -delete 2>/dev/null; du -sh "$o"
