#!/bin/bash
# round 3: the bench line with its transformer_layer section (short run), then rocprofv3 kernel statistics (csv) of the transformer layer at 64 x 256
o=gpurun_out/r03_tl3; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 150 python bench.py --steps 1 --warmup 1 --no-seam-level --no-cnn --no-sumcheck24 --no-cpu-baseline > "$o/bench.json" 2> "$o/bench.err"; echo "bench rc=$?" | tee -a "$o/summary.txt"
python - <<'P'
import json
b = json.load(open("gpurun_out/r03_tl3/bench.json")); print("value", b["value"], "transformer_layer:", b.get("transformer_layer"))
P
export GRAPH_MODEL=transformer_layer
cd /tmp && timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$o/prof" -o tl -- python "$GRAFT_REPO_ROOT/tools/graph_probe.py" 64 256 4 64 32 > "$GRAFT_REPO_ROOT/$o/rocprof.log" 2>&1; echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$o/summary.txt"
cd "$GRAFT_REPO_ROOT"; find "$o/prof" -type f | head; f=$(find "$o/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$o/kernel_stats.csv" && head -8 "$o/kernel_stats.csv" | cut -c1-160; rm -rf "$o/prof"
