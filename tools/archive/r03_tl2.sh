#!/bin/bash
# round 3: rocprofv3 kernel statistics of the transformer layer at 64 x 256 (one parity proof, three single proofs, 3 x 32 proofs in a batch)
o=gpurun_out/r03_tl2; mkdir -p "$o"; export TMPDIR=/tmp GRAPH_MODEL=transformer_layer
cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -o tl -- python "$GRAFT_REPO_ROOT/tools/graph_probe.py" 64 256 4 64 32 > "$GRAFT_REPO_ROOT/$o/rocprof.log" 2>&1; echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/$o/summary.txt"
cd "$GRAFT_REPO_ROOT"; grep -E "transformer_layer|single proof" "$o/rocprof.log"
f=$(find "$o/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$o/kernel_stats.csv" && head -12 "$o/kernel_stats.csv" | cut -c1-200; rm -rf "$o/prof"; du -sh "$o"
