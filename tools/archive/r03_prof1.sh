#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench's Dense-4M section on the final build (throughput mode, 448 in flight)
o=${1:-gpurun_out/r03_prof1}; mkdir -p "$o"; export TMPDIR=/tmp
cd /tmp && timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$o/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-sumcheck24 --no-cnn --no-seam-level > "$GRAFT_REPO_ROOT/$o/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"; echo "rc=$? $(tail -1 $o/bench.log | cut -c1-300)"
f=$(find "$o/prof" -name "*kernel_stats.csv" | head -1); echo "stats: $f"; cp "$f" "$o/kernel_stats.csv" 2>/dev/null; head -30 "$o/kernel_stats.csv"
find "$o/prof" -name "*kernel_trace.csv" -size +60M -delete; find "$o/prof" -name "*.csv" | xargs ls -la | head
