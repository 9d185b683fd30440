#!/bin/bash
# the default bench line under rocprofv3 --kernel-trace --stats + the cohort-regime trace analysis (tools/trace_analyze.py)
o=${1:-gpurun_out/r02_trace}; mkdir -p "$o"; export TMPDIR=/tmp
DP_BENCH_NO_TORCH=1 timeout 420 rocprofv3 --kernel-trace --stats -d "$o/bench_kt" -o x -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$o/bench_under_rocprof.json" 2> "$o/bench_under_rocprof.err"
db=$(find "$o/bench_kt" -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "$o/bench_kernel_stats.csv" > "$o/bench_kernel_stats.txt" 2>&1 && python tools/trace_analyze.py "$db" > "$o/bench_trace_analysis.txt" 2>&1
[ -n "$db" ] && rm -f "$db"
tail -2 "$o/bench_under_rocprof.err" | cut -c1-200; head -c 300 "$o/bench_under_rocprof.json"; echo; cat "$o/bench_trace_analysis.txt"
