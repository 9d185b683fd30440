#!/bin/bash
# kernel traces of the fewer-queues / larger-cohorts configurations (the dispatch-concurrency question)
out=${1:-gpurun_out/r02_call9}; mkdir -p "$out"; export TMPDIR=/tmp
for cfg in "32 8" "16 16"; do
  set -- $cfg; co=$1; q=$2
  DP_COHORT=$co timeout 150 rocprofv3 --kernel-trace --stats -d "$out/kt_c$co" -o x -- python tools/profile_batch.py dense_4m 256 > "$out/kt_c$co.log" 2>&1
  db=$(find "$out/kt_c$co" -name '*_results.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$out/kernel_stats_c$co.csv" > "$out/kernel_stats_c$co.txt" 2>&1 && python tools/trace_analyze.py "$db" > "$out/trace_analysis_c$co.txt" 2>&1
  [ -n "$db" ] && [ "$(stat -c %s "$db")" -gt 30000000 ] && rm -f "$db"
  grep proofs "$out/kt_c$co.log"; tail -16 "$out/trace_analysis_c$co.txt"
done
