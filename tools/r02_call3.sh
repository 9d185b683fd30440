#!/bin/bash
# kernel traces of the new throughput default (fused protocol kernels): solo durations (DP_DEVICE_FS=1 forces the device-side
# transcript for a single proof) against durations with 192 proofs in flight; launch sequence of one cohort step; host accounting
out=${1:-gpurun_out/r02_call3}; mkdir -p "$out"; export TMPDIR=/tmp
cd /tmp
DP_DEVICE_FS=1 timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/solo" -o x -- python "$GRAFT_REPO_ROOT/tools/one_proof_cwd.py" > "$GRAFT_REPO_ROOT/$out/solo.log" 2>&1
cd "$GRAFT_REPO_ROOT"
db=$(find "$out/solo" -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "$out/solo_kernel_stats.csv" > "$out/solo_kernel_stats.txt" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$out/kt" -o x -- python tools/profile_batch.py dense_4m 192 > "$out/kt.log" 2>&1
db=$(find "$out/kt" -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "$out/cohort_kernel_stats.csv" > "$out/cohort_kernel_stats.txt" 2>&1 && python tools/trace_analyze.py "$db" --sequence > "$out/trace_analysis.txt" 2>&1
[ -n "$db" ] && [ "$(stat -c %s "$db")" -gt 30000000 ] && rm -f "$db"
DP_TIMING=1 timeout 120 python tools/profile_batch.py dense_4m 192 > "$out/timing.log" 2> "$out/timing.err"
timeout 120 python tools/profile_batch.py cnn_264k 192 > "$out/cnn.log" 2>&1
tail -2 "$out/kt.log" "$out/timing.log" "$out/cnn.log"; grep "dp timing" "$out/timing.err" | tail -12 | cut -c1-300; head -30 "$out/trace_analysis.txt" | cut -c1-200
