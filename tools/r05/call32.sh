#!/bin/bash
# r05 call 32: where a CNN-264k cohort pass goes (0.67 x the Dense-4M rate): kernel trace of two 448-proof batches, host accounting
o=gpurun_out/r05_call32; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o c448 -- python "$R/tools/profile_batch.py" cnn_264k 448 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/rocpd_summary.py "$db" $o/r05_cnn448_kernel_stats.csv > $o/summary.err 2>&1; head -24 $o/r05_cnn448_kernel_stats.csv | cut -c1-120
  python tools/trace_analyze.py "$db" > $o/r05_trace_analysis_cnn448.txt 2>&1; sed -n 1,16p $o/r05_trace_analysis_cnn448.txt | cut -c1-160
fi
find $o -name '*.db' -size +2M -delete
DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 cnn_264k > $o/hoststats.txt 2>&1; grep -E "proofs/s|cohort|host phases|witness|upload" $o/hoststats.txt | tail -30 | cut -c1-300
