#!/bin/bash
# r05 call 9: rocprofv3 --kernel-trace --stats of the Dense-4M cohort regime (one latency-mode proof + two batches of 448 in flight) after the LDS-assembled
# messages; queue analysis; DP_TIMING cohort accounting
o=gpurun_out/r05_call9; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o b448 -- python "$R/tools/profile_batch.py" dense_4m 448 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -2 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1); [ -n "$db" ] || { echo "no rocpd database"; exit 1; }; echo "db: $db $(stat -c %s $db)"
python tools/rocpd_summary.py "$db" $o/kernel_stats.csv > $o/summary.err 2>&1; head -14 $o/kernel_stats.csv | cut -c1-120; tail -3 $o/summary.err
python tools/trace_analyze.py "$db" > $o/trace_analysis.txt 2>&1; cat $o/trace_analysis.txt | cut -c1-200
find $o/prof -name '*.db' -size +30M -delete
DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/timing_448.txt 2>&1
grep -E "cohort:|proofs/s" $o/timing_448.txt | tail -8 | cut -c1-260
