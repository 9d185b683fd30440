#!/bin/bash
# r06 call 51 (final sources, csrc bb19fdd1fc305af2): plain rocprofv3 --kernel-trace --stats (no counters) of three latency-mode Dense-4M proofs and of five 2^24 sumchecks: the
# per-kernel durations the bench line's `roofline.avg_launch_us` (HIP events) has to agree with
o=gpurun_out/r06_call51; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$R/$o/proofs" -o x -- python "$R/tools/proof_only.py" dense_4m 3 > "$R/$o/proofs.log" 2>&1; echo "proofs rc=$?"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$R/$o/sc24" -o x -- python "$R/tools/sumcheck24_only.py" 5 > "$R/$o/sc24.log" 2>&1; echo "sc24 rc=$?"
cd "$R"
db=$(find $o/proofs -name "*.db" | head -1); [ -n "$db" ] && python tools/r04/stats_after_marker.py "$db" k_merkle_paths $o/r06_dense4m_latency_proofs_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/proof_only.py dense_4m 3 (launches after the k_merkle_paths marker; final sources, tools/r06/call51.sh)" > $o/s1.txt 2>&1; head -12 $o/r06_dense4m_latency_proofs_kernel_stats.csv | cut -c1-120
db=$(find $o/sc24 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" $o/r06_sumcheck24_kernel_stats.csv > $o/s2.txt 2>&1; head -8 $o/r06_sumcheck24_kernel_stats.csv | cut -c1-120
tail -3 $o/s1.txt | cut -c1-200
find $o -name '*.db' -size +2M -delete
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$R/$o/sc26" -o x -- python "$R/tools/sumcheck24_only.py" 5 26 > "$R/$o/sc26.log" 2>&1; echo "sc26 rc=$?"
cd "$R"
db=$(find $o/sc26 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" $o/r06_sumcheck26_kernel_stats.csv > $o/s3.txt 2>&1; head -8 $o/r06_sumcheck26_kernel_stats.csv | cut -c1-120
find $o -name '*.db' -size +2M -delete
