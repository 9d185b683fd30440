#!/bin/bash
# r04 call 24: rocprofv3 --kernel-trace --stats of three latency-mode Dense-4M proofs (launches after the k_merkle_paths marker) — the population of bench.py's roofline
o=gpurun_out/r04_call24; mkdir -p $o; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o x -- python "$R/tools/proof_only.py" dense_4m 3 > "$R/$o/prof.log" 2>&1; echo "rc=$?"; cd "$R"
db=$(find $o/prof -name '*_results.db' | head -1); [ -n "$db" ] || { echo "no db"; tail -5 $o/prof.log; exit 1; }
python tools/r04/stats_after_marker.py "$db" k_merkle_paths $o/dense4m_latency_proofs_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/proof_only.py dense_4m 3: the launches after the k_merkle_paths marker (3 latency-mode proofs, no setup), final build of round 4 (tools/r04/call24.sh)"
grep "prove wall" $o/prof.log | tail -3
find $o -name '*_results.db' -delete
