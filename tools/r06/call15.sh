#!/bin/bash
# r06 call 15: (a) kernel trace of the cohort regime read as a timeline over ALL queues (tools/timeline_occupancy.py): are the cohorts in phase? (b) host work of a member
# between two device waits by the launches issued there (DP_TIMING=3); (c) the GELU letter switch on the device
o=gpurun_out/r06_call15; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace -d "$R/$o/prof" -o b448 -- python "$R/tools/profile_batch.py" dense_4m 448 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/timeline_occupancy.py "$db" 4 > $o/timeline_448.txt 2>&1; head -16 $o/timeline_448.txt | cut -c1-200
  python tools/trace_analyze.py "$db" --sequence > $o/trace_analysis_448.txt 2>&1
fi
find $o -name '*.db' -size +2M -delete
DP_TIMING=3 timeout -s KILL 200 python tools/archive/conc_hoststats.py 448 > $o/hostwork_448.txt 2>&1; grep "proofs/s" $o/hostwork_448.txt
timeout -s KILL 600 python -m pytest tests/test_gpu_zzzzzz_gelu.py -x -q 2>&1 | tail -3
