#!/bin/bash
# r06 call 4: kernel traces of the cohort regime, default against DP_WIDE_WG_CAP=128 DP_MERKLE_WG_CAP=512 (what do the caps change per kernel?)
o=gpurun_out/r06_call4; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tr() { tag=$1; shift
  cd /tmp && env "$@" timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof_$tag" -o b448 -- python "$R/tools/profile_batch.py" dense_4m 448 > "$R/$o/prof_$tag.log" 2>&1; echo "rocprof $tag rc=$?"
  cd "$R"; tail -1 $o/prof_$tag.log | cut -c1-200
  db=$(find $o/prof_$tag -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python tools/rocpd_summary.py "$db" $o/kernel_stats_$tag.csv > $o/summary_$tag.err 2>&1
    python tools/trace_analyze.py "$db" > $o/trace_analysis_$tag.txt 2>&1; sed -n 1,24p $o/trace_analysis_$tag.txt | cut -c1-200
  fi
  find $o -name '*.db' -size +2M -delete
}
tr base X=1
tr caps DP_WIDE_WG_CAP=128 DP_MERKLE_WG_CAP=512
