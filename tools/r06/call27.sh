#!/bin/bash
# r06 call 27: does a staggered start keep the cohorts out of phase, and what do the queues run side by side then? Kernel traces of 4-wave batches at 704 in flight read
# as timelines (tools/timeline_occupancy.py), default start against cohort c starting c x 28 ms late
o=gpurun_out/r06_call27; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tr() { tag=$1; shift
  cd /tmp && env "$@" timeout -s KILL 600 rocprofv3 --kernel-trace -d "$R/$o/prof_$tag" -o t -- python "$R/tools/profile_batch.py" dense_4m 704 4 > "$R/$o/prof_$tag.log" 2>&1; echo "rocprof $tag rc=$?"
  cd "$R"; tail -1 $o/prof_$tag.log | cut -c1-200
  db=$(find $o/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/timeline_occupancy.py "$db" 20 > $o/timeline_$tag.txt 2>&1 && sed -n 1,12p $o/timeline_$tag.txt | cut -c1-200
  find $o -name '*.db' -size +2M -delete
}
tr inphase X=1
tr stagger DP_COHORT_STAGGER_MS=28
