#!/bin/bash
# r06 call 17: (a) the query section assembled straight from the gather image (proof.h queries_ser): A/B is against call 16's base on another box, so the sha256 and the
# host-work accounting are what counts; (b) ONE cohort alone on the chip (21 proofs in flight): the solo duration of every merged launch = the chip time a cohort needs
o=gpurun_out/r06_call17; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run new1 dense_4m 448 6 X=1
run new2 dense_4m 448 6 X=1
run new12 dense_4m 448 12 X=1
run new_cnn cnn_264k 448 4 X=1
DP_TIMING=3 timeout -s KILL 200 python tools/archive/conc_hoststats.py 448 > $o/hostwork_448.txt 2>&1; grep "proofs/s" $o/hostwork_448.txt
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace -d "$R/$o/prof" -o one -- python "$R/tools/profile_batch.py" dense_4m 21 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" $o/one_cohort_kernel_stats.csv && python tools/trace_analyze.py "$db" --sequence > $o/one_cohort_trace.txt 2>&1
find $o -name '*.db' -size +2M -delete
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_zzz_batch_commit.py tests/test_gpu_zz_cohorts.py -x -q 2>&1 | tail -3
