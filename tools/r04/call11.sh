#!/bin/bash
# r04 call 11: async engine accounting (queue time / body time per call, client-thread time), persistent-buffer pool; SQ instruction pass
o=gpurun_out/r04_call11; mkdir -p $o tests/support/_build; export TMPDIR=/tmp
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd || exit 1
DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 14 6 0 > $o/seam_blocking.txt 2>&1; echo "blocking 14 threads: $(tail -1 $o/seam_blocking.txt | cut -c1-200)"
for g in 1 3; do for n in 64 256; do
  DP_TIMING=1 DP_ASYNC_GROUPS_PER_THREAD=$g DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench $n 3 3 > $o/seam_${n}_g$g.txt 2>&1
  echo "async $n in flight, $g groups/thread: $(grep -E 'seam_level|async engine' $o/seam_${n}_g$g.txt | cut -c1-420)"
done; done
DP_TIMING=1 DP_ASYNC_GROUP=1 DP_ASYNC_GROUPS_PER_THREAD=24 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_nomerge.txt 2>&1; echo "no merging, 24 calls per thread: $(grep -E 'seam_level|async engine' $o/seam_nomerge.txt | cut -c1-420)"
timeout -s KILL 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $o/sq -o x -- python tools/profile_batch.py dense_4m 448 > $o/sq.log 2>&1; echo "sq rc=$?"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $o/sqp -o x -- python tools/r04/probe_compress.py > $o/sqp.log 2>&1; echo "sqp rc=$?"
f=$(find $o/sq -name '*_results.db' | head -1); g=$(find $o/sqp -name '*_results.db' | head -1)
[ -n "$f" ] && python tools/pmc_sq_job.py "$f" 896 556.3 $o/pmc_sq_bench448.json "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python tools/profile_batch.py dense_4m 448 (cohort launches of the two 448-proof batches)" "$g" 2097152 > $o/pmc_sq.txt 2>&1
head -16 $o/pmc_sq.txt | cut -c1-250; grep -E "proofs/s" $o/sq.log | cut -c1-200; tail -2 $o/sqp.log | cut -c1-200
find $o -name '*.db' -size +8M -delete
