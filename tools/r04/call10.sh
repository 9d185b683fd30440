#!/bin/bash
# r04 call 10: async engine with several groups per thread (seam_bench), async parity tests, DP_WIDE_LDS on the Merkle kernels only, SQ instruction pass
o=gpurun_out/r04_call10; mkdir -p $o tests/support/_build; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_zz_async.py -m gpu -x -q > $o/pytest_async.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_async.txt
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd || exit 1
for cfg in "64 4 3" "128 3 3" "256 3 3" "384 2 3"; do
  set -- $cfg
  DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench $1 $2 $3 > $o/seam_$1_$3.txt 2>&1; echo "seam_bench $cfg: rc=$? $(tail -1 $o/seam_$1_$3.txt | cut -c1-300)"
done
DP_ASYNC_GROUPS_PER_THREAD=6 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_256_g6.txt 2>&1; echo "groups/thread 6: $(tail -1 $o/seam_256_g6.txt | cut -c1-300)"
DP_ASYNC_GROUPS_PER_THREAD=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_256_g1.txt 2>&1; echo "groups/thread 1: $(tail -1 $o/seam_256_g1.txt | cut -c1-300)"
for w in "0 0" "36864 1" "36864 1" "0 0"; do
  set -- $w
  DP_WIDE_LDS=$1 DP_WIDE_LDS_MERKLE_ONLY=$2 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wide_$1.txt 2>&1; echo "DP_WIDE_LDS=$1 merkle-only=$2: $(grep -E 'proofs/s' $o/wide_$1.txt | tail -1 | cut -c1-60)"
done
cd /tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OLDPWD/$o/sq -o x -- python $OLDPWD/tools/profile_batch.py dense_4m 448 > $OLDPWD/$o/sq.log 2>&1; echo "sq rc=$?"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OLDPWD/$o/sqp -o x -- python $OLDPWD/tools/r04/probe_compress.py > $OLDPWD/$o/sqp.log 2>&1; echo "sqp rc=$?"
cd $OLDPWD
f=$(find $o/sq -name '*_results.db' | head -1); g=$(find $o/sqp -name '*_results.db' | head -1)
rate=$(grep -E 'proofs/s' $o/wide_0.txt | tail -1 | sed 's/.*conc=448 \([0-9.]*\) proofs.*/\1/')
[ -n "$f" ] && python tools/pmc_sq_job.py "$f" 896 "$rate" $o/pmc_sq_bench448.json "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python tools/profile_batch.py dense_4m 448 (cohort launches of the two 448-proof batches)" "$g" 2097152 > $o/pmc_sq.txt 2>&1
head -16 $o/pmc_sq.txt | cut -c1-250; tail -3 $o/sq.log | cut -c1-200
find $o -name '*.db' -size +8M -delete
