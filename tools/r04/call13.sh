#!/bin/bash
# r04 call 13: PCS::commit(&poly) as one ticket (upload + commit): async parity test, seam bench from one thread
o=gpurun_out/r04_call13; mkdir -p $o tests/support/_build; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_zz_async.py tests/test_gpu_c_consumer.py -m gpu -x -q > $o/pytest_async.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_async.txt | cut -c1-300
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd || exit 1
for n in 64 128 256 384 512; do
  DP_TIMING=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench $n 3 3 > $o/seam_async_$n.txt 2>&1
  echo "async $n in flight: $(grep -E 'seam_level|async engine' $o/seam_async_$n.txt | cut -c1-460)"
done
DP_ASYNC_LINGER_US=300 DP_TIMING=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_async_256_l300.txt 2>&1; echo "linger 300: $(grep -E 'seam_level' $o/seam_async_256_l300.txt | cut -c1-300)"
DP_ASYNC_GROUP=64 DP_TIMING=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_async_256_g64.txt 2>&1; echo "group 64: $(grep -E 'seam_level' $o/seam_async_256_g64.txt | cut -c1-300)"
DP_HOST_THREADS=22 DP_TIMING=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 256 3 3 > $o/seam_async_256_t22.txt 2>&1; echo "22 engine threads: $(grep -E 'seam_level' $o/seam_async_256_t22.txt | cut -c1-300)"
