#!/bin/bash
# seam-level host on plain contexts in THROUGHPUT mode (dp_ctx_set_throughput_mode): parity test, then tests/support/seam_bench.c
o=${1:-gpurun_out/r03_seam3}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_zz_cohorts.py tests/test_gpu_c_consumer.py -m gpu -x -q > "$o/tests.log" 2>&1; echo "tests rc=$? $(tail -1 $o/tests.log)"
gcc -std=c11 -Wall -O2 -o /tmp/seam_bench tests/support/seam_bench.c -Ldeep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd -Iinclude 2>&1 | tail -3
export DP_ARENA_BYTES=$((2<<30))
for cfg in "14 6 0" "14 6 2" "28 4 2" "56 3 2" "112 2 2"; do
  set -- $cfg
  y=0; [ $1 -gt 14 ] && y=1
  DP_WAIT_YIELD=$y timeout -s KILL 150 /tmp/seam_bench $1 $2 $3 > "$o/seam_$1_$3.log" 2>&1; echo "T=$1 per=$2 mode=$3 yield=$y: rc=$? $(grep '^{' $o/seam_$1_$3.log | tail -1)"
done
