#!/bin/bash
o=${1:-gpurun_out/r03_seam1}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 500 python -m pytest tests/test_gpu_zzzz_rx.py -m gpu -q -x > "$o/rx_tests.log" 2>&1; echo "rx tests rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/rx_tests.log"
export DP_ARENA_BYTES=2147483648
B=tests/support/_build/seam_bench
timeout -s KILL 200 $B 4 2 1 > "$o/seam_t4_x.log" 2>&1; echo "t4 executor rc=$?"; tail -1 "$o/seam_t4_x.log"
timeout -s KILL 200 $B 14 6 1 > "$o/seam_t14_x.log" 2>&1; echo "t14 executor rc=$?"; tail -1 "$o/seam_t14_x.log"
timeout -s KILL 200 $B 14 6 0 > "$o/seam_t14_s.log" 2>&1; echo "t14 streams rc=$?"; tail -1 "$o/seam_t14_s.log"
DP_WAIT_YIELD=1 timeout -s KILL 240 $B 48 4 1 > "$o/seam_t48_x.log" 2>&1; echo "t48 executor rc=$?"; tail -1 "$o/seam_t48_x.log"
DP_WAIT_YIELD=1 timeout -s KILL 240 $B 48 4 0 > "$o/seam_t48_s.log" 2>&1; echo "t48 streams rc=$?"; tail -1 "$o/seam_t48_s.log"
