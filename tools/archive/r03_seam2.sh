#!/bin/bash
o=${1:-gpurun_out/r03_seam2}; mkdir -p "$o"; export TMPDIR=/tmp
export DP_ARENA_BYTES=2147483648
B=tests/support/_build/seam_bench
timeout -s KILL 60 $B 4 2 1 > "$o/seam_t4_x.log" 2>&1; rc=$?; echo "t4 executor rc=$rc"; tail -2 "$o/seam_t4_x.log"
[ $rc -eq 0 ] || exit 0
timeout -s KILL 90 $B 14 6 1 > "$o/seam_t14_x.log" 2>&1; echo "t14 executor rc=$?"; tail -1 "$o/seam_t14_x.log"
DP_WAIT_YIELD=1 timeout -s KILL 120 $B 48 4 1 > "$o/seam_t48_x.log" 2>&1; echo "t48 executor rc=$?"; tail -1 "$o/seam_t48_x.log"
DP_WAIT_YIELD=1 timeout -s KILL 150 $B 128 3 1 > "$o/seam_t128_x.log" 2>&1; echo "t128 executor rc=$?"; tail -1 "$o/seam_t128_x.log"
