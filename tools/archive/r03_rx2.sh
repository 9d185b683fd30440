#!/bin/bash
o=${1:-gpurun_out/r03_rx2}; mkdir -p "$o"; export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -s KILL 200 python tools/rx_probe2.py 64 4 > "$o/$name.log" 2>&1; echo "$name rc=$?" | tee -a "$o/summary.txt"; tail -12 "$o/$name.log"; }
run default DP_X=0
run nofused DP_DEVICE_LOGUP=0 DP_DEVICE_CLASSIC=0 DP_DEVICE_DENSE=0 DP_DEVICE_EQSUM=0 DP_DEVICE_COMMIT=0
run nofused_hostfs DP_DEVICE_LOGUP=0 DP_DEVICE_CLASSIC=0 DP_DEVICE_DENSE=0 DP_DEVICE_EQSUM=0 DP_DEVICE_COMMIT=0 DP_DEVICE_FS=0
run one_big DP_RX_STREAM_PER_CU=1
run stagewise DP_NTT_STAGEWISE=1
