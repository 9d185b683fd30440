#!/bin/bash
# r05 call 13: single-proof latency with the whole-CU (latency-mode) kernels compiled and launched for 512 threads (release), 1024 (round 4) and 256
o=gpurun_out/r05_call13; mkdir -p $o; export TMPDIR=/tmp
for v in release lat1024 lat256 release lat1024; do
  if [ $v = release ]; then unset DP_LIB_VARIANT; else export DP_LIB_VARIANT=$v; fi
  DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_$v.txt 2>&1; echo "$v rc=$?"; grep -E "sc-debug" $o/lat_$v.txt | tail -1 | cut -c1-200
  timeout -s KILL 200 python tools/archive/latency_probe.py > $o/latp_$v.txt 2>&1; grep -E "proof [0-9]" $o/latp_$v.txt | tail -3 | tr '\n' ';'; echo
done
unset DP_LIB_VARIANT
timeout -s KILL 300 python tools/r04/ab_batch.py cnn_264k 64 1 > $o/cnn.txt 2>&1; tail -1 $o/cnn.txt | cut -c1-200
