#!/bin/bash
# r05 call 17: where the fold phase of a latency-mode round goes (barriers against the fold loop)
o=gpurun_out/r05_call17; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t2.txt 2>&1; echo "rc=$?"; grep -E "sc-debug" $o/lat_t2.txt | tail -2 | cut -c1-300
