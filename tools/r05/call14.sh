#!/bin/bash
# r05 call 14: single-proof latency with the persistent sumcheck's arguments copied to LDS (no scalar loads from the kernarg segment inside the rounds)
o=gpurun_out/r05_call14; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t2.txt 2>&1; echo "rc=$?"; grep -E "sc-debug" $o/lat_t2.txt | tail -1 | cut -c1-200
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat.txt 2>&1; grep -E "proof [0-9]" $o/lat.txt | tail -3 | tr '\n' ';'; echo
timeout -s KILL 300 python tools/r04/ab_batch.py cnn_264k 64 1 > $o/cnn.txt 2>&1; tail -1 $o/cnn.txt | cut -c1-200
timeout -s KILL 300 python tools/r04/ab_batch.py dense_4m 448 3 > $o/d4m.txt 2>&1; tail -1 $o/d4m.txt | cut -c1-200
