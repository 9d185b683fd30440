#!/bin/bash
# r05 call 12: single-proof latency — where a Fiat-Shamir round of the latency-mode persistent kernel spends its cycles (DP_TIMING=2), and the device-side
# transcript (DP_DEVICE_FS=1: fused tails, whole-CU form) as the alternative for ONE proof
o=gpurun_out/r05_call12; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_timing2.txt 2>&1; echo "rc=$?"; grep -E "sc-debug|proof [0-9]|sumcheck rounds" $o/lat_timing2.txt | tail -12 | cut -c1-220
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_plain.txt 2>&1; echo "plain rc=$?"; grep -E "proof [0-9]" $o/lat_plain.txt | tail -4
DP_DEVICE_FS=1 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_devfs.txt 2>&1; echo "devfs rc=$?"; grep -E "proof [0-9]" $o/lat_devfs.txt | tail -4
