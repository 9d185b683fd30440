#!/bin/bash
# r06 call 38: one proof through the fused protocol tails (prove_batch of 1) against latency-mode prove(), sponge on the device / served by the host
o=gpurun_out/r06_call38; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -s KILL 200 python tools/r06/lat_probe.py dense_4m > $o/$tag.txt 2>&1; tail -1 $o/$tag.txt | cut -c1-250; }
run base X=1
run hostsponge DP_HOST_SPONGE=1
run hostsponge_t2 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=2
run cap0 DP_WIDE_WG_CAP=0
run cap0_hostsponge DP_WIDE_WG_CAP=0 DP_HOST_SPONGE=1
