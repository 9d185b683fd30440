#!/bin/bash
# do two processes on one GPU scale? (per-process bottleneck vs device-wide)
export GPU_MAX_HW_QUEUES=32 DP_HOST_THREADS=${1:-7}
python tools/conc_sweep.py 32 > /tmp/p1.log 2>&1 &
python tools/conc_sweep.py 32 > /tmp/p2.log 2>&1 &
wait
tail -1 /tmp/p1.log; tail -1 /tmp/p2.log
