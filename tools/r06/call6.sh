#!/bin/bash
# r06 call 6: host accounting of a 448-in-flight batch (DP_TIMING=1): how many cores are really busy when idle threads sleep?
o=gpurun_out/r06_call6; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/t_default.txt 2>&1
DP_IDLE_SLEEP_US=20 DP_HOST_THREADS=22 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/t_sleep_t22.txt 2>&1
DP_IDLE_SLEEP_US=20 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/t_sleep_t14.txt 2>&1
for f in default sleep_t22 sleep_t14; do echo "== $f"; grep -E "proofs/s|cohort:|device context|host phases of one proof" $o/t_$f.txt | tail -8 | cut -c1-420; done
