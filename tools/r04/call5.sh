#!/bin/bash
# r04 call 5: where does a k_logup_tail member's time go under load (diagnostic library, -DDP_WG_TIMES), and the cohorts' device / host phases (DP_TIMING), 448 in flight
o=gpurun_out/r04_call5; mkdir -p $o; export TMPDIR=/tmp
DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 400 python tools/archive/conc_hoststats.py 448 > $o/wgtimes_448.txt 2>&1; echo "rc=$?"
grep -E "wg-times|proofs/s|cohort:" $o/wgtimes_448.txt | cut -c1-400 | head -40
DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 200 python tools/archive/conc_hoststats.py 2 > $o/wgtimes_2.txt 2>&1; grep -E "wg-times|proofs/s" $o/wgtimes_2.txt | cut -c1-400
