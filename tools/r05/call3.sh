#!/bin/bash
# r05 call 3: member timing of k_logup_tail by phase, members with columns of at most / more than 1024 rows apart (diagnostic build)
o=gpurun_out/r05_call3; mkdir -p $o; export TMPDIR=/tmp
DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wgtimes_448.txt 2>&1
echo "wgtimes rc=$?"; grep -E "wg-times|proofs/s" $o/wgtimes_448.txt | tail -5 | cut -c1-560
DP_LOGUP_TAIL_MAX_N=4096 DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wgtimes_448_max4096.txt 2>&1
echo "wgtimes (DP_LOGUP_TAIL_MAX_N=4096) rc=$?"; grep -E "wg-times|proofs/s" $o/wgtimes_448_max4096.txt | tail -5 | cut -c1-560
