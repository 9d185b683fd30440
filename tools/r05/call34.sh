#!/bin/bash
# r05 call 34: the host work of ONE CNN-264k proof between its device waits (DP_TIMING=2 lists the long host stretches and the launches around them)
o=gpurun_out/r05_call34; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py cnn_264k > $o/cnn_t2.txt 2>&1
grep -c "dp chunk" $o/cnn_t2.txt
# the last proof's chunks: sort by host time
awk '/proof 4:/{p=1} p&&/dp chunk/{print}' $o/cnn_t2.txt | sed 's/.*before wait \([0-9]*\): \([0-9.]*\) us of host work, launches \(.*\)/\2 us  wait \1  \3/' | sort -rn | head -40
awk '/proof 4:/{p=1} p&&/dp chunk/{print}' $o/cnn_t2.txt | sed 's/.*before wait \([0-9]*\): \([0-9.]*\) us.*/\2/' | awk '{s+=$1; n++} END {print "chunks", n, "total host us", s}'
grep -E "proof 5|host transcript|witness" $o/cnn_t2.txt | tail -12 | cut -c1-200
