#!/bin/bash
# r06 call 11: the 2^24 sumcheck: default against DP_MULTI_MID=1 (the streaming rounds hand over to the multi-workgroup persistent phase at 2^18 entries)
o=gpurun_out/r06_call11; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  SC24_PROFILE=1 timeout -s KILL 200 python tools/sumcheck24_only.py 6 > $o/base_$rep.txt 2>&1; echo "base $rep:"; head -6 $o/base_$rep.txt | cut -c1-80 | tr '\n' ';'; echo
  DP_MULTI_MID=1 SC24_PROFILE=1 timeout -s KILL 200 python tools/sumcheck24_only.py 6 > $o/mid_$rep.txt 2>&1; echo "mid $rep:"; head -6 $o/mid_$rep.txt | cut -c1-80 | tr '\n' ';'; echo
done
tail -9 $o/base_1.txt; tail -9 $o/mid_1.txt
