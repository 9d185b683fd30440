#!/bin/bash
# r05 call 15: proofs in flight / cohort size after the tails shrank (the round-4 optimum was 448 in 22 cohorts)
o=gpurun_out/r05_call15; mkdir -p $o; export TMPDIR=/tmp
for conc in 256 352 448 544 448; do
  timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m $conc 3 > $o/ab_$conc.txt 2>&1; echo "in flight $conc: $(tail -1 $o/ab_$conc.txt | cut -c1-150)"
done
for co in 11 16 32; do
  DP_COHORT=$co timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_co$co.txt 2>&1; echo "DP_COHORT=$co: $(tail -1 $o/ab_co$co.txt | cut -c1-150)"
done
