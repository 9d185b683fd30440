#!/bin/bash
# r05 call 22: Merkle layers of 512 .. 4096 parents with one node per lane in throughput mode (DP_LP_MAX_TP=512, the new default) against the 8-lane kernel (4096)
o=gpurun_out/r05_call22; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  for v in 512 4096 256; do
    DP_LP_MAX_TP=$v timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_${v}_$rep.txt 2>&1; echo "DP_LP_MAX_TP=$v $rep: $(tail -1 $o/ab_${v}_$rep.txt | cut -c1-150)"
  done
done
for v in 512 4096; do
  DP_LP_MAX_TP=$v timeout -s KILL 200 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/ab_cnn_$v.txt 2>&1; echo "cnn DP_LP_MAX_TP=$v: $(tail -1 $o/ab_cnn_$v.txt | cut -c1-150)"
done
