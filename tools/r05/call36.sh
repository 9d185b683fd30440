#!/bin/bash
# r05 call 36: the threshold of the 512-thread k_logup_tail form: lookups of >= 1024 rows (most of Dense-4M's) against >= 2048 (the default), Dense-4M and the transformer layer
o=gpurun_out/r05_call36; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout -s KILL 300 python tools/r04/ab_batch.py dense_4m 448 3 > $o/dense_2048_$rep.txt 2>&1; echo "dense >=2048 $rep: $(tail -1 $o/dense_2048_$rep.txt | cut -c1-110)"
  DP_LOGUP_WIDE_N=1024 timeout -s KILL 300 python tools/r04/ab_batch.py dense_4m 448 3 > $o/dense_1024_$rep.txt 2>&1; echo "dense >=1024 $rep: $(tail -1 $o/dense_1024_$rep.txt | cut -c1-110)"
done
for rep in 1 2; do
  timeout -s KILL 300 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/tl_2048_$rep.txt 2>&1; echo "tl >=2048 $rep: $(tail -1 $o/tl_2048_$rep.txt | cut -c1-110)"
  DP_LOGUP_WIDE_N=0 timeout -s KILL 300 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/tl_0_$rep.txt 2>&1; echo "tl never $rep: $(tail -1 $o/tl_0_$rep.txt | cut -c1-110)"
done
