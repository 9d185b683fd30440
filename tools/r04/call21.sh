#!/bin/bash
# r04 call 21: DP_GRID_CAP sweep (workgroups per proof and launch of every chip-wide kernel), Dense-4M 448 in flight; 0 = uncapped (up to 4096)
o=gpurun_out/r04_call21; mkdir -p $o; export TMPDIR=/tmp
for c in 0 24 8 64 256 0 24; do
  DP_GRID_CAP=$c timeout -s KILL 120 python tools/r04/ab_batch.py dense_4m 448 3 2>> $o/ab.err | sed "s/^/DP_GRID_CAP=$c /" >> $o/ab.txt
done
DP_GRID_CAP=24 timeout -s KILL 120 python tools/r04/ab_batch.py transformer_layer 320 3 2>> $o/ab.err | sed "s/^/DP_GRID_CAP=24 /" >> $o/ab.txt
DP_GRID_CAP=24 timeout -s KILL 120 python tools/r04/ab_batch.py cnn_264k 448 3 2>> $o/ab.err | sed "s/^/DP_GRID_CAP=24 /" >> $o/ab.txt
cat $o/ab.txt | cut -c1-260; tail -5 $o/ab.err | cut -c1-300
