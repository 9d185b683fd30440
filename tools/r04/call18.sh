#!/bin/bash
# r04 call 18: wave priority 2 for every chip-wide kernel that is not a Poseidon2 hash kernel (KF_WIDE) against the same build with DP_WIDE_PRIO=0, alternating on one box
o=gpurun_out/r04_call18; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  for v in "" wp0; do
    DP_LIB_VARIANT=$v timeout -s KILL 120 python tools/r04/ab_batch.py dense_4m 448 3 >> $o/ab.txt 2>> $o/ab.err; echo "rc=$?" >> $o/ab.txt
  done
done
for v in "" wp0; do
  DP_LIB_VARIANT=$v timeout -s KILL 120 python tools/r04/ab_batch.py transformer_layer 320 3 >> $o/ab.txt 2>> $o/ab.err; echo "rc=$?" >> $o/ab.txt
  DP_LIB_VARIANT=$v timeout -s KILL 120 python tools/r04/ab_batch.py cnn_264k 448 3 >> $o/ab.txt 2>> $o/ab.err; echo "rc=$?" >> $o/ab.txt
done
cat $o/ab.txt; tail -5 $o/ab.err | cut -c1-300
