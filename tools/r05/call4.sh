#!/bin/bash
# r05 call 4: member timing of k_logup_tail, entry -> exit only (no per-permutation clock reads: an s_memrealtime round trip costs 1-2 us under load), the round-5
# kernel against the round-4 kernel (wgtimesv1), alternating
o=gpurun_out/r05_call4; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
for v in wgtimes wgtimesv1; do
  DP_LIB_VARIANT=$v DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/${v}_$rep.txt 2>&1
  echo "$v $rep rc=$?"; grep -E "wg-times|proofs/s" $o/${v}_$rep.txt | tail -2 | cut -c1-460
done
done
