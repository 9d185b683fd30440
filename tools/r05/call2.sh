#!/bin/bash
# r05 call 2 (wc_io: one inlined permutation per transcript run; v1 = round-4 k_logup_tail with the new sc_fs_round): the LDS-resident k_logup_tail (round 5) against the round-4 form (library variant v1): parity tests of the fused kernels, then the A/B probe
# alternating on one box, then the member timing of the diagnostic build (DP_WG_TIMES)
o=gpurun_out/r05_call2; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q > $o/pytest_fused.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_fused.txt
for rep in 1 2; do
  for v in release v1; do
    if [ $v = release ]; then unset DP_LIB_VARIANT; else export DP_LIB_VARIANT=$v; fi
    timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_${v}_$rep.txt 2>&1; echo "$v $rep rc=$? $(tail -1 $o/ab_${v}_$rep.txt | cut -c1-330)"
  done
done
unset DP_LIB_VARIANT
DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wgtimes_448.txt 2>&1
echo "wgtimes rc=$?"; grep -E "wg-times|proofs/s" $o/wgtimes_448.txt | tail -4 | cut -c1-520
