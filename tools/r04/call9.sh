#!/bin/bash
# r04 call 9: async seam API parity + seam_bench (blocking threads vs one thread with N proofs in flight); k_logup_tail member timing with / without DP_WIDE_LDS
o=gpurun_out/r04_call9; mkdir -p $o tests/support/_build; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_zz_async.py tests/test_gpu_c_consumer.py -m gpu -x -q > $o/pytest_async.txt 2>&1; echo "pytest rc=$?"; tail -15 $o/pytest_async.txt
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd || exit 1
for cfg in "14 6 0" "32 4 3" "64 4 3" "128 3 3" "256 2 3"; do
  set -- $cfg
  DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench $1 $2 $3 > $o/seam_$1_$3.txt 2>&1; echo "seam_bench $cfg: rc=$? $(tail -1 $o/seam_$1_$3.txt | cut -c1-300)"
done
for w in 0 36864; do
  DP_WIDE_LDS=$w DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wgtimes_wide_$w.txt 2>&1
  echo "DP_WIDE_LDS=$w"; grep -E "wg-times|proofs/s" $o/wgtimes_wide_$w.txt | tail -3 | cut -c1-420
done
