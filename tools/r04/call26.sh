#!/bin/bash
# r04 call 26: the whole GPU suite on the round's last commit; DP_TIMING=1 accounting of the transformer layer's cohorts (where a pass goes)
o=gpurun_out/r04_call26; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -3
DP_TIMING=1 timeout -s KILL 200 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/tl_timing.txt 2>&1; grep -E "proofs/s|cohort:" $o/tl_timing.txt | head -8 | cut -c1-260
