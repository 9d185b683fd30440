#!/bin/bash
# r04 call 28: inference + host half of the witness generation prepared ahead by helper threads (DP_PREP_THREADS, default 2) against 0, same box: goldens, proofs equal, rates
o=gpurun_out/r04_call28; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 200 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "full_size" > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -1 $o/pytest.txt | cut -c1-200
for v in 2 0; do DP_PREP_THREADS=$v timeout -s KILL 100 python tools/r04/ab_batch.py transformer_layer 320 2 2>/dev/null | sed "s/^/DP_PREP_THREADS=$v /" | cut -c1-330; done
DP_PREP_THREADS=2 timeout -s KILL 100 python tools/r04/ab_batch.py dense_4m 448 3 2>/dev/null | sed "s/^/DP_PREP_THREADS=2 /" | cut -c1-330
