#!/bin/bash
# r04 call 27: lookup-table columns cached per context + array histogram of the range table (host side of the witness generation): device parity of the transformer / Dense-4M / CNN goldens, rates
o=gpurun_out/r04_call27; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_zzzzz_mha.py tests/test_gpu_model.py -m gpu -x -q -k "mha or transformer or config2 or config3 or golden" > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt | cut -c1-200
DP_TIMING=1 timeout -s KILL 120 python tools/r04/ab_batch.py transformer_layer 320 3 > $o/tl.txt 2>&1; grep -E "proofs/s" $o/tl.txt | cut -c1-260; grep -E "witness: (host|mult)" $o/tl.txt | tail -2
timeout -s KILL 120 python tools/r04/ab_batch.py dense_4m 448 3 2>/dev/null | cut -c1-260
