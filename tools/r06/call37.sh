#!/bin/bash
# r06 call 37: k_sc_terms' base-table path with the any-representative multiplication, wide sums and two pairs per trip: parity (primitives / fused / model tests) and the 2^24 / 2^26 sumcheck
o=gpurun_out/r06_call37; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fused.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.txt 2>&1; tail -3 $o/pytest.txt
SC24_PROFILE=1 timeout -s KILL 200 python tools/sumcheck24_only.py 8 > $o/sc24.txt 2>&1; cat $o/sc24.txt | tail -16
SC24_PROFILE=1 timeout -s KILL 200 python tools/sumcheck24_only.py 5 26 > $o/sc26.txt 2>&1; cat $o/sc26.txt | tail -12
