#!/bin/bash
# r05 call 24 (speculation bounded by its cost per lane): the persistent sumcheck computes round j + 1 symbolically in the challenge of round j while that challenge is awaited: latency-mode parity, single-proof latency
o=gpurun_out/r05_call24; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $o/pytest.txt | cut -c1-200
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t2.txt 2>&1; echo "rc=$?"; grep -E "sc-debug|sumcheck rounds" $o/lat_t2.txt | tail -2 | cut -c1-250
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat.txt 2>&1; grep -E "proof [0-9]" $o/lat.txt | tail -3 | tr '\n' ';'; echo
timeout -s KILL 200 python tools/sumcheck24_only.py 5 > $o/sc24.txt 2>&1; tail -5 $o/sc24.txt | tr '\n' ';'; echo
timeout -s KILL 300 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/cnn.txt 2>&1; tail -1 $o/cnn.txt | cut -c1-200
