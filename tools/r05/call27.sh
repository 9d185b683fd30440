#!/bin/bash
# r05 call 27: the challenge mailbox in device memory (CPU writes through the BAR, the kernel polls HBM) and the one-round-trip poll: latency-mode parity, single-proof latency A/B
o=gpurun_out/r05_call27; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=1 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t1.txt 2>&1; echo "rc=$?"; grep -E "mailbox" $o/lat_t1.txt | head -2; grep -E "proof [0-9]" $o/lat_t1.txt | tail -3 | tr '\n' ';'; echo
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat.txt 2>&1; echo "vram mailbox:"; grep -E "proof [0-9]" $o/lat.txt | tail -4 | tr '\n' ';'; echo
DP_MAILBOX_VRAM=0 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_host.txt 2>&1; echo "host mailbox:"; grep -E "proof [0-9]" $o/lat_host.txt | tail -4 | tr '\n' ';'; echo
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t2.txt 2>&1; grep -E "sc-debug|sumcheck rounds" $o/lat_t2.txt | tail -2 | cut -c1-250
timeout -s KILL 200 python tools/archive/latency_probe.py cnn_264k > $o/lat_cnn.txt 2>&1; echo "cnn:"; grep -E "proof [0-9]" $o/lat_cnn.txt | tail -3 | tr '\n' ';'; echo
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.txt | cut -c1-200
timeout -s KILL 200 python tools/sumcheck24_only.py 5 > $o/sc24.txt 2>&1; tail -5 $o/sc24.txt | tr '\n' ';'; echo
