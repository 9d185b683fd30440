#!/bin/bash
# r05 call 28: Activation::Gelu on the device (golden cases 15-17), the 8-lane Merkle layer kernel up to wider layers in latency mode (DP_LP_MAX sweep)
o=gpurun_out/r05_call28; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_zzzzzz_gelu.py -m gpu -x -q > $o/pytest_gelu.txt 2>&1; echo "gelu pytest rc=$?"; tail -4 $o/pytest_gelu.txt | cut -c1-220
for lp in 4096 16384 65536; do
  DP_LP_MAX=$lp timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_lp$lp.txt 2>&1; echo "DP_LP_MAX=$lp:"; grep -E "proof [0-9]" $o/lat_lp$lp.txt | tail -3 | tr '\n' ';'; echo
done
DP_LP_MAX=16384 timeout -s KILL 200 python tools/archive/latency_probe.py cnn_264k > $o/lat_cnn_lp16k.txt 2>&1; echo "cnn lp 16384:"; grep -E "proof [0-9]" $o/lat_cnn_lp16k.txt | tail -2 | tr '\n' ';'; echo
timeout -s KILL 200 python tools/archive/latency_probe.py cnn_264k > $o/lat_cnn.txt 2>&1; echo "cnn default:"; grep -E "proof [0-9]" $o/lat_cnn.txt | tail -2 | tr '\n' ';'; echo
