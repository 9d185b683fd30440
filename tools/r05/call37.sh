#!/bin/bash
# r05 call 37: the knobs added in the last hours as cases of the knob-by-knob parity test (batch proof 0 = the single proof, both verified), then smoke()
o=gpurun_out/r05_call37; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "wide or mailbox" > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.txt | cut -c1-200
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.txt | cut -c1-200
