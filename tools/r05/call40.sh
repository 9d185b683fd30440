#!/bin/bash
# r05 call 40: the whole GPU suite on the tree at HEAD (the four knob cases added after final.sh included) and smoke()
o=gpurun_out/r05_call40; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -2
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $o/smoke.txt
