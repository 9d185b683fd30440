#!/bin/bash
# r06 call 8: GPU suite on the new defaults (idle sleep, one thread per cohort, NUMA pinning per device, capped-grid knobs off) + A/B probe
o=gpurun_out/r06_call8; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -3
for rep in 1 2; do timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_$rep.txt 2>&1; echo "dense $rep: $(tail -1 $o/ab_$rep.txt | cut -c1-200)"; done
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat.txt 2>&1; grep -E 'proof [3-5]' $o/lat.txt | sed 's/.*library //' | tr '\n' ' '; echo
