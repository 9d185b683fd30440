#!/bin/bash
# round 3: first device run of the Mha node (graph golden cases 11..13, tests/test_gpu_zzzzz_mha.py), then the whole GPU suite and smoke on the final build
o=gpurun_out/r03_mha1; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 150 python -m pytest tests/test_gpu_zzzzz_mha.py -q > "$o/mha.log" 2>&1; echo "mha cases rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/mha.log"
timeout -s KILL 330 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_zzzzz_mha.py > "$o/gpu_suite.log" 2>&1; echo "gpu suite rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/gpu_suite.log"
timeout -s KILL 60 python -c "import __graft_entry__ as g; g.smoke()" > "$o/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/smoke.log"
