#!/bin/bash
# round 3: the library as it is at the end of the round (DP_LOGUP_TAIL_MAX_N knob in): the Mha / transformer-layer golden cases and smoke once more
o=gpurun_out/r03_last; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 40 python -m pytest tests/test_gpu_zzzzz_mha.py -q > "$o/mha.log" 2>&1; echo "mha cases rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/mha.log"
timeout -s KILL 25 python -c "import __graft_entry__ as g; g.smoke()" > "$o/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/smoke.log"
