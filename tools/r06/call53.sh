#!/bin/bash
# r06 call 53: pre-launched fused rounds (DP_SC_PRELAUNCH: round i + 1 queued behind round i, its challenge through the mailbox): parity and timings with / without
o=gpurun_out/r06_call53; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; nv=$2; reps=$3; shift 3; env SC24_PROFILE=1 "$@" timeout -s KILL 120 python tools/sumcheck24_only.py $reps $nv > $o/$tag.txt 2>&1; echo "== $tag rc=$?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $o/$tag.txt | tail -10; }
run pre_22 22 4 X=1
run pre_24_a 24 8 X=1
run nopre_24_a 24 8 DP_SC_PRELAUNCH=0
run pre_24_b 24 8 X=1
run nopre_24_b 24 8 DP_SC_PRELAUNCH=0
run pre_26 26 5 X=1
run nopre_26 26 5 DP_SC_PRELAUNCH=0
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py tests/test_gpu_model.py -m gpu -x -q > $o/pytest.txt 2>&1; grep -E "passed|failed|error" $o/pytest.txt | tail -3
