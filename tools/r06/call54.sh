#!/bin/bash
# r06 call 54: the pre-launched fused rounds as a VARIANT library (tools/r06/experiments/prelaunch_fused_rounds.patch, libdeepprove_hip_prelaunch.so; the mailbox's
# probe launch moved before the round's own launch) against the release library, then the release's smoke + default bench on the committed sources
o=gpurun_out/r06_call54; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; nv=$2; reps=$3; shift 3; env SC24_PROFILE=1 "$@" timeout -s KILL 120 python tools/sumcheck24_only.py $reps $nv > $o/$tag.txt 2>&1; echo "== $tag rc=$?"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $o/$tag.txt | tail -7; }
run pre_22 22 4 DP_LIB_VARIANT=prelaunch
run rel_22 22 4 X=1
run pre_24_a 24 8 DP_LIB_VARIANT=prelaunch
run rel_24_a 24 8 X=1
run pre_24_b 24 8 DP_LIB_VARIANT=prelaunch
run rel_24_b 24 8 X=1
run pre_26 26 5 DP_LIB_VARIANT=prelaunch
run rel_26 26 5 X=1
DP_LIB_VARIANT=prelaunch timeout -s KILL 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_sharded.py -m gpu -x -q > $o/pytest_pre.txt 2>&1; echo "variant pytest:"; grep -E "passed|failed|error" $o/pytest_pre.txt | tail -3
echo "== release: smoke + bench"
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -s KILL 600 python bench.py --steps 5 > $o/bench.txt 2>$o/bench.err; echo "bench rc=$?"; tail -1 $o/bench.txt | cut -c1-1500
