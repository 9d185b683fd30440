#!/bin/bash
# r06 call 47: grid of the fused fold + sum rounds (k_sc_fused / k_sc_fused2: 4096 workgroups at most; the last workgroup reads every workgroup's four block sums past its L2):
# -DDP_FUSED_GRID_CAP=1024 / 512 builds against the release, 2^24 and 2^26 sumchecks, alternating
o=gpurun_out/r06_call47; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; nv=$2; reps=$3; shift 3; env SC24_PROFILE=1 "$@" timeout -s KILL 200 python tools/sumcheck24_only.py $reps $nv > $o/$tag.txt 2>&1; echo "== $tag"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $o/$tag.txt | tail -9; }
run rel_a 24 8 X=1
run cap1024_a 24 8 DP_LIB_VARIANT=fcap1024
run cap512_a 24 8 DP_LIB_VARIANT=fcap512
run rel_b 24 8 X=1
run cap1024_b 24 8 DP_LIB_VARIANT=fcap1024
run cap512_b 24 8 DP_LIB_VARIANT=fcap512
run rel_26 26 5 X=1
run cap1024_26 26 5 DP_LIB_VARIANT=fcap1024
