#!/bin/bash
# r06 call 10: seam-level host, 14 blocking threads with a context each: latency mode / throughput mode / throughput-mode workgroups with the HOST sponge
o=gpurun_out/r06_call10; mkdir -p $o; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
sb() { tag=$1; shift; timeout -s KILL 200 env DP_ARENA_BYTES=$((2<<30)) "$@" > $o/sb_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/sb_$tag.txt | cut -c1-200)"; }
B=tests/support/_build/seam_bench
sb lat14 $B 14 6 0
sb tp14 $B 14 6 2
sb tp14_hostfs DP_DEVICE_FS=0 $B 14 6 2
sb tp28_hostfs DP_DEVICE_FS=0 DP_WAIT_YIELD=1 $B 28 4 2
sb lat28 DP_WAIT_YIELD=1 $B 28 4 0
sb lat14_b $B 14 6 0
