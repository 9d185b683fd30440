#!/bin/bash
# r06 call 1: two hypotheses about the cohort plateau, Dense-4M at 448 in flight, alternating on one box (tools/r04/ab_batch.py prints proofs/s + proof sha):
#  (a) DP_MERKLE_WG_CAP: an uncapped merged Merkle layer takes every wave slot of the chip for milliseconds; the other queues' workgroups wait for slots
#  (b) DP_IDLE_SLEEP_US + DP_HOST_THREADS=22: one host thread per cohort, sleeping instead of polling when every member waits for the device
o=gpurun_out/r06_call1; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run base1 X=1
run cap256 DP_MERKLE_WG_CAP=256
run cap512 DP_MERKLE_WG_CAP=512
run cap1024 DP_MERKLE_WG_CAP=1024
run cap128 DP_MERKLE_WG_CAP=128
run base2 X=1
run sleep20_t22 DP_IDLE_SLEEP_US=20 DP_HOST_THREADS=22
run sleep20_t14 DP_IDLE_SLEEP_US=20
run sleep5_t22 DP_IDLE_SLEEP_US=5 DP_HOST_THREADS=22
run t22 DP_HOST_THREADS=22
run cap512_sleep20_t22 DP_MERKLE_WG_CAP=512 DP_IDLE_SLEEP_US=20 DP_HOST_THREADS=22
run base3 X=1
