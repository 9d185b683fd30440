#!/bin/bash
# r06 call 43: is the seam-level submit / poll rate (one client thread, ~400 proofs/s) bounded by the client thread? Two / three seam_bench mode-3 processes side by side
# (each its own context, engine and client thread) against one, and one at 704 in flight.
o=gpurun_out/r06_call43; mkdir -p $o; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
mkdir -p tests/support/_build
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd
B=tests/support/_build/seam_bench
one() { tag=$1; shift; DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 200 $B "$@" > $o/sb_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/sb_$tag.txt | cut -c1-260)"; }
par() { tag=$1; k=$2; shift 2; for i in $(seq 1 $k); do DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 200 $B "$@" > $o/sb_${tag}_$i.txt 2>&1 & done; wait; for i in $(seq 1 $k); do echo "${tag}_$i: $(tail -1 $o/sb_${tag}_$i.txt | cut -c1-200)"; done; }
nproc
one a384 384 3 3
par p2x192 2 192 4 3
par p2x384 2 384 3 3
par p3x256 3 256 3 3
one a704 704 3 3
par p4x192 4 192 4 3
