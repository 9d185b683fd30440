#!/bin/bash
# r06 call 49: the seam-level submit / poll client (seam_bench mode 3) by the number of proofs in flight — call 43 saw 445 proofs/s at 192 against 407 at 384
o=gpurun_out/r06_call49; mkdir -p $o; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
mkdir -p tests/support/_build
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd
B=tests/support/_build/seam_bench
one() { tag=$1; shift; DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 200 $B "$@" > $o/sb_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/sb_$tag.txt | cut -c1-175)"; }
one a384_a 384 3 3
one a192_a 192 4 3
one a256_a 256 4 3
one a160 160 4 3
one a224 224 4 3
one a192_b 192 6 3
one a384_b 384 3 3
one a256_b 256 4 3
one a128 128 6 3
