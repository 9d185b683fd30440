#!/bin/bash
# r06 call 50: grouping policy of the async engine (dp_async): linger while a group holds fewer than DP_ASYNC_LINGER_BELOW calls (default 2 = a lone call), for at most DP_ASYNC_LINGER_US (100),
# groups of at most DP_ASYNC_GROUP_MAX (32) — the seam-level submit / poll client at 192 / 256 / 384 / 128 in flight
o=gpurun_out/r06_call50; mkdir -p $o; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24 DP_LIB_VARIANT=asyncknobs
mkdir -p tests/support/_build
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench_ak tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip_asyncknobs -lpthread -Wl,-rpath,$PWD/deep-prove_amd
B=tests/support/_build/seam_bench_ak
one() { tag=$1; t=$2; per=$3; shift 3; env DP_ARENA_BYTES=$((2<<30)) "$@" timeout -s KILL 200 $B $t $per 3 > $o/sb_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/sb_$tag.txt | cut -c1-150)"; }
for T in 256 384 192 128; do
  per=3; [ $T -le 256 ] && per=4
  one base_$T $T $per X=1
  one b16_100_$T $T $per DP_ASYNC_LINGER_BELOW=16
  one b16_300_$T $T $per DP_ASYNC_LINGER_BELOW=16 DP_ASYNC_LINGER_US=300
  one b32_300_$T $T $per DP_ASYNC_LINGER_BELOW=32 DP_ASYNC_LINGER_US=300
  one b32_1000_$T $T $per DP_ASYNC_LINGER_BELOW=32 DP_ASYNC_LINGER_US=1000
done
one g64_b32_300_384 384 3 DP_ASYNC_GROUP_MAX=64 DP_ASYNC_LINGER_BELOW=32 DP_ASYNC_LINGER_US=300
one g16_384 384 3 DP_ASYNC_GROUP_MAX=16
one g24_384 384 3 DP_ASYNC_GROUP_MAX=24
