#!/bin/bash
# r06 call 9: blocking seam calls routed to an engine (dp_ctx_route_to_engine): parity test + seam_bench modes 0 / 3 / 4
o=gpurun_out/r06_call9; mkdir -p $o; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=24
timeout -s KILL 600 python -m pytest tests/test_gpu_zz_async.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.txt
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd
sb() { tag=$1; shift; DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 200 tests/support/_build/seam_bench "$@" > $o/sb_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/sb_$tag.txt | cut -c1-330)"; }
sb streams14 14 6 0
sb async128 128 3 3
sb routed14 14 6 4
sb routed32 32 4 4
sb routed64 64 3 4
sb routed128 128 3 4
sb routed256 256 2 4
