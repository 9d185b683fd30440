#!/bin/bash
# r04 call 3: first library with the gfx950 asm multiplication, un-spilled throughput-mode tails, no executor: GPU suite + short bench
o=gpurun_out/r04_call3; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest_gpu.txt
timeout -s KILL 600 python bench.py --steps 3 --warmup 1 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -c 3000 $o/bench.json; tail -5 $o/bench.err
