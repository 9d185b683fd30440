#!/bin/bash
# graph models (QKV / ConcatMatMul / MatMul, Add of two inputs) on the device; then the bench's Dense-4M section at 256 and 448 proofs in flight
o=${1:-gpurun_out/r03_graph1}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "graph" > "$o/tests_graph.log" 2>&1; echo "tests graph rc=$? $(tail -1 $o/tests_graph.log)"
grep -E "Error|error|assert" "$o/tests_graph.log" | head -20
for rep in 1 2; do
  for conc in 256 448; do
    timeout -s KILL 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sumcheck24 --no-cnn --no-seam-level --concurrency $conc > "$o/bench_${conc}_$rep.log" 2>&1
    echo "bench conc $conc rep $rep: $(python -c "import json,sys; d=json.loads(open('$o/bench_${conc}_$rep.log').read().strip().splitlines()[-1]); print(d['value'], d.get('golden_sha256_ok'), d['ms_per_step'])" 2>&1 | tail -1)"
  done
done
