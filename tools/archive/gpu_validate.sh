#!/bin/bash
# One GPU-box pass: the parity suite, the bench line, the PMC passes of the standalone 2^24 sumcheck, the multiplier
# micro-benchmark. usage (from the repo root on the GPU box): bash tools/gpu_validate.sh gpurun_out/<tag> [skip-tests]
out=${1:-gpurun_out/validate}; mkdir -p "$out"; export TMPDIR=/tmp
if [ "$2" != "skip-tests" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"
fi
timeout 500 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$c" -o x -- python tools/sumcheck24_only.py 3 > "$out/pmc_$c.log" 2>&1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mulbench tools/mulbench.hip 2>/dev/null && timeout 120 /tmp/mulbench > "$out/mulbench.log" 2>&1
tail -4 "$out/pytest.log" 2>/dev/null; tail -2 "$out/bench.err"; cat "$out/mulbench.log"; head -c 2500 "$out/bench.json"
