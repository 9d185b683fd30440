#!/bin/bash
# First GPU call of round 3: what was built after round 2's GPU budget ran out goes on hardware FIRST, then the full suite and a bench line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r03_call1.sh'
o=${1:-gpurun_out/r03_call1}; mkdir -p "$o"; export TMPDIR=/tmp
# 1. the one kernel never run on a GPU (k_batch_row_hash) and the general-evaluation batch_open (existing kernels, new host path)
timeout 300 python -m pytest tests/test_gpu_zzz_batch_commit.py -m gpu -q -x > "$o/zzz_batch_commit.log" 2>&1; echo "zzz rc=$?" | tee -a "$o/summary.txt"
# 2. smoke + the whole GPU suite (the refactored pcs_batch_open wrapper and the stricter verifier sit under every model proof)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$o/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$o/summary.txt"
timeout 600 python -m pytest tests -m gpu -q > "$o/gpu_suite.log" 2>&1; echo "gpu suite rc=$?" | tee -a "$o/summary.txt"
# 3. a bench line as the driver runs it (host verifier leg now multi-threaded: verify_ms_per_proof should drop from ~490 to ~50)
DP_BENCH_NO_TORCH=1 timeout 420 python bench.py > "$o/bench.json" 2> "$o/bench.err"; echo "bench rc=$?" | tee -a "$o/summary.txt"
tail -3 "$o/zzz_batch_commit.log"; tail -3 "$o/gpu_suite.log"; tail -1 "$o/smoke.log"; head -c 900 "$o/bench.json"; echo
