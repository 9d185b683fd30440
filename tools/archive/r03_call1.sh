#!/bin/bash
# First GPU call of round 3: config 5 pinned at size (new tests), the whole GPU suite, a bench line, and the round's first evidence:
# rocprofv3 --kernel-trace --stats of the 2^24 sumcheck + FETCH/WRITE PMC passes (sumcheck and latency-mode Dense-4M proofs, setup excluded).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r03_call1.sh'
o=${1:-gpurun_out/r03_call1}; mkdir -p "$o"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -k "config5" > "$o/config5.log" 2>&1; echo "config5 rc=$?" | tee -a "$o/summary.txt"
timeout 600 python -m pytest tests -m gpu -q > "$o/gpu_suite.log" 2>&1; echo "gpu suite rc=$?" | tee -a "$o/summary.txt"
DP_BENCH_NO_TORCH=1 timeout 500 python bench.py > "$o/bench.json" 2> "$o/bench.err"; echo "bench rc=$?" | tee -a "$o/summary.txt"
timeout 100 rocprofv3 --kernel-trace --stats -d "$o/sc24_kt" -o x -- python tools/sumcheck24_only.py 5 > "$o/sc24_kt.log" 2>&1
db=$(find "$o/sc24_kt" -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$o/sumcheck24_kernel_stats.csv" > "$o/sumcheck24_kernel_stats.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 80 rocprofv3 --kernel-trace --pmc $c -d "$o/sc24_$c" -o x -- python tools/sumcheck24_only.py 2 > "$o/sc24_$c.log" 2>&1
  timeout 100 rocprofv3 --kernel-trace --pmc $c -d "$o/proof_$c" -o x -- python tools/proof_only.py dense_4m 3 > "$o/proof_$c.log" 2>&1
done
f=$(find "$o/sc24_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/sc24_WRITE_SIZE" -name '*_results.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_summary.py --population sumcheck24 --units 2 "$f" "$w" "$o/pmc_sumcheck24.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/sumcheck24_only.py 2 (2 repetitions)" k_sc > "$o/pmc_sumcheck24.txt" 2>&1
f=$(find "$o/proof_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/proof_WRITE_SIZE" -name '*_results.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_summary.py --after-marker k_merkle_paths --population dense_4m_latency_proofs --units 3 "$f" "$w" "$o/pmc_dense4m_proofs.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/proof_only.py dense_4m 3 (launches after the k_merkle_paths marker: 3 latency-mode proofs, no setup)" > "$o/pmc_dense4m_proofs.txt" 2>&1
find "$o" -name '*_results.db' -size +8M -delete
cat "$o/summary.txt"; tail -3 "$o/config5.log"; tail -3 "$o/gpu_suite.log"; tail -2 "$o/bench.err" | cut -c1-300; head -c 1200 "$o/bench.json"; echo; head -8 "$o/sumcheck24_kernel_stats.txt"; cat "$o/pmc_sumcheck24.txt" "$o/pmc_dense4m_proofs.txt" 2>/dev/null | cut -c1-200 | head -20
