#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of three latency-mode Dense-4M proofs on the round's FINAL build (factored eq tables in the batch opening)
o=${1:-gpurun_out/r03_pmc2}; mkdir -p "$o"; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c -d "$o/proof_$c" -o x -- python tools/proof_only.py dense_4m 3 > "$o/proof_$c.log" 2>&1; echo "$c rc=$?"
done
f=$(find "$o/proof_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/proof_WRITE_SIZE" -name '*_results.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_summary.py --after-marker k_merkle_paths --population dense_4m_latency_proofs --units 3 "$f" "$w" "$o/pmc_dense4m_proofs_final.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/proof_only.py dense_4m 3 (launches after the k_merkle_paths marker: 3 latency-mode proofs, no setup; final build of round 3)" > "$o/pmc_dense4m_proofs_final.txt" 2>&1
find "$o" -name '*_results.db' -size +8M -delete
head -40 "$o/pmc_dense4m_proofs_final.txt" | cut -c1-160
