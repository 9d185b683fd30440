#!/bin/bash
# r04 call 22 (final build of the round): the whole GPU suite; FETCH_SIZE / WRITE_SIZE passes of three latency-mode Dense-4M proofs (the population bench.py's
# roofline times) -> profiles/r04_pmc_dense4m_proofs_final.json on the box before the bench reads it; the default bench -> profiles/r04_final_bench.json
o=gpurun_out/r04_call22; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -3
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c -d "$R/$o/proof_$c" -o x -- python "$R/tools/proof_only.py" dense_4m 3 > "$R/$o/proof_$c.log" 2>&1; echo "$c rc=$?"
done
cd "$R"
f=$(find "$o/proof_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/proof_WRITE_SIZE" -name '*_results.db' | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  python tools/pmc_summary.py --after-marker k_merkle_paths --population dense_4m_latency_proofs --units 3 "$f" "$w" "$o/pmc_dense4m_proofs_final.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/proof_only.py dense_4m 3 (launches after the k_merkle_paths marker: 3 latency-mode proofs, no setup; final build of round 4, tools/r04/call22.sh)" > "$o/pmc_dense4m_proofs_final.txt" 2>&1
  cp "$o/pmc_dense4m_proofs_final.json" profiles/r04_pmc_dense4m_proofs_final.json && echo "pmc summary written"
  grep -A3 '"k_merkle_layer"' "$o/pmc_dense4m_proofs_final.json" | head -8
fi
find "$o" -name '*_results.db' -delete
timeout -s KILL 900 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_call22/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'steps', d.get('step_ms_min_median_max'))
print('clocks', json.dumps(d['config'].get('gpu_clocks_timed_region'))[:300])
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'], d['sumcheck24'].get('roofline',{}) and d['sumcheck24']['roofline'].get('frac'))
t=d.get('transformer_layer') or {}
print('tl', {k:t.get(k) for k in ('value','proofs_in_flight','single_proof_latency_ms','golden_sha256_ok','error')})
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
r=d['roofline']
print('roofline', {k:r.get(k) for k in ('achieved','peak','frac','job_frac','job_frac_of_sustained_peak','traffic','traffic_source','avg_launch_us','peak_valu_bound','frac_of_valu_bound','valu_issue_util','valu_issue_util_at_sampled_clock')})
print('cpu', d['cpu_baseline'] and {k:d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
PY
