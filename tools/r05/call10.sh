#!/bin/bash
# r05 call 10: the default bench (driver contract) on the build with LDS-assembled messages
o=gpurun_out/r05_call10; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1200 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call10/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'steps', d.get('step_ms_min_median_max'))
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'], d['sumcheck24'].get('roofline',{}) and d['sumcheck24']['roofline'].get('frac'))
t=d.get('transformer_layer') or {}
print('tl', {k:t.get(k) for k in ('value','proofs_in_flight','single_proof_latency_ms','golden_sha256_ok','error')})
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
r=d['roofline']
print('roofline', {k:r.get(k) for k in ('achieved','peak','frac','job_frac','job_frac_of_sustained_peak','traffic','traffic_source','avg_launch_us','peak_valu_bound','frac_of_valu_bound','valu_issue_util','valu_issue_util_at_sampled_clock')})
print('cpu', d['cpu_baseline'] and {k:d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
PY
