#!/bin/bash
# r04 call 15: device ex_mul / ex_lerp on the asm primitives + merged knobs: GPU suite, bench
o=gpurun_out/r04_call15; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/pytest_gpu.txt | tail -2
timeout -s KILL 900 python bench.py --steps 3 --warmup 1 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_call15/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'])
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'], d['sumcheck24'].get('roofline',{}) and d['sumcheck24']['roofline'].get('frac'), d['sumcheck24']['profiled_records_median_repetition'])
print('tl', d['transformer_layer'].get('value'), d['transformer_layer'].get('golden_sha256_ok'), d['transformer_layer'].get('single_proof_latency_ms'), d['transformer_layer'].get('error'))
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','peak','frac','job_frac','peak_valu_bound','frac_of_valu_bound','probe_frac_of_valu_bound','valu_issue_util','avg_launch_us','gpu_busy_ms_per_proof_latency_mode')})
PY
