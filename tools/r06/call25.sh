#!/bin/bash
# r06 call 25: GPU suite and the default bench on the sources with the new throughput-mode defaults (wave priority 2, grid cap 256, 336 MB worker arenas, 704 in flight)
o=gpurun_out/r06_call25; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.txt
timeout -s KILL 1500 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call25/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'steps', d.get('step_ms_min_median_max'), 'in flight', d['config'].get('proofs_in_flight_per_gpu'))
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'])
print('batch64', d.get('batch64'))
t=d.get('transformer_layer') or {}
print('tl', {k:t.get(k) for k in ('value','proofs_in_flight','single_proof_latency_ms','golden_sha256_ok','error')})
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
r=d['roofline']
print('roofline', {k:r.get(k) for k in ('achieved','peak','frac','job_frac','avg_launch_us')})
print('cpu', d['cpu_baseline'] and {k:d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
PY
