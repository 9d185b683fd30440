#!/bin/bash
# r06 call 31: what the driver runs at round end — smoke() and the default bench — on the committed tree (bench.py no longer pins DP_HOST_THREADS for one rank)
o=gpurun_out/r06_call31; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -s KILL 1500 python bench.py --gpus 1 --steps 5 --warmup 1 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call31/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'host threads', d['config']['host_threads_per_rank'], 'knobs', d['config']['env_knobs'])
print('cnn', d['cnn_264k']['value'], 'tl', (d.get('transformer_layer') or {}).get('value'), 'b64', d['batch64']['ms_per_batch'], 'sc24', d['sumcheck24']['wall_ms'])
print('roofline', {k:d['roofline'].get(k) for k in ('frac','job_frac','traffic','valu_issue_util')}, 'tail', (d.get('tail_roofline') or {}).get('frac_merged_launch'))
PY
