#!/bin/bash
# r04 call 7: limb-form one-wave sponge (7.3 us per permutation), fence-free ticket in k_sc_fused, transformer layer pinned at 64 x 256: GPU suite + bench
o=gpurun_out/r04_call7; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest_gpu.txt
timeout -s KILL 700 python bench.py --steps 3 --warmup 1 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -5 $o/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_call7/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'])
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'], d['sumcheck24'].get('roofline'), d['sumcheck24']['profiled_records_median_repetition'])
print('tl', d['transformer_layer'])
print('seam', d['seam_level'])
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','peak','frac','avg_launch_us','job_frac','gpu_busy_ms_per_proof_latency_mode')})
PY
