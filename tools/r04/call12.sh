#!/bin/bash
# r04 call 12: after the prune + device-wide persistent pool + fix_high / eval tickets: GPU suite, seam bench (blocking vs async), full bench
o=gpurun_out/r04_call12; mkdir -p $o tests/support/_build; export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_gpu.txt | cut -c1-300
gcc -std=c11 -Wall -O2 -o tests/support/_build/seam_bench tests/support/seam_bench.c -L deep-prove_amd -ldeepprove_hip -lpthread -Wl,-rpath,$PWD/deep-prove_amd || exit 1
DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench 14 6 0 > $o/seam_blocking.txt 2>&1; echo "blocking 14 threads: $(tail -1 $o/seam_blocking.txt | cut -c1-200)"
for n in 32 64 128 256; do
  DP_TIMING=1 DP_ARENA_BYTES=$((2<<30)) timeout -s KILL 300 tests/support/_build/seam_bench $n 3 3 > $o/seam_async_$n.txt 2>&1
  echo "async $n in flight: $(grep -E 'seam_level|async engine' $o/seam_async_$n.txt | cut -c1-460)"
done
timeout -s KILL 900 python bench.py --steps 3 --warmup 1 > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_call12/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'])
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'])
print('tl', d['transformer_layer'].get('value'), d['transformer_layer'].get('golden_sha256_ok'), d['transformer_layer'].get('single_proof_latency_ms'))
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','peak','frac','job_frac','peak_valu_bound','frac_of_valu_bound','probe_frac_of_valu_bound','valu_issue_util','valu_instr_per_compress')})
PY
