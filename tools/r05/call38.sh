#!/bin/bash
# r05 call 38 (final sources): kernel trace of the transformer layer at 320 in flight; a second sample of the default bench on another box
o=gpurun_out/r05_call38; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o tl320 -- python "$R/tools/profile_batch.py" transformer_layer 320 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/rocpd_summary.py "$db" $o/r05_transformer_layer320_kernel_stats.csv > $o/summary.err 2>&1; head -14 $o/r05_transformer_layer320_kernel_stats.csv | cut -c1-120
  python tools/trace_analyze.py "$db" > $o/r05_trace_analysis_transformer_layer320.txt 2>&1; sed -n 1,12p $o/r05_trace_analysis_transformer_layer320.txt | cut -c1-160
fi
find $o -name '*.db' -size +2M -delete
timeout -s KILL 900 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call38/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], d['config']['golden_sha256_ok'], d['config']['single_proof_latency_ms'], d['step_ms_min_median_max'])
print('cnn', d['cnn_264k']['value'], d['cnn_264k']['single_proof_latency_ms']); t=d['transformer_layer']; print('tl', t['value'], t['single_proof_latency_ms'])
print('b64', d['batch64']['ms_per_batch'], 'sc24', d['sumcheck24']['wall_ms'], 'sc26', d['sumcheck26']['wall_ms'], 'roofline', d['roofline']['frac'], d['roofline']['job_frac'], d['roofline'].get('traffic'), 'tail', (d.get('tail_roofline') or {}).get('frac_member'))
PY
