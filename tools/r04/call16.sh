#!/bin/bash
# r04 call 16: shader clock / power of the GPU during the timed steps (bench.py ClockSampler), the sustained compress probe, transformer layer with 448 in flight
o=gpurun_out/r04_call16; mkdir -p $o; export TMPDIR=/tmp
(rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | head -40) > $o/smi_idle.txt
ls /sys/class/drm/ > $o/sysfs.txt 2>&1; for d in /sys/class/drm/card*/device; do echo $d; cat $d/vendor; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done >> $o/sysfs.txt 2>&1
DP_BENCH_TL_IN_FLIGHT=448 timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sumcheck24 --no-cnn --no-seam-level > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_call16/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'steps', d.get('step_ms_min_median_max'))
print('clocks', json.dumps(d['config'].get('gpu_clocks_timed_region')))
r=d['roofline']
print('peak', r['peak'], 'sustained', json.dumps(r.get('peak_sustained')))
print({k:r.get(k) for k in ('frac','job_frac','job_frac_of_sustained_peak','valu_issue_util','sclk_mhz_timed_region','valu_issue_util_at_sampled_clock')})
t=d.get('transformer_layer') or {}
print('tl', {k:t.get(k) for k in ('value','proofs_in_flight','single_proof_latency_ms','golden_sha256_ok','error','gpu_clocks_timed_region')})
PY
