#!/bin/bash
# First GPU call of the next round (about 15 minutes of box time): parity suite, the knob sweep — which is also the first
# hardware run of the experimental kernels (k_logup_tail tail / full mode, k_classic_tail), each configuration checked
# against the sequential proof — after the bench line of the default configuration; then a kernel trace and the analysis of
# the best configuration.
# usage (repo root, on the GPU box): bash tools/next_round_first_call.sh gpurun_out/r02_first
out=${1:-gpurun_out/r02_first}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"
# the default configuration first (validated code only), then the sweep: an experimental kernel that misbehaves cannot cost the baseline
timeout 400 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"
DP_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_zz_experimental.py -m gpu -q > "$out/pytest_experimental.log" 2>&1; tail -3 "$out/pytest_experimental.log"
timeout 560 python tools/knob_sweep.py dense_4m "$out/knob_sweep_dense4m.jsonl" 520 > "$out/knob_sweep.log" 2>&1
KNOB_ONLY=base_192,devall_192,devall_async_tailmax2048_192 timeout 200 python tools/knob_sweep.py cnn_264k "$out/knob_sweep_cnn264k.jsonl" 180 >> "$out/knob_sweep.log" 2>&1
best=$(python - "$out/knob_sweep_dense4m.jsonl" <<'PY'
import json, sys
recs = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
ok = [r for r in recs if r.get("batch0_equals_single") and "proofs_per_s" in r]
b = max(ok, key=lambda r: r["proofs_per_s"]) if ok else None
print(" ".join(f"{k}={v}" for k, v in (b["env"].items() if b else [])))
PY
)
echo "best configuration: $best" | tee "$out/best.txt"
env $best timeout 120 rocprofv3 --kernel-trace --stats -d "$out/kt" -o x -- python tools/profile_batch.py dense_4m 192 > "$out/kt.log" 2>&1
db=$(find "$out/kt" -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "$out/kernel_stats.csv" > "$out/kernel_stats.txt" 2>&1 && python tools/trace_analyze.py "$db" > "$out/trace_analysis.txt" 2>&1
[ -n "$db" ] && [ "$(stat -c %s "$db")" -gt 20000000 ] && rm -f "$db"
tail -3 "$out/pytest.log"; cat "$out/knob_sweep.log" | cut -c1-220; cat "$out/best.txt"; head -12 "$out/trace_analysis.txt"; head -c 1200 "$out/bench.json"
