#!/bin/bash
out=${1:-gpurun_out/r02_call18}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_primitives.py tests/test_gpu_zz_cohorts.py -m gpu -q -x -k "oracles_proof or default or merkle or commit or cohorts_match" > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log" | cut -c1-300
KNOB_WAVES=6 KNOB_ONLY=base_256 timeout 100 python tools/knob_sweep.py dense_4m "$out/knob_sweep.jsonl" 90 2>&1 | cut -c1-200
KNOB_WAVES=4 KNOB_ONLY=base_256 timeout 100 python tools/knob_sweep.py cnn_264k "$out/knob_sweep_cnn.jsonl" 90 2>&1 | cut -c1-200
DP_DEVICE_FS=1 timeout 100 rocprofv3 --kernel-trace --stats -d "$out/solo" -o x -- python tools/one_proof_cwd.py > "$out/solo.log" 2>&1
db=$(find "$out/solo" -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$out/solo_kernel_stats.csv" 2>&1 | head -8
