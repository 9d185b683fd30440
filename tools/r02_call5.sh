#!/bin/bash
# shared (non-exclusive) one-workgroup kernels: parity first (golden sha256 in throughput mode, knob matrix), then the sweep
out=${1:-gpurun_out/r02_call5}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "oracles_proof or default or logup=0+" > "$out/pytest_fused.log" 2>&1; tail -5 "$out/pytest_fused.log" | cut -c1-300
KNOB_ONLY=base_192,excl_192,shared128_192,shared512_192,shared1024_192,base_256,shared_cohort16_256,shared_cohort4_192,shared_nofused_192 timeout 300 python tools/knob_sweep.py dense_4m "$out/knob_sweep_dense4m.jsonl" 280 > "$out/knob_sweep.log" 2>&1
KNOB_ONLY=base_192,excl_192,shared512_192 timeout 120 python tools/knob_sweep.py cnn_264k "$out/knob_sweep_cnn264k.jsonl" 100 >> "$out/knob_sweep.log" 2>&1
cut -c1-260 "$out/knob_sweep.log"
