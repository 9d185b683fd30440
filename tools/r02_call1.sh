#!/bin/bash
# Round 2, first GPU call: the experimental fused-protocol kernels on hardware for the first time.
#  1. small models (mlp(2,64), cnn_tiny): every proof of a batch byte-compared with the sequential proof, one process per knob
#  2. full-size models at small concurrency (tests/test_gpu_zz_experimental.py)
#  3. the knob sweep at 192 in flight on the key configurations
out=${1:-gpurun_out/r02_call1}; mkdir -p "$out"; export TMPDIR=/tmp
FLAGSETS=("DP_DEVICE_LOGUP=1" "DP_DEVICE_LOGUP=2" "DP_DEVICE_CLASSIC=1" "DP_DEVICE_DENSE=1" "DP_DEVICE_EQSUM=1" "DP_DEVICE_COMMIT=1" "DP_ASYNC_UPLOAD=1" "DP_MERKLE_FUSE=4" "DP_TAIL_MAX=2048" "DP_COHORT_XCD=1")
for fs in "${FLAGSETS[@]}"; do
  tag=$(echo "$fs" | tr ' =' '__')
  env $fs timeout 90 python -m pytest tests/test_gpu_zz_cohorts.py -m gpu -q -x -k "mlp-8 or mlp and 8 or cnn" > "$out/small_$tag.log" 2>&1
  echo "small $fs rc=$? $(tail -1 "$out/small_$tag.log")" | tee -a "$out/small_summary.txt"
done
DP_TEST_EXPERIMENTAL=1 timeout 480 python -m pytest tests/test_gpu_zz_experimental.py -m gpu -q > "$out/pytest_experimental.log" 2>&1; tail -30 "$out/pytest_experimental.log" | cut -c1-400
KNOB_ONLY=base_192,devlogup_full_192,devall_192,devall_async_tailmax2048_192,tail256_192,cohort_noexcl_tail256_192,xcd_192,tailmax2048_192,async_upload_192 DP_TIMING=0 timeout 240 python tools/knob_sweep.py dense_4m "$out/knob_sweep_dense4m.jsonl" 210 > "$out/knob_sweep.log" 2>&1
cut -c1-330 "$out/knob_sweep.log"
cat "$out/small_summary.txt"
