#!/bin/bash
out=${1:-gpurun_out/r02_call13}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_zz_cohorts.py -m gpu -q -x -k "oracles_proof or default or cohorts" > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log" | cut -c1-300
DP_BENCH_NO_TORCH=1 timeout 400 python bench.py --steps 3 --warmup 1 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; tail -3 "$out/bench.err" | cut -c1-300
python - "$out/bench.json" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value",r["value"],"ms/step",r["ms_per_step"],"golden",r["config"]["golden_sha256_ok"],"cnn",r["cnn_264k"]["value"],r["cnn_264k"]["golden_sha256_ok"])
    rf=r["roofline"]; print({k:rf[k] for k in ("bound","kernel","achieved","peak","frac","job_frac","job_compress_per_s","merkle_nodes_per_proof","job_hbm_frac")})
    print("sc24",r["sumcheck24"]["wall_ms"],r["sumcheck24"]["roofline"]["frac"]); print("cpu",r["cpu_baseline"])
except Exception as e: print("parse failed",e)
PY
KNOB_WAVES=6 KNOB_ONLY=base_256,cohort8_256,noasync_256,base_192 timeout 200 python tools/knob_sweep.py dense_4m "$out/knob_sweep.jsonl" 180 2>&1 | cut -c1-200
