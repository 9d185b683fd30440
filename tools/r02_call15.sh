#!/bin/bash
out=${1:-gpurun_out/r02_call15}; mkdir -p "$out"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_c_consumer.py -m gpu -q -x -k "batch_verifier or c11" > "$out/pytest.log" 2>&1; tail -4 "$out/pytest.log" | cut -c1-300
DP_BENCH_NO_TORCH=1 timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; tail -3 "$out/bench.err" | cut -c1-300
python - "$out/bench.json" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=r["config"]
    print("value",r["value"],"ms/step",r["ms_per_step"],"golden",c["golden_sha256_ok"],"verified",c["verified_proofs_of_last_step"],"verify single ms",c["verify_ms_per_proof"],"batch ms/proof",c["verify_batch_ms_per_proof"])
    k=r["cnn_264k"]; print("cnn",k["value"],k["golden_sha256_ok"],k["verified_proofs_of_last_step"],k["verify_batch_ms_per_proof"])
    rf=r["roofline"]; print({x:rf[x] for x in ("bound","kernel","achieved","peak","frac","job_frac","job_compress_per_s","merkle_nodes_per_proof","job_hbm_frac")})
except Exception as e: print("parse failed",e)
PY
DP_BENCH_NO_TORCH=1 timeout 200 python bench.py --steps 3 --warmup 1 --batch 64 --no-cpu-baseline --no-sumcheck24 > "$out/bench_batch64.json" 2> "$out/bench_batch64.err"; echo "batch64 rc=$?"; python -c "
import json,sys
r=json.loads(open('$out/bench_batch64.json').read().strip().splitlines()[-1]); print('batch64: value',r['value'],'ms/step',r['ms_per_step'],r['scaling'],r['config']['proofs_per_step_all_gpus'],r['config']['golden_sha256_ok'])"
