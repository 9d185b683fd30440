#!/bin/bash
# what the driver runs at round end, in its order: smoke(), pytest -m gpu, bench.py (WITH torch, default flags + K/W it uses)
out=${1:-gpurun_out/r02_driver_like}; mkdir -p "$out"; export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 $out/smoke.log | cut -c1-200)"
t0=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s): $(grep -E 'passed|failed' $out/pytest_gpu.log | tail -1)"
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-2} > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"
tail -2 "$out/bench.err" | cut -c1-300
python - "$out/bench.json" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=r["config"]
print("value",r["value"],"ms/step",r["ms_per_step"],"steps",r["steps"],"golden",c["golden_sha256_ok"],"verified",c["verified_proofs_of_last_step"],"verify ms/proof",c["verify_batch_ms_per_proof"],"latency ms",c["single_proof_latency_ms"])
k=r["cnn_264k"]; print("cnn",k["value"],k["golden_sha256_ok"],k["steps"],"cpu",k["cpu_baseline"]["value"] if k["cpu_baseline"] else None)
rf=r["roofline"]; print({x:rf[x] for x in ("bound","kernel","achieved","peak","frac","job_frac","traffic")})
s=r["sumcheck24"]; print("sc24",s["wall_ms"],s["roofline"]["kernel"],s["roofline"]["frac"]); print("cpu",r["cpu_baseline"]["value"],r["cpu_baseline"]["cores"])
PY
