#!/bin/bash
out=${1:-gpurun_out/r02_bench_only}; mkdir -p "$out"; export TMPDIR=/tmp
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
