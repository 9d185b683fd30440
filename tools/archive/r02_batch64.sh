#!/bin/bash
# BASELINE config 4 at N = 1 (one batch of B proofs per step): cohort size sweep.  usage: r02_batch64.sh out "B:c B:c ..."
o=${1:-gpurun_out/r02_batch64}; mkdir -p "$o"; export TMPDIR=/tmp
for bc in ${2:-64:12 64:4 64:6 64:8 64:16}; do
  b=${bc%%:*}; c=${bc##*:}
  DP_BENCH_NO_TORCH=1 DP_COHORT=$c timeout 200 python bench.py --batch $b --steps 12 --warmup 2 --no-cpu-baseline --no-cnn --no-sumcheck24 > "$o/b_${b}_$c.json" 2> "$o/b_${b}_$c.err"
  python - "$o/b_${b}_$c.json" $b $c <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("batch",sys.argv[2],"cohort",sys.argv[3],"value",r["value"],"ms/step",r["ms_per_step"],"minmedmax",r.get("step_ms_min_median_max"),"golden",r["config"].get("golden_sha256_ok"),"in flight",r["config"]["proofs_in_flight_per_gpu"])
PY
done
