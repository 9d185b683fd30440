#!/bin/bash
# full validation of the round-3 state: the whole GPU suite, smoke, the bench line (default = cohorts)
o=${1:-gpurun_out/r03_final2}; mkdir -p "$o"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > "$o/gpu_suite.log" 2>&1; echo "gpu suite rc=$?" | tee -a "$o/summary.txt"; tail -5 "$o/gpu_suite.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$o/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/smoke.log"
timeout 900 python bench.py --steps 5 --warmup 2 > "$o/bench.json" 2> "$o/bench.err"; echo "bench rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/bench.err" | cut -c1-300; head -c 600 "$o/bench.json"; echo
python - <<'P'
import json
b=json.load(open("gpurun_out/r03_final2/bench.json"))
s=b["sumcheck24"]; print("sumcheck24:", s["wall_ms"], s["golden_sha256_ok"], s["roofline"] and (s["roofline"]["kernel"], s["roofline"]["frac"], s["roofline"]["avg_launch_us"], s["roofline"]["traffic"]), s["roofline_withheld"], s["profiled_kernel_total_ms"], s["profiled_non_kernel_records_ms"])
print("roofline:", b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline"]["traffic"], b["roofline"]["traffic_source"], b["roofline"]["job_frac"])
print("cnn:", b["cnn_264k"]["value"], "cpu:", b["cpu_baseline"]["value"], b["cpu_baseline"]["port_mt"])
P
