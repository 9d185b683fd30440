#!/bin/bash
# round 3: first device run of Softmax / transformer_block (graph golden cases 7..10) next to the other graph cases, smoke, a short bench
o=gpurun_out/r03_sm1; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 240 python -m pytest tests/test_gpu_model.py -q -k "graph_model_proof_bytes" > "$o/pytest.log" 2>&1; echo "graph cases rc=$?" | tee -a "$o/summary.txt"; tail -4 "$o/pytest.log"
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" > "$o/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$o/summary.txt"; tail -1 "$o/smoke.log"
timeout -s KILL 170 python bench.py --steps 2 --warmup 1 --no-seam-level > "$o/bench.json" 2> "$o/bench.err"; echo "bench rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/bench.err" | cut -c1-300; head -c 700 "$o/bench.json"; echo
