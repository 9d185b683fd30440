#!/bin/bash
# k_deleg_tail (the delegation chains of the convolution protocol in one launch each): parity, CNN-264k throughput and latency with and
# without it; then the host-side accounting of a Dense-4M batch (DP_TIMING: host work between device waits, per context)
o=${1:-gpurun_out/r03_cnn2}; mkdir -p "$o"; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fused.py -m gpu -x -q -k "cnn or fused or tail" > "$o/tests_cnn.log" 2>&1; echo "tests cnn rc=$? $(tail -1 $o/tests_cnn.log)"
for rep in 1 2; do
  for dg in 0 1; do
    DP_DEVICE_DELEG=$dg timeout -s KILL 300 python bench.py --workload cnn_264k --steps 2 --warmup 1 --no-cpu-baseline --no-sumcheck24 --no-seam-level > "$o/cnn_${dg}_$rep.log" 2>&1
    echo "cnn deleg=$dg rep $rep: $(python -c "import json; d=json.loads(open('$o/cnn_${dg}_$rep.log').read().strip().splitlines()[-1]); print(d['value'], 'proofs/s; single proof', d.get('single_proof_latency_ms'), 'ms; golden', d.get('golden_sha256_ok'))" 2>&1 | tail -1)"
  done
done
DP_TIMING=1 timeout -s KILL 300 python tools/rx_probe.py dense 448 2 0 > "$o/host_stats.log" 2>&1
echo "host stats: $(tail -1 $o/host_stats.log)"
grep "device context" "$o/host_stats.log" | awk '{for(i=1;i<=NF;i++){if($i=="launches,")l+=$(i-1); if($i=="waits,")w+=$(i-1)}; split($0,a,"host work between waits "); split(a[2],b," ms"); hw+=b[1]; split($0,c,"inside waits "); split(c[2],d," ms"); iw+=d[1]; n++} END {printf "contexts %d launches %d waits %d host-work-between-waits %.1f ms inside-waits %.1f ms\n", n, l, w, hw, iw}'
grep -c "device context" "$o/host_stats.log"
