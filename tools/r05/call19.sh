#!/bin/bash
# r05 call 19: two / three cohorts taking turns on one stream (DP_STREAM_SHARE) against one cohort per stream, alternating on one box; then the two-rank runs
# of the bench on this one GPU (gloo), which ran out of device memory in call 18
o=gpurun_out/r05_call19; mkdir -p $o; export TMPDIR=/tmp
for rep in 1 2; do
  for k in 1 2 3; do
    DP_STREAM_SHARE=$k timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_share${k}_$rep.txt 2>&1; echo "DP_STREAM_SHARE=$k $rep rc=$? $(tail -1 $o/ab_share${k}_$rep.txt | cut -c1-200)"
  done
done
DP_STREAM_SHARE=2 timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 640 3 > $o/ab_share2_640.txt 2>&1; echo "share 2, 640 in flight: $(tail -1 $o/ab_share2_640.txt | cut -c1-160)"
for k in 1 2; do
  DP_STREAM_SHARE=$k timeout -s KILL 200 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/ab_cnn_share$k.txt 2>&1; echo "cnn DP_STREAM_SHARE=$k: $(tail -1 $o/ab_cnn_share$k.txt | cut -c1-160)"
  DP_STREAM_SHARE=$k timeout -s KILL 200 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/ab_tl_share$k.txt 2>&1; echo "tl DP_STREAM_SHARE=$k: $(tail -1 $o/ab_tl_share$k.txt | cut -c1-160)"
done
export DP_DIST_BACKEND=gloo DP_FORCE_DEVICE=0
timeout -s KILL 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-seam-level --no-transformer > $o/bench_2ranks.json 2> $o/bench_2ranks.err; echo "2 ranks rc=$?"; grep -v "^\[W" $o/bench_2ranks.err | tail -2 | cut -c1-300
python - <<'PY'
import json
for f in ('bench_2ranks',):
    try:
        d=json.loads(open('gpurun_out/r05_call19/%s.json'%f).read().strip().split('\n')[-1])
        print(f, d['value'], d['n_gpus'], d['scaling'], d['config'].get('golden_sha256_ok'), d['config'].get('host_bound'), 'in flight', d['config'].get('proofs_in_flight_per_gpu'), 'sharded24', json.dumps(d.get('sumcheck24_sharded'))[:400], 'sharded26', json.dumps(d.get('sumcheck26_sharded'))[:300])
    except Exception as e: print(f, 'ERR', e)
PY
