#!/bin/bash
# r06 call 52 (final sources, csrc bb19fdd1fc305af2): the multi-rank path of the bench with TWO ranks sharing this one GPU over gloo: the weak-scaling line (config 5 sharded over the
# two ranks included) and BASELINE config 4 (--batch 64)
o=gpurun_out/r06_call52; mkdir -p $o; export TMPDIR=/tmp
export DP_DIST_BACKEND=gloo DP_FORCE_DEVICE=0
timeout -s KILL 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-seam-level --no-transformer > $o/bench_2ranks.json 2> $o/bench_2ranks.err; echo "2 ranks rc=$?"; tail -2 $o/bench_2ranks.err | cut -c1-300
timeout -s KILL 600 python bench.py --gpus 2 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-seam-level --no-transformer --no-sumcheck24 > $o/bench_2ranks_batch64.json 2> $o/bench_2ranks_batch64.err; echo "2 ranks batch 64 rc=$?"; tail -2 $o/bench_2ranks_batch64.err | cut -c1-300
python - <<'PY'
import json
for f in ('bench_2ranks','bench_2ranks_batch64'):
    try:
        d=json.loads(open('gpurun_out/r06_call52/%s.json'%f).read().strip().split('\n')[-1])
        print(f, d['value'], d['n_gpus'], d['scaling'], d['ms_per_step'], d['config'].get('golden_sha256_ok'), d['config'].get('host_bound'), 'sharded', json.dumps(d.get('sumcheck24_sharded'))[:400], json.dumps(d.get('sumcheck26_sharded'))[:300])
    except Exception as e: print(f, 'ERR', e)
PY
