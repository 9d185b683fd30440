#!/bin/bash
# r05 call 18: the instruction-rate table (review item 7); the multi-rank path of the bench with two ranks sharing this one GPU over gloo (review item 8): the
# weak-scaling line and BASELINE config 4 (--batch 64), config 5 sharded over the two ranks included
o=gpurun_out/r05_call18; mkdir -p $o; export TMPDIR=/tmp
tools/r05/instr_rate > $o/instr_rate.txt 2>&1; echo "instr_rate rc=$?"; cat $o/instr_rate.txt
export DP_DIST_BACKEND=gloo DP_FORCE_DEVICE=0
timeout -s KILL 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-cnn --no-seam-level --no-transformer > $o/bench_2ranks.json 2> $o/bench_2ranks.err; echo "2 ranks rc=$?"; tail -2 $o/bench_2ranks.err | cut -c1-300
timeout -s KILL 600 python bench.py --gpus 2 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-cnn --no-seam-level --no-transformer --no-sumcheck24 > $o/bench_2ranks_batch64.json 2> $o/bench_2ranks_batch64.err; echo "2 ranks batch 64 rc=$?"; tail -2 $o/bench_2ranks_batch64.err | cut -c1-300
python - <<'PY'
import json
for f in ('bench_2ranks','bench_2ranks_batch64'):
    try:
        d=json.loads(open('gpurun_out/r05_call18/%s.json'%f).read().strip().split('\n')[-1])
        print(f, d['value'], d['n_gpus'], d['scaling'], d['config'].get('golden_sha256_ok'), d['config'].get('host_bound'), 'sharded', json.dumps(d.get('sumcheck24_sharded'))[:300], json.dumps(d.get('sumcheck26_sharded'))[:200])
    except Exception as e: print(f, 'ERR', e)
PY
