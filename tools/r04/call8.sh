#!/bin/bash
# r04 call 8: DP_WIDE_LDS occupancy-limiter A/B at 448 in flight; transformer-layer probe (launches / waits per proof, throughput), kernel trace of the transformer layer
o=gpurun_out/r04_call8; mkdir -p $o; export TMPDIR=/tmp
for w in 0 36864 24576 0 36864; do
  DP_WIDE_LDS=$w timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/wide_$w.txt 2>&1; echo "DP_WIDE_LDS=$w: $(grep -E 'proofs/s' $o/wide_$w.txt | tail -1 | cut -c1-120)"
done
GRAPH_MODEL=transformer_layer GRAPH_NO_ORACLE=1 DP_TIMING=1 timeout -s KILL 300 python tools/graph_probe.py 64 256 4 64 192 > $o/tl_probe.txt 2>&1
grep -E "single proof|device context" $o/tl_probe.txt | head -6 | cut -c1-300
cd /tmp; GRAPH_MODEL=transformer_layer GRAPH_NO_ORACLE=1 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$o/tl_prof -o tl -- python $OLDPWD/tools/graph_probe.py 64 256 4 64 192 > $OLDPWD/$o/tl_prof.log 2>&1; cd $OLDPWD
f=$(find $o/tl_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
sys.path.insert(0,'tools')
from rocpd_summary import short
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print(f"{short(r['Name'])[:40]:40s} calls {r['Calls']:>7s} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
find $o -name '*.db' -size +8M -delete; find $o -name '*kernel_trace.csv' -size +8M -delete
