#!/bin/bash
# r06 call 40: CNN-264k at 674 in flight — where a pass goes: DP_TIMING=3 phase attribution, rocprofv3 kernel trace (stats, trace analysis, timeline)
o=gpurun_out/r06_call40; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DP_TIMING=3 timeout -s KILL 300 python tools/archive/conc_hoststats.py 674 cnn_264k > $o/phases_cnn.txt 2>&1; grep "proofs/s" $o/phases_cnn.txt
python - <<'PY'
import re,collections
H=collections.defaultdict(lambda:[0.0,0]); D=collections.defaultdict(lambda:[0.0,0]); nco=0
for l in open('gpurun_out/r06_call40/phases_cnn.txt'):
    m=re.match(r'\[dp cohort (host phase before the fire of|device phase ended by the result of)\]\s+([\d.]+) ms in\s+(\d+) phases,\s+([\d.]+) us each: (.*)',l)
    if m:
        t=H if m.group(1).startswith('host') else D
        t[m.group(5)][0]+=float(m.group(2)); t[m.group(5)][1]+=int(m.group(3))
    if 'wake-ups' in l: nco+=1
print('cohort stat blocks',nco)
for name,t in (('HOST phases (queue empty) by the launch that ends them',H),('DEVICE phases by the launch waited for',D)):
    tot=sum(v[0] for v in t.values())
    print(f'== {name}: {tot:.0f} ms summed over cohorts and passes')
    for k,v in sorted(t.items(),key=lambda kv:-kv[1][0])[:18]:
        print(f'  {100*v[0]/tot:5.1f} %  {v[0]/max(1,v[1])*1000:9.1f} us each x {v[1]:6d}  {k}')
PY
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o cnn -- python "$R/tools/profile_batch.py" cnn_264k 674 > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/rocpd_summary.py "$db" $o/r06_cnn674_kernel_stats.csv > $o/summary.err 2>&1; head -24 $o/r06_cnn674_kernel_stats.csv | cut -c1-120
  python tools/trace_analyze.py "$db" > $o/r06_trace_analysis_cnn674.txt 2>&1; sed -n 1,14p $o/r06_trace_analysis_cnn674.txt | cut -c1-160
  python tools/timeline_occupancy.py "$db" 5 > $o/r06_timeline_cnn674.txt 2>&1; sed -n 1,40p $o/r06_timeline_cnn674.txt | cut -c1-260
fi
find $o -name '*.db' -size +2M -delete
