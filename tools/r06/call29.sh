#!/bin/bash
# r06 call 29: where does a cohort's queue stand empty? DP_TIMING=3: host phases by the launch that ends them, device phases by the launch waited for (704 in flight, 4 waves)
o=gpurun_out/r06_call29; mkdir -p $o; export TMPDIR=/tmp
DP_TIMING=3 timeout -s KILL 300 python tools/archive/conc_hoststats.py 704 > $o/phases_704.txt 2>&1; grep "proofs/s" $o/phases_704.txt
python - <<'PY'
import re,collections
H=collections.defaultdict(lambda:[0.0,0]); D=collections.defaultdict(lambda:[0.0,0]); nco=0
for l in open('gpurun_out/r06_call29/phases_704.txt'):
    m=re.match(r'\[dp cohort (host phase before the fire of|device phase ended by the result of)\]\s+([\d.]+) ms in\s+(\d+) phases,\s+([\d.]+) us each: (.*)',l)
    if m:
        t=H if m.group(1).startswith('host') else D
        t[m.group(5)][0]+=float(m.group(2)); t[m.group(5)][1]+=int(m.group(3))
    if 'wake-ups' in l: nco+=1
print('cohort stat blocks',nco)
for name,t in (('HOST phases (queue empty) by the launch that ends them',H),('DEVICE phases by the launch waited for',D)):
    tot=sum(v[0] for v in t.values())
    print(f'== {name}: {tot:.0f} ms summed over cohorts and passes')
    for k,v in sorted(t.items(),key=lambda kv:-kv[1][0])[:16]:
        print(f'  {100*v[0]/tot:5.1f} %  {v[0]/max(1,v[1])*1000:9.1f} us each x {v[1]:6d}  {k}')
PY
