#!/bin/bash
# r05 call 26: ONE proof, sponge on the host, only the lookup / accumulation stretches fused (the dense layer, the commit rounds and the batch-opening rounds stay chip-wide)
o=gpurun_out/r05_call26; mkdir -p $o; export TMPDIR=/tmp
for off in "dense,commit,classic,deleg" "dense,commit,classic,deleg,eqsum"; do
  DP_FUSED_OFF=$off DP_DEVICE_FS=1 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=1 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat.txt 2>&1; echo "host sponge, fused off: $off rc=$?: $(grep -E 'proof [0-9]|rror' $o/lat.txt | tail -3 | tr '\n' ';')"
done
DP_FUSED_OFF=dense,commit,classic,deleg DP_DEVICE_FS=1 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=1 DP_TIMING=1 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t.txt 2>&1; grep -E "dp timing\]" $o/lat_t.txt | tail -34 | cut -c1-120
