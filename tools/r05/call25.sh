#!/bin/bash
# r05 call 25: ONE proof with the fused protocol kernels and the sponge on the HOST (DP_DEVICE_FS=1 DP_HOST_SPONGE=1): whole lookup arguments / accumulation
# sumchecks per launch, a mailbox round trip per Fiat-Shamir round instead of a device permutation chain or a kernel per layer
o=gpurun_out/r05_call25; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_default.txt 2>&1; echo "default: $(grep -E 'proof [0-9]' $o/lat_default.txt | tail -3 | tr '\n' ';')"
for st in 1 2 6; do
  DP_DEVICE_FS=1 DP_HOST_SPONGE=1 DP_SPONGE_THREADS=$st timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_hs_$st.txt 2>&1; echo "host sponge, $st server thread(s) rc=$?: $(grep -E 'proof [0-9]|rror' $o/lat_hs_$st.txt | tail -3 | tr '\n' ';')"
done
DP_DEVICE_FS=1 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_devfs.txt 2>&1; echo "device sponge: $(grep -E 'proof [0-9]' $o/lat_devfs.txt | tail -3 | tr '\n' ';')"
