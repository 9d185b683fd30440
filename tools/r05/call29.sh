#!/bin/bash
# r05 call 29: is the bimodal single-proof latency (30 vs 34.5 ms between processes) the NUMA node / core the proving thread lands on?
o=gpurun_out/r05_call29; mkdir -p $o; export TMPDIR=/tmp
{
echo "== topology"
lscpu | grep -E "Model name|Socket|NUMA|Core|Thread|^CPU\(s\)" | head -20
for d in /sys/class/drm/card*/device; do [ -e $d/numa_node ] && echo "$d: numa_node=$(cat $d/numa_node) local_cpulist=$(cat $d/local_cpulist) vendor=$(cat $d/vendor)"; done
for d in /sys/bus/pci/devices/*; do if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] && [ -e $d/numa_node ]; then echo "$d class=$(cat $d/class) numa=$(cat $d/numa_node) cpus=$(cat $d/local_cpulist)"; fi; done | head -20
echo "allowed cpus: $(grep Cpus_allowed_list /proc/self/status)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
numactl -H 2>/dev/null | head -20
} > $o/topology.txt 2>&1
cat $o/topology.txt | head -40
timeout -s KILL 100 python tools/archive/latency_probe.py > $o/warm.txt 2>&1
run() { echo "$1: $(grep -E 'proof [3-5]' $2 | sed 's/.*library //' | tr '\n' ' ')"; }
for rep in 1 2 3 4; do
  timeout -s KILL 100 python tools/archive/latency_probe.py > $o/free_$rep.txt 2>&1; run "free $rep" $o/free_$rep.txt
done
nn=$(ls -d /sys/devices/system/node/node* | wc -l)
for n in $(seq 0 $((nn-1))); do
  cl=$(cat /sys/devices/system/node/node$n/cpulist)
  for rep in 1 2; do
    timeout -s KILL 100 taskset -c $cl python tools/archive/latency_probe.py > $o/node${n}_$rep.txt 2>&1; run "node $n ($cl) $rep" $o/node${n}_$rep.txt
  done
done
