#!/bin/bash
# cohorts: proofs in flight A/B with repetition (one process per point, 6 waves, three repetitions interleaved against box drift)
o=${1:-gpurun_out/r03_co2}; mkdir -p "$o"; export TMPDIR=/tmp
for rep in 1 2 3; do
  for conc in 256 320 384; do
    timeout -s KILL 200 python tools/rx_probe.py dense $conc 6 0 > "$o/co_${conc}_$rep.log" 2>&1; echo "conc $conc rep $rep: $(tail -1 $o/co_${conc}_$rep.log)"
  done
  DP_WORKER_ARENA_BYTES=503316480 timeout -s KILL 200 python tools/rx_probe.py dense 448 5 0 > "$o/co_448a_$rep.log" 2>&1; echo "conc 448 (480 MB arenas) rep $rep: $(tail -1 $o/co_448a_$rep.log)"
done
