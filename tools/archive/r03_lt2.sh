#!/bin/bash
# round 3: the transformer layer at 64 x 256 with 192 proofs in flight (64 gave 84-89 proofs/s whatever the logup tail threshold)
o=gpurun_out/r03_lt2; mkdir -p "$o"; export TMPDIR=/tmp GRAPH_MODEL=transformer_layer GRAPH_NO_ORACLE=1
timeout -s KILL 75 python tools/graph_probe.py 64 256 4 64 192 > "$o/inflight_192.txt" 2>&1; echo "192 rc=$?" | tee -a "$o/summary.txt"; tail -2 "$o/inflight_192.txt"
