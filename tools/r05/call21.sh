#!/bin/bash
# r05 call 21: host work of a proof's two heavy steps cut (proof serialiser with block appends and a reused buffer; the query phase parses one flat gather buffer,
# one device wait less): parity + the A/B probe
o=gpurun_out/r05_call21; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_primitives.py tests/test_gpu_zzz_batch_commit.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt
for rep in 1 2 3; do
  timeout -s KILL 200 python tools/r04/ab_batch.py dense_4m 448 3 > $o/ab_$rep.txt 2>&1; echo "dense $rep: $(tail -1 $o/ab_$rep.txt | cut -c1-150)"
done
timeout -s KILL 200 python tools/r04/ab_batch.py cnn_264k 448 2 > $o/ab_cnn.txt 2>&1; echo "cnn: $(tail -1 $o/ab_cnn.txt | cut -c1-150)"
timeout -s KILL 200 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/ab_tl.txt 2>&1; echo "tl: $(tail -1 $o/ab_tl.txt | cut -c1-150)"
DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py 448 > $o/timing_448.txt 2>&1
grep -E "host phases of one proof|device context:" $o/timing_448.txt | tail -3 | cut -c1-300
grep -E "cohort:" $o/timing_448.txt | tail -4 | cut -c1-200
