#!/bin/bash
# r05 call 33: k_med_prepare in 64 KB blocks (two workgroups per 2^14-entry column instead of one that needs an empty CU): parity, then CNN-264k / transformer
# layer throughput with and without (DP_MED_SPLIT=0), alternating on one box
o=gpurun_out/r05_call33; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_model.py tests/test_gpu_fused.py tests/test_gpu_zzz_batch_commit.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.txt | cut -c1-200
for rep in 1 2; do
  timeout -s KILL 300 python tools/r04/ab_batch.py cnn_264k 448 3 > $o/cnn_split_$rep.txt 2>&1; echo "cnn split $rep: $(tail -1 $o/cnn_split_$rep.txt | cut -c1-150)"
  DP_MED_SPLIT=0 timeout -s KILL 300 python tools/r04/ab_batch.py cnn_264k 448 3 > $o/cnn_whole_$rep.txt 2>&1; echo "cnn DP_MED_SPLIT=0 $rep: $(tail -1 $o/cnn_whole_$rep.txt | cut -c1-150)"
done
timeout -s KILL 300 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/tl_split.txt 2>&1; echo "tl split: $(tail -1 $o/tl_split.txt | cut -c1-150)"
DP_MED_SPLIT=0 timeout -s KILL 300 python tools/r04/ab_batch.py transformer_layer 320 2 > $o/tl_whole.txt 2>&1; echo "tl DP_MED_SPLIT=0: $(tail -1 $o/tl_whole.txt | cut -c1-150)"
