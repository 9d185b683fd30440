#!/bin/bash
# r04 call 30: C consumer, asynchronous seam calls and the Mha / transformer goldens on the last commit
o=gpurun_out/r04_call30; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 44 python -m pytest tests/test_gpu_zz_async.py tests/test_gpu_zzzzz_mha.py tests/test_gpu_c_consumer.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -1 $o/pytest.txt | cut -c1-200
