#!/bin/bash
# r04 call 31: QKV inference on the int16 path (host code): the Mha / transformer goldens on the device once more
o=gpurun_out/r04_call31; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 22 python -m pytest tests/test_gpu_zzzzz_mha.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -1 $o/pytest.txt | cut -c1-200
