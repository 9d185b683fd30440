#!/bin/bash
# r04 call 25: HipDev::axpy_many refactored onto csrc/axpy_many.h (the plan shared with the kernel emulator): parity tests of the paths that use it
o=gpurun_out/r04_call25; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_primitives.py tests/test_gpu_zzz_batch_commit.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.txt | cut -c1-200
timeout -s KILL 400 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "full_size or axpy or default" > $o/pytest_fused.txt 2>&1; echo "fused rc=$?"; tail -2 $o/pytest_fused.txt | cut -c1-200
