#!/bin/bash
# r04 call 23: the GPU tests that call 22's -x stop did not reach (the async merge-count assertion was timing dependent; fixed)
o=gpurun_out/r04_call23; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_zz_async.py tests/test_gpu_zz_cohorts.py tests/test_gpu_zzz_batch_commit.py tests/test_gpu_zzzzz_mha.py -m gpu -q > $o/pytest_gpu_rest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error|FAILED" $o/pytest_gpu_rest.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
