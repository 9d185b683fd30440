#!/bin/bash
# r04 call 29: the cohort tests (every proof of a batch against its sequential proof, batches larger than the proofs in flight: the helper-prepared path) on the last commit
o=gpurun_out/r04_call29; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 60 python -m pytest tests/test_gpu_zz_cohorts.py -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -1 $o/pytest.txt | cut -c1-200
