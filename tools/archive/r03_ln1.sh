#!/bin/bash
# round 3: first device run of the LayerNorm layer (graph golden cases 5 / 6) next to the other graph cases
mkdir -p gpurun_out/r03_ln1
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "graph_model_proof_bytes" > gpurun_out/r03_ln1/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r03_ln1/pytest.log
tail -15 gpurun_out/r03_ln1/pytest.log
