#!/bin/bash
# the stream processor's tests on the GPU (select path + the aggregate path after the bin-key change)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sp_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3_sp_tests.txt
cat gpurun_out/r3_sp_tests.txt
