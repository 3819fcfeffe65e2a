#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest_gpu_final.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r3k/pytest_gpu_final.log | tail -3
