#!/bin/bash
# the stream processor's tests on the GPU + the select passes' times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_sp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/perf_sp_select.py 4000000 2>&1 | tail -4 | tee gpurun_out/r3k/perf_sp_select.txt
