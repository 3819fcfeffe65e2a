#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r4_perf1.py 10000000 16,12 0,1 > gpurun_out/r4_perf2.log 2>&1; cat gpurun_out/r4_perf2.log
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
