#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/r4_perf1.py 10000000 12 0
timeout 300 python tools/r4_perf1.py 10000000 12 0
timeout 600 python -m pytest tests/test_tile_gpu.py -x -q -m gpu 2>&1 | tail -2
