#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_small_call_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python tools/perf_mixed.py 40 2>&1 | grep -v amdgpu.ids | tail -12
