#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_tile_gpu.py tests/test_small_call_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/perf_host_level.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python tools/perf_mixed.py 20 2>&1 | grep -v amdgpu.ids | tail -3
