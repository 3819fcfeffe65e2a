#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_tile_gpu.py tests/test_small_call_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/perf_mixed.py 30 2>&1 | grep -v amdgpu.ids | tail -4
echo "== fix-up in the 1024-thread build"
FLBGPU_FIXUP_1024=1 timeout 600 python tools/perf_mixed.py 30 2>&1 | grep -v amdgpu.ids | tail -4
