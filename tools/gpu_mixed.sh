#!/bin/bash
# mixed record shapes through the pair on the GPU box: the suites that go through the register kernel, then tools/perf_mixed.py with the rows
# in the order of their lengths (the filter's own choice) and in chunk order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_tile_gpu.py tests/test_small_call_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/perf_mixed.py 30 2>&1 | grep -v amdgpu.ids | tail -3
echo "== chunk order only"
FLBGPU_SORT_ROWS=0 timeout 600 python tools/perf_mixed.py 20 2>&1 | grep -v amdgpu.ids | tail -3
