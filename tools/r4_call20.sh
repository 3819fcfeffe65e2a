#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== default (12 waves, no spills)"; timeout 300 python tools/r4_perf1.py 10000000 12 0
echo "== 16 waves"; timeout 300 python tools/r4_perf1.py 10000000 16 0
timeout 300 python tools/r4_mixed.py | head -3
FLBGPU_TILE_WAVES=16 timeout 300 python tools/r4_mixed.py | head -1
timeout 300 python tools/perf_host_phases.py 7000
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_kat_gpu.py tests/test_small_call_gpu.py -x -q -m gpu 2>&1 | tail -4
