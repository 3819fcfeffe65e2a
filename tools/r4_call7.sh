#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fx3 (default)"; timeout 300 python tools/r4_perf1.py 10000000 16,12 0
echo "== old tables"; FLBGPU_FX3=0 timeout 300 python tools/r4_perf1.py 10000000 16 0
echo "== mixed"; timeout 300 python tools/r4_mixed.py
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_kat_gpu.py -x -q -m gpu 2>&1 | tail -5
