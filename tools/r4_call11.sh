#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fx4 (default)"; timeout 300 python tools/r4_perf1.py 10000000 16 0
echo "== fx3"; FLBGPU_FX=3 timeout 300 python tools/r4_perf1.py 10000000 16 0
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_kat_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== mixed"; timeout 300 python tools/r4_mixed.py
echo "== timeline"; timeout 300 python tools/trace_reg.py 2>&1 | tail -25
