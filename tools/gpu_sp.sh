#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sp_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/perf_sp.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-500
