#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLBGPU_GREP_PROF=1 timeout 600 python tools/perf_config2.py 10000000 nocpu 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-900
