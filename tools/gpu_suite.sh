#!/bin/bash
# the whole GPU suite + the three quick timings, on the GPU box:  gpurun -- 'bash tools/gpu_suite.sh > gpurun_out/suite.log 2>&1'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/perf_host_phases.py
timeout 300 python tools/perf_reg.py 10000000 12 0
