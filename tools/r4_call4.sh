#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r4_perf1.py 10000000 > gpurun_out/r4_perf1.log 2>&1; cat gpurun_out/r4_perf1.log
