#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
( export PERF_TAIL_TRACE=1; echo "== tail skip on"; python tools/perf_tail.py 4000000; echo "== every position walked (FLBGPU_DEBUG_SKIP=128)"; FLBGPU_DEBUG_SKIP=128 python tools/perf_tail.py 4000000 ) > gpurun_out/r3k/perf_tail.txt 2>&1
cat gpurun_out/r3k/perf_tail.txt
