#!/bin/bash
# round 4, call 3: the whole -m gpu suite after the scratch fix (separate NFA scratch region)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/r4_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/r4_suite.log
tail -25 gpurun_out/r4_suite.log
