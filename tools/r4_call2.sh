#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r4_dbg_nfa.py 1 > gpurun_out/r4_dbg1.log 2>&1; tail -60 gpurun_out/r4_dbg1.log
