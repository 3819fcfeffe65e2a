#!/bin/bash
# the whole GPU suite and the default bench line, on the GPU box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 1500 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench rc $?"; tail -3 gpurun_out/bench_full.err
