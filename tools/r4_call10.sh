#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nfa_gpu.py tests/test_l2m_gpu.py tests/test_plugin_so.py -x -q -m gpu 2>&1 | tail -12
timeout 300 python tools/r4_perf1.py 10000000 16 0
