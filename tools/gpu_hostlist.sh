#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_rules_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 3000 python -m pytest tests -x -q -m gpu --deselect tests/test_host_rules_gpu.py 2>&1 | tail -8
