#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_multiline_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|split" | tail -8
