#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
rocm-smi --showuse 2>&1 | tail -4
timeout 200 python -m pytest tests/test_decoders_gpu.py tests/test_tile_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|Abort" | tail -3
