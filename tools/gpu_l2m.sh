#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_rccl_ranks_gpu.py tests/test_gpu_parity.py tests/test_host_rules_gpu.py tests/test_engine.py -x -q -m gpu 2>&1 | tail -25
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
