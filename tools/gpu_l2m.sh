#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_l2m_gpu.py tests/test_gpu_parity.py tests/test_host_rules_gpu.py tests/test_rccl_ranks_gpu.py -x -q -m gpu 2>&1 | tail -12
for v in 1 0; do FLBGPU_L2M_LANE=$v timeout 300 python tools/perf_l2m.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600; done
