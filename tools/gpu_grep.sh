#!/bin/bash
# filter_grep on the GPU box: parity suites that go through it, then BASELINE configs[2] with the one-pass kernel and with the three launches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_json_gpu.py tests/test_kat_gpu.py tests/test_host_rules_gpu.py tests/test_nfa_gpu.py -x -q -m gpu 2>&1 | tail -15
FLBGPU_GREP_PROF=1 timeout 600 python tools/perf_config2.py 20000000 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-1400
