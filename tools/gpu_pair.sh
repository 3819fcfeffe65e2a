#!/bin/bash
# two filter_grep instances as one pass: the pair's parity suite, the suites that go through chains, then BASELINE configs[2]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_grep_pair_gpu.py tests/test_gpu_parity.py tests/test_host_rules_gpu.py tests/test_small_call_gpu.py tests/test_index_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python tools/perf_config2.py 20000000 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-2400
