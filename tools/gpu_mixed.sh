#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
FLBGPU_DEBUG_LDS=1 timeout 600 python tools/perf_mixed.py 2 2>&1 | grep "k_parser_reg" | sort | uniq -c | head
echo "== 768 build of the fix-up"
FLBGPU_FIXUP_512=0 timeout 600 python tools/perf_mixed.py 30 2>&1 | grep -v amdgpu.ids | tail -3
