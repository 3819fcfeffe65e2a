#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/trace_fixup.py 2>&1 | grep -v amdgpu.ids | tail -16
