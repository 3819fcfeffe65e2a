#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "0 0" "60000 0" "90000 0" "0 40000"; do
  set -- $v
  echo "== size pad $1  emit pad $2"
  FLBGPU_FMT_SIZE_LDS=$1 FLBGPU_FMT_EMIT_LDS=$2 timeout 600 python tools/perf_fmt.py 10000000 2>&1 | grep -v amdgpu.ids | tail -1
done
