#!/bin/bash
# the JSON path on the GPU box: parity suite, then NDJSON -> events with the one-pass kernel, its phases, its counters, and the two-launch kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_json_gpu.py -x -q -m gpu 2>&1 | tail -15
for rep in 1 2; do
  timeout 300 python tools/perf_json.py 10000000 prof | cut -c1-330
done
timeout 300 python tools/perf_json.py 10000000 nolb | cut -c1-200
FLBGPU_JSON_TILE=0 timeout 300 python tools/perf_json.py 10000000 | cut -c1-200
timeout 600 bash tools/prof_kernel.sh pmc_json_lane k_json_lane -- python tools/perf_json.py 10000000
