#!/bin/bash
# the JSON path on the GPU box: parity suite, then NDJSON -> events with the tile pass and without it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_json_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python tools/perf_json.py 10000000
timeout 300 python tools/perf_json.py 10000000 prof
timeout 300 python tools/perf_json.py 10000000 nolb
FLBGPU_JSON_TILE=0 timeout 300 python tools/perf_json.py 10000000
