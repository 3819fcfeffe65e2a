#!/bin/bash
# the coalesced ingest (FLBGPU_STAGE=1): parity tests with it on, then the headline line with and without
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
FLBGPU_STAGE=1 timeout 1500 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 1 0; do
FLBGPU_STAGE=$v python bench.py --no-cpu --no-secondary --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('STAGE=$v value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'], 'verify', d['verify']['fused_equals_unfused'], d['verify']['fused_sha256'][:12])"
done
