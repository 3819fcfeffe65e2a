#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_small_call_gpu.py tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_index_gpu.py tests/test_json_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --no-cpu --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(p['value'], p['ms_per_step'], p['kernels'])"
FLBGPU_SCAN_LB=0 timeout 600 python bench.py --no-cpu --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; p=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three launches:', p['value'], p['ms_per_step'], p['kernels'])"
