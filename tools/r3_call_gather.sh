#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_kat_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python bench.py --no-cpu --steps 3 --warmup 1 --ndjson-lines 20000000 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['secondary']['config2_ndjson_grep32']['stages']
print({k: (v['ms_per_10M_lines'], v.get('kernel_ms')) for k, v in c.items()})
print('verify', d['verify']['fused_equals_unfused'], d['verify']['oracle_sample_matches'])"
