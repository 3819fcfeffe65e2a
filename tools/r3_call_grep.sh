#!/bin/bash
# the one-walk key lookup of k_grep_match: grep / log_to_metrics / parity tests, then configs[2]'s stages with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_l2m_gpu.py tests/test_tile_gpu.py tests/test_kat_gpu.py tests/test_json_gpu.py tests/test_index_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 0 1; do
if [ $v = 1 ]; then export FLBGPU_GREP_NO_HITS=1; fi
python bench.py --no-cpu --steps 3 --warmup 1 --ndjson-lines 20000000 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['secondary']['config2_ndjson_grep32']['stages']
print('NO_HITS=$v', {k: (v['ms_per_10M_lines'], v.get('kernel_ms')) for k, v in c.items()})"
done
