#!/bin/bash
# the walk's early end: tile parity tests + the headline line without the CPU / secondary legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_tile_gpu.py -m gpu -x -q > gpurun_out/r3k/pytest_tile.log 2>&1
tail -5 gpurun_out/r3k/pytest_tile.log
python bench.py --no-cpu --no-secondary --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3k/bench_headline.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3k/bench_headline.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'verify', (d.get('verify') or {}))"
