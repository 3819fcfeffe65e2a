#!/bin/bash
# the tail skip: the whole GPU suite + the headline line without the CPU / secondary legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest_gpu.log 2>&1
tail -6 gpurun_out/r3k/pytest_gpu.log
python bench.py --no-cpu --no-secondary --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3k/bench_headline_tail.json
python3 -c "
import json
d=json.load(open('gpurun_out/r3k/bench_headline_tail.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'verify', (d.get('verify') or {}))"
FLBGPU_DEBUG_SKIP=128 python bench.py --no-cpu --no-secondary --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('no tail exit: ms', d['ms_per_step'], 'kernel', d['roofline']['avg_launch_ms'])"
