#!/bin/bash
# after the PMC pass of the current kernel sources: the default bench line (traffic quoted), smoke, the stream-processor tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_sp_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3k/smoke.txt 2>&1; tail -2 gpurun_out/r3k/smoke.txt
python bench.py > gpurun_out/r3k/bench_final.json 2> gpurun_out/r3k/bench_final.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3k/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['verify'])
print(d['secondary']['flb_sp_select'])"
