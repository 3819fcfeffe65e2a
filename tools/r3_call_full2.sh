#!/bin/bash
# final state: the whole GPU suite, smoke, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest_gpu_final.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r3k/pytest_gpu_final.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3k/smoke.txt 2>&1; tail -1 gpurun_out/r3k/smoke.txt
python bench.py > gpurun_out/r3k/bench_final.json 2> gpurun_out/r3k/bench_final.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3k/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['verify']['fused_equals_unfused'], d['verify']['oracle_sample_matches'])
print({k: ('error' in v) for k, v in d['secondary'].items() if isinstance(v, dict)})
print(d['secondary']['config2_ndjson_grep32']['stages']['json_to_events'])"
