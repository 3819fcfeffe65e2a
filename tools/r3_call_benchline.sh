#!/bin/bash
# the default bench line of the final tree (traffic from profiles/r3_pmc_hbm_bench_10M.json: same kernel sources)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
python bench.py > gpurun_out/r3k/bench_final.json 2> gpurun_out/r3k/bench_final.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3k/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['verify'])
print({k: ('error' in v) for k, v in d['secondary'].items() if isinstance(v, dict)})
print(d['secondary']['flb_sp_select'])"
