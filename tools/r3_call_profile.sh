#!/bin/bash
# round-3 final numbers: default bench line + rocprofv3 stats + PMC passes of the same kernel sources
cd $GRAFT_REPO_ROOT
export SKIP_CAL=1
bash tools/profile_bench.sh > gpurun_out/profile_run.log 2>&1
tail -30 gpurun_out/profile_run.log | cut -c1-400
