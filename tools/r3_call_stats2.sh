#!/bin/bash
# rocprofv3 kernel stats of the bench with its secondary legs (final tree): the JSON kernels, the select kernels, the multiline kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k/stats2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $O/run.json 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r3k/kernel_stats_secondary.csv
head -40 $R/gpurun_out/r3k/kernel_stats_secondary.csv | cut -c1-150
rm -rf $O
