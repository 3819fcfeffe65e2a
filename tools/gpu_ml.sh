#!/bin/bash
# multiline (java parser) on the GPU box: per-kernel time of tools/perf_ml.py under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ml; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/perf_ml.py > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-60s calls %4s  avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -2 $O/run.log
