#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_parser_reg for library variants: tools/r4_pmc.sh <tag> <lib path>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; LIBP=$2
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && FLBGPU_LIB=$LIBP rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python $GRAFT_REPO_ROOT/tools/r4_perf1.py 10000000 16 0 > $O/run_$C.log 2>&1)
done
python - $O $TAG <<'PY'
import csv, sys, glob, os, collections
O, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in res.items():
    if "k_parser_reg" in k or "k_pg_emit" in k:
        f = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1); w = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
        print(tag, k, "FETCH x2 = %.3f GB  WRITE = %.3f GB  (launches %d)" % (f * 1024 * 2 / 1e9, w * 1024 / 1e9, len(cs.get("FETCH_SIZE", []))))
PY
tail -1 $O/run_FETCH_SIZE.log
