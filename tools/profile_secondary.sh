#!/bin/bash
# tools/profile_secondary.sh -- on the GPU box: HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, a pass each, kernel-trace only) of the
# kernels behind the secondary lines of bench.py, each with the algorithmic bytes of its launch beside it -> gpurun_out/pmc_secondary.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_secondary
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {   # name, command...
  local name=$1; shift
  (cd $R && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${name}_fetch -- "$@" > $O/${name}.out 2> $O/${name}.err)
  (cd $R && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${name}_write -- "$@" > /dev/null 2>&1)
}
run config2 python tools/perf_config2.py 10000000 nocpu
run l2m python tools/perf_l2m.py 10000000
run fmt python tools/perf_fmt.py 10000000
run sp python tools/perf_sp.py 4000000
python3 - $O $R/gpurun_out/pmc_secondary.json <<'PY'
import csv, sys, collections, json, glob, os, re
O, dst = sys.argv[1:3]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    run = os.path.relpath(f, O).split(os.sep)[0].rsplit("_", 1)[0]
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        res[(run, k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for (run, k), cs in sorted(res.items()):
    if not re.match(r"k_(json|grep|gather|l2m|fmt|sp|scan|tile_max)", k): continue
    f = cs.get("FETCH_SIZE", []); w = cs.get("WRITE_SIZE", [])
    e = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_per_launch": sum(f) / len(f) if f else None, "WRITE_SIZE_KiB_per_launch": sum(w) / len(w) if w else None}
    if f and w: e["traffic_bytes_per_launch"] = int(sum(f) / len(f) * 1024 * 2 + sum(w) / len(w) * 1024)
    out["%s/%s" % (run, k)] = e
json.dump({"unit": "FETCH_SIZE / WRITE_SIZE in KiB per launch, averaged over the launches of the command (different inputs for the two grep stages); traffic = FETCH x 2 + WRITE (MI355X guide: wide coalesced reads count half)",
           "commands": {"config2": "tools/perf_config2.py 10000000 nocpu", "l2m": "tools/perf_l2m.py 10000000", "fmt": "tools/perf_fmt.py 10000000", "sp": "tools/perf_sp.py 4000000"},
           "kernels": out}, open(dst, "w"), indent=1)
for k, v in out.items(): print(k, {a: (round(b / 1e6, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
for n in config2 l2m fmt sp; do echo "== $n"; grep -v amdgpu.ids $O/$n.out | tail -4 | cut -c1-700; done
