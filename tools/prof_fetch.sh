# tools/prof_fetch.sh NAME ... -- on the GPU box: FETCH_SIZE / WRITE_SIZE of k_parser_reg per launch (10 M records, tools/perf_fused.py)
# for builds of the library (tools/variant.sh names; "main" = libflbgpu.so); KiB as the counters report them
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for spec in "$@"; do
    name=${spec%%:*}; envs=""
    [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
    lib=$R/fluent-bit_amd/csrc/libflbgpu_$name.so; [ "$name" = main ] && lib=$R/fluent-bit_amd/csrc/libflbgpu.so
    O=$R/gpurun_out/prof_fetch_$name; rm -rf $O; mkdir -p $O
    # (one counter per pass -- FETCH_SIZE and WRITE_SIZE in one pass hung the box for the call's whole limit, round 5 -- and a limit of its own)
    env FLBGPU_LIB=$lib $envs timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/tools/perf_fused.py 10000000 > /dev/null 2>&1
    env FLBGPU_LIB=$lib $envs timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/tools/perf_fused.py 10000000 > /dev/null 2>&1
    python3 - $O "$spec" <<'PY'
import csv, sys, glob, os, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in res.items():
    if "parser_reg" in k or "pg_emit" in k:
        print("%-30s %-40s" % (sys.argv[2], k.split("::")[-1][:40]), {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
done
