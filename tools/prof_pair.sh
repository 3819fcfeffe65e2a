# tools/prof_pair.sh -- on the GPU box: rocprofv3 kernel stats + SQ instruction-mix counters of the fused pair
# (tools/perf_fused.py), summaries under gpurun_out/prof_pair/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_pair
rm -rf $O; mkdir -p $O
CMD="python3 $R/tools/perf_fused.py ${1:-4000000}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats_run.txt 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- $CMD > /dev/null 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python3 - $O <<'PY'
import csv, sys, collections, json, glob, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
for k, v in sorted(out.items()):
    if any(t in k for t in ("parser", "grep", "k_pg", "gather")): print(k.split("::")[-1][:28], {c.replace("SQ_", ""): round(x / 1e6, 2) for c, x in sorted(v.items())})
PY
head -14 $O/kernel_stats.csv | cut -c1-150
