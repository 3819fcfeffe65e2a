# tools/prof_reg.sh -- on the GPU box: SQ / LDS counters of the single-pass kernel (tools/perf_fused.py runs the pair),
# summaries under gpurun_out/prof_reg/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_reg
rm -rf $O; mkdir -p $O
CMD="python3 $R/tools/perf_fused.py ${1:-4000000}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/pmc3 -- $CMD > /dev/null 2>&1
python3 - $O <<'PY'
import csv, sys, collections, json, glob, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
for k, v in sorted(out.items()):
    if any(t in k for t in ("parser", "k_pg")): print(k.split("::")[-1][:28], {c.replace("SQ_", ""): round(x / 1e6, 2) for c, x in sorted(v.items())})
PY
