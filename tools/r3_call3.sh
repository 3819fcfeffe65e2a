# third GPU call: the restructured k_parser_reg (prefetched row offsets, value start from the record length, time text out of
# the registers, batched span reads): correctness, timeline, skip builds, counters
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_kat_gpu.py tests/test_plugin_so.py -m gpu -x -q > $O/pytest_subset.log 2>&1
tail -5 $O/pytest_subset.log
timeout 300 python tools/trace_reg.py 10000000 0 > $O/trace.log 2>&1
cat $O/trace.log
timeout 300 python tools/perf_stage.py 10000000 "0:16" > $O/perf_stage.log 2>&1
cat $O/perf_stage.log
cd /tmp
rocprofv3 --list-avail > $O/counters_avail.txt 2>&1
CMD="python3 $R/tools/perf_fused.py 10000000"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc1 -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc3 -- $CMD > /dev/null 2>&1
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
    if any(t in k for t in ("parser_reg", "k_pg")): print(k.split("::")[-1][:28], {c.replace("SQ_", ""): round(x / 1e6, 3) for c, x in sorted(v.items())})
PY
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
