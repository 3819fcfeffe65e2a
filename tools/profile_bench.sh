# tools/profile_bench.sh -- on the GPU box: the default bench line, rocprofv3 kernel stats and the FETCH/WRITE PMC
# passes of the same command (separate passes, kernel-trace only); summaries land in gpurun_out/profile/ and are
# copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/profile
rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --no-cpu --no-secondary --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats_run.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > /dev/null 2>&1
# counter calibration: kernels with a known HBM byte count per request shape (csrc/calib.hip); SKIP_CAL=1 leaves it out
if [ -z "$SKIP_CAL" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- python3 $R/tools/calib_counters.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- python3 $R/tools/calib_counters.py run > /dev/null 2>&1
python3 $R/tools/calib_counters.py summarize $(find $O/cal_fetch $O/cal_write -name "*counter_collection.csv") > $O/counter_calibration.json
fi
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python - $O <<'PY'
import csv, sys, collections, json, glob, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
sys.path.insert(0, "/root/repo")
import bench
json.dump({"command": "python bench.py --no-cpu --no-secondary --steps 5 --warmup 1", "unit": "KiB per launch (FETCH_SIZE / WRITE_SIZE), GRBM_GUI_ACTIVE cycles",
           "kernel_source_sha": bench.kernel_source_sha(), "kernels": out},
          open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    if "parser" in k or "grep" in k or "k_pg" in k: print(k, {c: round(x / 1e6, 3) for c, x in v.items()})
PY
# the default bench line LAST: it reads the PMC summary of the same kernel sources for roofline.traffic
cp $O/pmc_summary.json $R/$(python -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.PMC_FILE)")
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err      # (the driver's own command line)
head -12 $O/kernel_stats.csv | cut -c1-160
cat $O/bench_default.json | cut -c1-2600
