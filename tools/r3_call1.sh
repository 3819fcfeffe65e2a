# first GPU call of round 3: correctness of the staged ingest, where its time goes, counters, secondary-kernel baselines
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3a
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py tests/test_kat_gpu.py -m gpu -x -q > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
timeout 300 python tools/perf_stage.py 10000000 > $O/perf_stage.log 2>&1
cat $O/perf_stage.log
cd /tmp
CMD="python3 $R/tools/perf_fused.py 10000000"
for NB in 0 4; do
export FLBGPU_STAGE_NBUF=$NB
timeout 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_hbm_nb$NB -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_lds_nb$NB -- $CMD > /dev/null 2>&1
done
unset FLBGPU_STAGE_NBUF
python3 - $O <<'PY'
import csv, sys, collections, json, glob, os
O = sys.argv[1]
for nb in ("0", "4"):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_hbm_nb" + nb, "pmc_lds_nb" + nb):
        for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in res.items()}
    json.dump(out, open(os.path.join(O, "pmc_summary_nb%s.json" % nb), "w"), indent=1)
    for k, v in sorted(out.items()):
        if any(t in k for t in ("parser_reg", "k_pg")): print("nbuf", nb, k.split("::")[-1][:28], {c.replace("SQ_", ""): round(x / 1e6, 3) for c, x in sorted(v.items())})
PY
# secondary kernels: kernel stats of the bench's side measurements (before any r3 change to them)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_secondary -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_secondary.json 2> $O/bench_secondary.err
find $O/stats_secondary -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_secondary.csv
head -40 $O/kernel_stats_secondary.csv | cut -c1-150
rm -rf $O/stats_secondary $O/pmc_hbm_nb* $O/pmc_lds_nb*
