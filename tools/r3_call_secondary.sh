# secondary kernels: kernel stats + HBM / SQ counters of bench.py's side measurements, one pass each
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3d
rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --no-cpu --steps 3 --warmup 1 --ndjson-lines 20000000 --l2m-records 160000000"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_line.json 2> $O/bench.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq2 -- $CMD > /dev/null 2>&1
python3 - $O <<'PY'
import csv, sys, collections, json, glob, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=max(len(v) for v in cs.values())) for k, cs in res.items()}
json.dump({"command": "python3 bench.py --no-cpu --steps 3 --warmup 1 --ndjson-lines 20000000 --l2m-records 160000000", "unit": "per launch averages; FETCH_SIZE / WRITE_SIZE in KiB (FETCH x 2 on gfx950)", "kernels": out},
          open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
st = {}
for row in csv.DictReader(open(os.path.join(O, "kernel_stats.csv"))):
    st[row["Name"].split("(")[0]] = (int(row["Calls"]), float(row["AverageNs"]), float(row["MaxNs"]))
for k, v in sorted(out.items(), key=lambda kv: -st.get(kv[0], (0, 0, 0))[1]):
    if k.startswith("__amd") or "k_idx" in k or "k_scan" in k: continue
    calls, avg, mx = st.get(k, (0, 0, 0))
    print("%-42s calls %3d avg %8.3f ms max %8.3f | FETCHx2 %7.1f MB WRITE %7.1f MB | VALU %6.1fM SALU %6.1fM LDS %6.1fM VMEM %5.1fM | wavecyc %7.1fM wait %7.1fM busy %6.1fM" % (
        k.split("::")[-1][:42], calls, avg / 1e6, mx / 1e6, v.get("FETCH_SIZE", 0) * 2 / 1024, v.get("WRITE_SIZE", 0) / 1024,
        v.get("SQ_INSTS_VALU", 0) / 1e6, v.get("SQ_INSTS_SALU", 0) / 1e6, v.get("SQ_INSTS_LDS", 0) / 1e6, (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / 1e6,
        v.get("SQ_WAVE_CYCLES", 0) / 1e6, v.get("SQ_WAIT_ANY", 0) / 1e6, v.get("SQ_BUSY_CYCLES", 0) / 1e6))
PY
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
