cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|LDS[A-Za-z_0-9]*|VALU[A-Za-z_0-9]*|Occupancy[A-Za-z_0-9]*|MemUnit[A-Za-z]*|FetchSize|WriteSize|SALU[A-Za-z]*)\b" | sort -u | tr '\n' ' ' | head -c 6000
echo
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  d=$R/gpurun_out/pmc_$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $R/bench.py --steps 1 --warmup 0 --records 4000000 --no-cpu --no-secondary > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for row in csv.DictReader(open(sys.argv[1])):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for k, v in acc.items():
        if "k_parser" in k or "grep" in k: print(k, {c: round(x / cnt[(k, c)] / 1e6, 3) for c, x in v.items()})
except Exception as e:
    print("ERR", e)
PY
done
