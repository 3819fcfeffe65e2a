#!/bin/bash
# tools/prof_kernel.sh OUTNAME KERNEL_SUBSTRING -- CMD...   on the GPU box: SQ instruction / wait counters and FETCH / WRITE of the kernels
# whose name holds KERNEL_SUBSTRING, each counter group in a pass of its own (kernel-trace only), one JSON under gpurun_out/OUTNAME.json
name=$1; pat=$2; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$name
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- "$@" > $O/run$i.log 2>&1)
done
python3 - $O "$pat" $R/gpurun_out/$name.json "$*" <<'PY'
import csv, sys, collections, json, glob, os
O, pat, dst, cmd = sys.argv[1:5]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if pat in k: res[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=max(len(v) for v in cs.values())) for k, cs in res.items()}
json.dump({"command": cmd, "unit": "per launch; FETCH_SIZE / WRITE_SIZE in KiB (FETCH x 2 for wide coalesced reads, MI355X guide)", "kernels": out}, open(dst, "w"), indent=1)
for k, v in sorted(out.items()): print(k.split("::")[-1][:40], {c.replace("SQ_", ""): round(x / 1e6, 3) for c, x in sorted(v.items())})
PY
