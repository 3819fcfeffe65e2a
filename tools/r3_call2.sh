# second GPU call of round 3: k_parser_reg's timeline (s_memtime stamps), skip builds for both ingests, the reworked bench.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3b
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/trace_reg.py 10000000 0,3 > $O/trace.log 2>&1
cat $O/trace.log
timeout 300 python tools/perf_stage.py 10000000 "0:16,3:16" > $O/perf_stage.log 2>&1
cat $O/perf_stage.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["frac"], "verify", d.get("verify"))
    print("cpu", json.dumps(d.get("cpu_baseline"))[:900])
    for k, v in (d.get("secondary") or {}).items():
        print(k, json.dumps(v)[:1500])
except Exception as e:
    print("bench line unreadable", e)
PY
timeout 600 python bench.py --gpus 2 --records 2000000 --no-cpu --steps 3 --ndjson-lines 2000000 --l2m-records 32000000 > $O/bench_2rank.json 2> $O/bench_2rank.err
tail -5 $O/bench_2rank.err
python3 - $O/bench_2rank.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("2-rank: n_gpus", d["n_gpus"], "value", d["value"], "backend", d.get("collective_backend"), "rccl_ranks", d.get("rccl_ranks"))
    s = d.get("secondary") or {}
    for k in ("l2m_counter", "l2m_histogram", "l2m_merge_check", "flb_sp_group_by", "error"):
        if k in s: print(k, json.dumps(s[k])[:600])
except Exception as e:
    print("2-rank line unreadable", e)
PY
