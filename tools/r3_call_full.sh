# full GPU suite + the default bench line (round 3, after decoders + multiline)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3g
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"], "verify", d.get("verify"))
    print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
    for k, v in (d.get("secondary") or {}).items():
        print(k, json.dumps(v)[:700])
except Exception as e:
    print("bench line unreadable", e)
PY
