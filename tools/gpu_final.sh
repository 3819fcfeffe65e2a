#!/bin/bash
# the round's closing evidence on the GPU box: the whole -m gpu suite, then tools/profile_bench.sh (kernel stats, FETCH / WRITE passes of the
# headline command, the default bench line LAST so that it quotes the PMC summary of the same kernel sources)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/final_suite.log
cat gpurun_out/final_suite.log
SKIP_CAL=1 timeout 2400 bash tools/profile_bench.sh > gpurun_out/final_profile.log 2>&1
tail -5 gpurun_out/final_profile.log | cut -c1-600
python - <<'PY'
import json
try:
    p = json.loads(open("gpurun_out/profile/bench_default.json").read().strip().splitlines()[-1])
    print("value", p["value"], "ms/step", p["ms_per_step"], "roofline", p["roofline"]["frac"], p["roofline"]["traffic"])
    print("cpu", p["cpu_baseline"])
    s = p.get("secondary", {})
    for k in ("config2_ndjson_grep32", "mixed_shapes", "msgpack_to_json", "l2m_histogram", "l2m_histogram_reference_order", "engine_hosted", "host_level", "parser_only"):
        print(k, json.dumps(s.get(k))[:700])
except Exception as e:
    print("bench line not read:", e)
PY
