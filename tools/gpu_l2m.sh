#!/bin/bash
# log_to_metrics on the GPU box: its suites, then the histogram on msgpack floats in the reference's order (bench.measure_l2m_float: the
# sequential sum's chain, kernels_seqsum.hip) against the real cmetrics
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_l2m_gpu.py tests/test_rccl_ranks_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1500
import sys, json, types
sys.path.insert(0, "tests")
import flbamd_loader, torch
import bench as b
g = flbamd_loader.load(); g.init(0)
args = types.SimpleNamespace(ndjson_lines=10_000_000, no_cpu=True, steps=5, warmup=1)
out = b.measure_config2(g, torch, g.lib(), 0, 1, args)
args.no_cpu = False
r = b.measure_l2m_float(g, torch, g.lib(), out["_events_chunk"], 0, 1, args)
print(json.dumps(r))
PY
