#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1200 python -m pytest tests/test_json_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python tools/perf_json.py 4000000 2>&1 | tail -2 | tee gpurun_out/r3k/perf_json.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
