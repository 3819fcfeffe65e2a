#!/bin/bash
# round 4, call 1: the NFA engine on the device + the suites that share its code paths, then the headline bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nfa_gpu.py -x -q -m gpu > gpurun_out/r4_nfa.log 2>&1; echo "nfa rc=$?" >> gpurun_out/r4_nfa.log
tail -15 gpurun_out/r4_nfa.log
timeout 900 python -m pytest tests/test_kat_gpu.py tests/test_multiline_gpu.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_kat.log 2>&1; echo "kat rc=$?" >> gpurun_out/r4_kat.log
tail -8 gpurun_out/r4_kat.log
timeout 600 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err; echo "bench rc=$?"
head -c 1500 gpurun_out/r4_bench1.json
