#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C=$PWD/fluent-bit_amd/csrc
echo "== base (flat loads)"; timeout 200 python tools/r4_perf1.py 10000000 16 0,1
echo "== global loads"; FLBGPU_LIB=$C/libflbgpu_g.so timeout 200 python tools/r4_perf1.py 10000000 16,12 0,1
echo "== global loads + stream"; FLBGPU_LIB=$C/libflbgpu_gs.so timeout 200 python tools/r4_perf1.py 10000000 16,12 0
