#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C=$PWD/fluent-bit_amd/csrc
echo "== base (skip 4: the walk's result dropped)"; timeout 200 python tools/r4_perf2.py 10000000 4
for X in 1 2 3; do echo "== REG_X=$X"; FLBGPU_LIB=$C/libflbgpu_x$X.so timeout 200 python tools/r4_perf2.py 10000000 4; done
