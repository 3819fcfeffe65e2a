#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== stream, 4 waves/EU (128 VGPR), 16 waves x1"; timeout 300 python tools/r4_perf1.py 10000000 16 0
echo "== stream, 5 waves/EU (96 VGPR): 10 waves x2 WG"; FLBGPU_LIB=$PWD/fluent-bit_amd/csrc/libflbgpu_s5.so FLBGPU_TILE_GRID_MULT=2 timeout 300 python tools/r4_perf1.py 10000000 10 0,1
echo "== stream, 6 waves/EU (80 VGPR): 12 waves x2 WG"; FLBGPU_LIB=$PWD/fluent-bit_amd/csrc/libflbgpu_s6.so FLBGPU_TILE_GRID_MULT=2 timeout 300 python tools/r4_perf1.py 10000000 12 0,1
echo "== stream, 6 waves/EU lib but 16 waves x1"; FLBGPU_LIB=$PWD/fluent-bit_amd/csrc/libflbgpu_s6.so timeout 300 python tools/r4_perf1.py 10000000 16 0
