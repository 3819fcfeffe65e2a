#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel in a hipcc object file (the AMDGPU code object's metadata notes):
#   tools/kernel_regs.sh fluent-bit_amd/csrc/build/kernels_tile.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$1"
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co --unbundle
$LLVM/llvm-readelf --notes $tmp/dev.co | awk '
/\.private_segment_fixed_size:/ {scr=$2}
/\.group_segment_fixed_size:/ {lds=$2}
/\.sgpr_count:/ {sg=$2}
/\.symbol:/ {sym=$2}
/\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {sp=$2}
/\.wavefront_size:/ {printf "%-100s vgpr %3s sgpr %3s spill %3s scratch %5s lds %6s\n", sym, vg, sg, sp, scr, lds; sp=0}
' | c++filt | sed 's/(flbgpu::[A-Za-z]*)//; s/\.kd//'
rm -rf $tmp
