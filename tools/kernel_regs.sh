#!/bin/bash
# tools/kernel_regs.sh [object ...] -- registers, spills, scratch and LDS of every kernel in the gfx950 code objects of the
# kernel translation units (default: fluent-bit_amd/csrc/build/kernels*.o): the numbers the compiler wrote into the
# kernels' metadata.  No GPU needed; run before and after an edit to see that a tuned kernel kept its budget.
set -e
here=$(cd "$(dirname "$0")" && pwd)
objs=("$@")
[ ${#objs[@]} -eq 0 ] && objs=("$here"/../fluent-bit_amd/csrc/build/kernels*.o)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
for o in "${objs[@]}"; do
    /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$o" "$tmp/fat.bin"
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle
    /opt/rocm/lib/llvm/bin/llvm-readobj --notes "$tmp/dev.co" | awk -v obj="$(basename "$o")" '
    /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {sp=$2}
    /\.private_segment_fixed_size:/ {scr=$2}
    /\.wavefront_size:/ {printf "%-18s %-100s vgpr %4s spill %4s scratch %6s sgpr %4s lds %6s\n", obj, name, vg, sp, scr, sg, lds}'
done | sort
