#!/bin/bash
# tools/variant.sh NAME "-DMACRO ..." [unit]  -- an experiment build of ONE kernel translation unit (default kernels_tile) linked with
# the other objects of the regular build into fluent-bit_amd/csrc/libflbgpu_NAME.so; run it with FLBGPU_LIB=<that path>.
set -e
cd "$(dirname "$0")/../fluent-bit_amd/csrc"
name=$1; flags=$2; unit=${3:-kernels_tile}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c -o build/var_${unit}_$name.o $unit.hip 2>&1 | grep -E "error" || true
objs=""
for o in kernels kernels_pjson kernels_l2m kernels_misc kernels_fused kernels_tile kernels_tail kernels_fmt kernels_jtile kernels_glane kernels_l2mlane kernels_seqsum kernels_perm calib flbgpu packfmt tail ml l2m sp json index rx fx rx_capi dec_capi numconv_host; do
    if [ "$o" = "$unit" ]; then objs="$objs build/var_${unit}_$name.o"; else objs="$objs build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -o libflbgpu_$name.so $objs -lpthread
ls -la libflbgpu_$name.so
