#!/bin/bash
# tools/ab_libs.sh NAME[:ENV=V] ... -- on the GPU box: k_parser_reg's launch time (tools/perf_reg.py, 10 M records, 12 waves) for several
# builds of the library (fluent-bit_amd/csrc/libflbgpu_NAME.so from tools/variant.sh; "main" = libflbgpu.so), interleaved, three rounds:
# boxes differ by several percent, so builds are only compared within one call.
cd /root/repo
for rep in 1 2; do
    for spec in "$@"; do
        name=${spec%%:*}; envs=""
        [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
        lib=/root/repo/fluent-bit_amd/csrc/libflbgpu_$name.so
        [ "$name" = main ] && lib=/root/repo/fluent-bit_amd/csrc/libflbgpu.so
        printf "%-28s " "$spec"
        env FLBGPU_LIB=$lib $envs python tools/perf_reg.py 10000000 12 0 2>&1 | tail -1
    done
done
