#!/bin/bash
# build.sh -- builds the loadable plugins against the headers of a fluent-bit source tree (no cmake: the
# generated headers are the stubs of check_syntax.sh) plus the test host that loads them:
#   _build/flb-filter_grep_gpu.so  _build/flb-filter_parser_gpu.so  _build/flb-filter_log_to_metrics_gpu.so
#   _build/plugin_host
# The file names and the exported data symbols filter_<x>_gpu_plugin are what `fluent-bit -e <file>` /
# [PLUGINS] Path expect (src/flb_plugin.c:110-168).  Usage: build.sh [/path/to/fluent-bit] (default /root/reference)
set -e
R=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
B=$HERE/_build
T=$B/stub
mkdir -p $T/fluent-bit $T/monkey $T/mk_core $T/cfl $T/cmetrics $T/ctraces $T/cprofiles $T/chunkio
cat > $T/fluent-bit/flb_info.h <<'H'
#ifndef FLB_INFO_H
#define FLB_INFO_H
#define FLB_HAVE_REGEX 1
#define FLB_HAVE_RECORD_ACCESSOR 1
#define FLB_HAVE_PARSER 1
#define FLB_HAVE_METRICS 1
#endif
H
printf '#ifndef MK_CORE_INFO_H\n#define MK_CORE_INFO_H\n#define MK_HAVE_EVENTFD 1\n#define MK_HAVE_C_TLS 1\n#define MK_HAVE_UNISTD_H 1\n#define MK_HAVE_SYS_UIO_H 1\n#endif\n' > $T/mk_core/mk_core_info.h
printf '#ifndef MK_INFO_H\n#define MK_INFO_H\n#include <monkey/mk_core.h>\n#define MK_VERSION_STR "0"\n#define MK_PATH_CONF ""\n#endif\n' > $T/monkey/mk_info.h
printf '#ifndef CIO_INFO_H\n#define CIO_INFO_H\n#define CIO_HAVE_BACKEND_FILESYSTEM 1\n#endif\n' > $T/chunkio/cio_info.h
for lib in cfl:CFL:cfl cmetrics:CMT:cmt ctraces:CTR:ctr cprofiles:CPROF:cprof; do
  IFS=: read d P p <<< "$lib"
  printf "#ifndef ${P}_VERSION_H\n#define ${P}_VERSION_H\n#define ${P}_VERSION_MAJOR 0\n#define ${P}_VERSION_MINOR 0\n#define ${P}_VERSION_PATCH 0\n#define ${P}_VERSION_STR \"0\"\n#endif\n" > $T/$d/${p}_version.h
  printf "#ifndef ${P}_INFO_H\n#define ${P}_INFO_H\n#define CFL_HAVE_TIMESPEC_GET 1\n#define CFL_HAVE_GMTIME_R 1\n#define CFL_HAVE_CLOCK_GET_TIME 1\n#endif\n" > $T/$d/${p}_info.h
done
NG=$(dirname $(dirname $(find $R/lib -name nghttp2.h | head -1)))
MP=$(dirname $(dirname $(find $R/lib -name mpack.h | head -1)))
INC="-I$T -I$R/include -I$R/lib/monkey/include -I$R/lib/monkey/include/monkey -I$R/lib/cfl/include -I$R/lib/cfl/lib/xxhash
 -I$R/lib/cmetrics/include -I$R/lib/ctraces/include -I$R/lib/msgpack-c/include -I$R/lib/flb_libco -I$R/lib/onigmo -I$R/lib/cprofiles/include
 -I$R/lib/rbtree -I$R/lib/chunkio/include -I$R/lib/jsmn -I$R/lib/miniz -I$R/lib/tutf8e/include -I$NG -I$MP -I$HERE/../../include"
CS=$HERE/../csrc
i=1
for x in grep parser log_to_metrics; do
  gcc -O2 -fPIC -shared -Wall -Wno-unused-function -DFLBGPU_ONLY=$i $INC -o $B/flb-filter_${x}_gpu.so $HERE/filter_gpu_plugins.c \
      -L$CS -lflbgpu -Wl,-rpath,'$ORIGIN/../../csrc'
  i=$((i+1))
done
gcc -O2 -Wall -rdynamic $INC -o $B/plugin_host $HERE/plugin_host.c -ldl
echo "built: $(ls $B | grep -v stub | tr '\n' ' ')"
