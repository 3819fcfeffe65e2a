#!/bin/bash
# mkstubs.sh <dir> [stock] -- emits the few headers cmake would generate (flb_info.h, mk_info.h, cio_info.h, the
# *_version.h / *_info.h of cfl, cmetrics, ctraces, cprofiles) so that the headers of a fluent-bit source tree
# can be included without running its build system.  Used by build.sh / check_syntax.sh here and by
# oracle/Makefile (ref_packfmt).
set -e
T=$1
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
if [ "$2" = "stock" ]; then
# the switches of a stock engine build that move struct members in the headers a plugin includes (CMakeLists.txt:189,195,204,211:
# FLB_TLS, FLB_CHUNK_TRACE, FLB_HTTP_SERVER, FLB_STREAM_PROCESSOR default to Yes; flb_config.h:227,302,314, flb_input.h:261,519).
# A plugin loaded into an engine configured differently must be built against THAT engine's flb_info.h (build.sh FLB_INFO_DIR=...).
sed -i 's|^#endif$|#define FLB_HAVE_HTTP_SERVER 1\n#define FLB_HAVE_CHUNK_TRACE 1\n#define FLB_HAVE_TLS 1\n#define FLB_HAVE_STREAM_PROCESSOR 1\n#endif|' $T/fluent-bit/flb_info.h
fi
printf '#ifndef MK_CORE_INFO_H\n#define MK_CORE_INFO_H\n#define MK_HAVE_EVENTFD 1\n#define MK_HAVE_C_TLS 1\n#define MK_HAVE_UNISTD_H 1\n#define MK_HAVE_SYS_UIO_H 1\n#endif\n' > $T/mk_core/mk_core_info.h
printf '#ifndef MK_INFO_H\n#define MK_INFO_H\n#include <monkey/mk_core.h>\n#define MK_VERSION_STR "0"\n#define MK_PATH_CONF ""\n#endif\n' > $T/monkey/mk_info.h
printf '#ifndef CIO_INFO_H\n#define CIO_INFO_H\n#define CIO_HAVE_BACKEND_FILESYSTEM 1\n#endif\n' > $T/chunkio/cio_info.h
for lib in cfl:CFL:cfl cmetrics:CMT:cmt ctraces:CTR:ctr cprofiles:CPROF:cprof; do
  IFS=: read d P p <<< "$lib"
  printf "#ifndef ${P}_VERSION_H\n#define ${P}_VERSION_H\n#define ${P}_VERSION_MAJOR 0\n#define ${P}_VERSION_MINOR 0\n#define ${P}_VERSION_PATCH 0\n#define ${P}_VERSION_STR \"0\"\n#endif\n" > $T/$d/${p}_version.h
  printf "#ifndef ${P}_INFO_H\n#define ${P}_INFO_H\n#define CFL_HAVE_TIMESPEC_GET 1\n#define CFL_HAVE_GMTIME_R 1\n#define CMT_HAVE_TIMESPEC_GET 1\n#define CMT_HAVE_GMTIME_R 1\n#endif\n" > $T/$d/${p}_info.h
done
