#!/bin/bash
# oracle/build_engine.sh -- TEST INFRASTRUCTURE.  Builds the REFERENCE'S OWN ENGINE (libfluent-bit.so + bin/fluent-bit, with
# in_lib / in_dummy / in_emitter / out_lib / out_null and the three built-in filters of the path) so that the drop-in plugins can be
# loaded by the real flb_plugin_load (src/flb_plugin.c:194-320) and driven by the real flb_processor_run / flb_lib, and so that
# BASELINE.json configs[0] (in_dummy -> filter_grep -> out_null) can run.
#
# The reference's cmake writes generated headers INTO its source tree (include/fluent-bit/flb_info.h ...), so the build needs a
# writable copy: it lives in a scratch directory OUTSIDE the repository ($FLB_ENGINE_SCRATCH, default /tmp/flb_engine_build) and is
# deleted afterwards unless KEEP=1.  Nothing of the reference's sources enters the repository; what is kept, under oracle/_ref/engine/
# (git-ignored, travels to the GPU box):
#   lib/libfluent-bit.so   bin/fluent-bit   include/  (ONLY the headers the configure step generated: flb_info.h, *_info.h, *_version.h)
#   engine_host            (oracle/engine/engine_host.c linked against that libfluent-bit.so)
#   plugins/flb-filter_{grep,parser,log_to_metrics}_gpu.so   (fluent-bit_amd/plugin/filter_gpu_plugins.c compiled against THAT flb_info.h)
#
# What is NOT the reference's in that build:
#   * flex / bison are not in this image: oracle/engine/bin/{flex,bison} stand in for them and hand cmake the hand-written grammars
#     of oracle/grammar_ra.inc (src/record_accessor/ra.l, ra.y) and oracle/grammar_sql.inc (stream_processor/parser/sql.l, sql.y);
#   * src/flb_blob_db.c is emptied in the copy: with FLB_SQLDB=Off (lib/sqlite-amalgamation-*/sqlite3.c is absent) its stub block
#     does not compile (SURVEY.md probe table); nothing in this plugin set calls it;
#   * cmetrics' prometheus TEXT decoder is off (it needs a real flex / bison).
# Options: the core switches are the defaults of CMakeLists.txt (HTTP server, chunk trace, TLS, stream processor, record accessor,
# metrics ... on) -- the struct layouts are therefore the ones of a stock build; the plugin set is FLB_MINIMAL + what the path needs.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${REF:-/root/reference}
S=${FLB_ENGINE_SCRATCH:-/tmp/flb_engine_build}
OUT=$HERE/_ref/engine
[ -d "$REF/src" ] || { echo "build_engine.sh: no reference tree at $REF" >&2; exit 1; }
if [ -z "$FORCE" ] && [ -f $OUT/lib/libfluent-bit.so ] && [ -f $OUT/bin/fluent-bit ] && [ $OUT/lib/libfluent-bit.so -nt $HERE/grammar_ra.inc ] \
   && [ $OUT/lib/libfluent-bit.so -nt $HERE/grammar_sql.inc ] && [ $OUT/lib/libfluent-bit.so -nt $HERE/build_engine.sh ]; then
  echo "engine: up to date ($OUT/lib/libfluent-bit.so)"
else
  rm -rf $S; mkdir -p $S
  cp -r $REF $S/src
  echo '/* emptied by oracle/build_engine.sh (see its header) */ typedef int flb_blob_db_emptied;' > $S/src/src/flb_blob_db.c
  mkdir -p $S/build
  ( cd $S/build && PATH=$HERE/engine/bin:$PATH cmake -G Ninja ../src -DCMAKE_BUILD_TYPE=Release -DFLB_MINIMAL=On -DFLB_DEBUG=Off -DFLB_RELEASE=On \
      -DFLB_FILTER_GREP=On -DFLB_FILTER_PARSER=On -DFLB_FILTER_LOG_TO_METRICS=On -DFLB_IN_LIB=On -DFLB_OUT_LIB=On -DFLB_IN_DUMMY=On \
      -DFLB_OUT_NULL=On -DFLB_IN_EMITTER=On -DFLB_SQLDB=Off -DFLB_LUAJIT=Off -DFLB_WASM=Off -DFLB_KAFKA=Off -DFLB_CONFIG_YAML=Off \
      -DFLB_ZIG=Off -DFLB_PROXY_GO=Off -DFLB_EXAMPLES=Off -DFLB_BACKTRACE=Off -DFLB_CUSTOM_CALYPTIA=Off -DCMT_PROMETHEUS_TEXT_DECODER=Off \
      > cmake.log 2>&1 && PATH=$HERE/engine/bin:$PATH ninja > ninja.log 2>&1 ) || { tail -30 $S/build/cmake.log $S/build/ninja.log >&2; exit 1; }
  rm -rf $OUT; mkdir -p $OUT/lib $OUT/bin $OUT/include
  cp $S/build/lib/libfluent-bit.so $OUT/lib/
  cp $S/build/bin/fluent-bit $OUT/bin/
  strip --strip-debug $OUT/lib/libfluent-bit.so $OUT/bin/fluent-bit
  # the generated headers only: those in the copy that the reference tree does not have, and those of the build directory
  ( cd $S/src && for f in $(find include lib/cfl/include lib/cmetrics/include lib/ctraces/include lib/monkey/include -name "*.h"); do
      [ -f $REF/$f ] || { d=$OUT/include/$(echo $f | sed 's|^lib/[^/]*/include/||; s|^include/||'); mkdir -p $(dirname $d); cp $f $d; }; done )
  ( cd $S/build && for f in $(find generated/include lib/cprofiles/include lib/monkey/include lib/chunkio/include lib/nghttp2-*/lib/includes lib/miniz -name "*.h"); do
      d=$OUT/include/$(echo $f | sed 's|^generated/include/||; s|^lib/[^/]*/include/||; s|^lib/nghttp2-[^/]*/lib/includes/||; s|^lib/miniz/||'); mkdir -p $(dirname $d); cp $f $d; done )
  [ -n "$KEEP" ] || rm -rf $S
fi
NG=$(dirname $(dirname $(find $REF/lib -name nghttp2.h | head -1)))
MP=$(dirname $(dirname $(find $REF/lib -name mpack.h | head -1)))
INC="-I$OUT/include -I$OUT/include/monkey -I$REF/include -I$REF/lib/monkey/include -I$REF/lib/monkey/include/monkey -I$REF/lib/cfl/include -I$REF/lib/cfl/lib/xxhash
 -I$REF/lib/cmetrics/include -I$REF/lib/ctraces/include -I$REF/lib/msgpack-c/include -I$REF/lib/flb_libco -I$REF/lib/onigmo -I$REF/lib/cprofiles/include
 -I$REF/lib/rbtree -I$REF/lib/chunkio/include -I$REF/lib/jsmn -I$REF/lib/miniz -I$REF/lib/tutf8e/include -I$NG -I$MP -I$REF/lib/lwrb/lwrb/src/include"
gcc -O2 -g -Wall -Wno-unused-function $INC -o $OUT/engine_host $HERE/engine/engine_host.c -L$OUT/lib -lfluent-bit -Wl,-rpath,'$ORIGIN/lib' -lpthread -ldl -lm
# the drop-in plugins against the headers of THIS engine build
FLB_INFO_DIR=$OUT/include PLUGIN_OUT=$OUT/plugins PLUGIN_RPATH='$ORIGIN/../../../../fluent-bit_amd/csrc' NO_HOST=1 bash $HERE/../fluent-bit_amd/plugin/build.sh $REF
echo "engine: $(ls $OUT $OUT/plugins | tr '\n' ' ')"
