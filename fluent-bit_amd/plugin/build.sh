#!/bin/bash
# build.sh -- builds the loadable plugins against the headers of a fluent-bit source tree (no cmake: the
# generated headers are the stubs of mkstubs.sh -- with the layout switches of a STOCK engine build --, or, with FLB_INFO_DIR=<dir>, the
# ones a configured engine tree generated) plus the test host that loads them (tests/plugin_host.c: test infrastructure, not part of the
# package; NO_HOST=1 leaves it out):
#   _build/flb-filter_grep_gpu.so  _build/flb-filter_parser_gpu.so  _build/flb-filter_log_to_metrics_gpu.so
#   _build/plugin_host
# The file names and the exported data symbols filter_<x>_gpu_plugin are what `fluent-bit -e <file>` /
# [PLUGINS] Path expect (src/flb_plugin.c:110-168).  Usage: build.sh [/path/to/fluent-bit] (default /root/reference)
set -e
R=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
B=${PLUGIN_OUT:-$HERE/_build}
mkdir -p $B
if [ -n "$FLB_INFO_DIR" ]; then
  # the headers a configured engine tree generated (oracle/build_engine.sh: oracle/_ref/engine/include): the plugins are then built
  # against the struct layouts of exactly that engine
  T=$FLB_INFO_DIR
else
  T=$B/stub
  $HERE/mkstubs.sh $T stock
fi
NG=$(dirname $(dirname $(find $R/lib -name nghttp2.h | head -1)))
MP=$(dirname $(dirname $(find $R/lib -name mpack.h | head -1)))
INC="-I$T -I$T/monkey -I$R/include -I$R/lib/monkey/include -I$R/lib/monkey/include/monkey -I$R/lib/cfl/include -I$R/lib/cfl/lib/xxhash
 -I$R/lib/cmetrics/include -I$R/lib/ctraces/include -I$R/lib/msgpack-c/include -I$R/lib/flb_libco -I$R/lib/onigmo -I$R/lib/cprofiles/include
 -I$R/lib/rbtree -I$R/lib/chunkio/include -I$R/lib/jsmn -I$R/lib/miniz -I$R/lib/tutf8e/include -I$NG -I$MP -I$HERE/../../include"
CS=$HERE/../csrc
i=1
for x in grep parser log_to_metrics; do
  gcc -O2 -fPIC -shared -Wall -Wno-unused-function -DFLBGPU_ONLY=$i $INC -o $B/flb-filter_${x}_gpu.so $HERE/filter_gpu_plugins.c \
      -L$CS -lflbgpu -Wl,-rpath,${PLUGIN_RPATH:-'$ORIGIN/../../csrc'}
  i=$((i+1))
done
# the host exports the engine symbols the plugins import; with the reference's real cmetrics at hand (oracle/_ref, built by
# oracle/Makefile) also the ones filter_log_to_metrics_gpu needs
CMT=$HERE/../../oracle/_ref
if [ -n "$NO_HOST" ]; then
  :
elif [ -f $CMT/libcmetrics_ref.so ]; then
  gcc -O2 -Wall -rdynamic -DHOST_WITH_CMT $INC -I$CMT/stub3 -o $B/plugin_host $HERE/../../tests/plugin_host.c -L$CMT -lcmetrics_ref -lm -Wl,-rpath,'$ORIGIN/../../../oracle/_ref' -ldl
else
  gcc -O2 -Wall -rdynamic $INC -o $B/plugin_host $HERE/../../tests/plugin_host.c -ldl
fi
echo "built: $(ls $B | grep -v stub | tr '\n' ' ')"
