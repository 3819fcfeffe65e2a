#!/bin/bash
# host-only regex code of the product under ASan + UBSan (see tools/sanitize_rx.py); needs no GPU
set -e
cd "$(dirname "$0")/../fluent-bit_amd/csrc"
mkdir -p /tmp/san
cat > /tmp/san/stub.cpp <<'EOS'
#include <cstdarg>
#include <cstdio>
#include "host_int.hpp"
static thread_local char g_e[1024];
namespace flbgpu { void set_err(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_e, sizeof g_e, fmt, ap); va_end(ap); } }
extern "C" const char *flbgpu_last_error(void) { return g_e; }
EOS
g++ -O1 -g -std=c++17 -fPIC -shared -w -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I. -I../../include \
    rx.cpp fx.cpp rx_capi.cpp /tmp/san/stub.cpp -o /tmp/san/librx_san.so
cd ../..
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python tools/sanitize_rx.py -k "not deep_searches" "$@"
# (test_deep_searches_.. measures the matcher's own stack: ASan's frames are several times the real ones and the search ends on the
# stack limit -- an answer of the sanitized build, not a finding)
# ... and the matcher's reentrancy (flbgpu.cpp parallel_rows searches one program from up to 16 threads) under TSan
(cd fluent-bit_amd/csrc && g++ -O1 -g -std=c++17 -fPIC -shared -w -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I. -I../../include \
    rx.cpp fx.cpp rx_capi.cpp /tmp/san/stub.cpp -o /tmp/san/librx_tsan.so)
# (the matcher's per-thread stack is a thread_local with a destructor: TSan does not see the join that orders it before the C
# library frees the thread's TLS block -- _dl_deallocate_tls --, suppressed by name)
echo "race:_dl_deallocate_tls" > /tmp/san/tsan.supp
TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0:suppressions=/tmp/san/tsan.supp LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" python tools/tsan_rx.py
