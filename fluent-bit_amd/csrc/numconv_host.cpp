// numconv_host.cpp -- host instantiation of numconv.hpp (diagnostic entry points: the CPU tests fuzz
// these against glibc; the kernels use the device instantiation of the same header).
#include "numconv.hpp"
#include <string.h>

namespace flbgpu { namespace nc {
const uint64_t g_pow5_host[2 * (P5_QMAX - P5_QMIN + 1)] = {
#include "pow5_table.inc"
};
struct PtrSrc { const uint8_t *p; uint32_t operator[](uint32_t i) const { return p[i]; } };
struct BufDst { char *p; int n; void put(uint32_t c) { p[n++] = (char) c; } };
}}

using namespace flbgpu::nc;

extern "C" {
int flbgpu_nc_scan_double(const char *s, int len, int mode, int exact, double *out, int *consumed) {
    PtrSrc src{(const uint8_t *) s};
    ScanResult r = exact ? scan_double<true>(src, (uint32_t) len, mode) : scan_double<false>(src, (uint32_t) len, mode);
    if (out) memcpy(out, &r.bits, 8);
    if (consumed) *consumed = (int) r.consumed;
    return r.status;
}
int flbgpu_nc_fmt_f6(double v, char *buf, int cap) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    BufDst d{buf, 0};
    return fmt_f6(bits, d, cap);
}
int flbgpu_nc_fmt_json_double(double v, int nan_to_null, char *buf) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    BufDst d{buf, 0};
    return fmt_json_double(bits, nan_to_null != 0, d);
}
int flbgpu_nc_fmt_ld(long long v, char *buf) {
    BufDst d{buf, 0};
    return fmt_ld((int64_t) v, d);
}
}
