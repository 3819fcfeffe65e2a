// dec_capi.cpp -- host execution of dec.hpp (include/flb_gpu_dec.h): the decoders' string backends as the kernels will run
// them, for the CPU-only unit tests.  The filters never call this.
#include <stddef.h>
#include <stdint.h>
#include "dec.hpp"
#include "../../include/flb_gpu_dec.h"

namespace {
struct HostSrc {
    const uint8_t *p; uint32_t n;
    uint32_t operator[](uint32_t i) const { return i < n ? p[i] : 0u; }
};
struct HostSink {
    uint8_t *o; size_t cap, len;
    void put(uint32_t b) { if (o && len < cap) o[len] = (uint8_t) b; len++; }
};
}  // namespace

extern "C" int64_t flbgpu_dec_simulate(int backend, const void *in, size_t n, void *out, size_t cap)
{
    if ((!in && n) || n > 0xFFFFFFF0u) return -1;
    HostSrc s{(const uint8_t *) in, (uint32_t) n};
    HostSink k{(uint8_t *) out, out ? cap : 0, 0};
    uint32_t r;
    switch (backend) {
    case flbgpu::dec::BK_ESCAPED: r = flbgpu::dec::unescape_plain(s, (uint32_t) n, k); break;
    case flbgpu::dec::BK_MYSQL_QUOTED: r = flbgpu::dec::mysql_quoted(s, (uint32_t) n, k); break;
    case flbgpu::dec::BK_ESCAPED_UTF8: r = flbgpu::dec::unescape_utf8<false>(s, (uint32_t) n, k); break;
    case 102: r = flbgpu::dec::unescape_utf8<true>(s, (uint32_t) n, k); break;      // logfmt's use of it (strlen of the result)
    default: return -1;                                    // json: json_dev.inc
    }
    return r == k.len ? (int64_t) r : -1;
}
