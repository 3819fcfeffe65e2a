// dec.hpp -- the string backends of a parser's Decode_Field_As rules (src/flb_parser_decoder.c:85-147), written once for the
// host and for the device (like numconv.hpp): flb_unescape_string (src/flb_unescape.c:278-335), flb_mysql_unquote_string
// (:338-388) and decode_mysql_quoted's quote test (src/flb_parser_decoder.c:114-147).  `escaped_utf8` is kv_unescape of
// pkv_dev.inc (flb_unescape_string_utf8, already on the device for logfmt's quoted values).
//
// STATUS (round 2): these are not wired into write_record yet -- parsers with decoders are still refused
// (plugin/filter_gpu_plugins.c:266,390; DESIGN.md section 10 has the plan).  flbgpu_dec_simulate (dec_capi.cpp) runs them on
// the host so that tests/test_decoders_oracle.py can hold them against the oracle, which is pinned on the reference's own
// flb_unescape.c.
//
// Src: operator[](i) -> byte i of the value for i < n and 0 for i == n (the reference works on an sds copy: the byte behind
// the text is its NUL, and a trailing backslash makes flb_unescape_string copy exactly that byte).
// Sink: put(byte).  Every function returns the number of bytes it put, so a counting sink gives the size pass its answer and
// a writing sink the emit pass the same bytes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DEC_HD __host__ __device__ __forceinline__
#else
#define DEC_HD inline
#endif

namespace flbgpu {
namespace dec {

enum { BK_JSON = 0, BK_ESCAPED = 1, BK_ESCAPED_UTF8 = 2, BK_MYSQL_QUOTED = 3 };       // FLB_PARSER_DEC_* (flb_parser_decoder.h:32-35)

// flb_unescape_string: \n \a \b \t \v \f \r \\ are replaced; any other escaped character loses its backslash; a backslash
// that ends the text is followed by the byte behind the text (Src gives 0 there) and the returned length counts it
template <class Src, class Sink>
DEC_HD uint32_t unescape_plain(const Src &in, uint32_t n, Sink &o) {
    uint32_t i = 0, j = 0;
    while (i < n) {
        if (in[i] == '\\') {
            if (i + 1 < n) {
                const uint32_t c = in[i + 1];
                uint32_t r = 0x100;
                switch (c) {
                case 'n': r = '\n'; break;
                case 'a': r = '\a'; break;
                case 'b': r = '\b'; break;
                case 't': r = '\t'; break;
                case 'v': r = '\v'; break;
                case 'f': r = '\f'; break;
                case 'r': r = '\r'; break;
                case '\\': r = '\\'; break;
                default: break;
                }
                if (r != 0x100) { o.put((uint8_t) r); j++; i++; }
                i++;
                continue;
            }
            i++;
        }
        o.put((uint8_t) in[i]); j++; i++;
    }
    return j;
}

// flb_mysql_unquote_string: \n \r \t \\ \' \" \0 \Z; an unknown escape stays as it is; a backslash that ends the text stays
template <class Src, class Sink>
DEC_HD uint32_t mysql_unquote(const Src &in, uint32_t off, uint32_t n, Sink &o) {
    uint32_t i = 0, j = 0;
    while (i < n) {
        const uint32_t c = in[off + i++];
        if (c != '\\' || i >= n) { o.put((uint8_t) c); j++; continue; }
        const uint32_t e = in[off + i++];
        switch (e) {
        case 'n': o.put('\n'); j++; break;
        case 'r': o.put('\r'); j++; break;
        case 't': o.put('\t'); j++; break;
        case '\\': o.put('\\'); j++; break;
        case '\'': o.put('\''); j++; break;
        case '"': o.put('"'); j++; break;
        case '0': o.put(0); j++; break;
        case 'Z': o.put(0x1a); j++; break;
        default: o.put('\\'); o.put((uint8_t) e); j += 2; break;
        }
    }
    return j;
}

// decode_mysql_quoted: texts inside '...' or "..." are unquoted, anything else (and texts shorter than two bytes) is copied
template <class Src, class Sink>
DEC_HD uint32_t mysql_quoted(const Src &in, uint32_t n, Sink &o) {
    if (n >= 2 && ((in[0] == '\'' && in[n - 1] == '\'') || (in[0] == '"' && in[n - 1] == '"'))) return mysql_unquote(in, 1, n - 2, o);
    for (uint32_t i = 0; i < n; i++) o.put((uint8_t) in[i]);
    return n;
}

}  // namespace dec
}  // namespace flbgpu
