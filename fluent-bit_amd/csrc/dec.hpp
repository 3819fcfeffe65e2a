// dec.hpp -- the string backends of a parser's Decode_Field_As rules (src/flb_parser_decoder.c:85-147), written once for the
// host and for the device (like numconv.hpp): flb_unescape_string (src/flb_unescape.c:278-335), flb_mysql_unquote_string
// (:338-388), decode_mysql_quoted's quote test (src/flb_parser_decoder.c:114-147) and flb_unescape_string_utf8 (:186-277; the
// same steps pkv_dev.inc's kv_unescape takes for logfmt's quoted values, here over a byte source).
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

// ---- escaped_utf8: flb_unescape_string_utf8 (src/flb_unescape.c:186-277) with u8_read_escape_sequence (:78-184), the same
// steps as kv_unescape / kv_read_escape of pkv_dev.inc (logfmt's quoted values) over a Src.  STOP_AT_NUL: logfmt takes strlen()
// of the result (a decoded NUL ends the value); the decoder uses the returned length (decoded NULs stay).
DEC_HD bool is_hex(uint32_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'F') || (c >= 'a' && c <= 'f'); }
template <class Src>
DEC_HD uint32_t hexval(const Src &in, uint32_t at, uint32_t n) {
    uint32_t x = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = in[at + i];
        x = x * 16 + (c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
    }
    return x;
}
// at: index of the character behind the backslash; size: bytes from there to the end of the value; returns the characters consumed
template <class Src>
DEC_HD uint32_t read_escape(const Src &in, uint32_t at, uint32_t size, uint32_t *dest) {
    const uint32_t c0 = in[at];
    uint32_t ch = (uint32_t) (int32_t) (int8_t) c0;          // `char` is signed in the reference build
    uint32_t i = 1, dno = 0;
    if (c0 == 'n') ch = '\n';
    else if (c0 == 't') ch = '\t';
    else if (c0 == 'r') ch = '\r';
    else if (c0 == 'b') ch = '\b';
    else if (c0 == 'f') ch = '\f';
    else if (c0 == 'v') ch = '\v';
    else if (c0 == 'a') ch = 7;
    else if (c0 >= '0' && c0 <= '7') {
        uint32_t x = 0;
        i = 0;
        do { x = x * 8 + (in[at + i] - '0'); i++; dno++; } while (i < size && in[at + i] >= '0' && in[at + i] <= '7' && dno < 3);
        ch = x;
    }
    else if (c0 == 'x') {
        while (i < size && is_hex(in[at + i]) && dno < 2) { i++; dno++; }
        if (dno > 0) ch = hexval(in, at + 1, dno);
    }
    else if (c0 == 'u') {
        while (i < size && is_hex(in[at + i]) && dno < 4) { i++; dno++; }
        if (dno != 4 && dno > 0) ch = 0xFFFD;                                   // incomplete
        else {
            ch = hexval(in, at + 1, dno);                                       // (no digit at all: strtol("") = 0)
            if (ch >= 0xDC00 && ch <= 0xDFFF) ch = 0xFFFD;                      // low surrogate first
            else if (ch >= 0xD800 && ch <= 0xDBFF) {
                if (i + 2 < size && in[at + i] == '\\' && in[at + i + 1] == 'u') {
                    dno = 0;
                    i += 2;
                    const uint32_t ls = i;
                    while (i < size && is_hex(in[at + i]) && dno < 4) { i++; dno++; }
                    if (dno != 4 && dno > 0) ch = 0xFFFD;
                    else {
                        const uint32_t low = hexval(in, at + ls, dno);
                        if (low >= 0xDC00 && low <= 0xDFFF) ch = 0x10000 + (((ch - 0xD800) << 10) | (low - 0xDC00));
                        else ch = 0xFFFD;
                    }
                }
                else ch = 0xFFFD;
            }
        }
    }
    else if (c0 == 'U') {
        while (i < size && is_hex(in[at + i]) && dno < 8) { i++; dno++; }
        if (dno > 0) ch = hexval(in, at + 1, dno);
    }
    *dest = ch;
    return i;
}
template <bool STOP_AT_NUL, class Src, class Sink>
DEC_HD uint32_t unescape_utf8(const Src &in, uint32_t sz, Sink &o) {
    uint32_t pos = 0, count_out = 0;
    while (pos < sz && in[pos] != 0) {
        uint32_t ch, esc_in;
        const uint32_t c = in[pos];
        if (pos + 1 < sz && c == '\\') {
            const uint32_t nx = in[pos + 1];
            esc_in = 2;
            switch (nx) {
            case '"': ch = '"'; break;
            case '\'': ch = '\''; break;
            case '\\': ch = '\\'; break;
            case '/': ch = '/'; break;
            case 'n': ch = '\n'; break;
            case 'b': ch = '\b'; break;
            case 't': ch = '\t'; break;
            case 'f': ch = '\f'; break;
            case 'r': ch = '\r'; break;
            default: esc_in = read_escape(in, pos + 1, sz - (pos + 1), &ch) + 1;
            }
        }
        else { ch = (uint32_t) (int32_t) (int8_t) c; esc_in = 1; }
        pos += esc_in;
        // u8_wc_toutf8 (:40-64); a code point it refuses is stored as its low byte
        const uint32_t nb = ch < 0x80 ? 1 : ch < 0x800 ? 2 : ch < 0x10000 ? 3 : ch < 0x110000 ? 4 : 0;
        if (nb > sz - count_out) break;
        if (nb <= 1) {
            const uint32_t b = ch & 0xff;
            if (STOP_AT_NUL && b == 0) break;
            o.put((uint8_t) b);
            count_out += 1;
        }
        else if (nb == 2) { o.put((uint8_t) ((ch >> 6) | 0xC0)); o.put((uint8_t) ((ch & 0x3F) | 0x80)); count_out += 2; }
        else if (nb == 3) { o.put((uint8_t) ((ch >> 12) | 0xE0)); o.put((uint8_t) (((ch >> 6) & 0x3F) | 0x80)); o.put((uint8_t) ((ch & 0x3F) | 0x80)); count_out += 3; }
        else { o.put((uint8_t) ((ch >> 18) | 0xF0)); o.put((uint8_t) (((ch >> 12) & 0x3F) | 0x80)); o.put((uint8_t) (((ch >> 6) & 0x3F) | 0x80)); o.put((uint8_t) ((ch & 0x3F) | 0x80)); count_out += 4; }
    }
    return count_out;
}

}  // namespace dec
}  // namespace flbgpu
