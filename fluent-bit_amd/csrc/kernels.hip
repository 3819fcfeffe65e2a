// kernels.hip -- hand-written HIP kernels (gfx950 / MI355X) for the Fluent Bit filter hot path.
//
// Layout in HBM: a chunk is the reference wire format itself (concatenated msgpack log events,
// src/flb_log_event_encoder.c:195-217) as one contiguous byte column plus a row-offset column
// (u64 [n+1]).  One record is processed per lane; the automaton tables (byte classes, reverse
// DFA, viable-position bit sets, priority lists) are staged in LDS and stepped one input byte
// per lane.  Nothing here is a dense contraction, so there is no MFMA: the bound is HBM/LDS.
//
//   k_parser_locate / k_parser_rx / k_parser_finish (/ k_parser_generic)
//                    filter_parser pass 1 in phases: decode event + locate Key_Name, run the
//                    capture program, parse the time field and compute the output size
//                    (plugins/filter_parser/filter_parser.c:226-323, src/flb_parser_regex.c:114-227)
//   k_parser_emit    filter_parser pass 2: write the V2 record at its scanned offset
//                    (filter_parser.c:325-413, src/flb_log_event_encoder.c:195-217)
//   k_grep_match     filter_grep: rule evaluation -> keep flag / kept length
//                    (plugins/filter_grep/grep.c:167-194,250-284, src/flb_ra_key.c:374-434)
//   k_gather         copy of the kept records (flb_log_event_encoder_emit_raw_record)
//   k_scan_*         exclusive prefix sums (u32 -> u64) used for the write offsets
//   k_index_*        record boundary discovery helpers
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

namespace nc {
__device__ const uint64_t g_pow5_dev[2 * (P5_QMAX - P5_QMIN + 1)] = {
#include "pow5_table.inc"
};
}

#define DEV __device__ __forceinline__
#define LDS_AS __attribute__((address_space(3)))

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

// ------------------------------------------------------------------------------------------
// byte access
// ------------------------------------------------------------------------------------------
DEV uint32_t ld8(const uint8_t *p) { return *p; }
// unaligned 32-bit load: gfx950 global/flat loads accept any byte address, so a lane can fetch
// four consecutive bytes with one instruction wherever they start
DEV uint32_t ldu32(const uint8_t *p) {
    typedef uint32_t u32u __attribute__((aligned(1)));
    return *(const u32u *) p;
}
DEV uint32_t ldbe32(const uint8_t *p) { return __builtin_bswap32(ldu32(p)); }
DEV uint64_t ldbe64(const uint8_t *p) {
    typedef uint64_t u64u __attribute__((aligned(1)));
    return __builtin_bswap64(*(const u64u *) p);
}
DEV uint32_t ldbe16(const uint8_t *p) { return (ld8(p) << 8) | ld8(p + 1); }

// ------------------------------------------------------------------------------------------
// output sinks: CountSink sizes, ByteSink writes
// ------------------------------------------------------------------------------------------
struct CountSink {
    uint64_t n = 0;
    bool need_exact = false;          // set by the size pass when a Types float literal is a hard rounding case
    DEV void note_exact() { need_exact = true; }
    DEV void put(uint32_t) { n++; }
    DEV void copy(const uint8_t *, uint32_t len) { n += len; }
    DEV void words(const uint32_t *, uint32_t nbytes) { n += nbytes; }
    DEV void put32(uint32_t) { n += 4; }
    DEV void finish() {}
};

struct ByteSink {
    uint8_t *p;
    DEV void note_exact() {}
    DEV explicit ByteSink(uint8_t *dst) : p(dst) {}
    DEV void put(uint32_t b) { *p++ = (uint8_t) b; }
    DEV void copy(const uint8_t *src, uint32_t len) { for (uint32_t i = 0; i < len; i++) *p++ = (uint8_t) ld8(src + i); }
    DEV void words(const uint32_t *w, uint32_t nbytes) { for (uint32_t i = 0; i < nbytes; i++) *p++ = (uint8_t) (w[i >> 2] >> (8 * (i & 3))); }
    DEV void put32(uint32_t v) {
        typedef uint32_t u32u __attribute__((aligned(1)));
        *(u32u *) p = v;                  // gfx950 global stores accept any byte address
        p += 4;
    }
    DEV void finish() {}
};

// stages a record into LDS; source bytes are pulled through a one-dword cache
struct LdsSink {
    __attribute__((address_space(3))) uint8_t *p;
    const uint8_t *src_end = nullptr;   // end of the readable source buffer: a 16-byte load may run past a
                                        // field but never past this (nullptr: no over-read at all)
    __attribute__((address_space(3))) uint8_t *limit = nullptr;   // end of this record's staging region: the
                                        // neighbouring lane owns what follows, nothing may be written there
    DEV explicit LdsSink(__attribute__((address_space(3))) uint8_t *dst) : p(dst) {}
    DEV void note_exact() {}
    DEV void put(uint32_t b) { *p++ = (uint8_t) b; }
    DEV void copy(const uint8_t *src, uint32_t len) {
        // gfx950 accepts unaligned dword accesses to LDS and to global memory alike.  The tail of a
        // field is copied with ONE more 16-byte load + four dword stores: the bytes written past
        // `len` are scratch that the next put()/copy() of the same record overwrites, which
        // replaces up to six dependent dword/byte loads per field.
        typedef uint32_t u32u __attribute__((aligned(1)));
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        typedef v4 v4u __attribute__((aligned(1)));
        uint32_t i = 0;
        for (; i + 16 <= len; i += 16) {
            v4 w = *(const v4u *) (src + i);
            *(LDS_AS u32u *) (p + i) = w.x; *(LDS_AS u32u *) (p + i + 4) = w.y;
            *(LDS_AS u32u *) (p + i + 8) = w.z; *(LDS_AS u32u *) (p + i + 12) = w.w;
        }
        if (i < len) {
            if (src + i + 16 <= src_end && p + i + 16 <= limit) {
                v4 w = *(const v4u *) (src + i);
                *(LDS_AS u32u *) (p + i) = w.x; *(LDS_AS u32u *) (p + i + 4) = w.y;
                *(LDS_AS u32u *) (p + i + 8) = w.z; *(LDS_AS u32u *) (p + i + 12) = w.w;
            }
            else {
                for (; i + 4 <= len; i += 4) *(LDS_AS u32u *) (p + i) = *(const u32u *) (src + i);
                for (; i < len; i++) p[i] = (uint8_t) ld8(src + i);
            }
        }
        p += len;
    }
    // `nbytes` bytes held little-endian in dwords at a wave-uniform address (scalar loads)
    DEV void words(const uint32_t *w, uint32_t nbytes) {
        typedef uint32_t u32u __attribute__((aligned(1)));
        const uint32_t nw = (nbytes + 3) / 4;
        if (p + 4 * nw <= limit) for (uint32_t i = 0; i < nw; i++) *(LDS_AS u32u *) (p + 4 * i) = w[i];
        else for (uint32_t i = 0; i < nbytes; i++) p[i] = (uint8_t) (w[i >> 2] >> (8 * (i & 3)));
        p += nbytes;
    }
    DEV void put32(uint32_t v) {
        typedef uint32_t u32u __attribute__((aligned(1)));
        *(LDS_AS u32u *) p = v;
        p += 4;
    }
    DEV void finish() {}
};

// msgpack packers over a sink (lib/msgpack-c/cmake/pack_template.h.in smallest-encoding rules)
template <class S> DEV void pk_be(S &s, uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) s.put((uint32_t) (v >> (8 * i))); }
template <class S> DEV void pk_uint(S &s, uint64_t v) {
    if (v < 128) s.put((uint32_t) v);
    else if (v < 256) { s.put(0xcc); s.put((uint32_t) v); }
    else if (v < 65536) { s.put(0xcd); pk_be(s, v, 2); }
    else if (v < (1ull << 32)) { s.put(0xce); pk_be(s, v, 4); }
    else { s.put(0xcf); pk_be(s, v, 8); }
}
template <class S> DEV void pk_int(S &s, int64_t v) {
    if (v >= 0) { pk_uint(s, (uint64_t) v); return; }
    if (v >= -32) s.put((uint32_t) (uint8_t) v);
    else if (v >= -128) { s.put(0xd0); s.put((uint32_t) (uint8_t) v); }
    else if (v >= -32768) { s.put(0xd1); pk_be(s, (uint64_t) v, 2); }
    else if (v >= -2147483648LL) { s.put(0xd2); pk_be(s, (uint64_t) v, 4); }
    else { s.put(0xd3); pk_be(s, (uint64_t) v, 8); }
}
template <class S> DEV void pk_str_hdr(S &s, uint32_t n) {
    if (n < 32) s.put(0xa0 | n);
    else if (n < 256) { s.put(0xd9); s.put(n); }
    else if (n < 65536) { s.put(0xda); pk_be(s, n, 2); }
    else { s.put(0xdb); pk_be(s, n, 4); }
}
template <class S> DEV void pk_bin_hdr(S &s, uint32_t n) {
    if (n < 256) { s.put(0xc4); s.put(n); }
    else if (n < 65536) { s.put(0xc5); pk_be(s, n, 2); }
    else { s.put(0xc6); pk_be(s, n, 4); }
}
template <class S> DEV void pk_ext_hdr(S &s, uint32_t n, uint32_t type) {
    if (n == 1) s.put(0xd4);
    else if (n == 2) s.put(0xd5);
    else if (n == 4) s.put(0xd6);
    else if (n == 8) s.put(0xd7);
    else if (n == 16) s.put(0xd8);
    else if (n < 256) { s.put(0xc7); s.put(n); }
    else if (n < 65536) { s.put(0xc8); pk_be(s, n, 2); }
    else { s.put(0xc9); pk_be(s, n, 4); }
    s.put(type);
}
template <class S> DEV void pk_array_hdr(S &s, uint32_t n) {
    if (n < 16) s.put(0x90 | n);
    else if (n < 65536) { s.put(0xdc); pk_be(s, n, 2); }
    else { s.put(0xdd); pk_be(s, n, 4); }
}
template <class S> DEV void pk_map_hdr(S &s, uint32_t n) {
    if (n < 16) s.put(0x80 | n);
    else if (n < 65536) { s.put(0xde); pk_be(s, n, 2); }
    else { s.put(0xdf); pk_be(s, n, 4); }
}

// ------------------------------------------------------------------------------------------
// msgpack token reader
// ------------------------------------------------------------------------------------------
enum { T_NIL, T_BOOL, T_UINT, T_NINT, T_F32, T_F64, T_STR, T_BIN, T_EXT, T_ARRAY, T_MAP, T_BAD };

struct Tok {
    int type;
    uint32_t len;        // payload length (str/bin/ext) or element count (array/map)
    uint64_t u;          // integer value bits / float bits / bool / ext type
    const uint8_t *next; // first byte after the header (payload start for str/bin/ext)
};

DEV uint64_t ldu64(const uint8_t *p) {
    typedef uint64_t u64u __attribute__((aligned(1)));
    return *(const u64u *) p;
}

// reads one token header at p (p < end); returns T_BAD on truncation or the reserved byte 0xc1.
// The header (first byte + up to 4 length bytes + ext type) is fetched with ONE unaligned 8-byte
// load whenever 8 bytes remain before `end`; the 0xc0..0xdf family is decoded from two packed
// nibble tables (type, number of length bytes) instead of a 32-way switch, which keeps this
// function -- inlined at every token of every walker -- small.
// w: the 8 bytes at p, little-endian (bytes at or past `end` read as 0); p < end
DEV Tok mp_tok_w(uint64_t w, const uint8_t *p, const uint8_t *end) {
    Tok t;
    t.type = T_BAD; t.len = 0; t.u = 0; t.next = p;
    const uint32_t c = (uint32_t) (w & 0xff);
    p++;
    uint32_t need = 0;
    if (c <= 0x7f) { t.type = T_UINT; t.u = c; }
    else if (c >= 0xe0) { t.type = T_NINT; t.u = (uint64_t) (int64_t) (int8_t) c; }
    else if (c >= 0xa0 && c <= 0xbf) { t.type = T_STR; t.len = c & 31; }
    else if (c >= 0x90 && c <= 0x9f) { t.type = T_ARRAY; t.len = c & 15; }
    else if (c >= 0x80 && c <= 0x8f) { t.type = T_MAP; t.len = c & 15; }
    else {
        const uint32_t sh = 4 * (c & 15);
        const uint64_t ttab = c < 0xd0 ? 0x22225488877711b0ull : 0xaa99666888883333ull;      // type per first byte
        const uint64_t ntab = c < 0xd0 ? 0x8421844214210000ull : 0x4242421000008421ull;      // length bytes per first byte
        t.type = (int) ((ttab >> sh) & 15);
        need = (uint32_t) ((ntab >> sh) & 15);
        if (t.type == T_BAD) return t;                                  // 0xc1
        if (c >= 0xd4 && c <= 0xd8) t.len = 1u << (c - 0xd4);           // fixext 1/2/4/8/16
        if (c == 0xc3) t.u = 1;
        if ((uint64_t) (end - p) < need) { t.type = T_BAD; return t; }
        if (need) {
            uint64_t v;
            if (need == 8) v = ldbe64(p);                       // 64-bit ints / doubles (rare)
            else {
                // big-endian value of `need` bytes that follow the first byte, taken from w
                uint32_t x = (uint32_t) (w >> 8);
                v = need == 1 ? (x & 0xff) : need == 2 ? (((x & 0xff) << 8) | ((x >> 8) & 0xff)) : __builtin_bswap32(x);
            }
            p += need;
            if (t.type == T_UINT || t.type == T_F32 || t.type == T_F64) t.u = v;
            else if (t.type == T_NINT) {
                int64_t sv = need == 1 ? (int64_t) (int8_t) v : need == 2 ? (int64_t) (int16_t) v
                           : need == 4 ? (int64_t) (int32_t) v : (int64_t) v;
                // non-negative values of the signed family are POSITIVE_INTEGER
                // (lib/msgpack-c/src/unpack.c template_callback_int*)
                if (sv >= 0) t.type = T_UINT;
                t.u = (uint64_t) sv;
            }
            else t.len = (uint32_t) v;
        }
        if (t.type == T_EXT) {
            if (p >= end) { t.type = T_BAD; return t; }
            // ext type byte: byte 1 + need of the header
            t.u = (1 + need) < 8 ? (uint32_t) ((w >> (8 * (1 + need))) & 0xff) : ld8(p);
            p++;
        }
    }
    if (t.type == T_STR || t.type == T_BIN || t.type == T_EXT) {
        if ((uint64_t) (end - p) < t.len) { t.type = T_BAD; return t; }
    }
    t.next = p;
    return t;
}

DEV Tok mp_tok(const uint8_t *p, const uint8_t *end) {
    if (p >= end) {
        Tok t;
        t.type = T_BAD; t.len = 0; t.u = 0; t.next = p;
        return t;
    }
    uint64_t w;
    const uint32_t avail = (uint32_t) ((uint64_t) (end - p) < 9 ? (end - p) : 9);
    if (avail >= 8) w = ldu64(p);
    else { w = 0; for (uint32_t q = 0; q < avail; q++) w |= (uint64_t) ld8(p + q) << (8 * q); }
    return mp_tok_w(w, p, end);
}

// skips one complete object; nullptr when malformed / truncated
DEV const uint8_t *mp_skip(const uint8_t *p, const uint8_t *end) {
    uint64_t remaining = 1;
    while (remaining > 0) {
        Tok t = mp_tok(p, end);
        if (t.type == T_BAD) return nullptr;
        remaining--;
        p = t.next;
        if (t.type == T_STR || t.type == T_BIN || t.type == T_EXT) p += t.len;
        else if (t.type == T_ARRAY) remaining += t.len;
        else if (t.type == T_MAP) remaining += 2ull * t.len;
    }
    return p;
}

// end of the object whose header token `t` was read at `p`: scalars and str/bin/ext need no second
// decode, only containers are walked
DEV const uint8_t *mp_end_of(const Tok &t, const uint8_t *p, const uint8_t *end) {
    if (t.type == T_BAD) return nullptr;
    if (t.type == T_ARRAY || t.type == T_MAP) return t.len == 0 ? t.next : mp_skip(p, end);
    if (t.type == T_STR || t.type == T_BIN || t.type == T_EXT) return t.next + t.len;
    return t.next;
}

// size of the canonical re-pack of a non-container token (msgpack_pack_object, smallest encodings)
DEV uint32_t mp_canon_size_scalar(const Tok &t) {
    CountSink cs;
    switch (t.type) {
    case T_NIL: case T_BOOL: return 1;
    case T_UINT: pk_uint(cs, t.u); break;
    case T_NINT: pk_int(cs, (int64_t) t.u); break;
    case T_F32: return 5;
    case T_F64: return 9;
    case T_STR: pk_str_hdr(cs, t.len); cs.n += t.len; break;
    case T_BIN: pk_bin_hdr(cs, t.len); cs.n += t.len; break;
    case T_EXT: pk_ext_hdr(cs, t.len, (uint32_t) t.u); cs.n += t.len; break;
    default: break;
    }
    return (uint32_t) cs.n;
}

// canonical re-pack of one object (msgpack_pack_object, lib/msgpack-c/src/objectc.c:39-126)
template <class S> DEV const uint8_t *mp_canon(const uint8_t *p, const uint8_t *end, S &s) {
    uint64_t remaining = 1;
    while (remaining > 0) {
        Tok t = mp_tok(p, end);
        if (t.type == T_BAD) return nullptr;
        remaining--;
        p = t.next;
        switch (t.type) {
        case T_NIL: s.put(0xc0); break;
        case T_BOOL: s.put(t.u ? 0xc3 : 0xc2); break;
        case T_UINT: pk_uint(s, t.u); break;
        case T_NINT: pk_int(s, (int64_t) t.u); break;
        case T_F32: s.put(0xca); pk_be(s, t.u, 4); break;
        case T_F64: s.put(0xcb); pk_be(s, t.u, 8); break;
        case T_STR: pk_str_hdr(s, t.len); s.copy(p, t.len); p += t.len; break;
        case T_BIN: pk_bin_hdr(s, t.len); s.copy(p, t.len); p += t.len; break;
        case T_EXT: pk_ext_hdr(s, t.len, (uint32_t) t.u); s.copy(p, t.len); p += t.len; break;
        case T_ARRAY: pk_array_hdr(s, t.len); remaining += t.len; break;
        case T_MAP: pk_map_hdr(s, t.len); remaining += 2ull * t.len; break;
        }
    }
    return p;
}

// ------------------------------------------------------------------------------------------
// log event decode (src/flb_log_event_decoder.c:182-330)
// ------------------------------------------------------------------------------------------
struct Event {
    uint32_t flags;               // RF_*
    int64_t sec, nsec;
    const uint8_t *meta;          // nullptr => synthetic empty map (legacy format)
    const uint8_t *meta_end;
    const uint8_t *body;          // at the map header
    const uint8_t *body_end;
};

// lazy_body: the body map is NOT walked here; ev.body_end is the end of the row and the caller's
// own walk over the map (a key lookup) has to end exactly there, which is the same validation at
// no extra cost (map_find_last's `whole` flag).
DEV Event decode_event(const uint8_t *rec, const uint8_t *end, bool lazy_body = false) {
    Event ev;
    ev.flags = RF_BAD; ev.sec = 0; ev.nsec = 0; ev.meta = nullptr; ev.meta_end = nullptr; ev.body = nullptr; ev.body_end = nullptr;
    // an empty row is a record an earlier filter dropped (device chunks keep one row per input
    // record): there are no bytes, so there is nothing to decode -- invisible like a group marker
    if (rec == end) { ev.flags = RF_VALID | RF_SKIP; return ev; }
    // the encoder's own layout first: 92 92 d7 00 <sec32> <nsec32> 80 <body map>  (one compare
    // instead of five token decodes)
    if ((uint64_t) (end - rec) >= 14 && ldu32(rec) == 0x00d79292u && ld8(rec + 12) == 0x80) {
        const uint32_t s0 = ldbe32(rec + 4), ns0 = ldbe32(rec + 8);
        Tok b0 = mp_tok(rec + 13, end);
        if (b0.type != T_MAP) return ev;
        if (s0 == 0xffffffffu || s0 == 0xfffffffeu) {
            if (ns0 != 0) return ev;
            ev.sec = s0 == 0xffffffffu ? -1 : -2;
        }
        else {
            if (ns0 >= 1000000000u) return ev;
            ev.sec = s0; ev.nsec = ns0;
        }
        ev.meta = rec + 12; ev.meta_end = rec + 13;
        ev.body = rec + 13;
        ev.body_end = lazy_body ? end : mp_skip(rec + 13, end);
        if (!ev.body_end) return ev;
        ev.flags = RF_VALID;
        if (ev.sec < 0) ev.flags |= RF_SKIP;
        return ev;
    }
    Tok root = mp_tok(rec, end);
    if (root.type != T_ARRAY || root.len != 2) return ev;
    const uint8_t *p = root.next;
    Tok h = mp_tok(p, end);
    if (h.type == T_BAD) return ev;
    Tok ts;
    const uint8_t *after_header;
    if (h.type == T_ARRAY) {
        if (h.len != 2) return ev;
        ts = mp_tok(h.next, end);
        if (ts.type == T_BAD) return ev;
        const uint8_t *ts_end = ts.next + ((ts.type == T_EXT || ts.type == T_STR || ts.type == T_BIN) ? ts.len : 0);
        if (ts.type == T_ARRAY || ts.type == T_MAP) return ev;      // wrong timestamp type
        Tok m = mp_tok(ts_end, end);
        if (m.type != T_MAP) return ev;
        ev.meta = ts_end;
        ev.meta_end = mp_skip(ts_end, end);
        if (!ev.meta_end) return ev;
        after_header = ev.meta_end;
    }
    else {
        ts = h;
        if (ts.type == T_MAP) return ev;
        after_header = mp_skip(p, end);
        if (!after_header) return ev;
    }
    if (ts.type != T_UINT && ts.type != T_F64 && ts.type != T_EXT) return ev;
    Tok b = mp_tok(after_header, end);
    if (b.type != T_MAP) return ev;
    ev.body = after_header;
    ev.body_end = lazy_body ? end : mp_skip(after_header, end);
    if (!ev.body_end) return ev;
    // timestamp value (flb_log_event_decoder_decode_timestamp)
    if (ts.type == T_UINT) { ev.sec = (int64_t) ts.u; ev.nsec = 0; }
    else if (ts.type == T_F64) {
        double f = __longlong_as_double((long long) ts.u);
        ev.sec = (int64_t) f;
        ev.nsec = (int64_t) ((f - (double) ev.sec) * 1000000000);
    }
    else {
        if (ts.u != 0 || ts.len != 8) return ev;
        uint32_t s = ldbe32(ts.next), ns = ldbe32(ts.next + 4);
        if (s == 0xffffffffu || s == 0xfffffffeu) {
            if (ns != 0) return ev;
            ev.sec = s == 0xffffffffu ? -1 : -2;
            ev.nsec = 0;
        }
        else {
            if (ns >= 1000000000u) return ev;     // flb_time_is_valid_eventtime
            ev.sec = s; ev.nsec = ns;
        }
    }
    ev.flags = RF_VALID;
    if (ev.sec < 0) ev.flags |= RF_SKIP;          // group markers / invalid negative markers
    return ev;
}

// ------------------------------------------------------------------------------------------
// key lookup
// ------------------------------------------------------------------------------------------
DEV bool bytes_eq(const uint8_t *a, const char *b, uint32_t n) {
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint32_t x = ldu32(a + i);
        uint32_t y = (uint8_t) b[i] | ((uint32_t) (uint8_t) b[i + 1] << 8) | ((uint32_t) (uint8_t) b[i + 2] << 16) | ((uint32_t) (uint8_t) b[i + 3] << 24);
        if (x != y) return false;
    }
    for (; i < n; i++) if (ld8(a + i) != (uint8_t) b[i]) return false;
    return true;
}

// src/flb_ra_key.c:108-135: LAST entry whose key is a STR equal to `key`; returns the value ptr
// *whole (optional) is set when the walk decoded every key and value of the map and ended exactly
// at `end`: the map is then a well-formed object that fills [map, end)
DEV const uint8_t *map_find_last(const uint8_t *map, const uint8_t *end, const char *key, uint32_t klen, bool *whole = nullptr) {
    Tok m = mp_tok(map, end);
    if (m.type != T_MAP) return nullptr;
    const uint8_t *p = m.next, *found = nullptr;
    for (uint32_t i = 0; i < m.len; i++) {
        Tok k = mp_tok(p, end);
        const uint8_t *kend = mp_end_of(k, p, end);
        if (!kend) return nullptr;
        if (k.type == T_STR && k.len == klen && bytes_eq(k.next, key, klen)) found = kend;
        Tok v = mp_tok(kend, end);
        p = mp_end_of(v, kend, end);
        if (!p) return nullptr;
    }
    if (whole) *whole = (p == end);
    return found;
}

// src/flb_ra_key.c:151-236 + :374-434: resolves `$key['a'][1]`; returns pointer to the value
// object or nullptr.  *plain is set when the top-level value was used as is.
DEV const uint8_t *ra_resolve(const DevKey &k, const uint8_t *body, const uint8_t *end, bool *whole = nullptr) {
    const uint8_t *val = map_find_last(body, end, k.key, (uint32_t) k.key_len, whole);
    if (!val) return nullptr;
    Tok t = mp_tok(val, end);
    if ((t.type == T_MAP || t.type == T_ARRAY) && k.nsub > 0) {
        const uint8_t *cur = val;
        int matched = 0;
        for (int s = 0; s < k.nsub; s++) {
            Tok c = mp_tok(cur, end);
            if (k.sub_is_index[s]) {
                if (c.type != T_ARRAY) return nullptr;
                if ((uint32_t) k.sub_index[s] >= c.len) return nullptr;
                const uint8_t *p = c.next;
                for (int i = 0; i < k.sub_index[s]; i++) { p = mp_skip(p, end); if (!p) return nullptr; }
                cur = p;
                matched++;
                if (matched == k.nsub) break;
                continue;
            }
            if (c.type != T_MAP) break;
            const uint8_t *v = map_find_last(cur, end, k.sub_str + k.sub_off[s], (uint32_t) k.sub_len[s]);
            if (!v) continue;                      // "try next entry" (never completes the levels)
            cur = v;
            matched++;
            if (matched == k.nsub) break;
        }
        if (matched == 0 || matched != k.nsub) return nullptr;
        return cur;
    }
    return val;
}

// ------------------------------------------------------------------------------------------
// regex: match-only DFA
// ------------------------------------------------------------------------------------------
constexpr int RX_NOMATCH = 0, RX_MATCH = 1, RX_POISON = 2;

template <class T8, class T16>
DEV int dfa_match(const T8 *cls, const T16 *ddelta, const uint8_t *d_final, int ncls, int d_init,
                  const uint8_t *s, uint32_t len) {
    uint32_t st = (uint32_t) d_init;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t c = cls[ld8(s + i)];
        uint32_t n = ddelta[st * (uint32_t) ncls + c];
        if (n == 0xFFFF) return RX_MATCH;
        if (n == 0xFFFE) return RX_POISON;
        st = n;
    }
    return d_final[st] ? RX_MATCH : RX_NOMATCH;
}

// ------------------------------------------------------------------------------------------
// regex: capture program
// ------------------------------------------------------------------------------------------
DEV int utf8_seq_len_dev(const uint8_t *s, uint32_t i, uint32_t len) {
    uint32_t b0 = ld8(s + i);
    int rem = (int) (len - i - 1);
    int need;
    uint32_t lo1 = 0x80, hi1 = 0xbf;
    if (b0 >= 0xc2 && b0 <= 0xdf) need = 1;
    else if (b0 >= 0xe0 && b0 <= 0xef) { need = 2; if (b0 == 0xe0) lo1 = 0xa0; if (b0 == 0xed) hi1 = 0x9f; }
    else if (b0 >= 0xf0 && b0 <= 0xf4) { need = 3; if (b0 == 0xf0) lo1 = 0x90; if (b0 == 0xf4) hi1 = 0x8f; }
    else return 1;
    for (int k = 1; k <= need; k++) {
        if (k > rem) return need + 1;
        uint32_t b = ld8(s + i + k);
        if (k == 1) { if (b < lo1 || b > hi1) return 1; }
        else if (b < 0x80 || b > 0xbf) return 1;
    }
    return need + 1;
}

// Hot tables of one table set: the arrays touched once per input byte.  The pointers refer
// either to LDS (staged copy, address space 3 so that the compiler emits ds_read) or to global
// memory.
template <bool LDS> struct HotPtr;
template <> struct HotPtr<true> {
    typedef const LDS_AS uint8_t *p8; typedef const LDS_AS uint16_t *p16; typedef const LDS_AS uint32_t *p32;
};
template <> struct HotPtr<false> {
    typedef const uint8_t *p8; typedef const uint16_t *p16; typedef const uint32_t *p32;
};
template <bool LDS> struct HotTabs {
    typename HotPtr<LDS>::p8 cls;         // byte -> class (reverse pass)
    typename HotPtr<LDS>::p8 col;         // byte -> kind << fc_shift | class (forward pass)
    typename HotPtr<LDS>::p16 rdelta;     // rows of 1 << cls_shift entries
    typename HotPtr<LDS>::p32 ft;         // rows of 1 << wsh entries
    typename HotPtr<LDS>::p32 ft2;        // rows of 1 << fc_shift entries
    int ncls, NK, NKp, kind_edge, r_init, nX, nR, ascii_only, cls_shift, fc_shift, wsh, col_eot;
};

template <bool LDS> DEV void hot_scalars(HotTabs<LDS> &h, const DevCap &d) {
    h.ncls = d.ncls; h.NK = d.NK; h.NKp = d.NKp; h.kind_edge = d.kind_edge; h.r_init = d.r_init; h.nX = d.nX; h.nR = d.nR;
    h.ascii_only = d.ascii_only; h.cls_shift = d.cls_shift; h.fc_shift = d.fc_shift; h.wsh = d.wsh; h.col_eot = d.col_eot;
}

DEV HotTabs<false> hot_global(const DevCap &d) {
    HotTabs<false> h;
    h.cls = d.cls; h.col = d.col; h.rdelta = d.rdelta; h.ft = d.ft; h.ft2 = d.ft2;
    hot_scalars(h, d);
    return h;
}

DEV HotTabs<true> hot_lds(const DevCap &d, LDS_AS uint8_t *lds) {
    HotTabs<true> h;
    h.rdelta = (const LDS_AS uint16_t *) (lds + d.off_rdelta);
    h.ft = (const LDS_AS uint32_t *) (lds + d.off_ft);
    h.ft2 = (const LDS_AS uint32_t *) (lds + d.off_ft2);
    h.cls = lds + d.off_cls;
    h.col = lds + d.off_col;
    hot_scalars(h, d);
    return h;
}

// 16 consecutive bytes with one unaligned dwordx4 load
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
DEV v4u32 ldu128(const uint8_t *p) {
    typedef v4u32 v4u32u __attribute__((aligned(1)));
    return *(const v4u32u *) p;
}

// bytes pos..pos+3 of s (little endian in the result); bytes at or beyond len read as zero and
// are never dereferenced (no over-read past the value)
DEV uint32_t load4(const uint8_t *s, uint32_t pos, uint32_t len) {
    if (pos + 4 <= len) return ldu32(s + pos);
    uint32_t w = 0;
    for (uint32_t q = 0; q < 4; q++) if (pos + q < len) w |= ld8(s + pos + q) << (8 * q);
    return w;
}

// bytes pos..pos+15 (zero beyond len, never dereferenced)
DEV v4u32 load16(const uint8_t *s, uint32_t pos, uint32_t len) {
    if (pos + 16 <= len) return ldu128(s + pos);
    v4u32 r;
    r.x = load4(s, pos, len); r.y = load4(s, pos + 4, len); r.z = load4(s, pos + 8, len); r.w = load4(s, pos + 12, len);
    return r;
}

// Pass 1 over one value: the reverse automaton, one byte per step from the last byte to the
// first.  Bytes are fetched four at a time with one unaligned dword load issued one group ahead
// (its latency hides behind the previous group's steps); the four class lookups of a group are
// independent and issue together, only the four state transitions form a dependent chain.  Every
// CHK_STEP boundaries (counted from the END of the value, so that all lanes of a wave store in
// the same iteration) the state id is kept in chk[(t / CHK_STEP) * 64] ([checkpoint][lane]
// layout: a wave's stores coalesce).  Returns the leftmost viable start boundary, -1 for no
// match, -2 when a byte >= 0x80 poisoned the ASCII tables.
template <bool LDS>
DEV int rx_reverse(const HotTabs<LDS> &t, const uint8_t *r_info, const uint8_t *s, uint32_t len, uint16_t *chk) {
    uint32_t R = (uint32_t) t.r_init;
    int best = -1, h1 = -1, h2 = -1;
    static_assert(CHK_STEP == 16, "the 16-byte reverse trip stores one checkpoint per trip");
    const uint32_t sh = (uint32_t) t.cls_shift, poison = (uint32_t) t.nR;
    if (chk) chk[0] = (uint16_t) R;
    uint32_t tt = 0;
    if (t.ascii_only) {
        // one group = 4 bytes held in a dword; `top` is the index of the byte in its high lane
#define RX_REV_GROUP(w, top)                                                                        \
        {                                                                                           \
            uint32_t c0 = t.cls[(w) >> 24], c1 = t.cls[((w) >> 16) & 255];                          \
            uint32_t c2 = t.cls[((w) >> 8) & 255], c3 = t.cls[(w) & 255];                           \
            uint32_t e0 = t.rdelta[(R << sh) + c0];                                                 \
            uint32_t e1 = t.rdelta[((e0 & 0x7FFF) << sh) + c1];                                     \
            uint32_t e2 = t.rdelta[((e1 & 0x7FFF) << sh) + c2];                                     \
            uint32_t e3 = t.rdelta[((e2 & 0x7FFF) << sh) + c3];                                     \
            if ((e0 | e1 | e2 | e3) & 0x8000) {          /* start candidates are rare */           \
                if (e0 & 0x8000) best = (int) (top) + 1;                                            \
                if (e1 & 0x8000) best = (int) (top);                                                \
                if (e2 & 0x8000) best = (int) (top) - 1;                                            \
                if (e3 & 0x8000) best = (int) (top) - 2;                                            \
            }                                                                                       \
            R = e3 & 0x7FFF;                                                                        \
        }
        // 64 bytes per trip: four unaligned dwordx4 loads issued TOGETHER one trip ahead, so the
        // 128-byte line they share is brought into the vector L1 once and the other three loads hit
        // it before another wave evicts it (16 B loads spread over time re-fetched every line 8x)
        const uint32_t n64 = len / 64;
        v4u32 n0, n1, n2, n3;
        n0 = n1 = n2 = n3 = (v4u32) (0);
        if (n64) {
            const uint8_t *q = s + len - 64;
            n0 = ldu128(q); n1 = ldu128(q + 16); n2 = ldu128(q + 32); n3 = ldu128(q + 48);
        }
        for (uint32_t g = 0; g < n64; g++) {
            const v4u32 w0 = n0, w1 = n1, w2 = n2, w3 = n3;
            if (g + 1 < n64) {
                const uint8_t *q = s + len - 64 * (g + 2);
                n0 = ldu128(q); n1 = ldu128(q + 16); n2 = ldu128(q + 32); n3 = ldu128(q + 48);
            }
            const uint32_t top = len - 1 - 64 * g;
#define RX_REV_16(vv, tp)                                                                           \
            RX_REV_GROUP((vv).w, (tp)) RX_REV_GROUP((vv).z, (tp) - 4) RX_REV_GROUP((vv).y, (tp) - 8) \
            RX_REV_GROUP((vv).x, (tp) - 12)                                                         \
            if (chk) chk[(size_t) ((len - ((tp) - 15)) / CHK_STEP) * 64] = (uint16_t) R;
            RX_REV_16(w3, top)
            RX_REV_16(w2, top - 16)
            RX_REV_16(w1, top - 32)
            RX_REV_16(w0, top - 48)
#undef RX_REV_16
            // POISON (row nR) is absorbing and carries no start flags: one test per trip
            if (R == poison) return -2;
        }
        tt = 64 * n64;
        // remaining 16-byte groups
        while (tt + 16 <= len) {
            const v4u32 w = ldu128(s + len - tt - 16);
            const uint32_t top = len - 1 - tt;
            RX_REV_GROUP(w.w, top)
            RX_REV_GROUP(w.z, top - 4)
            RX_REV_GROUP(w.y, top - 8)
            RX_REV_GROUP(w.x, top - 12)
            if (R == poison) return -2;
            tt += 16;                                    // CHK_STEP == 16: a checkpoint every 16 bytes
            if (chk) chk[(size_t) (tt / CHK_STEP) * 64] = (uint16_t) R;
        }
        // remaining 4-byte groups
        while (tt + 4 <= len) {
            const uint32_t w = ldu32(s + len - tt - 4);
            RX_REV_GROUP(w, len - 1 - tt)
            if (R == poison) return -2;
            tt += 4;
            if (chk && (tt & (CHK_STEP - 1)) == 0) chk[(size_t) (tt / CHK_STEP) * 64] = (uint16_t) R;
        }
#undef RX_REV_GROUP
    }
    for (tt = tt + 1; tt <= len; tt++) {
        uint32_t i = len - tt;
        uint32_t b = ld8(s + i);
        uint32_t e = t.rdelta[(R << sh) + t.cls[b]];
        if ((e & 0x7FFF) == poison) return -2;
        int before = best;
        if (e & 0x8000) best = (int) i + 1;
        R = e & 0x7FFF;
        if (chk && (tt & (CHK_STEP - 1)) == 0) chk[(size_t) (tt / CHK_STEP) * 64] = (uint16_t) R;
        if (!t.ascii_only) {
            if (b >= 0xc2) {
                int L = utf8_seq_len_dev(s, i, len);
                if (L == 2) best = before;
                else if (L == 3) best = h1;
                else if (L == 4) best = h2;
            }
            h2 = h1; h1 = before;
        }
    }
    if (r_info[R] & 0x80) best = 0;
    return best;
}

// Where capture spans live while a value is being matched.  CapLds keeps them in LDS as u16
// ([slot][thread] layout, conflict-free) -- no divergent global stores in the walk; CapGlobal
// writes the u32 row in global memory directly (values of 65535 bytes or more).
struct CapLds {
    LDS_AS uint8_t *base;       // this thread's column; slot 0 is a dummy, span index ci lives in slot ci+1
    uint32_t stride_b;          // BYTES between slots (2 * threads per workgroup): address = one 24-bit mad
    DEV LDS_AS uint16_t *at(uint32_t slot) const { return (LDS_AS uint16_t *) (base + __umul24(slot, stride_b)); }
    DEV void set_raw(uint32_t slot, uint32_t j) { *at(slot) = (uint16_t) j; }
    DEV void set(uint32_t ci, uint32_t j) { *at(ci + 1) = (uint16_t) j; }
    DEV uint32_t get(uint32_t ci) const { uint32_t v = *at(ci + 1); return v == 0xFFFF ? CAP_UNSET : v; }
};
struct CapGlobal {
    uint32_t *base;             // [span][n] column block
    uint64_t n, r;
    DEV void set_raw(uint32_t slot, uint32_t j) { if (slot) base[(uint64_t) (slot - 1) * n + r] = j; }
    DEV void set(uint32_t ci, uint32_t j) { base[(uint64_t) ci * n + r] = j; }
    DEV uint32_t get(uint32_t ci) const { return base[(uint64_t) ci * n + r]; }
};

// several candidates remain for this byte (or its capture writes do not fit the packed entry):
// rebuild the reverse state of boundary j from the nearest checkpoint to its right, take the
// first viable candidate and apply its tag sequence.  Returns the target core, TG_MATCH for
// MATCH, TG_DEAD on inconsistency.
template <bool LDS, class CAP>
DEV uint32_t rx_resolve_multi(const DevCap &d, const HotTabs<LDS> &t, const uint8_t *s, uint32_t len, uint32_t j, uint32_t li,
                              const uint16_t *chk, const uint8_t *slot2cap, CAP &caps) {
    uint32_t t0 = ((len - j) / CHK_STEP) * CHK_STEP, b0 = len - t0;
    uint32_t r = chk[(size_t) (t0 / CHK_STEP) * 64];
    for (uint32_t i = b0; i > j; i--) r = t.rdelta[(r << t.cls_shift) + t.cls[ld8(s + i - 1)]] & 0x7FFF;
    for (uint32_t k = d.list_off[li]; k < d.list_off[li + 1]; k++) {
        uint32_t ent = d.list_ent[k], tg = ent & 0xFFFF;
        if (tg == 0xFFFF || ((d.vmask[r * (uint32_t) d.VW + (tg >> 5)] >> (tg & 31)) & 1)) {
            uint32_t ts = ent >> 16;
            for (uint32_t q = d.tag_off[ts]; q < d.tag_off[ts + 1]; q++) {
                uint32_t ci = slot2cap[d.tag_data[q]];
                if (ci != 0xFF) caps.set(ci, j);
            }
            return tg == 0xFFFF ? TG_MATCH : tg;
        }
    }
    return TG_DEAD;
}

// column codes of the four value bytes in w (positions pos..pos+3); positions at or beyond len
// get the end-of-text column
template <bool LDS>
DEV uint32_t pack_col(const HotTabs<LDS> &t, uint32_t w, uint32_t pos, uint32_t len) {
    uint32_t v0 = t.col[w & 255], v1 = t.col[(w >> 8) & 255], v2 = t.col[(w >> 16) & 255], v3 = t.col[w >> 24];
    if (pos + 4 > len) {
        const uint32_t eot = (uint32_t) t.col_eot;
        if (pos >= len) v0 = eot;
        if (pos + 1 >= len) v1 = eot;
        if (pos + 2 >= len) v2 = eot;
        v3 = eot;
    }
    return v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
}

// the rare entries of the forward walk: one-byte lookahead, MATCH, the multi-candidate resolution
// and dead ends.  Returns the plain entry to continue with; after a MATCH / dead end that is the
// absorbing row (no capture writes, every column maps to itself), so the caller's loop needs no
// early exit -- a lane that is done idles there until the wave's last position.
template <bool LDS, class CAP>
DEV uint32_t rx_forward_special(const DevCap &d, const HotTabs<LDS> &t, const uint8_t *s, uint32_t len, uint32_t j, uint32_t Sidx,
                                uint32_t colc, uint32_t next_code, uint32_t e, const uint16_t *chk, const uint8_t *slot2cap,
                                CAP &caps, int &endpos, bool &fail) {
    const uint32_t fsh = (uint32_t) t.fc_shift, wsh = (uint32_t) t.wsh;
    const uint32_t absorb = ((uint32_t) t.nX * (uint32_t) t.NKp) << wsh;       // plain entries carry row << wsh
    const uint32_t S = Sidx >> wsh;
    uint32_t ty = (e >> 28) & 7;
    if (ty == FT_LOOK) {
        const uint32_t cn = next_code & ((1u << fsh) - 1);           // class of the next byte / EOT
        e = t.ft2[((e & 0xFFFFFF) << fsh) + cn];
        if (!(e & FT_SPECIAL)) return e;
        ty = (e >> 28) & 7;
    }
    if (ty == FT_MATCH) {
        caps.set_raw((e >> 12) & 63, j);
        caps.set_raw((e >> 18) & 63, j);
        if (endpos < 0) endpos = (int) j;
        return absorb;
    }
    if (ty == FT_MULTI && chk) {
        const uint32_t x = S / (uint32_t) t.NKp, pkk = S % (uint32_t) t.NKp, nk = colc >> fsh;
        const uint32_t li = (x * (uint32_t) t.NK + pkk) * (uint32_t) t.NK + nk;
        const uint32_t tg = rx_resolve_multi(d, t, s, len, j, li, chk, slot2cap, caps);
        if (tg == TG_MATCH) { if (endpos < 0) endpos = (int) j; return absorb; }
        if (tg != TG_DEAD) return (tg * (uint32_t) t.NKp + nk) << wsh;  // plain entry without capture writes
    }
    // dead end, or a multi-candidate cell during the forward-first attempt (no reverse states yet)
    fail = true;
    return absorb;
}

// Pass 2: deterministic leftmost-first walk from boundary `start`.  Value bytes are fetched through
// 64-byte windows (four unaligned dwordx4 loads issued together, a window ahead of use) and turned
// into packed queues of 4 column codes one group ahead, so that a plain step is: index = row << wsh
// | column, ONE dependent LDS read, and the entry IS the next row, plus two unconditional capture
// writes (slot 0 is a dummy column).  Four steps are unrolled per trip; everything rare sits behind
// one "special" bit test per step and never leaves the loop (rx_forward_special).  The loop runs to
// the end-of-text column for every lane.  Returns the end boundary of the match (>= 0) or -1.
template <bool LDS, class CAP>
DEV int rx_forward(const DevCap &d, const HotTabs<LDS> &t, const uint8_t *s, uint32_t len, int start, const uint16_t *chk,
                   const uint8_t *slot2cap, CAP &caps) {
    uint32_t j = (uint32_t) start;
    const uint32_t fsh = (uint32_t) t.fc_shift, wsh = (uint32_t) t.wsh;
    uint32_t pk = j == 0 ? (uint32_t) t.kind_edge : (uint32_t) (t.col[ld8(s + j - 1)] >> fsh);
    uint32_t S = (((uint32_t) t.nX - 1) * (uint32_t) t.NKp + pk) << wsh;   // dword index of the current row
    uint32_t gb = j;                                            // position of the current 4-byte group
    v4u32 c0 = load16(s, gb, len), c1 = load16(s, gb + 16, len), c2 = load16(s, gb + 32, len), c3 = load16(s, gb + 48, len);
    v4u32 x0 = load16(s, gb + 64, len), x1 = load16(s, gb + 80, len), x2 = load16(s, gb + 96, len), x3 = load16(s, gb + 112, len);
    uint32_t vq = pack_col(t, c0.x, gb, len);                   // codes of positions gb..gb+3
    uint32_t vqn = pack_col(t, c0.y, gb + 4, len);
    uint32_t sub = 2;                                           // dword of the window feeding the NEXT refill
    int endpos = -1;
    bool fail = false;
    for (;;) {
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t colc = (vq >> (8 * k)) & 255;
            uint32_t e = t.ft[S + colc];
            if (e & FT_SPECIAL)
                e = rx_forward_special(d, t, s, len, j, S, colc, k < 3 ? (vq >> (8 * (k + 1))) & 255 : vqn & 255, e, chk, slot2cap, caps,
                                       endpos, fail);
            caps.set_raw((e >> 19) & 63, j);
            caps.set_raw((e >> 25) & 63, j);
            S = e & 0x7FFFF;
            j++;
        }
        if (j > len) break;                                      // the end-of-text column has been consumed
        gb += 4;
        vq = vqn;
        // codes of positions gb+4..gb+7: dword `sub` of the current window, or the first dword of the
        // next window once the current one is used up
        if (sub == 16) {
            c0 = x0; c1 = x1; c2 = x2; c3 = x3;
            const uint32_t nb = gb + 4 + 64;
            x0 = load16(s, nb, len); x1 = load16(s, nb + 16, len); x2 = load16(s, nb + 32, len); x3 = load16(s, nb + 48, len);
            sub = 0;
        }
        const uint32_t q4 = sub >> 2, q1 = sub & 3;
        const v4u32 cv = q4 == 0 ? c0 : q4 == 1 ? c1 : q4 == 2 ? c2 : c3;
        const uint32_t w = q1 == 0 ? cv.x : q1 == 1 ? cv.y : q1 == 2 ? cv.z : cv.w;
        vqn = pack_col(t, w, gb + 4, len);
        sub++;
    }
    return fail ? -1 : endpos;
}

// ------------------------------------------------------------------------------------------
// strptime (src/flb_strptime.c:253-907, C locale) + time lookup (src/flb_parser.c:1876-2065)
//
// Written as a compact table-driven interpreter: every numeric directive goes through ONE
// conv_num call site, the text through ONE windowed reader -- the straightforward "switch with a
// conv_num per case" version inlines into tens of thousands of instructions and thrashes the
// instruction cache.
// ------------------------------------------------------------------------------------------
struct Tm {
    int year, mon, mday, hour, min, sec, yday, wday;
    long gmtoff;
    int century, relyear, fields;
    int have_epoch; int64_t epoch;      // %s
};
enum { F_MON = 1, F_MDAY = 2, F_WDAY = 4, F_YDAY = 8, F_YEAR = 16 };

DEV bool d_isspace(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }
DEV bool d_isdigit(uint32_t c) { return c >= '0' && c <= '9'; }
DEV uint32_t d_lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }

// input text is [s, e); reads past e yield NUL (the reference works on a NUL-terminated copy).
// The text is read through a 16-byte register window (one unaligned dwordx4 load per 16 bytes
// instead of one divergent byte load per character).
struct TStr {
    const uint8_t *s, *e;
    const uint8_t *wb = nullptr;     // window base, nullptr = empty
    v4u32 w;
    // LDS mode: the text was copied into this lane's private LDS slot (from k_parser_rx's time
    // column); s is then only a position origin and is never dereferenced.  One ds_read_u8 per
    // character instead of a window check + select chain: strptime calls at() hundreds of times.
    LDS_AS const uint8_t *lds = nullptr;
    bool in_lds = false;             // (LDS offset 0 is a valid slot: the pointer cannot be the flag)
    DEV uint32_t at(const uint8_t *p) {
        if (p >= e) return 0;
        if (in_lds) return lds[(uint32_t) (p - s)];
        if (wb == nullptr || p < wb || p >= wb + 16) {
            wb = p;
            w = load16(p, 0, (uint32_t) (e - p));
        }
        uint32_t o = (uint32_t) (p - wb);
        uint32_t dw = o < 4 ? w.x : o < 8 ? w.y : o < 12 ? w.z : w.w;
        return (dw >> (8 * (o & 3))) & 0xff;
    }
};

// directive table, indexed by the conversion character (uniform across lanes => scalar loads)
enum { DK_BAD = 0, DK_NUM, DK_NAME, DK_AMPM, DK_TZ, DK_EPOCH, DK_WS, DK_PCT, DK_G };
enum { TF_MDAY = 0, TF_HOUR, TF_MIN, TF_SEC, TF_MON1, TF_YEAR, TF_RELYEAR, TF_CENTURY, TF_YDAY1, TF_WDAY, TF_WDAY7, TF_IGNORE,
       TF_MONNAME, TF_DAYNAME };
struct DirInfo { uint8_t kind, field, eatspace, pad; uint16_t lo, hi; };
__constant__ DirInfo c_dir[128] = {};
__constant__ int c_mon_len[2][12] = { { 31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 },
                                      { 31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 } };

DEV bool d_isleap(int y) { return (y % 4) == 0 && ((y % 100) != 0 || (y % 400) == 0); }

// days since 1970-01-01 of year/month(1..12)/day
DEV int64_t days_from_civil(int64_t y64, int m, int d) {
    // the year is a parsed %Y/%C%y (|y| < 2^20): 32-bit divisions, the 64-bit ones are emulated
    int y = (int) y64;
    y -= m <= 2;
    int era = (y >= 0 ? y : y - 399) / 400;
    int yoe = y - era * 400;
    int doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    int doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return (int64_t) era * 146097 + doe - 719468;
}

// one flb_strptime() call (initialize = 1).  fmt holds only primitive directives (the host
// expands %T %D %F %R %r %c %x %X).  Returns the new input pointer or nullptr.
template <class FP>
DEV const uint8_t *d_strptime(TStr &in, const uint8_t *bp, FP fmt, Tm &tm) {
    tm.century = 1900; tm.relyear = -1; tm.fields = 0; tm.gmtoff = 0;
    uint32_t c;
    while ((c = (uint8_t) *fmt) != 0) {
        if (d_isspace(c)) {
            while (d_isspace(in.at(bp))) bp++;
            fmt++;
            continue;
        }
        uint32_t cur = in.at(bp);
        if (cur == 0) return nullptr;
        c = (uint8_t) *fmt++;
        if (c != '%') {
            if (c != cur) return nullptr;
            bp++;
            continue;
        }
        c = (uint8_t) *fmt++;
        while (c == 'E' || c == 'O') c = (uint8_t) *fmt++;
        const DirInfo di = c_dir[c & 127];
        switch (di.kind) {
        case DK_PCT:
            if (cur != '%') return nullptr;
            bp++;
            break;
        case DK_WS:
            while (d_isspace(in.at(bp))) bp++;
            break;
        case DK_NUM: {
            if (di.eatspace && d_isspace(cur)) { bp++; cur = in.at(bp); }
            // _conv_num (src/flb_strptime.c:819-840)
            // rulim /= 10 per digit reaches 0 after as many digits as the upper limit has
            int result = 0, left = di.hi >= 1000 ? 4 : di.hi >= 100 ? 3 : di.hi >= 10 ? 2 : 1;
            if (cur < '0' || cur > '9') return nullptr;
            for (;;) {
                result = result * 10 + (int) (cur - '0');
                bp++;
                left--;
                cur = in.at(bp);
                if (!((result * 10 <= (int) di.hi) && left && cur >= '0' && cur <= '9')) break;
            }
            if (result < (int) di.lo || result > (int) di.hi) return nullptr;
            switch (di.field) {
            case TF_MDAY: tm.mday = result; tm.fields |= F_MDAY; break;
            case TF_HOUR: tm.hour = result; break;
            case TF_MIN: tm.min = result; break;
            case TF_SEC: tm.sec = result; break;
            case TF_MON1: tm.mon = result - 1; tm.fields |= F_MON; break;
            case TF_YEAR: tm.relyear = -1; tm.year = result - 1900; tm.fields |= F_YEAR; break;
            case TF_RELYEAR: tm.relyear = result; break;
            case TF_CENTURY: tm.century = result * 100; break;
            case TF_YDAY1: tm.yday = result - 1; tm.fields |= F_YDAY; break;
            case TF_WDAY: tm.wday = result; tm.fields |= F_WDAY; break;
            case TF_WDAY7: tm.wday = result % 7; tm.fields |= F_WDAY; break;
            default: break;
            }
            break;
        }
        case DK_NAME: {
            // full name first, then the 3-letter abbreviation, case-insensitively (:381-419)
            const int count = di.field == TF_MONNAME ? 12 : 7;
            int i, len = 0;
            // The three-letter abbreviation is a prefix of the full name and unique, so the index is
            // found by comparing the first three characters (lower-cased, packed) against immediates;
            // the rest of the full name decides between len = full and len = 3.
            const uint32_t key = d_lower(in.at(bp)) | (d_lower(in.at(bp + 1)) << 8) | (d_lower(in.at(bp + 2)) << 16);
            #define P3(a, b, c) ((uint32_t) (a) | ((uint32_t) (b) << 8) | ((uint32_t) (c) << 16))
            #define P8(str) (uint64_t) ((uint64_t) (uint8_t) (str)[0] | ((uint64_t) (uint8_t) (str)[1] << 8) | ((uint64_t) (uint8_t) (str)[2] << 16) | \
                             ((uint64_t) (uint8_t) (str)[3] << 24) | ((uint64_t) (uint8_t) (str)[4] << 32) | ((uint64_t) (uint8_t) (str)[5] << 40))
            uint64_t rest = 0;            // the characters of the full name after the abbreviation, NUL padded
            i = count;
            if (di.field == TF_MONNAME) {
                switch (key) {
                case P3('j', 'a', 'n'): i = 0; rest = P8("uary\0\0"); break;
                case P3('f', 'e', 'b'): i = 1; rest = P8("ruary\0"); break;
                case P3('m', 'a', 'r'): i = 2; rest = P8("ch\0\0\0\0"); break;
                case P3('a', 'p', 'r'): i = 3; rest = P8("il\0\0\0\0"); break;
                case P3('m', 'a', 'y'): i = 4; rest = 0; break;
                case P3('j', 'u', 'n'): i = 5; rest = P8("e\0\0\0\0\0"); break;
                case P3('j', 'u', 'l'): i = 6; rest = P8("y\0\0\0\0\0"); break;
                case P3('a', 'u', 'g'): i = 7; rest = P8("ust\0\0\0"); break;
                case P3('s', 'e', 'p'): i = 8; rest = P8("tember"); break;
                case P3('o', 'c', 't'): i = 9; rest = P8("ober\0\0"); break;
                case P3('n', 'o', 'v'): i = 10; rest = P8("ember\0"); break;
                case P3('d', 'e', 'c'): i = 11; rest = P8("ember\0"); break;
                default: break;
                }
            }
            else {
                switch (key) {
                case P3('s', 'u', 'n'): i = 0; rest = P8("day\0\0\0"); break;
                case P3('m', 'o', 'n'): i = 1; rest = P8("day\0\0\0"); break;
                case P3('t', 'u', 'e'): i = 2; rest = P8("sday\0\0"); break;
                case P3('w', 'e', 'd'): i = 3; rest = P8("nesday"); break;
                case P3('t', 'h', 'u'): i = 4; rest = P8("rsday\0"); break;
                case P3('f', 'r', 'i'): i = 5; rest = P8("day\0\0\0"); break;
                case P3('s', 'a', 't'): i = 6; rest = P8("urday\0"); break;
                default: break;
                }
            }
            #undef P3
            #undef P8
            if (i < count) {
                len = 3;
                int q = 0;
                while (q < 6 && ((rest >> (8 * q)) & 0xff) != 0 && d_lower(in.at(bp + 3 + q)) == (uint32_t) ((rest >> (8 * q)) & 0xff)) q++;
                if (q == 6 || ((rest >> (8 * q)) & 0xff) == 0) len = 3 + q;       // the whole name was there
            }
            if (i == count) return nullptr;
            if (di.field == TF_MONNAME) { tm.mon = i; tm.fields |= F_MON; }
            else { tm.wday = i; tm.fields |= F_WDAY; }
            bp += len;
            break;
        }
        case DK_AMPM: {
            uint32_t c1 = d_lower(cur), c2 = d_lower(in.at(bp + 1));
            if ((c1 != 'a' && c1 != 'p') || c2 != 'm') return nullptr;
            if (tm.hour > 12) return nullptr;
            if (c1 == 'a') { if (tm.hour == 12) tm.hour = 0; }
            else if (tm.hour < 12) tm.hour += 12;
            bp += 2;
            break;
        }
        case DK_EPOCH: {
            // _conv_num64 (:842-876) + gmtime_r
            int64_t result = 0, rulim = INT64_MAX;
            if (cur < '0' || cur > '9') return nullptr;
            for (;;) {
                if (result > 922337203685477580LL) return nullptr;
                result *= 10;
                if (result > 9223372036854775760LL) return nullptr;
                result += (int64_t) (cur - '0');
                bp++;
                rulim /= 10;
                if (result >= 922337203685477580LL) return nullptr;
                cur = in.at(bp);
                if (!(rulim && cur >= '0' && cur <= '9')) break;
            }
            if (result > 67767976233532799LL) return nullptr;        // gmtime_r overflows tm_year
            tm.have_epoch = 1; tm.epoch = result;
            tm.gmtoff = 0;
            tm.fields = 0xffff;
            break;
        }
        case DK_G:
            do bp++; while (d_isdigit(in.at(bp)));
            break;
        case DK_TZ: {
            while (d_isspace(in.at(bp))) bp++;
            uint32_t z = in.at(bp++);
            if (z == 'G') {
                if (in.at(bp++) != 'M') return nullptr;
                if (in.at(bp++) != 'T') return nullptr;
                tm.gmtoff = 0;
                break;
            }
            if (z == 'U') {
                if (in.at(bp++) != 'T') return nullptr;
                if (in.at(bp) == 'C') bp++;
                tm.gmtoff = 0;
                break;
            }
            if (z == 'Z') { tm.gmtoff = 0; break; }
            if (z != '+' && z != '-') {
                --bp;
                // RFC-822 North American zones: E/C/M/P + S/D + T
                uint32_t a = d_lower(in.at(bp)), b2 = d_lower(in.at(bp + 1)), c2 = d_lower(in.at(bp + 2));
                int zi = a == 'e' ? 0 : a == 'c' ? 1 : a == 'm' ? 2 : a == 'p' ? 3 : -1;
                if (zi >= 0 && c2 == 't' && (b2 == 's' || b2 == 'd')) {
                    tm.gmtoff = (b2 == 's' ? (-5 - zi) : (-4 - zi)) * 3600L;
                    bp += 3;
                    break;
                }
                return nullptr;
            }
            uint32_t d0 = in.at(bp), d1 = in.at(bp + 1);
            if (!d_isdigit(d0) || !d_isdigit(d1)) return nullptr;
            int offs = ((int) (d0 - '0') * 10 + (int) (d1 - '0')) * 3600;
            bp += 2;
            if (in.at(bp) == ':') bp++;
            d0 = in.at(bp);
            if (d_isdigit(d0)) {
                d1 = in.at(bp + 1);
                if (!d_isdigit(d1)) return nullptr;
                offs += ((int) (d0 - '0') * 10 + (int) (d1 - '0')) * 60;
                bp += 2;
            }
            tm.gmtoff = z == '-' ? -offs : offs;
            break;
        }
        default:
            return nullptr;
        }
    }
    if (tm.relyear != -1) {
        if (tm.century == 1900) tm.year = tm.relyear <= 68 ? tm.relyear + 2000 - 1900 : tm.relyear;
        else tm.year = tm.relyear + tm.century - 1900;
        tm.fields |= F_YEAR;
    }
    if ((tm.fields & F_YEAR) && !tm.have_epoch) {
        const int year = tm.year + 1900;
        const int lp = d_isleap(year) ? 1 : 0;
        if (!(tm.fields & F_YDAY) && (tm.fields & F_MON) && (tm.fields & F_MDAY)) {
            tm.yday = tm.mday - 1;
            for (int i = 0; i < tm.mon; i++) tm.yday += c_mon_len[lp][i];
            tm.fields |= F_YDAY;
        }
        if (tm.fields & F_YDAY) {
            int days = tm.yday;
            if (!(tm.fields & F_MON)) {
                tm.mon = 0;
                while (tm.mon < 12 && days >= c_mon_len[lp][tm.mon]) days -= c_mon_len[lp][tm.mon++];
            }
            if (!(tm.fields & F_MDAY)) tm.mday = days + 1;
        }
    }
    return bp;
}

// timegm(tm) - gmtoff (include/fluent-bit/flb_parser.h:80-94, use_system_timezone == FALSE)
DEV int64_t tm2time(const Tm &tm) {
    if (tm.have_epoch) return tm.epoch - tm.gmtoff;
    int64_t y = (int64_t) tm.year + 1900 + (tm.mon >= 0 ? tm.mon / 12 : -((11 - tm.mon) / 12));
    int m = tm.mon >= 0 ? tm.mon % 12 : 11 - ((11 - tm.mon) % 12);
    int64_t days = days_from_civil(y, m + 1, 1) + (tm.mday - 1);
    return days * 86400 + (int64_t) tm.hour * 3600 + (int64_t) tm.min * 60 + tm.sec - tm.gmtoff;
}

// flb_parser_time_lookup + tm2time for formats that carry the year.  Returns -1 (field is
// dropped, time stays 0), or 0 with *sec / *frac set.
// fmt1 / fmt2: the parser's two format halves, through whatever pointer type the caller staged them in
template <class FP>
DEV int time_lookup_in(const DevParser &ps, TStr &in, const uint8_t *v, uint32_t vlen, int64_t *sec, double *frac, FP fmt1, FP fmt2) {
    Tm tm;
    tm.year = tm.mon = tm.mday = tm.hour = tm.min = tm.sec = tm.yday = tm.wday = 0;
    tm.gmtoff = 0; tm.have_epoch = 0; tm.epoch = 0;
    *frac = 0;
    // the reference copies the text into a NUL-terminated buffer and works on strlen() of it:
    // an embedded NUL ends the string
    in.s = v; in.e = v + vlen;
    uint32_t n = 0;
    while (n < vlen && in.at(v + n) != 0) n++;
    in.e = v + n;
    // the two flb_strptime() calls that bracket %L share ONE call site (pass 0: fmt1, pass 1: fmt2)
    const uint8_t *p = v;
    bool ok = true;
    for (int pass = 0; pass < 2 && ok; pass++) {
        if (pass == 1) {
            if (!ps.has_frac) break;
            // parse_subseconds: strtod("0." + up to 9 chars) -- digits (+ an exponent inside the window)
            uint32_t avail = (uint32_t) (in.e - p);
            uint32_t digits = avail < 9 ? avail : 9, k = 0;
            uint64_t num = 0;
            while (k < digits && d_isdigit(in.at(p + k))) { num = num * 10 + (in.at(p + k) - '0'); k++; }
            if (k == 0) { ok = false; break; }
            // correctly rounded: num < 2^53 and 10^k <= 1e9 are exact doubles, one IEEE division
            double pw = 1.0;
            for (uint32_t q = 0; q < k; q++) pw *= 10.0;
            double f = (double) num / pw;
            uint32_t consumed = k;
            if (k < digits && (in.at(p + k) == 'e' || in.at(p + k) == 'E')) {
                uint32_t q = k + 1;
                int eneg = 0;
                if (q < digits && (in.at(p + q) == '+' || in.at(p + q) == '-')) { eneg = in.at(p + q) == '-'; q++; }
                if (q < digits && d_isdigit(in.at(p + q))) {
                    int ex = 0;
                    while (q < digits && d_isdigit(in.at(p + q))) { ex = ex * 10 + (int) (in.at(p + q) - '0'); q++; }
                    double sc = 1.0;
                    for (int z = 0; z < ex && z < 400; z++) sc *= 10.0;
                    f = eneg ? f / sc : f * sc;
                    consumed = q;
                }
            }
            *frac = f;
            p += consumed;
        }
        // (the second call re-initialises gmtoff/century/relyear/fields but keeps the tm fields)
        const uint8_t *p2 = d_strptime(in, p, pass == 0 ? fmt1 : fmt2, tm);
        if (!p2) ok = false;
        else p = p2;
    }
    if (!ok) {
        if (ps.time_strict) return -1;
        // non-strict: the reference returns 0 before applying the fixed offset and keeps
        // whatever strptime filled in so far (src/flb_parser.c:2004-2013)
        *sec = tm2time(tm);
        return 0;
    }
    if (!ps.time_with_tz) tm.gmtoff = ps.time_offset;
    *sec = tm2time(tm);
    return 0;
}
// The format's fixed-layout plan on a time text of exactly plan.len bytes in the lane's LDS slot
// (zero padded to 32).  true: *sec is what time_lookup_in would return for this text (and frac 0);
// false: let the interpreter decide.  Every byte of the text is checked by some op, so there is no
// embedded NUL; a two-digit number whose first digit alone exceeds the limit / 10 is left to the
// interpreter (it reads one digit there), and so is a month whose next character continues the
// full name.
DEV bool time_fast(const DevParser &ps, LDS_AS const uint8_t *t, int64_t *sec) {
    const TimePlan &pl = ps.plan;
    Tm tm;
    tm.year = tm.mon = tm.mday = tm.hour = tm.min = tm.sec = tm.yday = tm.wday = 0;
    tm.gmtoff = 0; tm.have_epoch = 0; tm.epoch = 0;
    for (int k = 0; k < pl.nops; k++) {
        const TimeOp op = pl.ops[k];
        const uint32_t o = op.off;
        switch (op.kind) {
        case TP_LIT:
            if (t[o] != op.a) return false;
            break;
        case TP_SPACE:
            if (!d_isspace(t[o])) return false;
            break;
        case TP_NUM2: {
            const uint32_t d0 = (uint32_t) t[o] - '0', d1 = (uint32_t) t[o + 1] - '0';
            if (d0 > 9 || d1 > 9 || d0 * 10 > op.b) return false;
            const int v = (int) (d0 * 10 + d1);
            if (v < (int) pl.lo[k] || v > (int) op.b) return false;
            if (op.a == TPF_MDAY) tm.mday = v;
            else if (op.a == TPF_HOUR) tm.hour = v;
            else if (op.a == TPF_MIN) tm.min = v;
            else if (op.a == TPF_SEC) tm.sec = v;
            else tm.mon = v - 1;
            break;
        }
        case TP_YEAR4: {
            const uint32_t d0 = (uint32_t) t[o] - '0', d1 = (uint32_t) t[o + 1] - '0', d2 = (uint32_t) t[o + 2] - '0', d3 = (uint32_t) t[o + 3] - '0';
            if (d0 > 9 || d1 > 9 || d2 > 9 || d3 > 9) return false;
            tm.year = (int) (d0 * 1000 + d1 * 100 + d2 * 10 + d3) - 1900;
            break;
        }
        case TP_MON3: {
            const uint32_t key = d_lower(t[o]) | (d_lower(t[o + 1]) << 8) | (d_lower(t[o + 2]) << 16);
            const uint32_t nxt = o + 3 < (uint32_t) pl.len ? d_lower(t[o + 3]) : 0;
            #define P3(a, b, c) ((uint32_t) (a) | ((uint32_t) (b) << 8) | ((uint32_t) (c) << 16))
            int m; uint32_t cont;                  // first character of the rest of the full name
            switch (key) {
            case P3('j', 'a', 'n'): m = 0; cont = 'u'; break;
            case P3('f', 'e', 'b'): m = 1; cont = 'r'; break;
            case P3('m', 'a', 'r'): m = 2; cont = 'c'; break;
            case P3('a', 'p', 'r'): m = 3; cont = 'i'; break;
            case P3('m', 'a', 'y'): m = 4; cont = 0; break;
            case P3('j', 'u', 'n'): m = 5; cont = 'e'; break;
            case P3('j', 'u', 'l'): m = 6; cont = 'y'; break;
            case P3('a', 'u', 'g'): m = 7; cont = 'u'; break;
            case P3('s', 'e', 'p'): m = 8; cont = 't'; break;
            case P3('o', 'c', 't'): m = 9; cont = 'o'; break;
            case P3('n', 'o', 'v'): m = 10; cont = 'e'; break;
            case P3('d', 'e', 'c'): m = 11; cont = 'e'; break;
            default: return false;
            }
            #undef P3
            if (cont && nxt == cont) return false;
            tm.mon = m;
            break;
        }
        case TP_TZ5: {
            const uint32_t sg = t[o];
            const uint32_t d0 = (uint32_t) t[o + 1] - '0', d1 = (uint32_t) t[o + 2] - '0', d2 = (uint32_t) t[o + 3] - '0', d3 = (uint32_t) t[o + 4] - '0';
            if ((sg != '+' && sg != '-') || d0 > 9 || d1 > 9 || d2 > 9 || d3 > 9) return false;
            const long offs = (long) (d0 * 10 + d1) * 3600 + (long) (d2 * 10 + d3) * 60;
            tm.gmtoff = sg == '-' ? -offs : offs;
            break;
        }
        default:
            return false;
        }
    }
    if (!ps.time_with_tz) tm.gmtoff = ps.time_offset;
    *sec = tm2time(tm);
    return true;
}

DEV int time_lookup(const DevParser &ps, const uint8_t *v, uint32_t vlen, int64_t *sec, double *frac) {
    TStr in;
    return time_lookup_in(ps, in, v, vlen, sec, frac, (const char *) ps.fmt1, (const char *) ps.fmt2);
}

// ------------------------------------------------------------------------------------------
// Types casts (src/flb_parser.c:2067-2164): atoll / strtoull(16) / bool
// ------------------------------------------------------------------------------------------
DEV int64_t d_atoll(const uint8_t *s, uint32_t n) {
    uint32_t i = 0;
    while (i < n && d_isspace(ld8(s + i))) i++;
    bool neg = false;
    if (i < n && (ld8(s + i) == '+' || ld8(s + i) == '-')) { neg = ld8(s + i) == '-'; i++; }
    uint64_t v = 0;
    bool ovf = false;
    uint64_t lim = neg ? (uint64_t) 1 << 63 : (uint64_t) INT64_MAX;
    while (i < n && d_isdigit(ld8(s + i))) {
        uint32_t d = ld8(s + i) - '0';
        if (v > (lim - d) / 10) ovf = true;
        v = v * 10 + d;
        i++;
    }
    if (ovf) return neg ? INT64_MIN : INT64_MAX;
    return neg ? (int64_t) (0 - v) : (int64_t) v;
}

DEV uint64_t d_strtoull16(const uint8_t *s, uint32_t n) {
    uint32_t i = 0;
    while (i < n && d_isspace(ld8(s + i))) i++;
    bool neg = false;
    if (i < n && (ld8(s + i) == '+' || ld8(s + i) == '-')) { neg = ld8(s + i) == '-'; i++; }
    auto hv = [](uint32_t c) -> int {
        if (c >= '0' && c <= '9') return (int) c - '0';
        if (c >= 'a' && c <= 'f') return (int) c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return (int) c - 'A' + 10;
        return -1;
    };
    if (i + 1 < n && ld8(s + i) == '0' && (ld8(s + i + 1) == 'x' || ld8(s + i + 1) == 'X') &&
        i + 2 < n && hv(ld8(s + i + 2)) >= 0) i += 2;
    uint64_t v = 0;
    bool ovf = false;
    while (i < n && hv(ld8(s + i)) >= 0) {
        if (v >> 60) ovf = true;
        v = (v << 4) | (uint64_t) hv(ld8(s + i));
        i++;
    }
    if (ovf) return UINT64_MAX;
    return neg ? (0 - v) : v;
}

// RecInfo lives in HBM as REC_NCOLS columns of n words each
DEV void rec_store(uint32_t *cols, uint64_t n, uint64_t r, const RecInfo &ri) {
    cols[0 * n + r] = ri.flags; cols[1 * n + r] = ri.val_off; cols[2 * n + r] = ri.val_len; cols[3 * n + r] = ri.key_index;
    cols[4 * n + r] = ri.ts_sec; cols[5 * n + r] = ri.ts_nsec; cols[6 * n + r] = ri.body_off; cols[7 * n + r] = ri.body_len;
    cols[8 * n + r] = ri.meta_off; cols[9 * n + r] = ri.meta_len; cols[10 * n + r] = (uint32_t) ri.parser_idx;
    cols[11 * n + r] = ri.nkept; cols[12 * n + r] = ri.drop_mask; cols[13 * n + r] = ri.meta_canon;
}
DEV RecInfo rec_load(const uint32_t *cols, uint64_t n, uint64_t r) {
    RecInfo ri;
    ri.flags = cols[0 * n + r]; ri.val_off = cols[1 * n + r]; ri.val_len = cols[2 * n + r]; ri.key_index = cols[3 * n + r];
    ri.ts_sec = cols[4 * n + r]; ri.ts_nsec = cols[5 * n + r]; ri.body_off = cols[6 * n + r]; ri.body_len = cols[7 * n + r];
    ri.meta_off = cols[8 * n + r]; ri.meta_len = cols[9 * n + r]; ri.parser_idx = (int32_t) cols[10 * n + r];
    ri.nkept = cols[11 * n + r]; ri.drop_mask = cols[12 * n + r]; ri.meta_canon = cols[13 * n + r];
    ri.pad_[0] = ri.pad_[1] = 0;
    return ri;
}
// one record's capture spans inside the [span][n] column block
struct CapsView {
    const uint32_t *base;
    uint64_t n, r;
    DEV uint32_t operator[](uint32_t i) const { return base[(uint64_t) i * n + r]; }
};

#include "json_kernels.inc"
#include "pjson_dev.inc"
#include "pkv_dev.inc"

// ------------------------------------------------------------------------------------------
// parsed-record body writer shared by the size pass and the emit pass
// ------------------------------------------------------------------------------------------
struct FieldSrc {
    const uint8_t *p;
    DEV uint32_t operator[](uint32_t i) const { return ld8(p + i); }
};

template <bool EXACT, class S>
DEV void write_field_value(S &s, int type, const uint8_t *v, uint32_t vlen) {
    switch (type) {
    case TY_FLOAT: {
        // atof(strndup(val)) (src/flb_parser.c:2111-2118): strtod, 0.0 when nothing converts
        FieldSrc src{v};
        nc::ScanResult r = nc::scan_double<EXACT>(src, vlen, nc::MODE_STRTOD);
        if (r.status == nc::NC_NEED_EXACT) s.note_exact();      // k_parser_emit_exact rewrites this record
        s.put(0xcb);
        pk_be(s, r.status == nc::NC_OK ? r.bits : 0ull, 8);
        break;
    }
    case TY_INT: pk_int(s, d_atoll(v, vlen)); break;
    case TY_HEX: pk_uint(s, d_strtoull16(v, vlen)); break;
    case TY_BOOL:
        if (vlen >= 4 && d_lower(ld8(v)) == 't' && d_lower(ld8(v + 1)) == 'r' && d_lower(ld8(v + 2)) == 'u' && d_lower(ld8(v + 3)) == 'e') s.put(0xc3);
        else if (vlen >= 5 && d_lower(ld8(v)) == 'f' && d_lower(ld8(v + 1)) == 'a' && d_lower(ld8(v + 2)) == 'l' && d_lower(ld8(v + 3)) == 's' && d_lower(ld8(v + 4)) == 'e') s.put(0xc2);
        else { pk_str_hdr(s, vlen); s.copy(v, vlen); }
        break;
    default:
        pk_str_hdr(s, vlen); s.copy(v, vlen);
    }
}


// Writes (or sizes) the complete output record for `rec`.
template <bool EXACT = false, int NW = 1, class S>
DEV void write_record(S &s, const FParserCfg &cfg, const DevParser *parsers, const uint8_t *rec, const uint8_t *rec_end,
                      const RecInfo &ri, const CapsView &caps, uint64_t null_mask) {
    // 92 92 d7 00 <sec> <nsec>   (src/flb_log_event_encoder.c:195-217)
    s.put32(0x00d79292u); s.put32(__builtin_bswap32(ri.ts_sec)); s.put32(__builtin_bswap32(ri.ts_nsec));
    if (ri.meta_len > 1) mp_canon(rec + ri.meta_off, rec + ri.meta_off + ri.meta_len, s);
    else s.put(0x80);                                        // no metadata, or the one-byte empty map
    const uint8_t *body = rec + ri.body_off, *body_end = body + ri.body_len;
    if (!(ri.flags & RF_PARSED)) {
        mp_canon(body, body_end, s);
        return;
    }
    const DevParser &ps = parsers[ri.parser_idx];
    // kvs to append after the parsed ones (filter_parser.c:343-395)
    Tok bm;
    bm.type = T_MAP; bm.len = 0; bm.u = 0; bm.next = body;
    uint32_t nappend = 0;
    bool plain_key = !cfg.key.is_ra;
    if (cfg.reserve_data || cfg.preserve_key) bm = mp_tok(body, body_end);
    if (cfg.reserve_data) {
        nappend = bm.len;
        if (plain_key && !cfg.preserve_key) {
            for (uint32_t i = 0; i < bm.len && i < 64; i++) if ((null_mask >> i) & 1) nappend--;
        }
    }
    else if (cfg.preserve_key && plain_key) nappend = 1;
    if (nappend > 0 || ps.is_json) pk_map_hdr(s, ri.nkept + nappend);   // flb_msgpack_expand_map / flb_parser_json_do repack
    else {
        // header patched in place: width of the ORIGINAL count is kept (flb_parser_regex.c:182-199)
        uint32_t n0 = (uint32_t) ps.nregs_minus1;
        if (n0 < 16) s.put(0x80 | ri.nkept);
        else if (n0 < 65536) { s.put(0xde); pk_be(s, ri.nkept, 2); }
        else { s.put(0xdf); pk_be(s, ri.nkept, 4); }
    }
    const uint8_t *val = rec + ri.val_off;
    if (ps.kv_format) pkv_emit_pairs(ps, val, ri.val_len, s);          // Format logfmt / ltsv: the kept pairs of the text
    else if (ps.is_json) {
        // the kept pairs of the JSON object (drop_mask holds the index of the time pair that goes away);
        // the first capture columns of a json parser hold the record's container counts
        JsonCounts cc;
        cc.col = const_cast<uint32_t *>(caps.base); cc.n = caps.n; cc.r = caps.r; cc.store = false;
        pjson_emit_pairs<EXACT, NW>(val, ri.val_len, ri.drop_mask, s, cc);
    }
    else
    for (int f = 0; f < ps.nfields; f++) {
        if ((ri.drop_mask >> f) & 1) continue;
        uint32_t b = caps[2 * f], e = caps[2 * f + 1];
        uint32_t vlen = (b == CAP_UNSET || e == CAP_UNSET) ? 0 : e - b;
        const uint8_t *v = (b == CAP_UNSET || e == CAP_UNSET) ? val : val + b;
        s.words(ps.keywords + ps.kw_off[f], (uint32_t) ps.kw_bytes[f]);
        write_field_value<EXACT>(s, ps.field_type[f], v, vlen);
    }
    if (nappend > 0) {
        const uint8_t *p = bm.next;
        for (uint32_t i = 0; i < bm.len; i++) {
            const uint8_t *kend = mp_skip(p, body_end);
            const uint8_t *vend = kend ? mp_skip(kend, body_end) : nullptr;
            if (!vend) return;
            bool take;
            if (cfg.reserve_data) take = !(plain_key && !cfg.preserve_key && i < 64 && ((null_mask >> i) & 1));
            else take = (i == ri.key_index);
            if (take) { mp_canon(p, kend, s); mp_canon(kend, vend, s); }
            p = vend;
        }
    }
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// one parser attempt on one value.  Returns true on success (flb_parser_do >= 0).
// `hot` are parser ps's ASCII hot tables (LDS copy for parser 0, global otherwise).
template <bool LDS, class CAP>
DEV bool try_parser(const DevParser &ps, const HotTabs<LDS> &hot, const uint8_t *val, uint32_t vlen, uint16_t *chk, uint32_t chk_len,
                    CAP &caps, int64_t *tsec, int64_t *tnsec, uint32_t *nkept, uint32_t *drop_mask, uint32_t dbg) {
    if (vlen / CHK_STEP + 2 > chk_len) return false;      // scratch too small: the host sizes it to fit
    if (dbg & 1) return false;
    bool use_utf8 = false;
    int best = rx_reverse(hot, ps.ascii.r_info, val, vlen, chk);
    HotTabs<false> hu = hot_global(ps.utf8);
    if (best == -2) {
        use_utf8 = true;
        best = rx_reverse(hu, ps.utf8.r_info, val, vlen, chk);
    }
    if (best < 0) return false;
    if (ps.nregs_minus1 <= 0) return false;               // flb_parser_regex_do: n <= 0
    for (int f = 0; f < 2 * ps.nfields; f++) caps.set((uint32_t) f, CAP_UNSET);
    if (dbg & 2) return false;
    int endb = use_utf8 ? rx_forward(ps.utf8, hu, val, vlen, best, chk, ps.slot2cap, caps)
                        : rx_forward(ps.ascii, hot, val, vlen, best, chk, ps.slot2cap, caps);
    if (endb < 0) return false;
    // named groups that did not participate stay CAP_UNSET
    bool any = false;
    uint32_t kept = 0, drop = 0;
    int64_t sec = 0; double frac = 0;
    for (int f = 0; f < ps.nfields; f++) {
        uint32_t b = caps.get(2 * f), e = caps.get(2 * f + 1);
        bool set = (b != CAP_UNSET && e != CAP_UNSET);
        if (!set) { caps.set(2 * f, CAP_UNSET); caps.set(2 * f + 1, CAP_UNSET); }
        if (set) any = true;                               // last_pos (src/flb_regex.c:52-54)
        uint32_t fl = set ? e - b : 0;
        if (fl == 0 && ps.skip_empty) { drop |= 1u << f; continue; }
        if (ps.field_is_time[f] && !(dbg & 4)) {
            int64_t s2; double f2;
            if (time_lookup(ps, set ? val + b : val, fl, &s2, &f2) == -1) { drop |= 1u << f; continue; }
            sec = s2; frac = f2;
            if (!ps.time_keep) { drop |= 1u << f; continue; }
        }
        kept++;
    }
    if (!any) return false;
    *tsec = sec;
    *tnsec = (int64_t) (frac * 1000000000);
    *nkept = kept;
    *drop_mask = drop;
    return true;
}

// ------------------------------------------------------------------------------------------
// filter_parser pass 1 is split into phase kernels so that every wave of a launch runs the same
// small piece of code (register pressure and instruction-cache footprint of one phase only):
//
//   k_parser_locate   decode the event, find the value Key_Name designates, size the record as
//                     if it stayed unparsed (records staged through LDS tiles, coalesced reads)
//   k_parser_rx       parser 0's capture program on the located value (tables + spans in LDS)
//   k_parser_finish   named fields, time lookup, size of the parsed record
//   k_parser_generic  everything the fast phases do not cover (UTF-8 input, several parsers or
//                     candidate keys, huge values): the complete per-record algorithm, run only
//                     on the records flagged RF_GENERIC (normally none)
// ------------------------------------------------------------------------------------------
constexpr int LOC_BLOCK = 256;
constexpr int LOC_TILE = 18432;             // LDS bytes per wave (64 records of 277 B + slack)

DEV void recinfo_init(RecInfo &ri) {
    ri.flags = 0; ri.val_off = 0; ri.val_len = 0; ri.key_index = 0; ri.ts_sec = 0; ri.ts_nsec = 0;
    ri.body_off = 0; ri.body_len = 0; ri.meta_off = 0; ri.meta_len = 0; ri.parser_idx = -1; ri.nkept = 0; ri.drop_mask = 0;
    ri.meta_canon = 1; ri.pad_[0] = ri.pad_[1] = 0;
}

// per-record part of k_parser_locate; rec may point into LDS (generic pointer)
DEV uint32_t locate_one(const ParserMatchArgs &a, uint64_t r, const uint8_t *rec, const uint8_t *rec_end, uint32_t &n_dec) {
    RecInfo ri;
    recinfo_init(ri);
    // the body map is validated by the candidate search below (it walks every key and value)
    Event ev = decode_event(rec, rec_end, true);
    ri.flags = ev.flags;
    if (ev.flags & RF_BAD) {
        atomicMin(a.first_bad, (unsigned long long) r);
        rec_store(a.info, a.n, r, ri);
        return 0;
    }
    if (ev.flags & RF_SKIP) {
        if (rec != rec_end && mp_skip(ev.body, rec_end) != rec_end) {        // a marker with a broken body is a decoder error too
            ri.flags = RF_BAD;
            atomicMin(a.first_bad, (unsigned long long) r);
        }
        rec_store(a.info, a.n, r, ri);
        return 0;
    }
    n_dec++;
    ri.body_off = (uint32_t) (ev.body - rec); ri.body_len = (uint32_t) (ev.body_end - ev.body);
    if (ev.meta) { ri.meta_off = (uint32_t) (ev.meta - rec); ri.meta_len = (uint32_t) (ev.meta_end - ev.meta); }
    // candidate values (plugins/filter_parser/filter_parser.c:259-323)
    uint32_t ncand = 0;
    bool whole = false, have_canon = false;
    CountSink canon;                       // canonical size of the body (the record if no parser matches)
    if (a.cfg.key.is_ra) {
        const uint8_t *v = ra_resolve(a.cfg.key, ev.body, ev.body_end, &whole);
        if (v) {
            Tok t = mp_tok(v, ev.body_end);
            if (t.type == T_STR || t.type == T_BIN) { ri.val_off = (uint32_t) (t.next - rec); ri.val_len = t.len; ncand = 1; }
        }
    }
    else {
        // the same walk validates the body, finds the candidates and sizes the canonical re-pack
        Tok bm = mp_tok(ev.body, ev.body_end);
        const uint8_t *p = bm.next;
        pk_map_hdr(canon, bm.len);
        have_canon = true;
        for (uint32_t i = 0; i < bm.len; i++) {
            Tok kt = mp_tok(p, ev.body_end);
            const uint8_t *kend;
            if (kt.type == T_ARRAY || kt.type == T_MAP) kend = mp_canon(p, ev.body_end, canon);
            else { kend = mp_end_of(kt, p, ev.body_end); canon.n += mp_canon_size_scalar(kt); }
            if (!kend) { p = nullptr; break; }
            Tok vt = mp_tok(kend, ev.body_end);
            if (vt.type == T_ARRAY || vt.type == T_MAP) p = mp_canon(kend, ev.body_end, canon);
            else { p = mp_end_of(vt, kend, ev.body_end); canon.n += mp_canon_size_scalar(vt); }
            if (!p) break;
            if ((kt.type == T_STR || kt.type == T_BIN) && kt.len == (uint32_t) a.cfg.key.key_len &&
                bytes_eq(kt.next, a.cfg.key.key, kt.len) && (vt.type == T_STR || vt.type == T_BIN)) {
                if (ncand == 0) { ri.val_off = (uint32_t) (vt.next - rec); ri.val_len = vt.len; ri.key_index = i; }
                ncand++;
            }
        }
        whole = (p == ev.body_end);
    }
    if (!whole) {
        // the body does not decode to exactly the rest of the row: decoder error, the loop stops here
        n_dec--;
        recinfo_init(ri);
        ri.flags = RF_BAD;
        atomicMin(a.first_bad, (unsigned long long) r);
        rec_store(a.info, a.n, r, ri);
        return 0;
    }
    if (ncand == 1 && a.caps_in_lds && ri.val_len < 0xFFFF && (ri.val_len / CHK_STEP + 2) <= a.chk_len) ri.flags |= RF_CAND;
    else if (ncand >= 1) { ri.flags |= RF_GENERIC; atomicAdd(&a.counts[2], 1ull); }
    // size of the record if no parser matches: 12 + canonical metadata + canonical body; an event
    // time outside the EventTime range makes the encoder reject the record
    uint32_t out_len = 0;
    if (ev.sec < 0 || (uint64_t) ev.sec > 0xffffffffull || ev.nsec < 0 || ev.nsec >= 1000000000LL) ri.flags |= RF_BADTS;
    else {
        ri.ts_sec = (uint32_t) ev.sec; ri.ts_nsec = (uint32_t) ev.nsec;
        CountSink cs;
        cs.n = 12;
        if (ev.meta) mp_canon(ev.meta, ev.meta_end, cs); else cs.n += 1;
        ri.meta_canon = (uint32_t) cs.n - 12;
        if (have_canon) cs.n += canon.n; else mp_canon(ev.body, ev.body_end, cs);
        out_len = (uint32_t) cs.n;
    }
    rec_store(a.info, a.n, r, ri);
    return out_len;
}

__global__ void __launch_bounds__(LOC_BLOCK) k_parser_locate(ParserMatchArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *tile = (LDS_AS uint8_t *) g_lds + (size_t) wave * LOC_TILE;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    const uint8_t *data_end = a.data + a.bytes;
    uint32_t n_dec = 0;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        const uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.row_off[r]; o1 = a.row_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            const uint64_t g0 = __shfl(o0, (int) lo, 64);
            const uint32_t align = (uint32_t) (g0 & 15);
            const bool fit = lane >= lo && lane < cnt && (o1 - g0 + align) <= (uint64_t) LOC_TILE;
            const uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            const bool direct = (m == 0);                 // one record larger than the tile: parse it in place
            if (direct) m = 1;
            if (!direct) {
                const uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - g0) + align;
                const uint8_t *src = a.data + (g0 - align);
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                for (uint32_t u = lane; u * 16 < total; u += 64) {
                    const uint8_t *p = src + (size_t) u * 16;
                    v4 v;
                    if (p + 16 <= data_end) v = *(const v4 *) p;
                    else {
                        uint32_t t4[4] = {0, 0, 0, 0};
                        for (int q = 0; q < 16 && p + q < data_end; q++) t4[q >> 2] |= (uint32_t) p[q] << (8 * (q & 3));
                        v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
                    }
                    *(LDS_AS v4 *) (tile + (size_t) u * 16) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            if (lane >= lo && lane < lo + m) {
                const uint8_t *rec, *rec_end;
                if (direct) { rec = a.data + o0; rec_end = a.data + o1; }
                else { rec = (const uint8_t *) (tile + align + (uint32_t) (o0 - g0)); rec_end = rec + (o1 - o0); }
                a.out_len[r] = locate_one(a, r, rec, rec_end, n_dec);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lo += m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_dec += __shfl_down(n_dec, o, 64);
    if (lane == 0 && n_dec) atomicAdd(&a.counts[0], (unsigned long long) n_dec);
}

// parser 0's capture program on the located values
template <bool LDS>
__global__ void __launch_bounds__(MATCH_BLOCK) k_parser_rx(ParserMatchArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint16_t *chk = a.chk + ((size_t) wave_slot * a.chk_len) * 64 + lane;
    const DevParser &ps = a.parsers[0];
    // stage the hot ASCII tables into LDS (one coalesced 16 B/lane copy per workgroup)
    HotTabs<LDS> hot0;
    if constexpr (LDS) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u *src = (const v4u *) ps.ascii.hot_base;
        LDS_AS v4u *dst = (LDS_AS v4u *) g_lds;
        for (uint32_t i = threadIdx.x; i < a.lds_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        hot0 = hot_lds(ps.ascii, (LDS_AS uint8_t *) g_lds);
    }
    else hot0 = hot_global(ps.ascii);
    CapLds capl;
    capl.base = (LDS_AS uint8_t *) (g_lds + a.caps_lds_off) + 2 * threadIdx.x;
    capl.stride_b = 2 * blockDim.x;
    const int ncap = 2 * ps.nfields;
    uint32_t n_gen = 0;
    for (uint64_t base = (uint64_t) wave_slot * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        if (r >= a.n) continue;
        const uint32_t flags = a.info[r];                     // column 0
        if (!(flags & RF_CAND)) continue;
        const uint32_t vlen = a.info[2 * a.n + r];
        const uint8_t *val = a.data + a.row_off[r] + a.info[1 * a.n + r];
        // phase 0: forward walk from boundary 0 with no reverse pass (start-anchored patterns: a match
        // that starts at 0 is the leftmost one, and every choice the walk makes is forced by the
        // byte / the next byte whenever a match exists); phase 1: reverse pass; phase 2: forward
        // walk from the leftmost viable start.  One call site each keeps the kernel small.
        int phase = (ps.fwd_first && ps.nregs_minus1 > 0) ? 0 : 1;
        int best = 0, endb = -1;
        for (;;) {
            if (phase == 1) {
                best = rx_reverse(hot0, ps.ascii.r_info, val, vlen, chk);
                if (best < 0 || ps.nregs_minus1 <= 0) break;
                phase = 2;
            }
            for (int c = 0; c < ncap; c++) capl.set((uint32_t) c, CAP_UNSET);
            endb = rx_forward(ps.ascii, hot0, val, vlen, phase == 0 ? 0 : best, phase == 0 ? (const uint16_t *) nullptr : chk, ps.slot2cap, capl);
            if (endb >= 0 || phase == 2) break;
            phase = 1;
        }
        if (endb >= 0) {
            // publish the spans: [span][record] columns, a wave stores 64 consecutive words
            for (int c = 0; c < ncap; c++) a.caps[(uint64_t) c * a.n + r] = capl.get((uint32_t) c);
            a.info[r] = flags | RF_RXOK;
            if (ps.time_field >= 0) {
                // the time text (cache-hot here) goes to its own coalesced column
                const uint32_t tb = capl.get((uint32_t) (2 * ps.time_field)), te = capl.get((uint32_t) (2 * ps.time_field + 1));
                if (tb != CAP_UNSET && te != CAP_UNSET && te - tb <= 4 * TBUF_WORDS) {
                    const uint32_t tl = te - tb;
                    v4u32 w0 = load16(val + tb, 0, tl), w1 = load16(val + tb + 16, 0, tl > 16 ? tl - 16 : 0);
                    a.tbuf[0 * a.n + r] = w0.x; a.tbuf[1 * a.n + r] = w0.y; a.tbuf[2 * a.n + r] = w0.z; a.tbuf[3 * a.n + r] = w0.w;
                    a.tbuf[4 * a.n + r] = w1.x; a.tbuf[5 * a.n + r] = w1.y; a.tbuf[6 * a.n + r] = w1.z; a.tbuf[7 * a.n + r] = w1.w;
                }
            }
        }
        else if (best == -2 || a.cfg.nparsers > 1) {
            // a byte >= 0x80 (UTF-8 tables) or more parsers to try: the generic kernel takes over
            a.info[r] = flags | RF_GENERIC;
            n_gen++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_gen += __shfl_down(n_gen, o, 64);
    if (lane == 0 && n_gen) atomicAdd(&a.counts[2], (unsigned long long) n_gen);
}

// named fields, time lookup and size of the records parser 0 matched
constexpr int FIN_SLOT = 4 * TBUF_WORDS + 4;      // per-lane LDS slot for the time text (+4: spreads the banks)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_parser_finish(ParserMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t tls_mem[256 * FIN_SLOT];
    __shared__ char fmt_mem[2 * MAX_TIMEFMT];
    LDS_AS uint8_t *tls = (LDS_AS uint8_t *) tls_mem;
    const DevParser &ps = a.parsers[0];
    // the format strings are walked character by character for every record: from LDS, not through
    // a chain of dependent global loads
    for (uint32_t i = threadIdx.x; i < 2 * MAX_TIMEFMT; i += blockDim.x) fmt_mem[i] = i < MAX_TIMEFMT ? ps.fmt1[i] : ps.fmt2[i - MAX_TIMEFMT];
    __syncthreads();
    LDS_AS const char *lfmt1 = (LDS_AS const char *) fmt_mem, *lfmt2 = lfmt1 + MAX_TIMEFMT;
    uint32_t n_gen = 0;
    // When the record's size does not depend on bytes of the chunk (no reserved / preserved kvs, no
    // Types cast) and the time text sits in the tbuf column, this kernel reads and writes nothing
    // but coalesced columns.
    const bool size_from_columns = !a.cfg.reserve_data && !(a.cfg.preserve_key && !a.cfg.key.is_ra) && ps.plain_types;
    const uint64_t n = a.n;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t fl0 = a.info[r];
        if (!(fl0 & RF_RXOK) || (fl0 & RF_GENERIC)) { if (!(fl0 & RF_GENERIC)) a.null_mask[r] = 0; continue; }
        CapsView caps;
        caps.base = a.caps; caps.n = n; caps.r = r;
        bool any = false;
        uint32_t kept = 0, drop = 0, body_bytes = 0;
        int64_t sec = 0; double frac = 0;
        for (int f = 0; f < ps.nfields; f++) {
            uint32_t b = caps[2 * f], e = caps[2 * f + 1];
            bool set = (b != CAP_UNSET && e != CAP_UNSET);
            if (set) any = true;                               // last_pos (src/flb_regex.c:52-54)
            uint32_t fl = set ? e - b : 0;
            if (fl == 0 && ps.skip_empty) { drop |= 1u << f; continue; }
            if (ps.field_is_time[f]) {
                int64_t s2; double f2;
                TStr in;
                const uint8_t *tv;
                if (f == ps.time_field && fl <= 4 * TBUF_WORDS) {
                    LDS_AS uint32_t *slot = (LDS_AS uint32_t *) (tls + threadIdx.x * FIN_SLOT);
                    #pragma unroll
                    for (int k = 0; k < TBUF_WORDS; k++) slot[k] = fl > (uint32_t) (4 * k) ? a.tbuf[(uint64_t) k * n + r] : 0;
                    in.lds = (LDS_AS const uint8_t *) slot; in.in_lds = true;
                    tv = (const uint8_t *) 4096;               // position origin only, never dereferenced
                }
                else {
                    const uint8_t *val = a.data + a.row_off[r] + a.info[1 * n + r];
                    tv = set ? val + b : val;
                }
                if (in.in_lds && ps.plan.ok && fl == (uint32_t) ps.plan.len && time_fast(ps, in.lds, &s2)) f2 = 0;
                else if (time_lookup_in(ps, in, tv, fl, &s2, &f2, lfmt1, lfmt2) == -1) { drop |= 1u << f; continue; }
                sec = s2; frac = f2;
                if (!ps.time_keep) { drop |= 1u << f; continue; }
            }
            kept++;
            const uint32_t nl = (uint32_t) ps.field_name_len[f];
            body_bytes += (nl < 32 ? 1 : nl < 256 ? 2 : nl < 65536 ? 3 : 5) + nl + (fl < 32 ? 1 : fl < 256 ? 2 : fl < 65536 ? 3 : 5) + fl;
        }
        if (!any) {
            // flb_regex_parse found no participating named group: the parser fails
            if (a.cfg.nparsers > 1) { a.info[r] = fl0 | RF_GENERIC; n_gen++; }
            else a.null_mask[r] = 0;
            continue;
        }
        uint32_t flags = (fl0 | RF_PARSED) & ~(uint32_t) RF_BADTS;
        const uint32_t key_index = a.info[3 * n + r];
        uint64_t null_mask = (!a.cfg.key.is_ra && key_index < 64) ? 1ull << key_index : 0;
        int64_t tsec = a.info[4 * n + r], tnsec = a.info[5 * n + r];
        if (fl0 & RF_BADTS) { tsec = -1; tnsec = 0; }                        // the event time was out of range
        int64_t psec = sec, pnsec = (int64_t) (frac * 1000000000);
        bool have_parsed_time = ((uint64_t) psec * 1000000000ull + (uint64_t) pnsec) != 0;
        if (have_parsed_time) { tsec = psec; tnsec = pnsec; }
        a.null_mask[r] = null_mask;
        a.info[10 * n + r] = 0;                                // parser_idx
        a.info[11 * n + r] = kept;
        a.info[12 * n + r] = drop;
        // encoder timestamp check (src/flb_log_event_encoder.c:345-363)
        if (tsec < 0 || (uint64_t) tsec > 0xffffffffull || tnsec < 0 || tnsec >= 1000000000LL) {
            a.info[r] = flags | RF_BADTS;
            a.out_len[r] = 0;
            continue;
        }
        a.info[r] = flags;
        a.info[4 * n + r] = (uint32_t) tsec; a.info[5 * n + r] = (uint32_t) tnsec;
        if (size_from_columns) {
            // 92 92 d7 00 ts(8) + metadata + map header (width of the ORIGINAL count) + fields
            const uint32_t n0 = (uint32_t) ps.nregs_minus1;
            a.out_len[r] = 12 + a.info[13 * n + r] + (n0 < 16 ? 1 : n0 < 65536 ? 3 : 5) + body_bytes;
        }
        else {
            RecInfo ri = rec_load(a.info, n, r);
            const uint8_t *rec = a.data + a.row_off[r];
            CountSink cs;
            write_record(cs, a.cfg, a.parsers, rec, a.data + a.row_off[r + 1], ri, caps, null_mask);
            a.out_len[r] = (uint32_t) cs.n;
            if (cs.need_exact) { a.info[r] = flags | RF_EXACT; atomicAdd(&a.counts[3], 1ull); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_gen += __shfl_down(n_gen, o, 64);
    if ((threadIdx.x & 63) == 0 && n_gen) atomicAdd(&a.counts[2], (unsigned long long) n_gen);
}

// The complete per-record algorithm (every candidate key, every parser, UTF-8 tables, values of
// any length), for the records the fast phases flagged RF_GENERIC.
// The wide json walker (4096-level stack, exact decimals) is big: one out-of-line copy serves the
// slow-path kernels instead of one inlined copy per call site.
__device__ __noinline__ void pjson_try_wide(const DevParser *ps, const uint8_t *v, uint32_t vlen, uint32_t *caps_base, uint64_t n, uint64_t r,
                                            LDS_AS uint8_t *slot, PjsonTry *out) {
    JsonCounts cc;
    cc.col = caps_base; cc.n = n; cc.r = r; cc.store = true; cc.next = 0;
    *out = pjson_try<true, JSON_GENERIC_WORDS>(*ps, v, vlen, cc, slot);
}
__device__ __noinline__ uint32_t size_record_wide(const FParserCfg *cfg, const DevParser *parsers, const uint8_t *rec, const uint8_t *rec_end,
                                                  const RecInfo *ri, const uint32_t *caps_base, uint64_t n, uint64_t r, uint64_t null_mask) {
    CapsView caps;
    caps.base = caps_base; caps.n = n; caps.r = r;
    CountSink cs;
    write_record<true, JSON_GENERIC_WORDS>(cs, *cfg, parsers, rec, rec_end, *ri, caps, null_mask);
    return (uint32_t) cs.n;
}

__global__ void __launch_bounds__(256) k_parser_generic(ParserMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t gen_slots[256 * 64];       // per-lane scratch of a json parser's time text
    LDS_AS uint8_t *slot = (LDS_AS uint8_t *) gen_slots + threadIdx.x * 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint16_t *chk = a.chk + ((size_t) wave_slot * a.chk_len) * 64 + lane;
    for (uint64_t base = (uint64_t) wave_slot * 64; base < a.n; base += nwaves * 64) {
        uint64_t r = base + lane;
        if (r >= a.n) continue;
        if (!(a.info[r] & RF_GENERIC)) continue;
        const uint8_t *rec = a.data + a.row_off[r];
        const uint8_t *rec_end = a.data + a.row_off[r + 1];
        RecInfo ri;
        recinfo_init(ri);
        uint64_t null_mask = 0;
        CapsView caps;
        caps.base = a.caps; caps.n = a.n; caps.r = r;
        Event ev = decode_event(rec, rec_end);
        ri.flags = ev.flags;                                  // valid, not skipped (checked by locate)
        ri.body_off = (uint32_t) (ev.body - rec); ri.body_len = (uint32_t) (ev.body_end - ev.body);
        if (ev.meta) { ri.meta_off = (uint32_t) (ev.meta - rec); ri.meta_len = (uint32_t) (ev.meta_end - ev.meta); }
        int64_t tsec = ev.sec, tnsec = ev.nsec;
        bool have_out = false, last_ok = false;
        Tok bm = mp_tok(ev.body, ev.body_end);
        const uint8_t *p = bm.next;
        uint32_t nkv = a.cfg.key.is_ra ? 1 : bm.len;
        for (uint32_t i = 0; i < nkv; i++) {
            const uint8_t *vptr = nullptr;
            uint32_t vlen = 0;
            if (a.cfg.key.is_ra) {
                const uint8_t *v = ra_resolve(a.cfg.key, ev.body, ev.body_end);
                if (v) {
                    Tok t = mp_tok(v, ev.body_end);
                    if (t.type == T_STR || t.type == T_BIN) { vptr = t.next; vlen = t.len; }
                }
            }
            else {
                Tok kt = mp_tok(p, ev.body_end);
                const uint8_t *kend = mp_skip(p, ev.body_end);
                Tok vt = mp_tok(kend, ev.body_end);
                p = mp_skip(kend, ev.body_end);
                if ((kt.type == T_STR || kt.type == T_BIN) && kt.len == (uint32_t) a.cfg.key.key_len &&
                    bytes_eq(kt.next, a.cfg.key.key, kt.len) && (vt.type == T_STR || vt.type == T_BIN)) { vptr = vt.next; vlen = vt.len; }
            }
            if (!vptr) continue;
            for (int q = 0; q < a.cfg.nparsers; q++) {
                int64_t ps = 0, pn = 0; uint32_t nk = 0, dm = 0;
                CapGlobal capg;
                capg.base = a.caps; capg.n = a.n; capg.r = r;
                if (a.parsers[q].is_json) {
                    // Format json / logfmt / ltsv in a list of parsers: the walkers of pjson_dev.inc / pkv_dev.inc
                    // (json with the wide stack and the exact decimals: such a record is emitted by k_parser_emit_exact)
                    PjsonTry t;
                    if (a.parsers[q].kv_format) t = pkv_try(a.parsers[q], vptr, vlen);
                    else pjson_try_wide(&a.parsers[q], vptr, vlen, a.caps, a.n, r, slot, &t);
                    last_ok = t.ok;
                    ps = t.sec; pn = t.nsec; nk = t.npairs; dm = t.skip;
                }
                else
                last_ok = try_parser(a.parsers[q], hot_global(a.parsers[q].ascii), vptr, vlen, chk, a.chk_len, capg, &ps, &pn, &nk, &dm, 0u);
                if (last_ok) {
                    have_out = true;
                    ri.val_off = (uint32_t) (vptr - rec); ri.val_len = vlen; ri.parser_idx = q; ri.nkept = nk; ri.drop_mask = dm;
                    if (!a.cfg.key.is_ra) {
                        ri.key_index = i;
                        if (i < 64) null_mask |= 1ull << i;
                    }
                    if ((uint64_t) ps * 1000000000ull + (uint64_t) pn != 0) { tsec = ps; tnsec = pn; }
                    break;
                }
            }
        }
        if (have_out && last_ok) ri.flags |= RF_PARSED;
        // encoder timestamp check (src/flb_log_event_encoder.c:345-363)
        if (tsec < 0 || (uint64_t) tsec > 0xffffffffull || tnsec < 0 || tnsec >= 1000000000LL) {
            ri.flags |= RF_BADTS;
            rec_store(a.info, a.n, r, ri); a.out_len[r] = 0; a.null_mask[r] = null_mask;
            continue;
        }
        ri.ts_sec = (uint32_t) tsec; ri.ts_nsec = (uint32_t) tnsec;
        CountSink cs;
        const bool json_won = (ri.flags & RF_PARSED) && a.parsers[ri.parser_idx].is_json && !a.parsers[ri.parser_idx].kv_format;
        if (json_won) {
            cs.n = size_record_wide(&a.cfg, a.parsers, rec, rec_end, &ri, a.caps, a.n, r, null_mask);
            if (cs.n) cs.need_exact = true;                    // sized with the wide walker: emitted by the same one
        }
        else write_record(cs, a.cfg, a.parsers, rec, rec_end, ri, caps, null_mask);
        if (cs.need_exact) { ri.flags |= RF_EXACT; atomicAdd(&a.counts[3], 1ull); }
        rec_store(a.info, a.n, r, ri);
        a.null_mask[r] = null_mask;
        a.out_len[r] = (uint32_t) cs.n;
    }
}

// records with a non-empty output (flb_mp_count_log_records of the result)
__global__ void __launch_bounds__(256) k_count_nonzero(const uint32_t *len, uint64_t n, unsigned long long *out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) c += len[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long) c);
}

// Pass 2.  Each lane produces its record into a per-wave LDS staging area at the record's
// offset inside the wave's (contiguous) output range; the wave then flushes the staged bytes with
// 16 B per lane coalesced stores.  A record that does not fit the staging area is written directly.
constexpr int EMIT_BLOCK = 256;
constexpr int EMIT_STG = 18944;             // staging bytes per wave (64 records x 275 B + slack)

__global__ void __launch_bounds__(EMIT_BLOCK) k_parser_emit(ParserEmitArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *stg = (LDS_AS uint8_t *) g_lds + (size_t) wave * EMIT_STG;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        uint64_t r = base + lane;
        uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.out_off[r]; o1 = a.out_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            uint64_t batch_base = __shfl(o0, (int) lo, 64);
            uint32_t align = (uint32_t) (batch_base & 15);
            bool fit = lane >= lo && lane < cnt && (o1 - batch_base + align) <= (uint64_t) EMIT_STG;
            uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            if (m == 0) {
                // one record larger than the staging area: straight to global memory
                if (lane == lo && o1 > o0) {
                    ByteSink s(a.out + o0);
                    CapsView cv;
                    cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
                    write_record(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], rec_load(a.info, a.n_cols, r),
                                 cv, a.null_mask[r]);
                }
                lo += 1;
                continue;
            }
            if (lane >= lo && lane < lo + m && o1 > o0) {
                LdsSink s(stg + align + (uint32_t) (o0 - batch_base));
                s.src_end = a.data + a.bytes;
                s.limit = s.p + (uint32_t) (o1 - o0);
                CapsView cv;
                cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
                write_record(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], rec_load(a.info, a.n_cols, r),
                             cv, a.null_mask[r]);
            }
            uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - batch_base);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staged bytes visible to the wave
            // flush [align, align + total) of the staging area to out[batch_base ...]
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            uint8_t *dst = a.out + (batch_base - align);               // 16 B aligned
            uint32_t n16 = (align + total + 15) / 16;
            for (uint32_t u = lane; u < n16; u += 64) {
                uint32_t b0 = u * 16, b1 = b0 + 16;
                if (b0 >= align && b1 <= align + total) *(v4u *) (dst + b0) = *(LDS_AS v4u *) (stg + b0);
                else {
                    uint32_t s0 = b0 < align ? align : b0, s1 = b1 > align + total ? align + total : b1;
                    for (uint32_t q = s0; q < s1; q++) dst[q] = stg[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staging area reusable
            lo += m;
        }
    }
}

// records whose Types float literal is a hard rounding case (RF_EXACT): rewritten in place with the
// big-integer conversion, one record per lane straight to global memory (rare)
__global__ void __launch_bounds__(64) k_parser_emit_exact(ParserEmitArgs a) {
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += (uint64_t) gridDim.x * blockDim.x) {
        if (!(a.info[r] & RF_EXACT) || a.out_off[r + 1] == a.out_off[r]) continue;
        ByteSink s(a.out + a.out_off[r]);
        CapsView cv;
        cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
        write_record<true, JSON_GENERIC_WORDS>(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], rec_load(a.info, a.n_cols, r), cv,
                           a.null_mask[r]);
    }
}

// ------------------------------------------------------------------------------------------
// filter_grep
// ------------------------------------------------------------------------------------------

// flb_ra_regex_match (src/flb_record_accessor.c:753-765): > 0 match, <= 0 no match
DEV int rule_match(const GrepRule &ru, const uint8_t *body, const uint8_t *body_end, bool *whole = nullptr) {
    const uint8_t *v = ra_resolve(ru.key, body, body_end, whole);
    if (!v) return -1;
    Tok t = mp_tok(v, body_end);
    if (t.type != T_STR) return -1;
    int m = dfa_match(ru.dfa.cls, ru.dfa.ddelta, ru.dfa.d_final, ru.dfa.ncls, ru.dfa.d_init, t.next, t.len);
    if (m == RX_POISON) {
        HotTabs<false> hu = hot_global(ru.utf8);
        int best = rx_reverse(hu, ru.utf8.r_info, t.next, t.len, nullptr);
        m = best >= 0 ? RX_MATCH : RX_NOMATCH;
    }
    return m == RX_MATCH ? 1 : 0;
}

// rule evaluation for one decoded event
// *valid: the first rule's key lookup walks the whole body map -- that walk doubles as the
// decoder's validation of the body (decode_event(lazy_body))
DEV bool grep_decide(const GrepArgs &a, const Event &ev, bool *valid) {
    bool keep = true;
    if (a.logical_op == OP_LEGACY) {
        // plugins/filter_grep/grep.c:167-194
        for (int i = 0; i < a.nrules; i++) {
            const GrepRule &ru = a.rules[i];
            int ret = rule_match(ru, ev.body, ev.body_end, i == 0 ? valid : nullptr);
            if (ret <= 0) { if (ru.type == GREP_REGEX) { keep = false; break; } }
            else { keep = (ru.type != GREP_EXCLUDE); break; }
        }
    }
    else {
        // plugins/filter_grep/grep.c:250-284
        bool found = false;
        int last = 0;
        for (int i = 0; i < a.nrules; i++) {
            last = i;
            found = rule_match(a.rules[i], ev.body, ev.body_end, i == 0 ? valid : nullptr) > 0;
            if (a.logical_op == OP_OR && found) break;
            if (a.logical_op == OP_AND && !found) break;
        }
        if (a.nrules > 0) keep = (a.rules[last].type == GREP_REGEX) ? found : !found;
    }
    return keep;
}

// filter_grep.  One record per lane lets every lane touch its own cache lines ~20 times while it
// walks the msgpack tokens; with hundreds of KB of such lines in flight per CU the 32 KB vector L1
// thrashes and every touch goes back to L2/HBM (measured: 10x the chunk bytes fetched).  So the
// wave first copies its 64 records -- one contiguous byte range of the chunk -- into LDS with
// coalesced 16 B/lane loads (each chunk byte is read from HBM exactly once) and the lanes parse
// their records out of LDS.
constexpr int GREP_BLOCK = 256;
constexpr int GREP_TILE = 18432;            // LDS bytes per wave (64 records of 277 B + slack)

__global__ void __launch_bounds__(GREP_BLOCK) k_grep_match(GrepArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *tile = (LDS_AS uint8_t *) g_lds + (size_t) wave * GREP_TILE;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    const uint8_t *data_end = a.data + a.bytes;
    uint32_t n_dec = 0, n_keep = 0;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        const uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.row_off[r]; o1 = a.row_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            const uint64_t g0 = __shfl(o0, (int) lo, 64);
            const uint32_t align = (uint32_t) (g0 & 15);
            const bool fit = lane >= lo && lane < cnt && (o1 - g0 + align) <= (uint64_t) GREP_TILE;
            const uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            const bool direct = (m == 0);                 // one record larger than the tile: parse it in place
            if (direct) m = 1;
            if (!direct) {
                const uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - g0) + align;
                const uint8_t *src = a.data + (g0 - align);
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                for (uint32_t u = lane; u * 16 < total; u += 64) {
                    const uint8_t *p = src + (size_t) u * 16;
                    v4 v;
                    if (p + 16 <= data_end) v = *(const v4 *) p;
                    else {
                        uint32_t t4[4] = {0, 0, 0, 0};
                        for (int q = 0; q < 16 && p + q < data_end; q++) t4[q >> 2] |= (uint32_t) p[q] << (8 * (q & 3));
                        v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
                    }
                    *(LDS_AS v4 *) (tile + (size_t) u * 16) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            if (lane >= lo && lane < lo + m) {
                const uint8_t *rec, *rec_end;
                if (direct) { rec = a.data + o0; rec_end = a.data + o1; }
                else {
                    // generic pointer into the LDS aperture: the msgpack walkers are shared with
                    // the kernels that read from global memory
                    rec = (const uint8_t *) (tile + align + (uint32_t) (o0 - g0));
                    rec_end = rec + (o1 - o0);
                }
                Event ev = decode_event(rec, rec_end, true);
                bool keep = false;
                if (!(ev.flags & (RF_BAD | RF_SKIP))) {
                    bool valid = false;
                    keep = grep_decide(a, ev, &valid);
                    if (!valid) valid = mp_skip(ev.body, rec_end) == rec_end;     // no rule walked the map
                    if (!valid) ev.flags = RF_BAD;
                }
                else if ((ev.flags & RF_SKIP) && rec != rec_end && mp_skip(ev.body, rec_end) != rec_end) ev.flags = RF_BAD;
                a.status[r] = ev.flags;
                if (ev.flags & RF_BAD) { atomicMin(a.first_bad, (unsigned long long) r); a.keep_len[r] = 0; }
                else if (ev.flags & RF_SKIP) a.keep_len[r] = 0;
                else {
                    a.keep_len[r] = keep ? (uint32_t) (o1 - o0) : 0;
                    n_dec++;
                    if (keep) n_keep++;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lo += m;
        }
    }
    // one pair of atomics per wave (same-address atomics serialise at ~12 ns each)
    for (int o = 32; o > 0; o >>= 1) { n_dec += __shfl_down(n_dec, o, 64); n_keep += __shfl_down(n_keep, o, 64); }
    if (lane == 0) {
        if (n_dec) atomicAdd(&a.counts[0], (unsigned long long) n_dec);
        if (n_keep) atomicAdd(&a.counts[1], (unsigned long long) n_keep);
    }
}

// copy of the kept records: one wave per record, byte granular

// One wave per 64 rows: the kept rows of the tile are copied one after the other, each by the
// whole wave with 16 B per lane (unaligned vector loads/stores), so a 275 B record is one load
// and one store instruction.
__global__ void __launch_bounds__(256) k_gather(GatherArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef v4 v4un __attribute__((aligned(1)));
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        uint32_t len = 0;
        uint64_t so = 0, dof = 0;
        if (r < a.n) { len = a.keep_len[r]; if (len) { so = a.row_off[r]; dof = a.out_off[r]; } }
        uint64_t mask = __ballot(len != 0);
        while (mask) {
            const int src_lane = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t l = __shfl(len, src_lane, 64);
            const uint8_t *src = a.data + __shfl(so, src_lane, 64);
            uint8_t *dst = a.out + __shfl(dof, src_lane, 64);
            for (uint32_t o = lane * 16; o < l; o += 64 * 16) {
                if (o + 16 <= l) *(v4un *) (dst + o) = *(const v4un *) (src + o);
                else for (uint32_t q = o; q < l; q++) dst[q] = src[q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64 (three small kernels; n up to 2^32)
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;            // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// tile_sums[b] = sum of tile b; tile_cnt (optional) [b] = its number of non-zero inputs
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile_sums(const uint32_t *in, uint64_t n, uint64_t *tile_sums, uint32_t *tile_cnt) {
    __shared__ uint64_t sh[SCAN_BLOCK / 64];
    uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
    uint32_t nz = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + (uint64_t) k * SCAN_BLOCK + threadIdx.x;
        if (i < n) { const uint32_t v = in[i]; s += v; nz += v != 0; }
    }
    if (tile_cnt) {
        uint32_t t = nz;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&tile_cnt[blockIdx.x], t);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) t += sh[w];
        tile_sums[blockIdx.x] = t;
    }
}

// single block: exclusive scan of the tile sums in place; total written to tile_sums[ntiles]
__global__ void __launch_bounds__(1024) k_scan_spine(uint64_t *tile_sums, uint64_t ntiles, const uint32_t *tile_cnt, unsigned long long *nonzero) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    if (tile_cnt) {
        unsigned long long c = 0;
        for (uint64_t i = threadIdx.x; i < ntiles; i += 1024) c += tile_cnt[i];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(nonzero, c);
    }
    for (uint64_t base = 0; base < ntiles; base += 1024) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < ntiles ? tile_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint64_t t = threadIdx.x >= (unsigned) o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        if (i < ntiles) tile_sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t *in, uint64_t n, const uint64_t *tile_sums, uint64_t *out) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; v[k] = i < n ? in[i] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        uint64_t t = threadIdx.x >= (unsigned) o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = tile_sums[blockIdx.x] + sh[threadIdx.x] - s;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = tile_sums[gridDim.x];
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__global__ void k_max_row_len(const uint64_t *row_off, uint64_t n, unsigned long long *out) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        unsigned long long l = row_off[i + 1] - row_off[i];
        if (l > m) m = l;
    }
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(m, o, 64); if (t > m) m = t; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------
// strptime directive table (src/flb_strptime.c:357-560: ranges of the numeric conversions)
bool upload_time_tables() {
    DirInfo h[128];
    memset(h, 0, sizeof(h));
    auto set = [&](char c, int kind, int field, int eat, int lo, int hi) {
        h[(int) c].kind = (uint8_t) kind; h[(int) c].field = (uint8_t) field; h[(int) c].eatspace = (uint8_t) eat;
        h[(int) c].lo = (uint16_t) lo; h[(int) c].hi = (uint16_t) hi;
    };
    set('d', DK_NUM, TF_MDAY, 0, 1, 31);   set('e', DK_NUM, TF_MDAY, 1, 1, 31);
    set('H', DK_NUM, TF_HOUR, 0, 0, 23);   set('k', DK_NUM, TF_HOUR, 0, 0, 23);
    set('I', DK_NUM, TF_HOUR, 0, 1, 12);   set('l', DK_NUM, TF_HOUR, 0, 1, 12);
    set('j', DK_NUM, TF_YDAY1, 0, 1, 366); set('M', DK_NUM, TF_MIN, 0, 0, 59);
    set('m', DK_NUM, TF_MON1, 0, 1, 12);   set('S', DK_NUM, TF_SEC, 0, 0, 60);
    set('U', DK_NUM, TF_IGNORE, 0, 0, 53); set('W', DK_NUM, TF_IGNORE, 0, 0, 53); set('V', DK_NUM, TF_IGNORE, 0, 0, 53);
    set('w', DK_NUM, TF_WDAY, 0, 0, 6);    set('u', DK_NUM, TF_WDAY7, 0, 1, 7);   set('g', DK_NUM, TF_IGNORE, 0, 0, 99);
    set('Y', DK_NUM, TF_YEAR, 0, 0, 9999); set('y', DK_NUM, TF_RELYEAR, 0, 0, 99); set('C', DK_NUM, TF_CENTURY, 0, 0, 99);
    set('A', DK_NAME, TF_DAYNAME, 0, 0, 0); set('a', DK_NAME, TF_DAYNAME, 0, 0, 0);
    set('B', DK_NAME, TF_MONNAME, 0, 0, 0); set('b', DK_NAME, TF_MONNAME, 0, 0, 0); set('h', DK_NAME, TF_MONNAME, 0, 0, 0);
    set('p', DK_AMPM, 0, 0, 0, 0); set('s', DK_EPOCH, 0, 0, 0, 0); set('z', DK_TZ, 0, 0, 0, 0);
    set('n', DK_WS, 0, 0, 0, 0); set('t', DK_WS, 0, 0, 0, 0); set('%', DK_PCT, 0, 0, 0, 0); set('G', DK_G, 0, 0, 0, 0);
    return hipMemcpyToSymbol(HIP_SYMBOL(c_dir), h, sizeof(h)) == hipSuccess;
}

void launch_parser_locate(const ParserMatchArgs &a, int cus, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = (size_t) (LOC_BLOCK / 64) * LOC_TILE;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_locate, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + LOC_BLOCK / 64 - 1) / (LOC_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_locate, dim3((unsigned) blocks), dim3(LOC_BLOCK), lds, st, a);
}
void launch_parser_rx(const ParserMatchArgs &a, int grid, int threads, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_rx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) k_parser_rx<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (a.lds_bytes) hipLaunchKernelGGL(k_parser_rx<true>, dim3(grid), dim3(threads), a.lds_total, st, a);
    else hipLaunchKernelGGL(k_parser_rx<false>, dim3(grid), dim3(threads), a.lds_total, st, a);
}
void launch_parser_finish(const ParserMatchArgs &a, int cus, hipStream_t st) {
    uint64_t blocks = (a.n + 255) / 256, cap = (uint64_t) cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_finish, dim3((unsigned) blocks), dim3(256), 0, st, a);
}
void launch_parser_generic(const ParserMatchArgs &a, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_parser_generic, dim3(grid), dim3(256), 0, st, a);
}
void launch_count_nonzero(const uint32_t *len, uint64_t n, unsigned long long *out, hipStream_t st) {
    if (n == 0) return;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_count_nonzero, dim3((unsigned) blocks), dim3(256), 0, st, len, n, out);
}
void launch_parser_emit(const ParserEmitArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    static bool attr_set = false;
    const size_t lds = (size_t) (EMIT_BLOCK / 64) * EMIT_STG;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_emit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + EMIT_BLOCK / 64 - 1) / (EMIT_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;                 // a few waves of tiles per CU, grid-stride over the rest
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_emit, dim3((unsigned) blocks), dim3(EMIT_BLOCK), lds, st, a);
}
void launch_grep_match(const GrepArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    static bool attr_set = false;
    const size_t lds = (size_t) (GREP_BLOCK / 64) * GREP_TILE;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_grep_match, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + GREP_BLOCK / 64 - 1) / (GREP_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;                 // 2 resident workgroups per CU, a few rounds each
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_grep_match, dim3((unsigned) blocks), dim3(GREP_BLOCK), lds, st, a);
}
void launch_parser_emit_exact(const ParserEmitArgs &a, hipStream_t st) {
    uint64_t grid = (a.n + 63) / 64;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_parser_emit_exact, dim3((unsigned) grid), dim3(64), 0, st, a);
}
void launch_gather(const GatherArgs &a, hipStream_t st) {
    if (a.n == 0) return;
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_gather, dim3((unsigned) blocks), dim3(256), 0, st, a);
}
// exclusive scan: out[0..n] (n+1 entries); tmp must hold ntiles+1 u64
// tmp: ntiles + 1 tile sums, then (u32) ntiles tile counts
size_t scan_tmp_elems(uint64_t n) { const size_t t = (size_t) ((n + SCAN_TILE - 1) / SCAN_TILE); return t + 2 + (t + 1) / 2 + 1; }
void launch_scan(const uint32_t *in, uint64_t n, uint64_t *tmp, uint64_t *out, hipStream_t st, unsigned long long *nonzero) {
    uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles == 0) { (void) hipMemsetAsync(out, 0, sizeof(uint64_t), st); return; }
    uint32_t *tile_cnt = nullptr;
    if (nonzero) {
        tile_cnt = (uint32_t *) (tmp + ntiles + 2);
        (void) hipMemsetAsync(tile_cnt, 0, ntiles * sizeof(uint32_t), st);
    }
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned) ntiles), dim3(SCAN_BLOCK), 0, st, in, n, tmp, tile_cnt);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, tmp, ntiles, (const uint32_t *) tile_cnt, nonzero);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned) ntiles), dim3(SCAN_BLOCK), 0, st, in, n, tmp, out);
}
void launch_max_row_len(const uint64_t *row_off, uint64_t n, unsigned long long *out, hipStream_t st) {
    (void) hipMemsetAsync(out, 0, sizeof(unsigned long long), st);
    if (n == 0) return;
    unsigned grid = (unsigned) ((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_max_row_len, dim3(grid), dim3(256), 0, st, row_off, n, out);
}

#include "l2m_kernels.inc"
#include "pjson_kernels.inc"
#include "index_kernels.inc"

}  // namespace flbgpu
