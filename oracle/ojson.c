/*
 * oracle/ojson.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (oracle) of flb_pack_json(): JSON text -> msgpack, default backend
 * (src/flb_pack_json.c:41-46 -> src/flb_pack.c:389-508 pack_json_to_msgpack_yyjson), i.e. the
 * reader of lib/yyjson-0.12.0 with the flags STOP_WHEN_DONE | INSITU | ALLOW_INVALID_UNICODE |
 * REPLACE_INVALID_UNICODE followed by yyjson_val_to_msgpack (src/flb_pack.c:328-387):
 *
 *   documents   a stream of values separated by [ \t\n\r]*; the first value that does not parse ends
 *               the stream (error only if nothing was parsed)                    flb_pack.c:421-487
 *   structure   strict JSON: no trailing commas, string keys, any root type     yyjson.c:5445-5847
 *   numbers     -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][-+]?[0-9]+)?; integers that fit u64 (or i64 when
 *               negative) stay integers ("-0" is the integer 0), everything else is the correctly
 *               rounded binary64; a real that overflows is an error             yyjson.c:3816-4200
 *   strings     escapes \" \\ \/ \b \f \n \r \t \uXXXX (+ surrogate pairs); with the REPLACE flag a
 *               malformed \u keeps its text and swallows the offending character, a lone or broken
 *               surrogate becomes U+FFFD; raw control characters and any byte >= 0x80 (valid UTF-8
 *               or not) are copied through                                      yyjson.c:4660-5170
 *   msgpack     smallest encodings (lib/msgpack-c pack_template.h), reals always float64
 *
 * Pinned against the real reader (oracle/_ref/libyyjson_ref.so) by tests/test_json_oracle.py and on
 * the golden vectors of tests/golden/json_kat.json.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "omp.h"

typedef struct { const unsigned char *s; size_t len; } jsrc;

/* reads past the end see the zero padding the reference appends (flb_pack.c:414-416) */
static inline unsigned ch(const jsrc *j, size_t i) { return i < j->len ? j->s[i] : 0; }
static inline int is_ws(unsigned c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
static inline int is_dig(unsigned c) { return c >= '0' && c <= '9'; }
static inline int hexv(unsigned c)
{
    if (c >= '0' && c <= '9') return (int) c - '0';
    if (c >= 'a' && c <= 'f') return (int) c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return (int) c - 'A' + 10;
    return -1;
}
static int hex4(const jsrc *j, size_t i, unsigned *out)
{
    unsigned v = 0;
    int k;
    for (k = 0; k < 4; k++) {
        int h = hexv(ch(j, i + k));
        if (h < 0) return 0;
        v = (v << 4) | (unsigned) h;
    }
    *out = v;
    return 1;
}

static void put_utf8(omp_buf *b, unsigned u)
{
    unsigned char t[4];
    if (u < 0x80) { t[0] = (unsigned char) u; omp_buf_write(b, t, 1); }
    else if (u < 0x800) { t[0] = 0xC0 | (u >> 6); t[1] = 0x80 | (u & 0x3F); omp_buf_write(b, t, 2); }
    else if (u < 0x10000) { t[0] = 0xE0 | (u >> 12); t[1] = 0x80 | ((u >> 6) & 0x3F); t[2] = 0x80 | (u & 0x3F); omp_buf_write(b, t, 3); }
    else { t[0] = 0xF0 | (u >> 18); t[1] = 0x80 | ((u >> 12) & 0x3F); t[2] = 0x80 | ((u >> 6) & 0x3F); t[3] = 0x80 | (u & 0x3F); omp_buf_write(b, t, 4); }
}

/* string at *pos (the opening quote); decoded bytes appended to `out`.  0 ok, -1 error */
static int read_str(const jsrc *j, size_t *pos, omp_buf *out)
{
    size_t i = *pos + 1;
    static const unsigned char fffd[3] = { 0xEF, 0xBF, 0xBD };
    for (;;) {
        unsigned c = ch(j, i);
        if (c == '"') { *pos = i + 1; return 0; }
        if (c == '\\') {
            unsigned e = ch(j, i + 1), hi, lo;
            unsigned char o;
            switch (e) {
            case '"': case '\\': case '/': o = (unsigned char) e; omp_buf_write(out, &o, 1); i += 2; continue;
            case 'b': o = '\b'; omp_buf_write(out, &o, 1); i += 2; continue;
            case 'f': o = '\f'; omp_buf_write(out, &o, 1); i += 2; continue;
            case 'n': o = '\n'; omp_buf_write(out, &o, 1); i += 2; continue;
            case 'r': o = '\r'; omp_buf_write(out, &o, 1); i += 2; continue;
            case 't': o = '\t'; omp_buf_write(out, &o, 1); i += 2; continue;
            case 'u': break;
            default: return -1;                          /* invalid escaped sequence in string */
            }
            /* read_uni_esc, yyjson.c:4660-4826, with REPLACE_INVALID_UNICODE */
            i += 2;
            if (!hex4(j, i, &hi)) {
                size_t cnt = 0, k;
                unsigned nx;
                while (cnt < 4 && hexv(ch(j, i + cnt)) >= 0) cnt++;
                nx = ch(j, i + cnt);
                omp_buf_write(out, "\\u", 2);
                for (k = 0; k < cnt; k++) { o = (unsigned char) ch(j, i + k); omp_buf_write(out, &o, 1); }
                i += cnt;
                if (nx && nx != '"' && nx != '\'') i++;      /* the offending character is swallowed */
                continue;
            }
            i += 4;
            if ((hi & 0xF800) != 0xD800) { put_utf8(out, hi); continue; }
            if ((hi & 0xFC00) == 0xD800) {
                if (!(ch(j, i) == '\\' && ch(j, i + 1) == 'u')) { omp_buf_write(out, fffd, 3); continue; }
                if (!hex4(j, i + 2, &lo)) {
                    size_t cnt = 0;
                    i += 2;
                    while (cnt < 4 && hexv(ch(j, i + cnt)) >= 0) cnt++;
                    i += cnt;
                    omp_buf_write(out, fffd, 3);
                    continue;
                }
                if ((lo & 0xFC00) != 0xDC00) { i += 6; omp_buf_write(out, fffd, 3); continue; }
                put_utf8(out, (((hi - 0xD800) << 10) | (lo - 0xDC00)) + 0x10000);
                i += 6;
                continue;
            }
            omp_buf_write(out, fffd, 3);                 /* low surrogate without a high one */
            continue;
        }
        if (c < 0x20 && i >= j->len) return -1;          /* unclosed string (the padding was reached) */
        {
            unsigned char o = (unsigned char) c;         /* control characters and bytes >= 0x80 pass through */
            omp_buf_write(out, &o, 1);
            i++;
        }
    }
}

/* number at *pos; packs it.  0 ok, -1 error (yyjson.c:3816-4200) */
static int read_num(const jsrc *j, size_t *pos, omp_buf *out)
{
    size_t i = *pos, st = *pos, k;
    int neg = 0, is_real = 0;
    if (ch(j, i) == '-') { neg = 1; i++; }
    if (!is_dig(ch(j, i))) return -1;                    /* no digit after sign / '+' / '.' */
    if (ch(j, i) == '0') {
        i++;
        if (is_dig(ch(j, i))) return -1;                 /* number with leading zero is not allowed */
    }
    else while (is_dig(ch(j, i))) i++;
    if (ch(j, i) == '.') {
        is_real = 1;
        i++;
        if (!is_dig(ch(j, i))) return -1;                /* no digit after decimal point */
        while (is_dig(ch(j, i))) i++;
    }
    if (ch(j, i) == 'e' || ch(j, i) == 'E') {
        is_real = 1;
        i++;
        if (ch(j, i) == '-' || ch(j, i) == '+') i++;
        if (!is_dig(ch(j, i))) return -1;                /* no digit after exponent sign */
        while (is_dig(ch(j, i))) i++;
    }
    if (!is_real) {
        /* integer: u64 magnitude if it fits (yyjson.c:3963-3971,4011-4028) */
        uint64_t v = 0;
        int ovf = 0;
        for (k = st + neg; k < i; k++) {
            unsigned d = ch(j, k) - '0';
            if (v > (UINT64_MAX - d) / 10) { ovf = 1; break; }
            v = v * 10 + d;
        }
        if (!ovf && !(neg && v > ((uint64_t) 1 << 63))) {
            if (neg) omp_pack_int64(out, (int64_t) (~v + 1));
            else omp_pack_uint64(out, v);
            *pos = i;
            return 0;
        }
    }
    {
        /* real: the correctly rounded binary64 of the literal (libc strtod is) */
        char tmp[64], *buf = tmp;
        double d;
        size_t n = i - st;
        if (n + 1 > sizeof(tmp)) buf = malloc(n + 1);
        for (k = 0; k < n; k++) buf[k] = (char) ch(j, st + k);
        buf[n] = 0;
        d = strtod(buf, NULL);
        if (buf != tmp) free(buf);
        if (isinf(d)) return -1;                         /* number is infinity when parsed as double */
        omp_pack_double(out, d);
        *pos = i;
        return 0;
    }
}

static int lit(const jsrc *j, size_t i, const char *w)
{
    size_t k;
    for (k = 0; w[k]; k++) if (ch(j, i + k) != (unsigned char) w[k]) return 0;
    return 1;
}

/* one value at *pos (recursive: the oracle favours clarity; the reader itself is iterative) */
static int read_val(const jsrc *j, size_t *pos, omp_buf *out, int *type)
{
    unsigned c = ch(j, *pos);
    if (c == '{' || c == '[') {
        /* children are packed into a side buffer: the header needs their count first */
        omp_buf kids;
        size_t i = *pos + 1, n = 0;
        const int obj = c == '{';
        const unsigned close = obj ? '}' : ']';
        omp_buf_init(&kids);
        for (;;) {
            int t;
            while (is_ws(ch(j, i))) i++;
            if (ch(j, i) == close) {
                if (n != 0) goto bad;                    /* trailing comma */
                i++;
                break;
            }
            if (obj) {
                omp_buf key;
                if (ch(j, i) != '"') goto bad;
                omp_buf_init(&key);
                if (read_str(j, &i, &key) != 0) { omp_buf_free(&key); goto bad; }
                omp_pack_str_with_body(&kids, key.data, key.size);
                omp_buf_free(&key);
                while (is_ws(ch(j, i))) i++;
                if (ch(j, i) != ':') goto bad;
                i++;
                while (is_ws(ch(j, i))) i++;
            }
            if (read_val(j, &i, &kids, &t) != 0) goto bad;
            n++;
            while (is_ws(ch(j, i))) i++;
            if (ch(j, i) == ',') { i++; continue; }
            if (ch(j, i) == close) { i++; break; }
            goto bad;
        }
        if (obj) omp_pack_map(out, n); else omp_pack_array(out, n);
        omp_buf_write(out, kids.data, kids.size);
        omp_buf_free(&kids);
        *pos = i;
        *type = obj ? 1 : 2;
        return 0;
bad:
        omp_buf_free(&kids);
        return -1;
    }
    if (c == '"') {
        omp_buf s;
        omp_buf_init(&s);
        if (read_str(j, pos, &s) != 0) { omp_buf_free(&s); return -1; }
        omp_pack_str_with_body(out, s.data, s.size);
        omp_buf_free(&s);
        *type = 3;
        return 0;
    }
    *type = 4;
    if (c == 't') { if (!lit(j, *pos, "true")) return -1; omp_pack_bool(out, 1); *pos += 4; return 0; }
    if (c == 'f') { if (!lit(j, *pos, "false")) return -1; omp_pack_bool(out, 0); *pos += 5; return 0; }
    if (c == 'n') { if (!lit(j, *pos, "null")) return -1; omp_pack_nil(out); *pos += 4; return 0; }
    if (c == '-' || c == '+' || c == '.' || is_dig(c)) return read_num(j, pos, out);
    return -1;
}

/* flb_pack_json / flb_pack_json_recs: src/flb_pack.c:670-688 -> :389-508.
 * root_type uses jsmn's values (lib/jsmn/jsmn.h:47-51): 1 object, 2 array, 3 string, 4 primitive */
int ojson_pack(const char *js, size_t len, char **buffer, size_t *size, int *root_type, int *records, size_t *consumed)
{
    jsrc j = { (const unsigned char *) js, len };
    size_t pos = 0;
    int count = 0;
    omp_buf b;
    omp_buf_init(&b);
    while (pos < len) {
        size_t p, mark;
        int t = 0;
        while (pos < len && is_ws(ch(&j, pos))) pos++;
        if (pos >= len) break;
        p = pos;
        mark = b.size;
        if (read_val(&j, &p, &b, &t) != 0 || p == pos) {
            b.size = mark;                               /* a value that fails leaves nothing behind */
            if (count > 0) break;
            omp_buf_free(&b);
            return -1;
        }
        if (root_type && count == 0) *root_type = t;
        count++;
        pos = p;
    }
    if (records) *records = count;
    if (consumed) *consumed = pos;
    *buffer = b.data;
    *size = b.size;
    return 0;
}
