/*
 * oracle/omp.c -- TEST INFRASTRUCTURE ONLY (see omp.h).
 *
 * msgpack DOM unpack + canonical pack, restating lib/msgpack-c:
 *   unpack : lib/msgpack-c/include/msgpack/unpack_template.h, lib/msgpack-c/src/unpack.c
 *   pack   : lib/msgpack-c/cmake/pack_template.h.in
 *   object : lib/msgpack-c/src/objectc.c:39-126 (msgpack_pack_object)
 * and the log event decoder src/flb_log_event_decoder.c:182-511.
 */
#include <stdlib.h>
#include <string.h>
#include "omp.h"

/* ------------------------------------------------------------------ arena */
struct omp_arena_chunk { struct omp_arena_chunk *next; size_t cap; size_t used; };

void omp_arena_init(omp_arena *a) { memset(a, 0, sizeof(*a)); }

static void *arena_alloc(omp_arena *a, size_t n)
{
    struct omp_arena_chunk *c = a->chunks;
    n = (n + 15) & ~(size_t) 15;
    if (c == NULL || c->used + n > c->cap) {
        size_t cap = n > 65536 ? n : 65536;
        c = malloc(sizeof(*c) + cap);
        c->cap = cap; c->used = 0; c->next = a->chunks;
        a->chunks = c;
    }
    {
        void *p = (char *) (c + 1) + c->used;
        c->used += n;
        return p;
    }
}

void omp_arena_reset(omp_arena *a)
{
    /* keep the newest chunk, drop the rest */
    struct omp_arena_chunk *c = a->chunks;
    if (c == NULL) return;
    while (c->next) {
        struct omp_arena_chunk *n = c->next;
        c->next = n->next;
        free(n);
    }
    c->used = 0;
}

void omp_arena_free(omp_arena *a)
{
    struct omp_arena_chunk *c = a->chunks;
    while (c) { struct omp_arena_chunk *n = c->next; free(c); c = n; }
    a->chunks = NULL;
}

/* ------------------------------------------------------------------ unpack */
static uint64_t be(const unsigned char *p, int n)
{
    uint64_t v = 0; int i;
    for (i = 0; i < n; i++) v = (v << 8) | p[i];
    return v;
}

/* returns 1 ok, 0 need more data, -1 parse error */
static int unpack_obj(omp_arena *a, omp_obj *o, const unsigned char *d, size_t len, size_t *off,
                      int depth)
{
    size_t p = *off;
    unsigned char c;
    uint32_t n, i;
    int r;
    if (p >= len) return 0;
    c = d[p++];
#define NEED(k) do { if (len - p < (size_t) (k)) return 0; } while (0)
    if (c <= 0x7f) { o->type = OMP_POS; o->via.u64 = c; }
    else if (c >= 0xe0) { o->type = OMP_NEG; o->via.i64 = (int8_t) c; }
    else if (c >= 0xa0 && c <= 0xbf) { n = c & 0x1f; goto str_body; }
    else if (c >= 0x90 && c <= 0x9f) { n = c & 0x0f; goto arr_body; }
    else if (c >= 0x80 && c <= 0x8f) { n = c & 0x0f; goto map_body; }
    else switch (c) {
    case 0xc0: o->type = OMP_NIL; break;
    case 0xc2: o->type = OMP_BOOL; o->via.b = 0; break;
    case 0xc3: o->type = OMP_BOOL; o->via.b = 1; break;
    case 0xc4: NEED(1); n = d[p]; p += 1; goto bin_body;
    case 0xc5: NEED(2); n = (uint32_t) be(d + p, 2); p += 2; goto bin_body;
    case 0xc6: NEED(4); n = (uint32_t) be(d + p, 4); p += 4; goto bin_body;
    case 0xc7: NEED(1); n = d[p]; p += 1; goto ext_body;
    case 0xc8: NEED(2); n = (uint32_t) be(d + p, 2); p += 2; goto ext_body;
    case 0xc9: NEED(4); n = (uint32_t) be(d + p, 4); p += 4; goto ext_body;
    case 0xca: { uint32_t u; float f; NEED(4); u = (uint32_t) be(d + p, 4); p += 4;
                 memcpy(&f, &u, 4); o->type = OMP_F32; o->via.f64 = f; break; }
    case 0xcb: { uint64_t u; double f; NEED(8); u = be(d + p, 8); p += 8;
                 memcpy(&f, &u, 8); o->type = OMP_F64; o->via.f64 = f; break; }
    case 0xcc: NEED(1); o->type = OMP_POS; o->via.u64 = d[p]; p += 1; break;
    case 0xcd: NEED(2); o->type = OMP_POS; o->via.u64 = be(d + p, 2); p += 2; break;
    case 0xce: NEED(4); o->type = OMP_POS; o->via.u64 = be(d + p, 4); p += 4; break;
    case 0xcf: NEED(8); o->type = OMP_POS; o->via.u64 = be(d + p, 8); p += 8; break;
    /* signed families: non-negative values become POSITIVE_INTEGER
     * (lib/msgpack-c/src/unpack.c template_callback_int8..int64) */
    case 0xd0: { int64_t v; NEED(1); v = (int8_t) d[p]; p += 1; goto sint; sint:
                 if (v >= 0) { o->type = OMP_POS; o->via.u64 = (uint64_t) v; }
                 else { o->type = OMP_NEG; o->via.i64 = v; }
                 break;
    case 0xd1:   NEED(2); v = (int16_t) be(d + p, 2); p += 2; goto sint;
    case 0xd2:   NEED(4); v = (int32_t) be(d + p, 4); p += 4; goto sint;
    case 0xd3:   NEED(8); v = (int64_t) be(d + p, 8); p += 8; goto sint; }
    case 0xd4: n = 1; goto fixext;
    case 0xd5: n = 2; goto fixext;
    case 0xd6: n = 4; goto fixext;
    case 0xd7: n = 8; goto fixext;
    case 0xd8: n = 16; goto fixext;
    case 0xd9: NEED(1); n = d[p]; p += 1; goto str_body;
    case 0xda: NEED(2); n = (uint32_t) be(d + p, 2); p += 2; goto str_body;
    case 0xdb: NEED(4); n = (uint32_t) be(d + p, 4); p += 4; goto str_body;
    case 0xdc: NEED(2); n = (uint32_t) be(d + p, 2); p += 2; goto arr_body;
    case 0xdd: NEED(4); n = (uint32_t) be(d + p, 4); p += 4; goto arr_body;
    case 0xde: NEED(2); n = (uint32_t) be(d + p, 2); p += 2; goto map_body;
    case 0xdf: NEED(4); n = (uint32_t) be(d + p, 4); p += 4; goto map_body;
    default: /* 0xc1 */
        return -1;
    }
    *off = p;
    return 1;

str_body:
    NEED(n);
    o->type = OMP_STR; o->via.str.size = n; o->via.str.ptr = (const char *) d + p;
    *off = p + n;
    return 1;
bin_body:
    NEED(n);
    o->type = OMP_BIN; o->via.str.size = n; o->via.str.ptr = (const char *) d + p;
    *off = p + n;
    return 1;
fixext:
    NEED(1 + n);
    o->type = OMP_EXT; o->via.ext.type = (int8_t) d[p]; o->via.ext.size = n;
    o->via.ext.ptr = (const char *) d + p + 1;
    *off = p + 1 + n;
    return 1;
ext_body:
    NEED(1 + (size_t) n);
    o->type = OMP_EXT; o->via.ext.type = (int8_t) d[p]; o->via.ext.size = n;
    o->via.ext.ptr = (const char *) d + p + 1;
    *off = p + 1 + n;
    return 1;
arr_body:
    o->type = OMP_ARRAY; o->via.array.size = n; o->via.array.ptr = NULL;
    if (n > 0) {
        if ((size_t) n > len - p) return 0;        /* each element needs >= 1 byte */
        o->via.array.ptr = arena_alloc(a, sizeof(omp_obj) * (size_t) n);
        for (i = 0; i < n; i++) {
            r = unpack_obj(a, &o->via.array.ptr[i], d, len, &p, depth + 1);
            if (r != 1) return r;
        }
    }
    *off = p;
    return 1;
map_body:
    o->type = OMP_MAP; o->via.map.size = n; o->via.map.ptr = NULL;
    if (n > 0) {
        if ((size_t) n * 2 > len - p) return 0;
        o->via.map.ptr = arena_alloc(a, sizeof(omp_kv) * (size_t) n);
        for (i = 0; i < n; i++) {
            r = unpack_obj(a, &o->via.map.ptr[i].key, d, len, &p, depth + 1);
            if (r != 1) return r;
            r = unpack_obj(a, &o->via.map.ptr[i].val, d, len, &p, depth + 1);
            if (r != 1) return r;
        }
    }
    *off = p;
    return 1;
#undef NEED
}

/*
 * What msgpack-c's template_execute (lib/msgpack-c/include/msgpack/unpack_template.h:82-452) reports for
 * an object it cannot finish: the return code (0 = CONTINUE, -1 = PARSE_ERROR, -2 = NOMEM_ERROR) and the
 * offset it leaves in *off.  The executor consumes complete fields -- a type byte, a length field, a
 * payload -- and stops in front of the first field that is cut short (:242-247, :439-447), at the
 * reserved byte 0xc1 (:415-417), or at the 33rd open container (MSGPACK_EMBED_STACK_SIZE 32, :139-143).
 * A buffer that ends exactly on a field boundary therefore comes back as CONTINUE with *off == len, which
 * flb_log_event_decoder_get_last_result / filter_grep read as a clean end
 * (src/flb_log_event_decoder.c:334-342, plugins/filter_grep/grep.c:357-360).
 * Returns 1 if the object is complete after all (the caller's recursive reader said otherwise: never).
 */
#define OMP_STACK 32
static int exec_tail(const unsigned char *d, size_t len, size_t start, size_t *off)
{
    uint64_t count[OMP_STACK];
    int top = 0;
    size_t p = start;
    for (;;) {
        unsigned char c;
        size_t q, e, k = 0, lb = 0;
        uint64_t n = 0;
        int container = 0;
        if (p >= len) { *off = len; return 0; }
        c = d[p];
        q = p + 1;
        if (c <= 0x7f || c >= 0xe0 || c == 0xc0 || c == 0xc2 || c == 0xc3) e = q;
        else if (c == 0xc1) { *off = p; return -1; }
        else if (c >= 0xa0 && c <= 0xbf) { k = c & 0x1f; if (len - q < k) { *off = q; return 0; } e = q + k; }
        else if (c >= 0x90 && c <= 0x9f) { container = 1; n = c & 0x0f; e = q; }
        else if (c >= 0x80 && c <= 0x8f) { container = 1; n = 2u * (c & 0x0f); e = q; }
        else {
            switch (c) {
            case 0xcc: case 0xd0: k = 1; break;
            case 0xcd: case 0xd1: k = 2; break;
            case 0xce: case 0xd2: case 0xca: k = 4; break;
            case 0xcf: case 0xd3: case 0xcb: k = 8; break;
            case 0xd4: k = 2; break;
            case 0xd5: k = 3; break;
            case 0xd6: k = 5; break;
            case 0xd7: k = 9; break;
            case 0xd8: k = 17; break;
            case 0xc4: case 0xc7: case 0xd9: lb = 1; break;
            case 0xc5: case 0xc8: case 0xda: case 0xdc: case 0xde: lb = 2; break;
            default: lb = 4; break;                          /* c6 c9 db dd df */
            }
            if (lb == 0) { if (len - q < k) { *off = q; return 0; } e = q + k; }
            else {
                size_t q2;
                if (len - q < lb) { *off = q; return 0; }
                n = be(d + q, (int) lb);
                q2 = q + lb;
                if (c == 0xdc || c == 0xdd) { container = 1; e = q2; }
                else if (c == 0xde || c == 0xdf) { container = 1; n *= 2; e = q2; }
                else {
                    k = (size_t) n + ((c == 0xc7 || c == 0xc8 || c == 0xc9) ? 1 : 0);       /* ext: type byte + data */
                    if (len - q2 < k) { *off = q2; return 0; }
                    e = q2 + k;
                }
            }
        }
        if (container && n > 0) {
            if (top >= OMP_STACK) { *off = p; return -2; }
            count[top++] = n;
            p = e;
            continue;
        }
        if (container && top >= OMP_STACK) { *off = p; return -2; }          /* the check precedes the count test (:139-146) */
        /* a value is complete: close every container it finishes */
        p = e;
        for (;;) {
            if (top == 0) { *off = p; return 1; }
            if (--count[top - 1] > 0) break;
            top--;
        }
    }
}

int omp_unpack_next(omp_arena *a, omp_obj *out, const char *data, size_t len, size_t *off)
{
    size_t p = *off;
    int r;
    if (len <= p) return OMP_UNPACK_CONTINUE;
    /* the container stack of the real executor is 32 deep: deeper objects are NOMEM errors there */
    r = exec_tail((const unsigned char *) data, len, p, &p);
    if (r != 1) { *off = p; return r == 0 ? OMP_UNPACK_CONTINUE : r; }
    p = *off;
    r = unpack_obj(a, out, (const unsigned char *) data, len, &p, 0);
    if (r == 1) { *off = p; return OMP_UNPACK_SUCCESS; }
    return OMP_UNPACK_PARSE_ERROR;                           /* (not reached: exec_tail accepted the object) */
}

/* ------------------------------------------------------------------ pack */
void omp_buf_init(omp_buf *b) { b->data = NULL; b->size = 0; b->cap = 0; }
void omp_buf_free(omp_buf *b) { free(b->data); b->data = NULL; b->size = b->cap = 0; }

void omp_buf_write(omp_buf *b, const void *p, size_t n)
{
    if (b->size + n > b->cap) {
        size_t cap = b->cap ? b->cap * 2 : 8192;
        while (cap < b->size + n) cap *= 2;
        b->data = realloc(b->data, cap);
        b->cap = cap;
    }
    if (n) memcpy(b->data + b->size, p, n);
    b->size += n;
}

static void put1(omp_buf *b, unsigned char c) { omp_buf_write(b, &c, 1); }
static void putbe(omp_buf *b, unsigned char tag, uint64_t v, int n)
{
    unsigned char t[9]; int i;
    t[0] = tag;
    for (i = 0; i < n; i++) t[1 + i] = (unsigned char) (v >> (8 * (n - 1 - i)));
    omp_buf_write(b, t, 1 + n);
}

void omp_pack_nil(omp_buf *b) { put1(b, 0xc0); }
void omp_pack_bool(omp_buf *b, int v) { put1(b, v ? 0xc3 : 0xc2); }

void omp_pack_uint64(omp_buf *b, uint64_t v)
{
    if (v < (1ULL << 7)) put1(b, (unsigned char) v);
    else if (v < (1ULL << 8)) putbe(b, 0xcc, v, 1);
    else if (v < (1ULL << 16)) putbe(b, 0xcd, v, 2);
    else if (v < (1ULL << 32)) putbe(b, 0xce, v, 4);
    else putbe(b, 0xcf, v, 8);
}

void omp_pack_int64(omp_buf *b, int64_t v)
{
    if (v >= 0) { omp_pack_uint64(b, (uint64_t) v); return; }
    if (v >= -32) put1(b, (unsigned char) (int8_t) v);
    else if (v >= -128) putbe(b, 0xd0, (uint64_t) v, 1);
    else if (v >= -32768) putbe(b, 0xd1, (uint64_t) v, 2);
    else if (v >= -2147483648LL) putbe(b, 0xd2, (uint64_t) v, 4);
    else putbe(b, 0xd3, (uint64_t) v, 8);
}

void omp_pack_float(omp_buf *b, float v) { uint32_t u; memcpy(&u, &v, 4); putbe(b, 0xca, u, 4); }
void omp_pack_double(omp_buf *b, double v) { uint64_t u; memcpy(&u, &v, 8); putbe(b, 0xcb, u, 8); }

void omp_pack_str(omp_buf *b, size_t n)
{
    if (n < 32) put1(b, (unsigned char) (0xa0 | n));
    else if (n < 256) putbe(b, 0xd9, n, 1);
    else if (n < 65536) putbe(b, 0xda, n, 2);
    else putbe(b, 0xdb, n, 4);
}

void omp_pack_str_with_body(omp_buf *b, const char *s, size_t n)
{
    omp_pack_str(b, n);
    omp_buf_write(b, s, n);
}

void omp_pack_bin(omp_buf *b, size_t n)
{
    if (n < 256) putbe(b, 0xc4, n, 1);
    else if (n < 65536) putbe(b, 0xc5, n, 2);
    else putbe(b, 0xc6, n, 4);
}

void omp_pack_ext(omp_buf *b, size_t n, int8_t type)
{
    switch (n) {
    case 1: put1(b, 0xd4); break;
    case 2: put1(b, 0xd5); break;
    case 4: put1(b, 0xd6); break;
    case 8: put1(b, 0xd7); break;
    case 16: put1(b, 0xd8); break;
    default:
        if (n < 256) putbe(b, 0xc7, n, 1);
        else if (n < 65536) putbe(b, 0xc8, n, 2);
        else putbe(b, 0xc9, n, 4);
    }
    put1(b, (unsigned char) type);
}

void omp_pack_array(omp_buf *b, size_t n)
{
    if (n < 16) put1(b, (unsigned char) (0x90 | n));
    else if (n < 65536) putbe(b, 0xdc, n, 2);
    else putbe(b, 0xdd, n, 4);
}

void omp_pack_map(omp_buf *b, size_t n)
{
    if (n < 16) put1(b, (unsigned char) (0x80 | n));
    else if (n < 65536) putbe(b, 0xde, n, 2);
    else putbe(b, 0xdf, n, 4);
}

void omp_pack_object(omp_buf *b, const omp_obj *o)
{
    uint32_t i;
    switch (o->type) {
    case OMP_NIL: omp_pack_nil(b); break;
    case OMP_BOOL: omp_pack_bool(b, o->via.b); break;
    case OMP_POS: omp_pack_uint64(b, o->via.u64); break;
    case OMP_NEG: omp_pack_int64(b, o->via.i64); break;
    case OMP_F32: omp_pack_float(b, (float) o->via.f64); break;
    case OMP_F64: omp_pack_double(b, o->via.f64); break;
    case OMP_STR: omp_pack_str_with_body(b, o->via.str.ptr, o->via.str.size); break;
    case OMP_BIN: omp_pack_bin(b, o->via.str.size); omp_buf_write(b, o->via.str.ptr, o->via.str.size); break;
    case OMP_EXT: omp_pack_ext(b, o->via.ext.size, o->via.ext.type);
                  omp_buf_write(b, o->via.ext.ptr, o->via.ext.size); break;
    case OMP_ARRAY:
        omp_pack_array(b, o->via.array.size);
        for (i = 0; i < o->via.array.size; i++) omp_pack_object(b, &o->via.array.ptr[i]);
        break;
    case OMP_MAP:
        omp_pack_map(b, o->via.map.size);
        for (i = 0; i < o->via.map.size; i++) {
            omp_pack_object(b, &o->via.map.ptr[i].key);
            omp_pack_object(b, &o->via.map.ptr[i].val);
        }
        break;
    }
}

/* ------------------------------------------------------------------ log event decoder */
void oev_decoder_init(oev_decoder *d, const char *buf, size_t len)
{
    memset(d, 0, sizeof(*d));
    d->buf = buf; d->len = len; d->off = 0;
    d->last_result = OEV_ERR_INSUFFICIENT_DATA;
    omp_arena_init(&d->arena);
    omp_arena_init(&d->garena);
    d->empty_map.type = OMP_MAP;
    d->empty_map.via.map.size = 0;
    d->empty_map.via.map.ptr = NULL;
}

void oev_decoder_destroy(oev_decoder *d) { omp_arena_free(&d->arena); omp_arena_free(&d->garena); }

/* src/flb_log_event_decoder.c:182-245 */
static int decode_timestamp(const omp_obj *in, oev_time *out)
{
    out->sec = 0; out->nsec = 0;
    if (in->type == OMP_POS) {
        out->sec = (int64_t) in->via.u64;          /* time_t = u64 (wraps for >= 2^63) */
    }
    else if (in->type == OMP_F64) {
        out->sec = (int64_t) in->via.f64;
        out->nsec = (int64_t) ((in->via.f64 - (double) out->sec) * 1000000000);
    }
    else if (in->type == OMP_EXT) {
        uint32_t sec, nsec;
        if (in->via.ext.type != 0 || in->via.ext.size != 8) return OEV_ERR_WRONG_TIMESTAMP_TYPE;
        sec = (uint32_t) be((const unsigned char *) in->via.ext.ptr, 4);
        nsec = (uint32_t) be((const unsigned char *) in->via.ext.ptr + 4, 4);
        if (sec == 0xffffffffu) {
            if (nsec != 0) return OEV_ERR_WRONG_TIMESTAMP_TYPE;
            out->sec = -1; out->nsec = 0;
            return OEV_SUCCESS;
        }
        if (sec == 0xfffffffeu) {
            if (nsec != 0) return OEV_ERR_WRONG_TIMESTAMP_TYPE;
            out->sec = -2; out->nsec = 0;
            return OEV_SUCCESS;
        }
        /* flb_time_msgpack_to_time + flb_time_is_valid_eventtime (src/flb_time.c:284-298) */
        out->sec = sec; out->nsec = nsec;
        if (nsec >= 1000000000u) return OEV_ERR_WRONG_TIMESTAMP_TYPE;
    }
    else {
        return OEV_ERR_WRONG_TIMESTAMP_TYPE;
    }
    return OEV_SUCCESS;
}

/* src/flb_log_event_decoder.c:247-330 */
static int decode_object(oev_decoder *d, oev_event *ev, omp_obj *root, size_t prev_off)
{
    omp_obj *header, *ts, *meta, *body;
    int r;
    memset(ev, 0, sizeof(*ev));
    if (root->type != OMP_ARRAY) return OEV_ERR_WRONG_ROOT_TYPE;
    if (root->via.array.size != 2) return OEV_ERR_WRONG_ROOT_SIZE;
    header = &root->via.array.ptr[0];
    if (header->type == OMP_ARRAY) {
        if (header->via.array.size != 2) return OEV_ERR_WRONG_HEADER_SIZE;
        ts = &header->via.array.ptr[0];
        meta = &header->via.array.ptr[1];
    }
    else {
        ts = header;
        meta = &d->empty_map;
    }
    if (ts->type != OMP_POS && ts->type != OMP_F64 && ts->type != OMP_EXT)
        return OEV_ERR_WRONG_TIMESTAMP_TYPE;
    if (meta->type != OMP_MAP) return OEV_ERR_WRONG_METADATA_TYPE;
    body = &root->via.array.ptr[1];
    if (body->type != OMP_MAP) return OEV_ERR_WRONG_BODY_TYPE;
    r = decode_timestamp(ts, &ev->ts);
    if (r != OEV_SUCCESS) return r;
    ev->metadata = meta; ev->body = body; ev->root = root;
    ev->record_base = d->buf + prev_off;
    ev->record_length = d->off - prev_off;
    return OEV_SUCCESS;
}

/* src/flb_log_event_decoder.c:342-489 with read_groups == FALSE (the filters' and the formatters' setting).
 * The reference recurses once per skipped record (group markers, "invalid group markers") and refuses to
 * decode at depth 1000 (:27,:389-394): a record that follows 1000 consecutive skipped ones ends the walk. */
int oev_decoder_next(oev_decoder *d, oev_event *ev)
{
    d->recursion_depth = 0;
    for (;;) {
        size_t prev = d->off;
        int r;
        if (d->len == 0) { d->last_result = OEV_ERR_INSUFFICIENT_DATA; return d->last_result; }
        omp_arena_reset(&d->arena);
        r = omp_unpack_next(&d->arena, &d->root, d->buf, d->len, &d->off);
        if (r == OMP_UNPACK_CONTINUE) { d->last_result = OEV_ERR_INSUFFICIENT_DATA; return d->last_result; }
        if (r != OMP_UNPACK_SUCCESS) { d->last_result = OEV_ERR_DESERIALIZATION; return d->last_result; }
        d->last_result = decode_object(d, ev, &d->root, prev);
        if (d->last_result != OEV_SUCCESS) return d->last_result;
        if (d->recursion_depth >= 1000) { d->last_result = OEV_ERR_DESERIALIZATION; return d->last_result; }
        /* record type: src/flb_log_event_decoder.c:491-511.  sec >= 0 normal; -1/-2 group
         * markers (skipped); any other negative value: "invalid group marker", skipped with the
         * group state preserved (:396-413). */
        if (ev->ts.sec >= 0) {
            ev->group_metadata = d->cur_group_metadata;
            ev->group_attributes = d->cur_group_attributes;
            return OEV_SUCCESS;
        }
        if (ev->ts.sec == -1) {
            /* group opener: the decoder keeps this record's objects (:427-461) */
            size_t goff = prev;
            omp_arena_reset(&d->garena);
            omp_unpack_next(&d->garena, &d->groot, d->buf, d->len, &goff);
            {
                omp_obj *header = &d->groot.via.array.ptr[0];
                if (header->type == OMP_ARRAY && header->via.array.size == 2) d->cur_group_metadata = &header->via.array.ptr[1];
                else d->cur_group_metadata = &d->empty_map;
                d->cur_group_attributes = &d->groot.via.array.ptr[1];
            }
        }
        else if (ev->ts.sec == -2) {
            d->cur_group_metadata = NULL;
            d->cur_group_attributes = NULL;
        }
        d->recursion_depth++;
    }
}

int oev_count_records(const char *buf, size_t len)
{
    oev_decoder d;
    oev_event ev;
    int n = 0;
    oev_decoder_init(&d, buf, len);
    while (oev_decoder_next(&d, &ev) == OEV_SUCCESS) n++;
    oev_decoder_destroy(&d);
    return n;
}

/* ------------------------------------------------------------------ pin helper (tests/test_msgpack_pin.py)
 * Same loop as oracle/ref_msgpack_shim.c drives on the real msgpack-c: unpack every object of `data`,
 * re-pack it; codes[i] / ends[i] = return value and offset of the i-th omp_unpack_next call. */
int omp_roundtrip(const char *data, size_t len, char **out, size_t *out_size, int *codes, size_t *ends, int max_calls)
{
    omp_arena arena;
    omp_buf b;
    size_t off = 0;
    int n = 0;
    omp_arena_init(&arena);
    omp_buf_init(&b);
    while (n < max_calls) {
        omp_obj o;
        int r = omp_unpack_next(&arena, &o, data, len, &off);
        codes[n] = r;
        ends[n] = off;
        n++;
        if (r != OMP_UNPACK_SUCCESS) break;
        omp_pack_object(&b, &o);
    }
    omp_arena_free(&arena);
    *out = malloc(b.size ? b.size : 1);
    if (b.size) memcpy(*out, b.data, b.size);
    *out_size = b.size;
    omp_buf_free(&b);
    return n;
}
