/*
 * oracle/ref_yyjson_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL JSON reader of the reference (lib/yyjson-0.12.0/src/yyjson.c, compiled in place
 * by oracle/Makefile into _ref/libyyjson_ref.so) the way src/flb_pack.c:389-508
 * (pack_json_to_msgpack_yyjson) does, so that oracle/ojson.c -- the restatement -- can be pinned
 * on the real thing.  Only the glue is restated here (flb_pack.c needs the whole engine to link);
 * the msgpack bytes come from oracle/omp.c's packers.
 */
#include <stdlib.h>
#include <string.h>
#include <yyjson.h>
#include "omp.h"

/* src/flb_pack.c:328-387 yyjson_val_to_msgpack */
static void val_to_msgpack(yyjson_val *val, omp_buf *b)
{
    size_t idx, max;
    yyjson_val *key, *tmp;
    switch (yyjson_get_type(val)) {
    case YYJSON_TYPE_OBJ:
        omp_pack_map(b, yyjson_obj_size(val));
        yyjson_obj_foreach(val, idx, max, key, tmp) {
            omp_pack_str_with_body(b, yyjson_get_str(key), yyjson_get_len(key));
            val_to_msgpack(tmp, b);
        }
        break;
    case YYJSON_TYPE_ARR:
        omp_pack_array(b, yyjson_arr_size(val));
        yyjson_arr_foreach(val, idx, max, tmp) val_to_msgpack(tmp, b);
        break;
    case YYJSON_TYPE_STR: omp_pack_str_with_body(b, yyjson_get_str(val), yyjson_get_len(val)); break;
    case YYJSON_TYPE_BOOL: omp_pack_bool(b, yyjson_get_bool(val)); break;
    case YYJSON_TYPE_NULL: omp_pack_nil(b); break;
    case YYJSON_TYPE_NUM:
        if (yyjson_is_int(val)) {
            if (yyjson_is_sint(val)) omp_pack_int64(b, yyjson_get_sint(val));
            else omp_pack_uint64(b, yyjson_get_uint(val));
        }
        else omp_pack_double(b, yyjson_get_real(val));
        break;
    default: omp_pack_nil(b);
    }
}

/* src/flb_pack.c:389-508; root_type values are jsmn's (lib/jsmn/jsmn.h:47-51) */
int ref_pack_json(const char *js, size_t len, char **buffer, size_t *size, int *root_type, int *records, size_t *consumed)
{
    int count = 0;
    char *insitu = malloc(len + YYJSON_PADDING_SIZE);
    char *start = insitu, *end = insitu + len;
    omp_buf b;
    memcpy(insitu, js, len);
    memset(insitu + len, 0, YYJSON_PADDING_SIZE);
    omp_buf_init(&b);
    while (start < end) {
        yyjson_read_err err;
        yyjson_doc *doc;
        yyjson_val *root;
        size_t rd;
        while (start < end && (*start == ' ' || *start == '\t' || *start == '\n' || *start == '\r')) start++;
        if (start >= end) break;
        doc = yyjson_read_opts(start, (size_t) (end - start),
                               YYJSON_READ_STOP_WHEN_DONE | YYJSON_READ_INSITU | YYJSON_READ_ALLOW_INVALID_UNICODE |
                               YYJSON_READ_REPLACE_INVALID_UNICODE, NULL, &err);
        if (!doc) {
            if (count > 0) break;
            omp_buf_free(&b); free(insitu);
            return -1;
        }
        rd = yyjson_doc_get_read_size(doc);
        if (rd == 0) {
            yyjson_doc_free(doc);
            if (count == 0) { omp_buf_free(&b); free(insitu); return -1; }
            break;
        }
        root = yyjson_doc_get_root(doc);
        if (!root) { yyjson_doc_free(doc); omp_buf_free(&b); free(insitu); return -1; }
        val_to_msgpack(root, &b);
        if (root_type && count == 0) {
            switch (yyjson_get_type(root)) {
            case YYJSON_TYPE_OBJ: *root_type = 1; break;
            case YYJSON_TYPE_ARR: *root_type = 2; break;
            case YYJSON_TYPE_STR: *root_type = 3; break;
            default: *root_type = 4;
            }
        }
        yyjson_doc_free(doc);
        count++;
        start += rd;
    }
    if (records) *records = count;
    if (consumed) *consumed = (size_t) (start - insitu);
    *buffer = b.data;
    *size = b.size;
    free(insitu);
    return 0;
}

void ref_free(void *p) { free(p); }
