/* ref_packfmt_shim.c -- TEST INFRASTRUCTURE.  Drives the REAL flb_pack_msgpack_to_json_format
 * (/root/reference/src/flb_pack.c:1320-1600) compiled from where it lies, together with the real
 * flb_utils.c (flb_utils_write_str), flb_utf8.c, flb_sds.c, flb_log_event_decoder.c, flb_time.c, flb_mp.c
 * and msgpack-c.  Built by oracle/Makefile into oracle/_ref/ref_packfmt (an executable: the objects carry
 * references to engine parts this path never calls, which the link leaves unresolved).
 *
 *   ref_packfmt <json_format> <date_format> <date_key | -> <escape_unicode> <nan_to_null> < chunk.msgpack > out
 *
 * Output: "NULL\n" when the function returns NULL, else "OK <len>\n" followed by the bytes.
 * Batch mode (argv[1] == "batch"): stdin is a sequence of cases
 *   u32 json_format, u32 date_format, u32 escape, u32 nan_null, i32 date_key_len (-1 = NULL), key bytes,
 *   u64 data_len, data bytes
 * and stdout the sequence of  i64 len (-1 = NULL), bytes. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_sds.h>
#include <fluent-bit/flb_pack.h>

#include <fluent-bit/flb_config.h>

/* json.convert_nan_to_null reaches the formatter through flb_pack_init(config) (src/flb_pack.c:1740-1750) */
static void set_nan_to_null(int b)
{
    static struct flb_config cfg;
    cfg.convert_nan_to_null = b ? 1 : 0;
    flb_pack_init(&cfg);
}

/* what the objects reference from the logger: the worker context stays NULL and the print hooks do nothing */
#include <fluent-bit/flb_log.h>
#include <fluent-bit/flb_worker.h>
FLB_TLS_DEFINE(struct flb_worker, flb_worker_ctx);
void flb_log_print(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; }
int flb_log_is_truncated(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; return 0; }
int flb_errno_print(int errnum, const char *file, int line) { (void) errnum; (void) file; (void) line; return 0; }

static char *slurp(FILE *f, size_t *n)
{
    size_t cap = 1 << 16, len = 0;
    char *b = malloc(cap);
    for (;;) {
        size_t r = fread(b + len, 1, cap - len, f);
        len += r;
        if (r == 0) break;
        if (len == cap) { cap *= 2; b = realloc(b, cap); }
    }
    *n = len;
    return b;
}

static int rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "batch") == 0) {
        for (;;) {
            uint32_t h[4];
            int32_t kl;
            uint64_t dl;
            char *key = NULL, *data;
            flb_sds_t dk = NULL, out;
            int64_t ol;
            if (!rd(h, sizeof(h))) break;
            if (!rd(&kl, 4)) return 2;
            if (kl >= 0) { key = malloc((size_t) kl + 1); if (kl && !rd(key, (size_t) kl)) return 2; dk = flb_sds_create_len(key, kl); }
            if (!rd(&dl, 8)) return 2;
            data = malloc(dl ? dl : 1);
            if (dl && !rd(data, dl)) return 2;
            set_nan_to_null((int) h[3]);
            out = flb_pack_msgpack_to_json_format(data, dl, (int) h[0], (int) h[1], dk, (int) h[2]);
            ol = out ? (int64_t) flb_sds_len(out) : -1;
            fwrite(&ol, 8, 1, stdout);
            if (out) { fwrite(out, 1, (size_t) ol, stdout); flb_sds_destroy(out); }
            if (dk) flb_sds_destroy(dk);
            free(key);
            free(data);
        }
        return 0;
    }
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    {
        size_t n;
        char *data = slurp(stdin, &n);
        flb_sds_t dk = strcmp(argv[3], "-") == 0 ? NULL : flb_sds_create(argv[3]);
        flb_sds_t out;
        set_nan_to_null(atoi(argv[5]));
        out = flb_pack_msgpack_to_json_format(data, n, atoi(argv[1]), atoi(argv[2]), dk, atoi(argv[4]));
        if (!out) { printf("NULL\n"); return 0; }
        printf("OK %zu\n", flb_sds_len(out));
        fwrite(out, 1, flb_sds_len(out), stdout);
    }
    return 0;
}
