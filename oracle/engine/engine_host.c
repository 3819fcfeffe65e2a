/* engine_host.c -- TEST INFRASTRUCTURE.  A driver linked against the REFERENCE'S OWN ENGINE (oracle/_ref/engine/lib/libfluent-bit.so,
 * built by oracle/build_engine.sh from /root/reference with its cmake) -- nothing of the engine is restated here.  It loads the drop-in
 * plugins with the real flb_plugin_load_router (src/flb_plugin.c:194-320,322-369), instantiates them beside the built-in filters, and
 * drives them the two ways the engine does:
 *
 *   load <a.so> [<b.so> ...]
 *        flb_config_init + flb_plugin_load_router for each file; prints one line per filter plugin registered in
 *        config->filter_plugins: name, whether it came from a DSO, its config_map property names.
 *   processor [-e <plugin.so>]... [--parser 'name|regex|time_fmt|time_key|time_keep']... [--repeat N] <in.mp> <out.mp>
 *             --unit <filter> [k=v]... [--unit <filter> [k=v]...]...
 *        the engine-less route of tests/internal/processor.c:293-336: flb_processor_create / flb_processor_unit_create (a FILTER unit:
 *        src/flb_processor.c:1407-1437 calls the plugin's cb_filter under pu->lock) / flb_processor_run on the whole file as one chunk.
 *        Prints {"ret", "out_bytes", "seconds", "units": [{"name", "records", "dropped", "added"}]} (the per-filter cmetrics the
 *        reference's test asserts at :354-366); the output chunk goes to <out.mp>.
 *   lib [-e <plugin.so>]... [--parser ...]... [--metrics-tag T] [--batch N] <in.json> <out.bin> --filter <filter> [k=v]... [--filter ...]...
 *        the whole engine (flb_create / flb_start: event loop, in_lib -> flb_input_chunk_append_raw -> flb_filter_do -> router ->
 *        out_lib): every line of <in.json> ('[ts, {..}]') is pushed with flb_lib_push, the log chunks out_lib hands over
 *        (data_mode chunk) are concatenated into <out.bin> as  u32 kind = 0, u64 len, bytes ; with --metrics-tag a second out_lib
 *        matches the tag filter_log_to_metrics emits on, each metrics chunk is decoded with cmetrics' own msgpack decoder and
 *        written as prometheus text without timestamps (kind = 1): the LAST one is the state after all records.
 *   configs0 <records> <grep rule> <line text>
 *        BASELINE.json configs[0]: in_dummy -> filter_grep (built-in, one Regex rule) -> out_null inside the real engine; prints
 *        {"records", "seconds", "records_per_s", "filter_records", "filter_dropped"} read from the filter instance's cmetrics.
 *
 * Only tests/ and bench.py's cpu_baseline leg run this; nothing of the product links or loads it. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <time.h>
#include <fluent-bit.h>
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_lib.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_plugin.h>
#include <fluent-bit/flb_filter.h>
#include <fluent-bit/flb_parser.h>
#include <fluent-bit/flb_processor.h>
#include <fluent-bit/flb_time.h>
#include <cmetrics/cmetrics.h>
#include <cmetrics/cmt_counter.h>
#include <cmetrics/cmt_decode_msgpack.h>
#include <cmetrics/cmt_encode_prometheus.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static char *read_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    char *b;
    long n;
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    b = malloc(n + 1);
    if (n && fread(b, 1, n, f) != (size_t) n) { perror(path); exit(2); }
    b[n] = 0;
    fclose(f);
    *len = n;
    return b;
}

/* 'name|regex|time_fmt|time_key|time_keep' -> flb_parser_create (src/flb_parser.c:1003-1049), the values conf/parsers.conf's reader
 * would hand it: Skip_Empty_Values on, Time_Strict on */
static int add_parser(struct flb_config *config, const char *spec)
{
    /* the name ends at the first '|'; time_keep, time_key, time_fmt are split off from the right (the regex may hold '|') */
    char *s = strdup(spec), *f[5] = {NULL, NULL, NULL, NULL, NULL}, *q;
    int i;
    f[0] = s;
    q = strchr(s, '|');
    if (!q) { fprintf(stderr, "bad --parser %s\n", spec); return -1; }
    *q = 0;
    f[1] = q + 1;
    for (i = 4; i >= 2; i--) {
        q = strrchr(f[1], '|');
        if (!q) { fprintf(stderr, "bad --parser %s\n", spec); return -1; }
        *q = 0;
        f[i] = q + 1;
    }
    if (!flb_parser_create(f[0], "regex", f[1], FLB_TRUE, *f[2] ? f[2] : NULL, *f[3] ? f[3] : NULL, NULL,
                           atoi(f[4]) ? FLB_TRUE : FLB_FALSE, FLB_TRUE, FLB_FALSE, FLB_FALSE, NULL, 0, NULL, config)) {
        fprintf(stderr, "flb_parser_create failed for %s\n", f[0]);
        return -1;
    }
    free(s);
    return 0;
}

static void filter_counters(struct flb_filter_instance *f_ins, double *records, double *dropped, double *added)
{
    char *labels[1];
    labels[0] = (char *) flb_filter_name(f_ins);
    *records = *dropped = *added = 0;
    cmt_counter_get_val(f_ins->cmt_records, 1, labels, records);
    cmt_counter_get_val(f_ins->cmt_drop_records, 1, labels, dropped);
    cmt_counter_get_val(f_ins->cmt_add_records, 1, labels, added);
}

/* ------------------------------------------------------------------------------------------------ load */
static int cmd_load(int argc, char **argv)
{
    struct flb_config *config;
    struct mk_list *head, *h2;
    struct flb_filter_plugin *p;
    int i, rc = 0;
    flb_init_env();
    config = flb_config_init();
    if (!config) return 1;
    for (i = 0; i < argc; i++) {
        int r = flb_plugin_load_router(argv[i], config);
        printf("flb_plugin_load_router %s -> %d\n", argv[i], r);
        if (r != 0) rc = 1;
    }
    mk_list_foreach(head, &config->filter_plugins) {
        p = mk_list_entry(head, struct flb_filter_plugin, _head);
        printf("filter_plugin name=%s cb_init=%d cb_filter=%d cb_exit=%d props=", p->name, p->cb_init != NULL, p->cb_filter != NULL, p->cb_exit != NULL);
        if (p->config_map) {
            struct flb_config_map *m;
            for (m = p->config_map; m->name; m++) printf("%s,", m->name);
        }
        printf("\n");
    }
    (void) h2;
    flb_config_exit(config);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ processor */
static int cmd_processor(int argc, char **argv)
{
    struct flb_config *config;
    struct flb_processor *proc;
    struct flb_processor_unit *pu = NULL, *units[16];
    int nunits = 0, i, ret, repeat = 1, r;
    const char *in_path = NULL, *out_path = NULL;
    char *in;
    size_t in_len;
    void *out_buf = NULL;
    size_t out_size = 0;
    double t0, t1, t_first, first_s;
    FILE *fo;

    flb_init_env();
    config = flb_config_init();
    if (!config) return 1;
    proc = flb_processor_create(config, "engine_host", NULL, 0);
    if (!proc) return 1;
    for (i = 0; i < argc; i++) {
        if (!strcmp(argv[i], "-e") && i + 1 < argc) {
            if (flb_plugin_load_router(argv[++i], config) != 0) { fprintf(stderr, "flb_plugin_load_router(%s) failed\n", argv[i]); return 3; }
        }
        else if (!strcmp(argv[i], "--parser") && i + 1 < argc) {
            if (add_parser(config, argv[++i]) != 0) return 3;
        }
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--unit") && i + 1 < argc) {
            pu = flb_processor_unit_create(proc, FLB_PROCESSOR_LOGS, argv[++i]);
            if (!pu) { fprintf(stderr, "flb_processor_unit_create(%s) failed\n", argv[i]); return 3; }
            units[nunits++] = pu;
        }
        else if (pu && strchr(argv[i], '=')) {
            char *kv = strdup(argv[i]), *eq = strchr(kv, '=');
            *eq = 0;
            if (flb_processor_unit_set_property_str(pu, kv, eq + 1) != 0) { fprintf(stderr, "property %s refused\n", argv[i]); return 3; }
            free(kv);
        }
        else if (!in_path) in_path = argv[i];
        else if (!out_path) out_path = argv[i];
        else { fprintf(stderr, "unexpected argument %s\n", argv[i]); return 2; }
    }
    if (!in_path || !out_path || !nunits) { fprintf(stderr, "usage: processor ... <in.mp> <out.mp> --unit <filter> [k=v]...\n"); return 2; }
    if (flb_processor_init(proc) != 0) { printf("{\"ret\": -1, \"init\": false}\n"); return 4; }
    in = read_file(in_path, &in_len);
    t_first = now_s();
    /* the first call by itself (a filter's first chunk pays for its buffers; a GPU plugin's also for the code object and its streams):
     * "first_call_seconds"; "seconds" / "repeat" are the calls behind it when there are any */
    ret = flb_processor_run(proc, 0, FLB_PROCESSOR_LOGS, "t", 1, in, in_len, &out_buf, &out_size);
    t0 = now_s();
    first_s = t0 - t_first;
    if (repeat > 1) {
        for (r = 1; r < repeat; r++) {
            if (out_buf && out_buf != in) flb_free(out_buf);
            out_buf = NULL; out_size = 0;
            ret = flb_processor_run(proc, 0, FLB_PROCESSOR_LOGS, "t", 1, in, in_len, &out_buf, &out_size);
        }
        t1 = now_s();
        repeat -= 1;
    }
    else { t1 = t0; t0 = t_first; }
    fo = fopen(out_path, "wb");
    if (out_buf && out_size) fwrite(out_buf, 1, out_size, fo);
    fclose(fo);
    printf("{\"ret\": %d, \"init\": true, \"out_is_input\": %s, \"out_bytes\": %zu, \"seconds\": %.6f, \"repeat\": %d, \"first_call_seconds\": %.6f, \"units\": [", ret,
           out_buf == (void *) in ? "true" : "false", out_size, t1 - t0, repeat, first_s);
    for (i = 0; i < nunits; i++) {
        double a, b, c;
        struct flb_filter_instance *f_ins = units[i]->ctx;
        filter_counters(f_ins, &a, &b, &c);
        printf("%s{\"name\": \"%s\", \"records\": %.0f, \"dropped\": %.0f, \"added\": %.0f}", i ? ", " : "", flb_filter_name(f_ins), a, b, c);
    }
    printf("]}\n");
    fflush(stdout);
    if (out_buf && out_buf != in) flb_free(out_buf);
    flb_processor_destroy(proc);
    flb_config_exit(config);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ lib */
struct sink { FILE *f; int kind; size_t chunks; size_t bytes; double last; };
static pthread_mutex_t sink_lock = PTHREAD_MUTEX_INITIALIZER;

static int cb_sink(void *record, size_t size, void *data)
{
    struct sink *s = data;
    uint32_t kind = s->kind;
    uint64_t len;
    pthread_mutex_lock(&sink_lock);
    if (s->kind == 0) {
        len = size;
        fwrite(&kind, 4, 1, s->f); fwrite(&len, 8, 1, s->f); fwrite(record, 1, size, s->f);
    }
    else {
        /* a metrics chunk: cmetrics msgpack -> cmetrics' own decoder -> prometheus text without timestamps */
        size_t off = 0;
        struct cmt *cmt = NULL;
        while (cmt_decode_msgpack_create(&cmt, record, size, &off) == 0) {
            cfl_sds_t text = cmt_encode_prometheus_create(cmt, CMT_FALSE);
            len = cfl_sds_len(text);
            fwrite(&kind, 4, 1, s->f); fwrite(&len, 8, 1, s->f); fwrite(text, 1, len, s->f);
            cmt_encode_prometheus_destroy(text);
            cmt_destroy(cmt);
            cmt = NULL;
        }
    }
    s->chunks++;
    s->bytes += size;
    s->last = now_s();
    fflush(s->f);
    pthread_mutex_unlock(&sink_lock);
    return 0;
}

static int cmd_lib(int argc, char **argv)
{
    flb_ctx_t *ctx;
    int in_ffd, out_ffd, f_ffd = -1, i;
    const char *in_path = NULL, *out_path = NULL, *mtag = NULL;
    struct sink logs = {NULL, 0, 0, 0, 0.0}, mets = {NULL, 1, 0, 0, 0.0};
    double t_push0 = 0.0, t_push1 = 0.0;
    struct flb_lib_out_cb cb_logs, cb_mets;
    char *in, *p, *e;
    size_t in_len;
    FILE *fo;
    int pushed = 0, batch = 500;

    ctx = flb_create();
    if (!ctx) return 1;
    flb_service_set(ctx, "flush", "0.2", "grace", "1", "log_level", getenv("ENGINE_HOST_LOG") ? getenv("ENGINE_HOST_LOG") : "error", NULL);
    /* pass 1: plugins and parsers (they must exist before the filters are instantiated) */
    for (i = 0; i < argc; i++) {
        if (!strcmp(argv[i], "-e") && i + 1 < argc) {
            if (flb_plugin_load_router(argv[++i], ctx->config) != 0) { fprintf(stderr, "flb_plugin_load_router(%s) failed\n", argv[i]); return 3; }
        }
        else if (!strcmp(argv[i], "--parser") && i + 1 < argc) {
            if (add_parser(ctx->config, argv[++i]) != 0) return 3;
        }
        else if (!strcmp(argv[i], "--metrics-tag") && i + 1 < argc) mtag = argv[++i];
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--filter")) break;
        else if (!in_path) in_path = argv[i];
        else if (!out_path) out_path = argv[i];
    }
    if (!in_path || !out_path) { fprintf(stderr, "usage: lib ... <in.json> <out.bin> --filter <filter> [k=v]...\n"); return 2; }
    in_ffd = flb_input(ctx, "lib", NULL);
    flb_input_set(ctx, in_ffd, "tag", "t", NULL);
    for (; i < argc; i++) {
        if (!strcmp(argv[i], "--filter") && i + 1 < argc) {
            f_ffd = flb_filter(ctx, argv[++i], NULL);
            if (f_ffd < 0) { fprintf(stderr, "flb_filter(%s) failed\n", argv[i]); return 3; }
            flb_filter_set(ctx, f_ffd, "match", "t", NULL);
        }
        else if (f_ffd >= 0 && strchr(argv[i], '=')) {
            char *kv = strdup(argv[i]), *eq = strchr(kv, '=');
            *eq = 0;
            if (flb_filter_set(ctx, f_ffd, kv, eq + 1, NULL) != 0) { fprintf(stderr, "property %s refused\n", argv[i]); return 3; }
            free(kv);
        }
    }
    fo = fopen(out_path, "wb");
    logs.f = mets.f = fo;
    cb_logs.cb = cb_sink; cb_logs.data = &logs;
    out_ffd = flb_output(ctx, "lib", &cb_logs);
    flb_output_set(ctx, out_ffd, "match", "t", "data_mode", "chunk", NULL);
    if (mtag) {
        cb_mets.cb = cb_sink; cb_mets.data = &mets;
        out_ffd = flb_output(ctx, "lib", &cb_mets);
        flb_output_set(ctx, out_ffd, "match", mtag, "data_mode", "chunk", NULL);
    }
    if (flb_start(ctx) != 0) { printf("{\"started\": false}\n"); fclose(fo); flb_destroy(ctx); return 4; }
    in = read_file(in_path, &in_len);
    t_push0 = now_s();
    /* `batch` lines per flb_lib_push (in_lib's JSON state parser takes a stream of documents): one chunk append, i.e. one
     * flb_filter_do call, per push */
    for (p = in; p < in + in_len; ) {
        int k = 0;
        char *b0 = p;
        while (p < in + in_len && k < batch) {
            e = memchr(p, '\n', in + in_len - p);
            if (!e) e = in + in_len;
            if (e > p) k++;
            p = e + 1;
        }
        if (p > in + in_len) p = in + in_len;
        if (k) { flb_lib_push(ctx, in_ffd, b0, p - b0); pushed += k; }
    }
    t_push1 = now_s();
    /* the engine's flush timer (0.2 s) and the emitter's own collector get their turns: wait until nothing has arrived for 1.5 s */
    {
        size_t seen = (size_t) -1;
        int quiet = 0, waited = 0;
        while (quiet < 15 && waited < 600) {
            size_t cur;
            usleep(100000);
            waited++;
            pthread_mutex_lock(&sink_lock);
            cur = logs.chunks + mets.chunks;
            pthread_mutex_unlock(&sink_lock);
            if (cur == seen) quiet++; else { quiet = 0; seen = cur; }
        }
    }
    flb_stop(ctx);
    flb_destroy(ctx);
    fclose(fo);
    /* seconds: first push -> the last log chunk out_lib handed over (the flush timer's 0.2 s granularity is in it); push_seconds: the
     * pushes alone (in_lib's JSON packing and the filters run inside flb_lib_push's pipe reader, not here) */
    printf("{\"started\": true, \"pushed\": %d, \"log_chunks\": %zu, \"metric_chunks\": %zu, \"log_bytes\": %zu, \"seconds\": %.4f, \"push_seconds\": %.4f}\n",
           pushed, logs.chunks, mets.chunks, logs.bytes, logs.last > t_push0 ? logs.last - t_push0 : t_push1 - t_push0, t_push1 - t_push0);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ configs0 */
static int cmd_configs0(int argc, char **argv)
{
    flb_ctx_t *ctx;
    int in_ffd, out_ffd, f_ffd;
    long records, copies = 1000, samples;
    char dummy[4096], nbuf[32], cbuf[32];
    struct mk_list *head;
    struct flb_filter_instance *f_ins = NULL;
    double t0, t1, rec = 0, drop = 0, add = 0;
    const char *q;
    char *d;

    if (argc < 3) { fprintf(stderr, "usage: configs0 <records> <grep rule> <line text>\n"); return 2; }
    records = atol(argv[0]);
    samples = (records + copies - 1) / copies;
    d = dummy + sprintf(dummy, "{\"log\":\"");
    for (q = argv[2]; *q && d < dummy + sizeof(dummy) - 8; q++) { if (*q == '"' || *q == '\\') *d++ = '\\'; *d++ = *q; }
    strcpy(d, "\"}");
    ctx = flb_create();
    flb_service_set(ctx, "flush", "0.05", "grace", "1", "log_level", "error", NULL);
    in_ffd = flb_input(ctx, "dummy", NULL);
    snprintf(nbuf, sizeof(nbuf), "%ld", samples);
    snprintf(cbuf, sizeof(cbuf), "%ld", copies);
    flb_input_set(ctx, in_ffd, "tag", "t", "dummy", dummy, "samples", nbuf, "copies", cbuf, "rate", "100000", NULL);
    f_ffd = flb_filter(ctx, "grep", NULL);
    flb_filter_set(ctx, f_ffd, "match", "t", "regex", argv[1], NULL);
    out_ffd = flb_output(ctx, "null", NULL);
    flb_output_set(ctx, out_ffd, "match", "t", NULL);
    t0 = now_s();
    if (flb_start(ctx) != 0) { printf("{\"started\": false}\n"); return 4; }
    mk_list_foreach(head, &ctx->config->filters) { f_ins = mk_list_entry(head, struct flb_filter_instance, _head); break; }
    for (;;) {
        filter_counters(f_ins, &rec, &drop, &add);
        t1 = now_s();
        if (rec >= samples * copies || t1 - t0 > 120) break;
        usleep(2000);
    }
    printf("{\"started\": true, \"records\": %.0f, \"seconds\": %.4f, \"records_per_s\": %.0f, \"filter_records\": %.0f, \"filter_dropped\": %.0f, "
           "\"pipeline\": \"in_dummy -> filter_grep -> out_null (reference engine, 1 thread)\"}\n", rec, t1 - t0, rec / (t1 - t0), rec, drop);
    fflush(stdout);
    flb_stop(ctx);
    flb_destroy(ctx);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "load")) return cmd_load(argc - 2, argv + 2);
    if (argc >= 2 && !strcmp(argv[1], "processor")) return cmd_processor(argc - 2, argv + 2);
    if (argc >= 2 && !strcmp(argv[1], "lib")) return cmd_lib(argc - 2, argv + 2);
    if (argc >= 2 && !strcmp(argv[1], "configs0")) return cmd_configs0(argc - 2, argv + 2);
    fprintf(stderr, "usage: engine_host load|processor|lib|configs0 ... (see the header of oracle/engine/engine_host.c)\n");
    return 2;
}
