/* ref_filters_shim.c -- TEST INFRASTRUCTURE.  Drives the REAL filter plugins of the reference --
 * plugins/filter_grep/grep.c (cb_grep_init / cb_grep_filter) and plugins/filter_parser/filter_parser.c
 * (cb_parser_init / cb_parser_filter) -- compiled from where they lie together with everything under them:
 * src/flb_parser*.c, flb_regex.c, flb_strptime.c, flb_log_event_decoder.c, flb_log_event_encoder*.c, flb_mp.c, flb_pack.c,
 * flb_record_accessor.c, flb_ra_key.c, record_accessor/flb_ra_parser.c, flb_config_map.c, flb_kv.c, flb_time.c, flb_sds.c,
 * flb_utils.c, ..., the real Onigmo and the real msgpack-c (oracle/Makefile, _ref/ref_filters: an executable, because the
 * objects reference engine parts this path never calls, which the link leaves unresolved).
 *
 * What is NOT the reference's: (1) the two files flex / bison generate from src/record_accessor/ra.l / ra.y (neither tool
 * is in this image): the 40-line grammar -- '$' IDENTIFIER ( '[' STRING ']' | '[' INTEGER ']' )* -- is written out by
 * hand below with the lexer's exact token rules; (2) the engine around a filter instance: this file builds the
 * struct flb_filter_instance / struct flb_config the callbacks read (properties list, config map, parser list) the way
 * src/flb_filter.c does (flb_filter_set_property: flb_kv_item_create; flb_filter_init: flb_config_map_create).
 *
 * Protocol (stdin -> stdout, binary, one case after the other):
 *   u32 kind (1 grep, 2 parser, 3 bench-pair), u32 nprops, nprops x (u32 klen, key, u32 vlen, val),
 *   u32 nparsers, nparsers x 9 strings (name, format, regex, time_fmt, time_key, time_offset, types, skip_empty "0/1",
 *   flags "time_keep time_strict[ tz=<zone>][ systz][|decode_field[_as] backend field [action]]..."), u64 data_len, data        [kind 3: u32 iterations first]
 *   kind 4 (in_tail's line packing): props = key, path_key, path, offset_key, stream_offset, skip_empty_lines, sec, nsec; data = text:
 *   the loop of process_content (plugins/in_tail/tail_file.c:783-786,840-1000, plain path) restated here, every line packed by the
 *   REAL encoder through flb_tail_file_pack_line's call sequence (:552-604); answer ret = lines, out = records, then u64 processed
 *   kind 6 (filter_log_to_metrics, plugins/filter_log_to_metrics/log_to_metrics.c + lib/cmetrics + lib/cfl compiled in place): props = the
 *   plugin's own properties; data = chunks, each behind a u64 length; the answer is described at run_log_to_metrics.  The emitter input
 *   and the flush timer cb_init asks the engine for are stubs (see below): records in, cmetrics state out is what is pinned.
 *   kind 5 (multiline, src/multiline/*.c compiled in place): props = type regex|endswith|equal, match_string, negate, key_content, buffer_limit,
 *   builtin (java|go|python|ruby: the built-in parser of that name instead of rules), rule (repeated: from_states \x1f regex \x1f to_state),
 *   skip_empty_lines, final_flush; data = frames (u32 sec, u32 nsec, u32 len, text): every frame is what one read of in_tail appends to
 *   the file's buffer -- the loop of process_content cuts it into lines (what follows the last newline waits for the next frame) and
 *   hands each to the REAL flb_ml_append_text with the frame's time; answer ret = records flushed, out = what the flush callback received
 * answer: i32 ret (-100: cb_init failed), u64 out_len, out bytes; kind 3: f64 seconds, u64 records in, u64 records kept, then the
 * pair's output bytes of the first pass (outside the timed loop's cost: one memcpy) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_mem.h>
#include <fluent-bit/flb_sds.h>
#include <fluent-bit/flb_kv.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_config_map.h>
#include <fluent-bit/flb_filter.h>
#include <fluent-bit/flb_parser.h>
#include <fluent-bit/flb_parser_decoder.h>
#include <fluent-bit/flb_log.h>
#include <fluent-bit/flb_worker.h>
#include <fluent-bit/flb_slist.h>
#include <fluent-bit/flb_mp.h>
#include <fluent-bit/flb_utils.h>
#include <fluent-bit/flb_env.h>
#include <fluent-bit/flb_time.h>
#include <fluent-bit/flb_log_event_encoder.h>
#include <fluent-bit/record_accessor/flb_ra_parser.h>
#include <fluent-bit/multiline/flb_ml.h>
#include <fluent-bit/multiline/flb_ml_parser.h>
#include <fluent-bit/multiline/flb_ml_rule.h>

#include <fluent-bit/flb_input.h>
#include <fluent-bit/flb_scheduler.h>
#include <cmetrics/cmetrics.h>
#include <cmetrics/cmt_counter.h>
#include <cmetrics/cmt_gauge.h>
#include <cmetrics/cmt_histogram.h>
#include <cmetrics/cmt_map.h>
#include <cmetrics/cmt_metric.h>
#include "log_to_metrics.h"

extern struct flb_filter_plugin filter_grep_plugin;
extern struct flb_filter_plugin filter_parser_plugin;
extern struct flb_filter_plugin filter_log_to_metrics_plugin;

/* ---- what cb_log_to_metrics_init asks of the engine (plugins/filter_log_to_metrics/log_to_metrics.c:873-968): an emitter input and a
 * timer.  Neither is on the path that is pinned here -- records in, cmetrics state out --: the input is a zeroed instance nobody runs,
 * the timer is never armed, appended metric contexts are counted. */
static int g_metrics_appended;
int flb_input_name_exists(const char *name, struct flb_config *config) { (void) name; (void) config; return FLB_FALSE; }
struct flb_input_instance *flb_input_new(struct flb_config *config, const char *input, void *data, int public_only)
{ (void) config; (void) input; (void) data; (void) public_only; return flb_calloc(1, sizeof(struct flb_input_instance)); }
int flb_input_set_property(struct flb_input_instance *ins, const char *k, const char *v) { (void) ins; (void) k; (void) v; return 0; }
int flb_input_instance_init(struct flb_input_instance *ins, struct flb_config *config) { (void) ins; (void) config; return 0; }
int flb_storage_input_create(struct cio_ctx *cio, struct flb_input_instance *in) { (void) cio; (void) in; return 0; }
struct flb_sched *flb_sched_ctx_get(void) { static long fake; return (struct flb_sched *) &fake; }
int flb_sched_timer_cb_create(struct flb_sched *sched, int type, int ms, void (*cb)(struct flb_config *, void *), void *data, struct flb_sched_timer **out_timer)
{ (void) sched; (void) type; (void) ms; (void) cb; (void) data; if (out_timer) *out_timer = NULL; return 0; }
int flb_sched_timer_cb_disable(struct flb_sched_timer *timer) { (void) timer; return 0; }
int flb_sched_timer_destroy(struct flb_sched_timer *timer) { (void) timer; return 0; }
int flb_input_metrics_append(struct flb_input_instance *ins, const char *tag, size_t tag_len, struct cmt *cmt)
{ (void) ins; (void) tag; (void) tag_len; (void) cmt; g_metrics_appended++; return 0; }

/* ---- the logger: the worker context stays NULL and the print hooks do nothing */
FLB_TLS_DEFINE(struct flb_worker, flb_worker_ctx);
void flb_log_print(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; }
int flb_log_is_truncated(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; return 0; }
int flb_errno_print(int errnum, const char *file, int line) { (void) errnum; (void) file; (void) line; return 0; }
struct flb_worker *flb_worker_get(void) { return NULL; }
int flb_worker_log_level(struct flb_worker *worker) { (void) worker; return 0; }
int flb_log_cache_check_suppress(struct flb_log_cache *cache, char *msg_buf, size_t msg_size) { (void) cache; (void) msg_buf; (void) msg_size; return 0; }

/* ---- src/flb_filter.c:757-775 */
void flb_filter_set_context(struct flb_filter_instance *ins, void *context) { ins->context = context; }
const char *flb_filter_get_property(const char *key, struct flb_filter_instance *ins) { return flb_kv_get_key_value((char *) key, &ins->properties); }
const char *flb_filter_name(struct flb_filter_instance *ins) { return ins->alias ? ins->alias : ins->name; }

/* ---- src/record_accessor/ra.l + ra.y by hand (see the header of this file) */
#include "grammar_ra.inc"

/* ---- protocol helpers */
static int rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }
static char *rd_str(void)
{
    uint32_t n;
    char *s;
    if (!rd(&n, 4)) return NULL;
    s = malloc((size_t) n + 1);
    if (n && !rd(s, n)) { free(s); return NULL; }
    s[n] = '\0';
    return s;
}

struct inst {
    struct flb_config *config;
    struct flb_filter_instance ins;
    int ok;
};

/* one filter instance the way src/flb_filter.c builds it: properties (flb_filter_set_property :383-440), config map
 * (flb_filter_init :605-640: flb_config_map_create + flb_config_map_properties_check), cb_init */
static void inst_open(struct inst *it, struct flb_config *config, struct flb_filter_plugin *p, uint32_t nprops, char **keys, char **vals)
{
    uint32_t i;
    memset(&it->ins, 0, sizeof(it->ins));
    it->config = config;
    it->ins.p = p;
    it->ins.config = config;
    it->ins.log_level = -1;
    snprintf(it->ins.name, sizeof(it->ins.name), "%s.0", p->name);
    mk_list_init(&it->ins.properties);
    for (i = 0; i < nprops; i++) {
        /* (the engine expands ${ENV} through flb_env_var_translate first: no variables here) */
        flb_kv_item_create(&it->ins.properties, keys[i], vals[i]);
    }
    it->ins.config_map = p->config_map ? flb_config_map_create(config, p->config_map) : NULL;
    it->ok = p->cb_init(&it->ins, config, NULL) == 0;
}

static struct flb_config *make_config(void)
{
    struct flb_config *c = calloc(1, sizeof(*c));
    mk_list_init(&c->parsers);
    mk_list_init(&c->filters);
    mk_list_init(&c->multiline_parsers);
    c->env = flb_env_create();                  /* flb_config_init: the (empty) environment the config map translates defaults through */
    return c;
}

static void wr_answer(int32_t ret, const void *out, uint64_t n)
{
    fwrite(&ret, 4, 1, stdout);
    fwrite(&n, 8, 1, stdout);
    if (n) fwrite(out, 1, n, stdout);
}

/* a call into an engine part that was left unresolved lands on address 0: say where it came from */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_segv(int sig)
{
    void *bt[32];
    int n = backtrace(bt, 32);
    (void) sig;
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}

/* parser.decoders from "|<decode_field|decode_field_as> <backend> <field> [action]" entries behind the flags: the list
 * flb_parser_decoder_list_create (src/flb_parser_decoder.c:603-745) builds from a [PARSER] section -- same split (' ', 3), same
 * backend / action names, one flb_parser_dec per field (get_decoder_key_context :555-601), rules in configuration order.  That
 * function reads a struct flb_cf_section (the config-format library, not linked here); everything the list is USED by --
 * flb_parser_decoder_do -- is the reference's own object. */
static struct mk_list *make_decoders(const char *spec)
{
    struct mk_list *list = NULL;
    while (spec && *spec == '|') {
        const char *e = strchr(spec + 1, '|');
        size_t n = e ? (size_t) (e - spec - 1) : strlen(spec + 1);
        char *line = flb_strndup(spec + 1, n), *sp = strchr(line, ' ');
        struct mk_list *split, *head;
        struct flb_split_entry *ent[3] = {NULL, NULL, NULL};
        struct flb_parser_dec *dec = NULL;
        struct flb_parser_dec_rule *rule;
        int type, backend, cnt = 0;
        spec = e;
        if (!sp) { flb_free(line); continue; }
        *sp = 0;
        type = !strcasecmp(line, "decode_field_as") ? FLB_PARSER_DEC_AS : FLB_PARSER_DEC_DEFAULT;
        split = flb_utils_split(sp + 1, ' ', 3);
        mk_list_foreach(head, split) { if (cnt < 3) ent[cnt] = mk_list_entry(head, struct flb_split_entry, _head); cnt++; }
        if (cnt < 2) { flb_utils_split_free(split); flb_free(line); continue; }
        if (!strcasecmp(ent[0]->value, "json")) backend = FLB_PARSER_DEC_JSON;
        else if (!strcasecmp(ent[0]->value, "escaped")) backend = FLB_PARSER_DEC_ESCAPED;
        else if (!strcasecmp(ent[0]->value, "escaped_utf8")) backend = FLB_PARSER_DEC_ESCAPED_UTF8;
        else backend = FLB_PARSER_DEC_MYSQL_QUOTED;
        if (!list) { list = flb_malloc(sizeof(struct mk_list)); mk_list_init(list); }
        mk_list_foreach(head, list) {
            struct flb_parser_dec *d = mk_list_entry(head, struct flb_parser_dec, _head);
            if (flb_sds_cmp(d->key, ent[1]->value, strlen(ent[1]->value)) == 0) { dec = d; break; }
        }
        if (!dec) {
            dec = flb_malloc(sizeof(struct flb_parser_dec));
            dec->key = flb_sds_create_len(ent[1]->value, strlen(ent[1]->value));
            dec->buffer = flb_sds_create_size(FLB_PARSER_DEC_BUF_SIZE);
            dec->add_extra_keys = FLB_FALSE;
            mk_list_init(&dec->rules);
            mk_list_add(&dec->_head, list);
        }
        rule = flb_calloc(1, sizeof(struct flb_parser_dec_rule));
        if (type == FLB_PARSER_DEC_DEFAULT) dec->add_extra_keys = FLB_TRUE;
        rule->type = type;
        rule->backend = backend;
        if (cnt >= 3 && ent[2]) {
            if (!strcasecmp(ent[2]->value, "try_next")) rule->action = FLB_PARSER_ACT_TRY_NEXT;
            else if (!strcasecmp(ent[2]->value, "do_next")) rule->action = FLB_PARSER_ACT_DO_NEXT;
            else rule->action = FLB_PARSER_ACT_NONE;
        }
        mk_list_add(&rule->_head, &dec->rules);
        flb_utils_split_free(split);
        flb_free(line);
    }
    return list;
}

/* ---- kind 6: filter_log_to_metrics (plugins/filter_log_to_metrics/log_to_metrics.c compiled in place, the real cmetrics under it).
 * data = chunks, each behind a u64 length: cb_filter runs once per chunk on the same instance.  Answer: i32 ret of the LAST call (-100:
 * cb_init failed), then  u32 nchunks, per chunk i32 ret + u64 out_size;  u32 mode, u32 timer_mode, u32 metrics appended, u32 label keys +
 * strings, u32 buckets + f64 bounds, u32 series, per series: the label strings, f64 value | u64 buckets[nb + 1], u64 count, f64 sum --
 * the metric's map in list order (a map without label keys: its one static metric once it is set). */
static void wb(char **b, size_t *n, size_t *cap, const void *p, size_t len)
{
    if (*n + len > *cap) { *cap = (*n + len) * 2 + 256; *b = realloc(*b, *cap); }
    memcpy(*b + *n, p, len); *n += len;
}
static void wb_u32(char **b, size_t *n, size_t *cap, uint32_t v) { wb(b, n, cap, &v, 4); }
static void wb_str(char **b, size_t *n, size_t *cap, const char *s) { uint32_t l = s ? (uint32_t) strlen(s) : 0; wb_u32(b, n, cap, l); if (l) wb(b, n, cap, s, l); }

static void run_log_to_metrics(struct flb_config *config, uint32_t nprops, char **keys, char **vals, const char *data, uint64_t dlen)
{
    struct inst it;
    struct log_to_metrics_ctx *ctx;
    struct cmt_map *map;
    struct cfl_list *head, *lh;
    char *b = NULL;
    size_t n = 0, cap = 0, at = 0, cnt_at;
    uint32_t nchunks = 0, nb = 0, ns = 0, i;
    int32_t last = 0;
    g_metrics_appended = 0;
    inst_open(&it, config, &filter_log_to_metrics_plugin, nprops, keys, vals);
    if (!it.ok) { wr_answer(-100, NULL, 0); return; }
    ctx = it.ins.context;
    cnt_at = n; wb_u32(&b, &n, &cap, 0);
    while (at + 8 <= dlen) {
        uint64_t len;
        void *out = NULL;
        size_t out_size = 0;
        uint64_t o64;
        memcpy(&len, data + at, 8); at += 8;
        if (at + len > dlen) break;
        last = it.ins.p->cb_filter(data + at, len, "t", 1, &out, &out_size, &it.ins, NULL, it.ins.context, config);
        at += len;
        o64 = out_size;
        wb(&b, &n, &cap, &last, 4); wb(&b, &n, &cap, &o64, 8);
        nchunks++;
    }
    memcpy(b + cnt_at, &nchunks, 4);
    wb_u32(&b, &n, &cap, (uint32_t) ctx->mode); wb_u32(&b, &n, &cap, (uint32_t) ctx->timer_mode); wb_u32(&b, &n, &cap, (uint32_t) g_metrics_appended);
    map = ctx->mode == FLB_LOG_TO_METRICS_COUNTER ? ctx->c->map : ctx->mode == FLB_LOG_TO_METRICS_GAUGE ? ctx->g->map : ctx->h->map;
    wb_u32(&b, &n, &cap, (uint32_t) map->label_count);
    cfl_list_foreach(head, &map->label_keys) wb_str(&b, &n, &cap, cfl_list_entry(head, struct cmt_map_label, _head)->name);
    if (ctx->mode == FLB_LOG_TO_METRICS_HISTOGRAM) {
        nb = (uint32_t) ctx->histogram_buckets->count;
        wb_u32(&b, &n, &cap, nb);
        wb(&b, &n, &cap, ctx->histogram_buckets->upper_bounds, 8 * (size_t) nb);
    }
    else wb_u32(&b, &n, &cap, 0);
    if (map->label_count == 0) ns = map->metric_static_set ? 1 : 0;
    else ns = (uint32_t) cfl_list_size(&map->metrics);
    wb_u32(&b, &n, &cap, ns);
    for (i = 0; i < ns; i++) {
        struct cmt_metric *mt = NULL;
        uint32_t k = 0;
        if (map->label_count == 0) mt = &map->metric;
        else cfl_list_foreach(head, &map->metrics) { if (k++ == i) { mt = cfl_list_entry(head, struct cmt_metric, _head); break; } }
        if (map->label_count) cfl_list_foreach(lh, &mt->labels) wb_str(&b, &n, &cap, cfl_list_entry(lh, struct cmt_map_label, _head)->name);
        if (ctx->mode != FLB_LOG_TO_METRICS_HISTOGRAM) { double v = cmt_metric_get_value(mt); wb(&b, &n, &cap, &v, 8); }
        else {
            uint32_t q;
            uint64_t c;
            double sm;
            for (q = 0; q <= nb; q++) { uint64_t bv = cmt_metric_hist_get_value(mt, (int) q); wb(&b, &n, &cap, &bv, 8); }
            c = cmt_metric_hist_get_count_value(mt); wb(&b, &n, &cap, &c, 8);
            sm = cmt_metric_hist_get_sum_value(mt); wb(&b, &n, &cap, &sm, 8);
        }
    }
    wr_answer(last, b, n);
    free(b);
    if (it.ins.p->cb_exit) it.ins.p->cb_exit(it.ins.context, config);
}

/* ---- kind 5: the multiline core behind in_tail's line loop */
/* linked with --wrap=flb_time_get: a group flushed before any time was registered takes "now" (flb_ml.c:1619-1624); the case names it */
int __real_flb_time_get(struct flb_time *tm);
static int g_now_set; static struct flb_time g_now;
int __wrap_flb_time_get(struct flb_time *tm) { if (g_now_set) { *tm = g_now; return 0; } return __real_flb_time_get(tm); }

struct ml_out { char *buf; size_t len, cap; int records; };
static int ml_flush_cb(struct flb_ml_parser *parser, struct flb_ml_stream *mst, void *data, char *buf_data, size_t buf_size)
{
    struct ml_out *o = data;
    (void) parser; (void) mst;
    if (o->len + buf_size > o->cap) { o->cap = (o->len + buf_size) * 2 + 4096; o->buf = realloc(o->buf, o->cap); }
    memcpy(o->buf + o->len, buf_data, buf_size);
    o->len += buf_size;
    o->records++;
    return 0;
}

static void run_multiline(struct flb_config *config, uint32_t nprops, char **keys, char **vals, char *data, uint64_t dlen)
{
    const char *type = "regex", *match_string = NULL, *key_content = NULL, *builtin = NULL;
    int negate = 0, skip_empty = 0, final_flush = 0;
    uint32_t i;
    struct flb_ml *ml;
    struct flb_ml_parser *mlp = NULL;
    struct flb_ml_parser_ins *mlp_i;
    struct ml_out out = {0};
    uint64_t stream_id = 0, off = 0;
    char *pend = NULL;
    size_t pend_len = 0;
    for (i = 0; i < nprops; i++) {
        if (!strcmp(keys[i], "type")) type = vals[i];
        else if (!strcmp(keys[i], "match_string")) match_string = vals[i];
        else if (!strcmp(keys[i], "negate")) negate = atoi(vals[i]);
        else if (!strcmp(keys[i], "key_content")) key_content = vals[i];
        else if (!strcmp(keys[i], "builtin")) builtin = vals[i];
        else if (!strcmp(keys[i], "skip_empty_lines")) skip_empty = atoi(vals[i]);
        else if (!strcmp(keys[i], "final_flush")) final_flush = atoi(vals[i]);
        else if (!strcmp(keys[i], "buffer_limit")) config->multiline_buffer_limit = vals[i];
        else if (!strcmp(keys[i], "now_sec")) { g_now_set = 1; g_now.tm.tv_sec = (time_t) strtoull(vals[i], NULL, 10); }
        else if (!strcmp(keys[i], "now_nsec")) { g_now_set = 1; g_now.tm.tv_nsec = (long) strtoull(vals[i], NULL, 10); }
    }
    if (builtin) {
        if (flb_ml_parser_builtin_create(config) != 0) { wr_answer(-100, NULL, 0); return; }
    }
    else {
        mlp = flb_ml_parser_create(config, "t", flb_ml_type_lookup((char *) type), (char *) match_string, negate, 0, (char *) key_content, NULL, NULL, NULL, NULL);
        if (!mlp) { wr_answer(-100, NULL, 0); return; }
        for (i = 0; i < nprops; i++) {
            char *a, *b;
            if (strcmp(keys[i], "rule")) continue;
            a = strchr(vals[i], 0x1f);
            b = a ? strchr(a + 1, 0x1f) : NULL;
            if (!b) { wr_answer(-100, NULL, 0); return; }
            *a = 0; *b = 0;
            if (flb_ml_rule_create(mlp, vals[i], a + 1, b[1] ? b + 1 : NULL, NULL) != 0) { wr_answer(-100, NULL, 0); return; }
        }
        if (flb_ml_parser_init(mlp) != 0) { wr_answer(-100, NULL, 0); return; }
    }
    ml = flb_ml_create(config, "t");
    if (builtin && strchr(builtin, ',')) {
        /* a list, as in_tail's `multiline.parser docker, cri` builds it (plugins/in_tail/tail_config.c: one instance per name, in order) */
        char *names = flb_strdup(builtin), *nm = names;
        while (ml && nm) {
            char *c = strchr(nm, ',');
            if (c) *c = 0;
            while (*nm == ' ') nm++;
            mlp_i = flb_ml_parser_instance_create(ml, nm);
            if (!mlp_i) break;
            if (key_content) flb_ml_parser_instance_set(mlp_i, "key_content", (char *) key_content);
            nm = c ? c + 1 : NULL;
        }
        flb_free(names);
    }
    else mlp_i = ml ? flb_ml_parser_instance_create(ml, (char *) (builtin ? builtin : "t")) : NULL;
    if (!mlp_i || flb_ml_stream_create(ml, "f", -1, ml_flush_cb, &out, &stream_id) != 0) { wr_answer(-100, NULL, 0); return; }
    if (builtin && key_content && !strchr(builtin, ',')) flb_ml_parser_instance_set(mlp_i, "key_content", (char *) key_content);
    while (off + 12 <= dlen) {
        uint32_t sec, nsec, len;
        struct flb_time tm;
        char *d, *end, *nl;
        memcpy(&sec, data + off, 4); memcpy(&nsec, data + off + 4, 4); memcpy(&len, data + off + 8, 4);
        off += 12;
        pend = realloc(pend, pend_len + len + 1);
        memcpy(pend + pend_len, data + off, len);
        pend_len += len; off += len;
        tm.tm.tv_sec = sec; tm.tm.tv_nsec = nsec;
        d = pend; end = pend + pend_len;
        while (d < end && *d == '\0') d++;                                       /* tail_file.c:783-786 */
        while (d < end && (nl = memchr(d, '\n', end - d))) {                     /* :840 */
            size_t ll = nl - d;
            int crlf = 0;
            if (skip_empty) {                                                    /* :863-874 */
                if (ll == 0) { d++; continue; }
                else if (ll == 1 && d[0] == '\r') { d += 2; continue; }
            }
            if (ll >= 2) crlf = (d[ll - 1] == '\r');                             /* :877-884 */
            flb_ml_append_text(ml, stream_id, &tm, d, ll - crlf);                /* :893-898 */
            d += ll + 1;
        }
        pend_len = end - d;
        memmove(pend, d, pend_len);
    }
    if (final_flush) flb_ml_flush_pending_now(ml);
    wr_answer(out.records, out.buf, out.len);
    free(out.buf); free(pend);
    g_now_set = 0;
}

int main(void)
{
    signal(SIGSEGV, on_segv);
    for (;;) {
        uint32_t kind, nprops, nparsers, iters = 0, i;
        char **keys, **vals;
        struct flb_config *config;
        uint64_t dlen;
        char *data;
        if (!rd(&kind, 4)) break;
        if (kind == 3 && !rd(&iters, 4)) break;
        if (!rd(&nprops, 4)) break;
        keys = calloc(nprops + 1, sizeof(char *)); vals = calloc(nprops + 1, sizeof(char *));
        for (i = 0; i < nprops; i++) { keys[i] = rd_str(); vals[i] = rd_str(); }
        if (!rd(&nparsers, 4)) break;
        config = make_config();
        for (i = 0; i < nparsers; i++) {
            char *f[9];
            int k, time_keep = 0, time_strict = 1;
            struct flb_parser_types *types = NULL;
            struct mk_list *decoders = NULL;
            int types_len = 0;
            for (k = 0; k < 9; k++) f[k] = rd_str();
            sscanf(f[8], "%d %d", &time_keep, &time_strict);
            if (f[6][0]) {
                /* Types: src/flb_parser.c:1130-1182 (parser_types_create is static there; same split) */
                struct mk_list *split = flb_utils_split(f[6], ' ', 256);
                struct mk_list *head;
                int cnt = mk_list_size(split), q = 0;
                types = flb_calloc(cnt, sizeof(struct flb_parser_types));
                mk_list_foreach(head, split) {
                    struct flb_split_entry *e = mk_list_entry(head, struct flb_split_entry, _head);
                    char *colon = strchr(e->value, ':');
                    const char *ty;
                    if (!colon) continue;
                    types[q].key = flb_strndup(e->value, colon - e->value);
                    types[q].key_len = (int) (colon - e->value);
                    ty = colon + 1;
                    if (!strcasecmp(ty, "integer")) types[q].type = FLB_PARSER_TYPE_INT;
                    else if (!strcasecmp(ty, "bool")) types[q].type = FLB_PARSER_TYPE_BOOL;
                    else if (!strcasecmp(ty, "float")) types[q].type = FLB_PARSER_TYPE_FLOAT;
                    else if (!strcasecmp(ty, "hex")) types[q].type = FLB_PARSER_TYPE_HEX;
                    else types[q].type = FLB_PARSER_TYPE_STRING;
                    q++;
                }
                types_len = q;
                flb_utils_split_free(split);
            }
            decoders = make_decoders(strchr(f[8], '|'));
            {
                /* "... tz=<IANA name>" / "... systz" in front of the first '|': the parser section's Time_Zone / Time_System_Timezone */
                char zone[256] = "";
                int systz = FLB_FALSE;
                char *bar = strchr(f[8], '|'), *z;
                if (bar) *bar = 0;
                z = strstr(f[8], " tz=");
                if (z) { sscanf(z + 4, "%255s", zone); }
                if (strstr(f[8], " systz")) systz = FLB_TRUE;
                if (bar) *bar = '|';
                flb_parser_create_with_time_zone(f[0], f[1], f[2][0] ? f[2] : NULL, atoi(f[7]), f[3][0] ? f[3] : NULL, f[4][0] ? f[4] : NULL,
                                                 f[5][0] ? f[5] : NULL, time_keep, time_strict, systz, zone[0] ? zone : NULL, FLB_FALSE,
                                                 types, types_len, decoders, config);
            }
        }
        if (!rd(&dlen, 8)) break;
        data = malloc(dlen + 1);
        if (dlen && !rd(data, dlen)) break;
        if (kind == 1 || kind == 2) {
            struct inst it;
            void *out = NULL;
            size_t out_size = 0;
            int ret;
            inst_open(&it, config, kind == 1 ? &filter_grep_plugin : &filter_parser_plugin, nprops, keys, vals);
            if (!it.ok) { wr_answer(-100, NULL, 0); fflush(stdout); continue; }
            ret = it.ins.p->cb_filter(data, dlen, "t", 1, &out, &out_size, &it.ins, NULL, it.ins.context, config);
            wr_answer(ret, out, ret == FLB_FILTER_MODIFIED ? out_size : 0);
        }
        else if (kind == 5) run_multiline(config, nprops, keys, vals, data, dlen);
        else if (kind == 6) run_log_to_metrics(config, nprops, keys, vals, data, dlen);
        else if (kind == 4) {
            const char *key = "log", *path_key = NULL, *path = "", *offset_key = NULL;
            uint64_t stream_offset = 0, processed = 0;
            int skip_empty = 1, lines = 0;
            struct flb_time tm;
            struct flb_log_event_encoder *enc = flb_log_event_encoder_create(FLB_LOG_EVENT_FORMAT_DEFAULT);
            const char *d = data, *end = data + dlen, *nl;
            tm.tm.tv_sec = 0; tm.tm.tv_nsec = 0;
            for (i = 0; i < nprops; i++) {
                if (!strcmp(keys[i], "key")) key = vals[i];
                else if (!strcmp(keys[i], "path_key")) path_key = vals[i];
                else if (!strcmp(keys[i], "path")) path = vals[i];
                else if (!strcmp(keys[i], "offset_key")) offset_key = vals[i];
                else if (!strcmp(keys[i], "stream_offset")) stream_offset = strtoull(vals[i], NULL, 10);
                else if (!strcmp(keys[i], "skip_empty_lines")) skip_empty = atoi(vals[i]);
                else if (!strcmp(keys[i], "sec")) tm.tm.tv_sec = (time_t) strtoull(vals[i], NULL, 10);
                else if (!strcmp(keys[i], "nsec")) tm.tm.tv_nsec = (long) strtoull(vals[i], NULL, 10);
            }
            while (d < end && *d == '\0') { d++; processed++; }                     /* :783-786 flb_skip_leading_zeros_simd */
            while (d < end && (nl = memchr(d, '\n', end - d))) {                    /* :840 */
                size_t len = nl - d, line_len;
                int crlf = 0, r;
                if (skip_empty) {                                                   /* :863-874 */
                    if (len == 0) { d++; processed++; continue; }
                    else if (len == 1 && d[0] == '\r') { d += 2; processed += 2; continue; }
                }
                if (len >= 2) crlf = (d[len - 1] == '\r');                          /* :877-884 */
                line_len = len - crlf;
                /* flb_tail_file_pack_line (:552-604), with the timestamp given instead of "now" */
                r = flb_log_event_encoder_begin_record(enc);
                if (r == FLB_EVENT_ENCODER_SUCCESS) r = flb_log_event_encoder_set_timestamp(enc, &tm);
                if (path_key && r == FLB_EVENT_ENCODER_SUCCESS)
                    r = flb_log_event_encoder_append_body_values(enc, FLB_LOG_EVENT_CSTRING_VALUE(path_key), FLB_LOG_EVENT_STRING_VALUE(path, strlen(path)));
                if (offset_key) {
                    if (r == FLB_EVENT_ENCODER_SUCCESS) r = flb_log_event_encoder_append_body_values(enc, FLB_LOG_EVENT_CSTRING_VALUE(offset_key));
                    if (r == FLB_EVENT_ENCODER_SUCCESS) r = flb_log_event_encoder_append_body_uint64(enc, (uint64_t) (stream_offset + processed));
                }
                if (r == FLB_EVENT_ENCODER_SUCCESS)
                    r = flb_log_event_encoder_append_body_values(enc, FLB_LOG_EVENT_CSTRING_VALUE(key), FLB_LOG_EVENT_STRING_VALUE(d, line_len));
                if (r == FLB_EVENT_ENCODER_SUCCESS) r = flb_log_event_encoder_commit_record(enc);
                d += len + 1; processed += len + 1; lines++;                        /* go_next :985-992 */
            }
            { int32_t ret = lines; uint64_t n = enc->output_length + 8; fwrite(&ret, 4, 1, stdout); fwrite(&n, 8, 1, stdout);
              if (enc->output_length) fwrite(enc->output_buffer, 1, enc->output_length, stdout); fwrite(&processed, 8, 1, stdout); }
            flb_log_event_encoder_destroy(enc);
        }
        else if (kind == 3) {
            /* parser (the properties up to the first "--") then grep (the rest): flb_filter_do's loop over the two, timed */
            struct inst ip, ig;
            uint32_t cut = 0, it_n;
            struct timespec t0, t1;
            uint64_t rin = 0, rkept = 0;
            double secs;
            char *first_out = NULL;            /* the pair's output of the first pass: returned behind the 24-byte header */
            size_t first_len = 0;
            while (cut < nprops && strcmp(keys[cut], "--") != 0) cut++;
            inst_open(&ip, config, &filter_parser_plugin, cut, keys, vals);
            inst_open(&ig, config, &filter_grep_plugin, nprops - cut - 1, keys + cut + 1, vals + cut + 1);
            if (!ip.ok || !ig.ok) { wr_answer(-100, NULL, 0); fflush(stdout); continue; }
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (it_n = 0; it_n < iters; it_n++) {
                void *o1 = NULL, *o2 = NULL;
                size_t s1 = 0, s2 = 0;
                int r1 = ip.ins.p->cb_filter(data, dlen, "t", 1, &o1, &s1, &ip.ins, NULL, ip.ins.context, config);
                const void *d2 = r1 == FLB_FILTER_MODIFIED ? o1 : data;
                size_t n2 = r1 == FLB_FILTER_MODIFIED ? s1 : dlen;
                int r2 = ig.ins.p->cb_filter(d2, n2, "t", 1, &o2, &s2, &ig.ins, NULL, ig.ins.context, config);
                if (it_n == 0) {
                    rin = (uint64_t) flb_mp_count(data, dlen);
                    rkept = (uint64_t) flb_mp_count(r2 == FLB_FILTER_MODIFIED ? o2 : d2, r2 == FLB_FILTER_MODIFIED ? s2 : n2);
                    first_len = r2 == FLB_FILTER_MODIFIED ? s2 : n2;
                    first_out = malloc(first_len ? first_len : 1);
                    memcpy(first_out, r2 == FLB_FILTER_MODIFIED ? o2 : d2, first_len);
                }
                if (r1 == FLB_FILTER_MODIFIED) flb_free(o1);
                if (r2 == FLB_FILTER_MODIFIED) flb_free(o2);
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            secs = (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
            { int32_t z = 0; uint64_t n = 24 + first_len; fwrite(&z, 4, 1, stdout); fwrite(&n, 8, 1, stdout); fwrite(&secs, 8, 1, stdout); fwrite(&rin, 8, 1, stdout); fwrite(&rkept, 8, 1, stdout);
              if (first_len) fwrite(first_out, 1, first_len, stdout); }
            free(first_out);
        }
        fflush(stdout);
        free(data);
    }
    return 0;
}
