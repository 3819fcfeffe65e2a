/*
 * plugin_host.c -- TEST INFRASTRUCTURE: the smallest "engine" that can load the GPU filter plugins the way
 * Fluent Bit does and drive their callbacks, so that flb-filter_<x>_gpu.so is linked, dlopen'd and executed
 * without the reference's build system.
 *
 * What it restates of the engine (paths in the fluent-bit tree):
 *   - dynamic loading, src/flb_plugin.c:110-168,194-320: the file is named flb-filter_<x>.so, the registration
 *     structure is the DATA symbol filter_<x>_plugin, which is copied and used as the instance's plugin;
 *   - an instance with its property list (struct flb_filter_instance, include/fluent-bit/flb_filter.h:84-121);
 *     flb_filter_get_property / flb_filter_set_context / flb_filter_name (src/flb_filter.c:747-775);
 *   - flb_config_map_set (src/flb_config_map.c): every map entry with set_property gets its property value,
 *     or its default, written at `offset` of the plugin context (STR / BOOL / INT / SIZE / TIME);
 *   - a parser registry for flb_parser_get (src/flb_parser.c) holding struct flb_parser entries with the fields
 *     the parsers file would give them (conf/parsers.conf);
 *   - the cb_init / cb_filter / cb_exit call sequence of flb_filter_init_all / flb_filter_do
 *     (src/flb_filter.c:121-325,620-720).
 * Everything else the engine offers is absent: this host exports exactly the symbols the plugins import.  For
 * filter_log_to_metrics_gpu (HOST_WITH_CMT: built when oracle/_ref/libcmetrics_ref.so, the reference's real cmetrics, is
 * there to link against) that is also the hidden emitter input and the scheduler timer the plugin creates in cb_init
 * (plugins/filter_log_to_metrics/log_to_metrics.c:852-966) -- stubbed: the emitter's flb_input_metrics_append PRINTS the
 * struct cmt it is handed (series in map order, label values, values / buckets / sum / count with %.17g), which is what the
 * test compares with the oracle; the timer callback is kept and fired by hand after cb_filter when a flush interval is set.  Built against the reference's headers by plugin/build.sh (in the build container); the
 * binary travels to the GPU box with the plugin objects.
 *
 * usage: plugin_host <plugin.so> <symbol> inspect
 *        plugin_host <plugin.so> <symbol> run <in.mp> <out.mp> [key=value ...] [--parser name|regex|time_fmt|time_key]
 *        plugin_host <plugin.so> <symbol> parser_do <in.txt> <out.mp> --parser name|regex|time_fmt|time_key
 *                    (calls the exported flb_parser_do_gpu with the registry's first parser on the file's bytes)
 */
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_filter.h>
#include <fluent-bit/flb_filter_plugin.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_config_map.h>
#include <fluent-bit/flb_kv.h>
#include <fluent-bit/flb_mem.h>
#include <fluent-bit/flb_parser.h>
#include <fluent-bit/flb_sds.h>
#include <fluent-bit/flb_time.h>
#ifdef HOST_WITH_CMT
#include <fluent-bit/flb_input.h>
#include <fluent-bit/flb_scheduler.h>
#include <cmetrics/cmetrics.h>
#include <cmetrics/cmt_map.h>
#include <cmetrics/cmt_metric.h>
#include <cmetrics/cmt_counter.h>
#include <cmetrics/cmt_gauge.h>
#include <cmetrics/cmt_histogram.h>
#endif
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <strings.h>

/* ---- symbols the plugins import from the engine ------------------------------------------------ */
static flb_sds_t host_sds(const char *s)
{
    size_t n = strlen(s);
    struct flb_sds *h = calloc(1, FLB_SDS_HEADER_SIZE + n + 1);
    h->len = n; h->alloc = n;
    memcpy(h->buf, s, n);
    return h->buf;
}

void flb_log_print(int type, const char *file, int line, const char *fmt, ...)
{
    va_list ap;
    (void) file; (void) line;
    fprintf(stderr, "[host log %d] ", type);
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
}
int flb_errno_print(int errnum, const char *file, int line)
{
    fprintf(stderr, "[host errno %d] %s:%d\n", errnum, file, line);
    return 0;
}
struct flb_worker *flb_worker_get(void) { return NULL; }
int flb_log_cache_check_suppress(struct flb_log_cache *cache, char *msg_buf, size_t msg_size)
{
    (void) cache; (void) msg_buf; (void) msg_size;
    return FLB_FALSE;
}
const char *flb_filter_name(struct flb_filter_instance *ins) { return ins->alias ? ins->alias : ins->name; }
void flb_filter_set_context(struct flb_filter_instance *ins, void *context) { ins->context = context; }
const char *flb_filter_get_property(const char *key, struct flb_filter_instance *ins)
{
    struct mk_list *head;
    struct flb_kv *kv;
    const char *val = NULL;
    mk_list_foreach(head, &ins->properties) {
        kv = mk_list_entry(head, struct flb_kv, _head);
        if (strcasecmp(kv->key, key) == 0) val = kv->val;      /* flb_kv_get_key_value: the last one wins? first: */
    }
    return val;
}
/* `map` is the plugin's raw struct flb_config_map[] (this host hands it over as ins->config_map) */
int flb_config_map_set(struct flb_config *config, struct mk_list *properties, struct mk_list *map, void *context)
{
    struct flb_config_map *m = (struct flb_config_map *) map;
    (void) config;
    for (; m->name != NULL || m->type != 0; m++) {
        const char *val = NULL;
        struct mk_list *head;
        struct flb_kv *kv;
        char *base = context;
        if (!m->set_property) continue;
        mk_list_foreach(head, properties) {
            kv = mk_list_entry(head, struct flb_kv, _head);
            if (strcasecmp(kv->key, m->name) == 0) val = kv->val;
        }
        if (!val) val = m->def_value;
        if (!val) continue;
        switch (m->type) {
        case FLB_CONFIG_MAP_STR: *(flb_sds_t *) (base + m->offset) = host_sds(val); break;
        case FLB_CONFIG_MAP_BOOL:
            *(int *) (base + m->offset) = (!strcasecmp(val, "true") || !strcasecmp(val, "on") || !strcasecmp(val, "yes") || !strcmp(val, "1"));
            break;
        case FLB_CONFIG_MAP_INT: *(int *) (base + m->offset) = atoi(val); break;
        case FLB_CONFIG_MAP_SIZE: *(size_t *) (base + m->offset) = (size_t) atoll(val); break;
        default: break;
        }
    }
    return 0;
}

/* parser registry (conf/parsers.conf entries, filled from the command line) */
static struct flb_parser *registry[16];
static int n_registry = 0;
struct flb_parser *flb_parser_get(const char *name, struct flb_config *config)
{
    int i;
    (void) config;
    for (i = 0; i < n_registry; i++) if (strcmp(registry[i]->name, name) == 0) return registry[i];
    return NULL;
}

static void add_parser(char *spec)
{
    /* name|regex|time_fmt|time_key : the fields flb_parser_create fills (src/flb_parser.c:805-1049), conf-file
     * defaults for the rest (skip_empty on, time_keep off, time_strict on, :1277-1304) */
    char *f[4] = {NULL, NULL, NULL, NULL};
    int i = 0;
    struct flb_parser *p = calloc(1, sizeof(*p));
    for (f[0] = spec; i < 3; ) {
        char *bar = strchr(f[i], '|');
        if (!bar) break;
        *bar = 0;
        f[++i] = bar + 1;
    }
    p->type = FLB_PARSER_REGEX;
    p->name = f[0];
    p->p_regex = f[1];
    p->skip_empty = FLB_TRUE;
    p->time_fmt_full = (f[2] && *f[2]) ? f[2] : NULL;
    p->time_key = (f[3] && *f[3]) ? f[3] : NULL;
    p->time_keep = FLB_FALSE;
    p->time_strict = FLB_TRUE;
    registry[n_registry++] = p;
}

#ifdef HOST_WITH_CMT
/* ---- what filter_log_to_metrics_gpu imports beyond the above: the emitter input, the scheduler, the metrics hand-off ---- */
static void (*host_timer_cb)(struct flb_config *, void *) = NULL;
static void *host_timer_data = NULL;
static int host_appends = 0;

int flb_input_name_exists(const char *name, struct flb_config *config) { (void) name; (void) config; return FLB_FALSE; }
struct flb_input_instance *flb_input_new(struct flb_config *config, const char *input, void *data, int public_only)
{
    (void) config; (void) data; (void) public_only;
    if (strcmp(input, "emitter") != 0) return NULL;
    return calloc(1, sizeof(struct flb_input_instance));
}
int flb_input_set_property(struct flb_input_instance *ins, const char *k, const char *v)
{
    (void) ins;
    printf("emitter property %s=%s\n", k, v);
    return 0;
}
int flb_input_instance_init(struct flb_input_instance *ins, struct flb_config *config) { (void) ins; (void) config; return 0; }
int flb_storage_input_create(struct cio_ctx *cio, struct flb_input_instance *in) { (void) cio; (void) in; return 0; }
struct flb_sched *flb_sched_ctx_get(void) { static long dummy; return (struct flb_sched *) &dummy; }
int flb_sched_timer_cb_create(struct flb_sched *sched, int type, int ms, void (*cb)(struct flb_config *, void *), void *data,
                              struct flb_sched_timer **out_timer)
{
    (void) sched; (void) type;
    printf("timer created ms=%d\n", ms);
    host_timer_cb = cb; host_timer_data = data;
    if (out_timer) *out_timer = (struct flb_sched_timer *) &host_timer_cb;
    return 0;
}
int flb_sched_timer_cb_destroy(struct flb_sched_timer *timer) { (void) timer; host_timer_cb = NULL; return 0; }

static void dump_map(const char *kind, struct cmt_map *map, int nbuckets, struct cmt_histogram_buckets *hb)
{
    struct cfl_list *head, *lh;
    int b;
    printf("%s %s_%s_%s labels=%d\n", kind, map->opts->ns, map->opts->subsystem, map->opts->name, map->label_count);
    if (map->metric_static_set) {
        struct cmt_metric *m = &map->metric;
        printf("  series");
        if (hb) {
            for (b = 0; b <= nbuckets; b++) printf(" b%d=%llu", b, (unsigned long long) cmt_metric_hist_get_value(m, b));
            printf(" count=%llu sum=%.17g\n", (unsigned long long) cmt_metric_hist_get_count_value(m), cmt_metric_hist_get_sum_value(m));
        }
        else printf(" value=%.17g\n", cmt_metric_get_value(m));
    }
    cfl_list_foreach(head, &map->metrics) {
        struct cmt_metric *m = cfl_list_entry(head, struct cmt_metric, _head);
        printf("  series");
        cfl_list_foreach(lh, &m->labels) {
            struct cmt_map_label *l = cfl_list_entry(lh, struct cmt_map_label, _head);
            printf(" [%s]", l->name);
        }
        if (hb) {
            for (b = 0; b <= nbuckets; b++) printf(" b%d=%llu", b, (unsigned long long) cmt_metric_hist_get_value(m, b));
            printf(" count=%llu sum=%.17g\n", (unsigned long long) cmt_metric_hist_get_count_value(m), cmt_metric_hist_get_sum_value(m));
        }
        else printf(" value=%.17g\n", cmt_metric_get_value(m));
    }
}
/* the emitter's entry (src/flb_input_metric.c): here the hand-off point -- print the context the plugin filled */
int flb_input_metrics_append(struct flb_input_instance *ins, const char *tag, size_t tag_len, struct cmt *cmt)
{
    struct cfl_list *head;
    (void) ins;
    printf("metrics_append %d tag=%.*s\n", ++host_appends, (int) tag_len, tag);
    cfl_list_foreach(head, &cmt->counters) {
        struct cmt_counter *c = cfl_list_entry(head, struct cmt_counter, _head);
        dump_map("counter", c->map, 0, NULL);
    }
    cfl_list_foreach(head, &cmt->gauges) {
        struct cmt_gauge *g = cfl_list_entry(head, struct cmt_gauge, _head);
        dump_map("gauge", g->map, 0, NULL);
    }
    cfl_list_foreach(head, &cmt->histograms) {
        struct cmt_histogram *h = cfl_list_entry(head, struct cmt_histogram, _head);
        size_t i;
        printf("bounds");
        for (i = 0; i < h->buckets->count; i++) printf(" %.17g", h->buckets->upper_bounds[i]);
        printf("\n");
        dump_map("histogram", h->map, (int) h->buckets->count, h->buckets);
    }
    return 0;
}
#endif /* HOST_WITH_CMT */

/* ---- the engine's side of the plugin contract ---------------------------------------------------- */
int main(int argc, char **argv)
{
    void *h;
    struct flb_filter_plugin *sym, *p;
    struct flb_filter_instance *ins;
    struct flb_config_map *m;
    struct flb_config *config;
    FILE *fp;
    char *in;
    long bytes;
    void *out = NULL;
    size_t out_size = 0;
    int i, ret;

    if (argc < 4) { fprintf(stderr, "usage: plugin_host <so> <symbol> inspect|run ...\n"); return 2; }
    h = dlopen(argv[1], RTLD_LAZY | RTLD_GLOBAL);                          /* src/flb_plugin.c:236 */
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    sym = dlsym(h, argv[2]);                                               /* the data symbol filter_<x>_plugin */
    if (!sym) { fprintf(stderr, "dlsym: %s\n", dlerror()); return 4; }
    p = malloc(sizeof(*p));
    memcpy(p, sym, sizeof(*p));                                            /* src/flb_plugin.c:258-268 */
    printf("name=%s\ndescription=%s\ncb_init=%d cb_filter=%d cb_exit=%d\n", p->name, p->description,
           p->cb_init != NULL, p->cb_filter != NULL, p->cb_exit != NULL);
    for (m = p->config_map; m && (m->name != NULL || m->type != 0); m++)
        printf("config_map %s type=%d mult=%d set=%d default=%s\n", m->name, m->type, (m->flags & FLB_CONFIG_MAP_MULT) != 0, m->set_property,
               m->def_value ? m->def_value : "-");
    if (strcmp(argv[3], "inspect") == 0) return 0;
    if (argc < 6) return 2;

    config = calloc(1, sizeof(*config) + 4096);
    ins = calloc(1, sizeof(*ins));
    ins->p = p;
    ins->log_level = 3;
    ins->config = config;
    snprintf(ins->name, sizeof(ins->name), "%s.0", p->name);
    mk_list_init(&ins->properties);
    ins->config_map = (struct mk_list *) p->config_map;
    for (i = 6; i < argc; i++) {
        if (strcmp(argv[i], "--parser") == 0 && i + 1 < argc) { add_parser(argv[++i]); continue; }
        {
            char *eq = strchr(argv[i], '=');
            struct flb_kv *kv;
            if (!eq) continue;
            *eq = 0;
            kv = calloc(1, sizeof(*kv));
            kv->key = host_sds(argv[i]);
            kv->val = host_sds(eq + 1);
            mk_list_add(&kv->_head, &ins->properties);                     /* configuration order */
        }
    }
    if (strcmp(argv[3], "parser_do") == 0) {
        int (*pdo)(struct flb_parser *, const char *, size_t, void **, size_t *, struct flb_time *) = dlsym(h, "flb_parser_do_gpu");
        struct flb_time tm;
        if (!pdo || n_registry == 0) return 6;
        fp = fopen(argv[4], "rb");
        if (!fp) return 5;
        fseek(fp, 0, SEEK_END); bytes = ftell(fp); fseek(fp, 0, SEEK_SET);
        in = malloc(bytes ? bytes : 1);
        if (fread(in, 1, bytes, fp) != (size_t) bytes) return 5;
        fclose(fp);
        flb_time_zero(&tm);
        ret = pdo(registry[0], in, bytes, &out, &out_size, &tm);
        printf("flb_parser_do=%d out_size=%zu sec=%lld nsec=%ld\n", ret, ret >= 0 ? out_size : 0, (long long) tm.tm.tv_sec, (long) tm.tm.tv_nsec);
        fp = fopen(argv[5], "wb");
        if (ret >= 0 && out_size) fwrite(out, 1, out_size, fp);
        fclose(fp);
        if (ret >= 0) flb_free(out);
        return 0;
    }
    ret = p->cb_init(ins, config, NULL);                                   /* src/flb_filter.c:684-697 */
    printf("cb_init=%d\n", ret);
    if (ret != 0) return 10;
    fp = fopen(argv[4], "rb");
    if (!fp) return 5;
    fseek(fp, 0, SEEK_END); bytes = ftell(fp); fseek(fp, 0, SEEK_SET);
    in = malloc(bytes ? bytes : 1);
    if (fread(in, 1, bytes, fp) != (size_t) bytes) return 5;
    fclose(fp);
    ret = p->cb_filter(in, bytes, "test", 4, &out, &out_size, ins, NULL, ins->context, config);   /* src/flb_filter.c:194-211 */
    printf("cb_filter=%d out_size=%zu\n", ret, out_size);
#ifdef HOST_WITH_CMT
    if (host_timer_cb) {                                                   /* the flush timer fires (log_to_metrics.c:947-966) */
        host_timer_cb(config, host_timer_data);
        host_timer_cb(config, host_timer_data);                            /* a second tick without new data appends nothing */
    }
#endif
    fp = fopen(argv[5], "wb");
    if (ret == FLB_FILTER_MODIFIED && out_size) fwrite(out, 1, out_size, fp);
    fclose(fp);
    if (ret == FLB_FILTER_MODIFIED) flb_free(out);                         /* the engine releases it with flb_free (:235-237) */
    ret = p->cb_exit(ins->context, config);                                /* src/flb_filter.c flb_filter_exit */
    printf("cb_exit=%d\n", ret);
    return 0;
}
