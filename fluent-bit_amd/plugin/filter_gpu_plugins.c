/*
 * filter_gpu_plugins.c -- the reference-side binding: Fluent Bit filter plugins whose callbacks
 * forward to libflbgpu.so (include/flb_gpu.h).
 *
 * Built INSIDE a fluent-bit source tree / against its headers (it needs the generated
 * fluent-bit/flb_info.h), not as part of libflbgpu.so.  Two ways to load it:
 *   - as built-ins replacing the CPU plugins: add this directory to plugins/CMakeLists.txt and
 *     configure with -DFLB_FILTER_GREP=Off -DFLB_FILTER_PARSER=Off, keeping .name = "grep" /
 *     "parser" below so existing configuration files load unchanged;
 *   - as dynamic plugins: build two shared objects named flb-filter_grep_gpu.so /
 *     flb-filter_parser_gpu.so exporting filter_grep_gpu_plugin / filter_parser_gpu_plugin
 *     (src/flb_plugin.c:110-168,194-320) and start `fluent-bit -e <path>.so`; the filters are then
 *     selected with `Name grep_gpu` / `Name parser_gpu`.
 *
 * The property names, their meaning, the return codes, buffer ownership (flb_free == free of a
 * malloc'd buffer) and the error behaviour (log + FLB_FILTER_NOTOUCH, cb_init -1) are the
 * reference's: plugins/filter_grep/grep.c:196-434, plugins/filter_parser/filter_parser.c:96-499.
 */
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_filter.h>
#include <fluent-bit/flb_filter_plugin.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_kv.h>
#include <fluent-bit/flb_mem.h>
#include <fluent-bit/flb_parser.h>
#include <fluent-bit/flb_str.h>

#include <flb_gpu.h>

#ifndef FLBGPU_PLUGIN_SUFFIX
#define FLBGPU_PLUGIN_SUFFIX "_gpu"
#endif

/* One source, three shared objects (plugin/Makefile): FLBGPU_ONLY = 1 grep, 2 parser, 3 log_to_metrics keeps one
 * plugin per object, so that flb-filter_grep_gpu.so does not drag in the cmetrics / emitter symbols
 * log_to_metrics needs from the engine.  Undefined: all three (the built-in / static build). */
#if !defined(FLBGPU_ONLY) || FLBGPU_ONLY == 1
#define FLBGPU_WITH_GREP 1
#endif
#if !defined(FLBGPU_ONLY) || FLBGPU_ONLY == 2
#define FLBGPU_WITH_PARSER 1
#endif
#if !defined(FLBGPU_ONLY) || FLBGPU_ONLY == 3
#define FLBGPU_WITH_L2M 1
#endif

static int gpu_ready = 0;

/* per instance (no process-wide statics shared by instances and threads; one call at a time per instance,
 * SURVEY 8b "Threading"): what has been said already about the regex corners and the host matcher's limits */
struct gpu_watch {
    unsigned long calls;
    int corner_warned;
    uint64_t host_over;        /* flbgpu_filter_host_rules [2]: searches ended by the backtrack budget, last seen */
    uint64_t host_unhandled;   /* [3]: records a host parser did not take, last seen */
};

static int ensure_gpu(struct flb_filter_instance *ins)
{
    if (gpu_ready) {
        return 0;
    }
    if (flbgpu_init(0) != 0) {
        flb_plg_error(ins, "%s", flbgpu_last_error());
        return -1;
    }
    gpu_ready = 1;
    return 0;
}

/* ------------------------------------------------------------------ grep */
struct grep_gpu_ctx {
    flbgpu_filter *f;
    struct flb_filter_instance *ins;
    struct gpu_watch w;               /* f, ins, w: the head every context of this file starts with */
};


/* a pattern that is not a regular expression (look-around, back-references, ...) does not fail the filter: the host's backtracking
 * matcher answers it (include/flb_gpu.h flbgpu_filter_host_rules) -- a slow path, said at start-up */
static void gpu_note_host_rules(struct flb_filter_instance *f_ins, flbgpu_filter *f)
{
    uint64_t hr[4] = {0, 0, 0, 0};
    if (flbgpu_filter_host_rules(f, hr) == 0 && hr[0] > 0) {
        flb_plg_warn(f_ins, "%llu pattern(s) of this filter are not regular expressions (look-around, atomic groups, back-references): "
                     "their values are searched on the host, everything else stays on the GPU", (unsigned long long) hr[0]);
    }
}

/* After a run: what must never be silent.  (1) the two regex corners in which the reference's own answer depends on its search
 * optimizer (flb_gpu.h flbgpu_filter_regex_corners): said once per instance; looked at every 256 calls (a small device read).
 * (2) the host matcher's own limits (csrc/rxbt.inc: a backtrack budget the reference does not have; records a host parser did not
 * take): a search that ended on the budget was answered "no match", which is NOT what Onigmo would say -- warned every time the
 * counters grow (host counters: no device read). */
static void gpu_watch_after_run(struct flb_filter_instance *f_ins, flbgpu_filter *f, struct gpu_watch *w)
{
    uint64_t hr[4] = {0, 0, 0, 0};
    if (!w->corner_warned && (++w->calls & 255) == 1 && flbgpu_filter_regex_corners(f) > 0) {
        w->corner_warned = 1;
        flb_plg_warn(f_ins, "%llu value(s) held ill-formed UTF-8 behind a line / word anchor or a case-fold character whose UTF-8 "
                     "length differs: Onigmo's answer there depends on its search optimizer, the GPU path answered leftmost-first",
                     (unsigned long long) flbgpu_filter_regex_corners(f));
    }
    if (flbgpu_filter_host_rules(f, hr) == 0 && hr[0] > 0) {
        if (hr[2] > w->host_over) {
            flb_plg_warn(f_ins, "%llu search(es) of a pattern that is not a regular expression spent the host matcher's backtrack "
                         "budget and were answered 'no match' (Onigmo has no such budget: these records may be filtered differently)",
                         (unsigned long long) (hr[2] - w->host_over));
            w->host_over = hr[2];
        }
        if (hr[3] > w->host_unhandled) {
            flb_plg_warn(f_ins, "%llu record(s) were passed on unparsed by a parser that is not a regular expression",
                         (unsigned long long) (hr[3] - w->host_unhandled));
            w->host_unhandled = hr[3];
        }
    }
}

#ifdef FLBGPU_WITH_GREP
static int cb_grep_gpu_init(struct flb_filter_instance *f_ins, struct flb_config *config, void *data)
{
    int n = 0;
    int i = 0;
    const char **kinds;
    const char **vals;
    const char *op;
    struct mk_list *head;
    struct flb_kv *kv;
    struct grep_gpu_ctx *ctx;
    (void) config;
    (void) data;

    if (ensure_gpu(f_ins) != 0) {
        return -1;
    }
    /* rules are read by walking the instance properties in configuration order, exactly as
     * set_rules() does (plugins/filter_grep/grep.c:67-88) */
    mk_list_foreach(head, &f_ins->properties) {
        n++;
    }
    kinds = flb_calloc(n ? n : 1, sizeof(char *));
    vals = flb_calloc(n ? n : 1, sizeof(char *));
    if (!kinds || !vals) {
        flb_errno();
        return -1;
    }
    mk_list_foreach(head, &f_ins->properties) {
        kv = mk_list_entry(head, struct flb_kv, _head);
        if (strcasecmp(kv->key, "regex") != 0 && strcasecmp(kv->key, "exclude") != 0) {
            continue;
        }
        kinds[i] = kv->key;
        vals[i] = kv->val;
        i++;
    }
    op = flb_filter_get_property("logical_op", f_ins);

    ctx = flb_calloc(1, sizeof(struct grep_gpu_ctx));
    if (!ctx) {
        flb_errno();
        flb_free(kinds);
        flb_free(vals);
        return -1;
    }
    ctx->ins = f_ins;
    ctx->f = flbgpu_filter_grep_create(i, kinds, vals, op);
    flb_free(kinds);
    flb_free(vals);
    if (!ctx->f) {
        flb_plg_error(f_ins, "%s", flbgpu_last_error());
        flb_free(ctx);
        return -1;
    }
    gpu_note_host_rules(f_ins, ctx->f);
    flb_filter_set_context(f_ins, ctx);
    return 0;
}

#endif /* FLBGPU_WITH_GREP */

static int cb_gpu_filter(const void *data, size_t bytes, const char *tag, int tag_len,
                         void **out_buf, size_t *out_size,
                         struct flb_filter_instance *f_ins, struct flb_input_instance *i_ins,
                         void *context, struct flb_config *config)
{
    /* grep_gpu_ctx and parser_gpu_ctx share their first member */
    struct grep_gpu_ctx *ctx = context;
    int ret;
    (void) tag;
    (void) tag_len;
    (void) i_ins;
    (void) config;
    /* FLBGPU_FILTER_MODIFIED/NOTOUCH == FLB_FILTER_MODIFIED/NOTOUCH; the output buffer is
     * malloc'd, the engine releases it with flb_free (src/flb_filter.c:235-237) */
    ret = flbgpu_filter_run(ctx->f, data, bytes, out_buf, out_size);
    gpu_watch_after_run(f_ins, ctx->f, &ctx->w);
    return ret;
}

static int cb_gpu_exit(void *data, struct flb_config *config)
{
    struct grep_gpu_ctx *ctx = data;
    (void) config;
    if (!ctx) {
        return 0;
    }
    flbgpu_filter_destroy(ctx->f);
    flb_free(ctx);
    return 0;
}

#ifdef FLBGPU_WITH_GREP
static struct flb_config_map grep_config_map[] = {
    { FLB_CONFIG_MAP_STR, "regex", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0,
      "Keep records in which the content of KEY matches the regular expression." },
    { FLB_CONFIG_MAP_STR, "exclude", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0,
      "Exclude records in which the content of KEY matches the regular expression." },
    { FLB_CONFIG_MAP_STR, "logical_op", "legacy", 0, FLB_FALSE, 0,
      "legacy, AND or OR." },
    {0}
};

struct flb_filter_plugin filter_grep_gpu_plugin = {
    .name         = "grep" FLBGPU_PLUGIN_SUFFIX,
    .description  = "grep events by specified field values (MI355X)",
    .cb_init      = cb_grep_gpu_init,
    .cb_filter    = cb_gpu_filter,
    .cb_exit      = cb_gpu_exit,
    .config_map   = grep_config_map,
    .flags        = 0
};

#endif /* FLBGPU_WITH_GREP */

/* ------------------------------------------------------------------ parser */
#ifdef FLBGPU_WITH_PARSER
#define MAX_GPU_PARSERS 16

struct parser_gpu_ctx {
    flbgpu_filter *f;                       /* must stay first (cb_gpu_filter) */
    struct flb_filter_instance *ins;
    struct gpu_watch w;
    flbgpu_parser *parsers[MAX_GPU_PARSERS];
    int n_parsers;
    flb_sds_t key_name;
    int reserve_data;
    int preserve_key;
};

/* "key:type key:type" from struct flb_parser_types (src/flb_parser.c:1130-1182 in reverse) */
static char *types_to_str(struct flb_parser *p)
{
    int i;
    size_t len = 1;
    char *out;
    static const char *tn[] = { "", "integer", "float", "bool", "string", "hex" };

    for (i = 0; i < p->types_len; i++) {
        if (p->types[i].key) {
            len += p->types[i].key_len + 10;
        }
    }
    out = flb_calloc(1, len);
    if (!out) {
        return NULL;
    }
    for (i = 0; i < p->types_len; i++) {
        if (!p->types[i].key) {
            continue;
        }
        strncat(out, p->types[i].key, p->types[i].key_len);
        strcat(out, ":");
        strcat(out, tn[p->types[i].type]);
        strcat(out, " ");
    }
    return out;
}

/* the parser's Decode_Field / Decode_Field_As rules (struct flb_parser.decoders: one struct flb_parser_dec per key with its rules
 * in configuration order, src/flb_parser_decoder.c:593-776) handed to the device parser in the same order */
#include <fluent-bit/flb_parser_decoder.h>
static char gpu_shim_err[256];      /* what add_decoders refused (flbgpu_last_error would show an older message) */

static int add_decoders(flbgpu_parser *g, struct flb_parser *p)
{
    static const char *backends[] = { "json", "escaped", "escaped_utf8", "mysql_quoted" };
    struct mk_list *head;
    struct mk_list *r_head;
    struct flb_parser_dec *dec;
    struct flb_parser_dec_rule *rule;

    if (!p->decoders) {
        return 0;
    }
    mk_list_foreach(head, p->decoders) {
        dec = mk_list_entry(head, struct flb_parser_dec, _head);
        mk_list_foreach(r_head, &dec->rules) {
            rule = mk_list_entry(r_head, struct flb_parser_dec_rule, _head);
            if (rule->backend < 0 || rule->backend > 3) {
                snprintf(gpu_shim_err, sizeof(gpu_shim_err), "parser '%s': decoder backend %d of key '%s' is not one of json / escaped / "
                         "escaped_utf8 / mysql_quoted", p->name ? p->name : "", rule->backend, dec->key ? dec->key : "");
                return -2;
            }
            if (flbgpu_parser_add_decoder(g, rule->type == FLB_PARSER_DEC_AS, backends[rule->backend], dec->key,
                                          rule->action == FLB_PARSER_ACT_TRY_NEXT ? "try_next" :
                                          rule->action == FLB_PARSER_ACT_DO_NEXT ? "do_next" : NULL) != 0) {
                return -1;
            }
        }
    }
    return 0;
}

static int cb_parser_gpu_init(struct flb_filter_instance *f_ins, struct flb_config *config, void *data)
{
    char *types;
    char off[16];
    int ret;
    struct mk_list *head;
    struct flb_kv *kv;
    struct flb_parser *p;
    struct parser_gpu_ctx *ctx;
    (void) data;

    if (ensure_gpu(f_ins) != 0) {
        return -1;
    }
    ctx = flb_calloc(1, sizeof(struct parser_gpu_ctx));
    if (!ctx) {
        flb_errno();
        return -1;
    }
    ctx->ins = f_ins;
    if (flb_filter_config_map_set(f_ins, ctx) < 0) {
        flb_plg_error(f_ins, "configuration error");
        flb_free(ctx);
        return -1;
    }
    if (ctx->key_name == NULL) {
        flb_plg_error(f_ins, "missing 'key_name'");
        flb_free(ctx);
        return -1;
    }
    /* the parsers themselves stay in the engine's registry (flb_parser_get): the GPU twin is
     * created from the same struct flb_parser fields (include/fluent-bit/flb_parser.h:41-70) */
    mk_list_foreach(head, &f_ins->properties) {
        kv = mk_list_entry(head, struct flb_kv, _head);
        if (strcasecmp("parser", kv->key) != 0) {
            continue;
        }
        p = flb_parser_get(kv->val, config);
        if (!p) {
            flb_plg_error(f_ins, "requested parser '%s' not found", kv->val);
            continue;
        }
        if (p->type != FLB_PARSER_REGEX && p->type != FLB_PARSER_JSON && p->type != FLB_PARSER_LOGFMT &&
            p->type != FLB_PARSER_LTSV) {
            flb_plg_error(f_ins, "parser '%s': only Format regex / json / logfmt / ltsv is on the GPU path", kv->val);
            goto error;
        }
        if (ctx->n_parsers >= MAX_GPU_PARSERS) {
            goto error;
        }
        types = types_to_str(p);
        snprintf(off, sizeof(off), "%c%02d%02d", p->time_offset < 0 ? '-' : '+',
                 abs(p->time_offset) / 3600, (abs(p->time_offset) / 60) % 60);
        if (p->type == FLB_PARSER_JSON) {
            ctx->parsers[ctx->n_parsers] =
                flbgpu_parser_create_json(p->name, p->time_fmt_full, p->time_key, p->time_offset ? off : NULL,
                                          p->time_keep, p->time_strict);
        }
        else if (p->type == FLB_PARSER_LOGFMT || p->type == FLB_PARSER_LTSV) {
            ctx->parsers[ctx->n_parsers] =
                flbgpu_parser_create_kv(p->name, p->type == FLB_PARSER_LOGFMT ? "logfmt" : "ltsv", p->time_fmt_full,
                                        p->time_key, p->time_offset ? off : NULL, p->time_keep, p->time_strict,
                                        p->logfmt_no_bare_keys, types);
        }
        else {
            ctx->parsers[ctx->n_parsers] =
                flbgpu_parser_create(p->name, p->p_regex, p->skip_empty, p->time_fmt_full, p->time_key,
                                     p->time_offset ? off : NULL, p->time_keep, p->time_strict, types);
        }
        flb_free(types);
        if (!ctx->parsers[ctx->n_parsers]) {
            flb_plg_error(f_ins, "%s", flbgpu_last_error());
            goto error;
        }
        /* Time_System_Timezone / Time_Zone (src/flb_parser.c:986-1022): the twin reads the zone's file the way the parser did */
        if (flbgpu_parser_set_system_timezone(ctx->parsers[ctx->n_parsers], p->time_system_timezone) != 0 ||
            flbgpu_parser_set_time_zone(ctx->parsers[ctx->n_parsers], p->time_zone) != 0) {
            flb_plg_error(f_ins, "%s", flbgpu_last_error());
            ctx->n_parsers++;
            goto error;
        }
        ret = add_decoders(ctx->parsers[ctx->n_parsers], p);
        if (ret != 0) {
            flb_plg_error(f_ins, "%s", ret == -2 ? gpu_shim_err : flbgpu_last_error());
            ctx->n_parsers++;
            goto error;
        }
        ctx->n_parsers++;
    }
    if (ctx->n_parsers == 0) {
        flb_plg_error(f_ins, "Invalid 'parser'");
        goto error;
    }
    ctx->f = flbgpu_filter_parser_create(ctx->key_name, ctx->reserve_data, ctx->preserve_key,
                                         ctx->n_parsers, ctx->parsers);
    if (!ctx->f) {
        flb_plg_error(f_ins, "%s", flbgpu_last_error());
        goto error;
    }
    gpu_note_host_rules(f_ins, ctx->f);
    flb_filter_set_context(f_ins, ctx);
    return 0;

error:
    while (ctx->n_parsers > 0) {
        flbgpu_parser_destroy(ctx->parsers[--ctx->n_parsers]);
    }
    flb_free(ctx);
    return -1;
}

static int cb_parser_gpu_exit(void *data, struct flb_config *config)
{
    struct parser_gpu_ctx *ctx = data;
    (void) config;
    if (!ctx) {
        return 0;
    }
    flbgpu_filter_destroy(ctx->f);
    while (ctx->n_parsers > 0) {
        flbgpu_parser_destroy(ctx->parsers[--ctx->n_parsers]);
    }
    flb_free(ctx);
    return 0;
}

static struct flb_config_map parser_config_map[] = {
    { FLB_CONFIG_MAP_STR, "Key_Name", NULL, 0, FLB_TRUE, offsetof(struct parser_gpu_ctx, key_name),
      "Specify field name in record to parse." },
    { FLB_CONFIG_MAP_STR, "Parser", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0,
      "Specify the parser name to interpret the field (repeatable)." },
    { FLB_CONFIG_MAP_BOOL, "Preserve_Key", "false", 0, FLB_TRUE, offsetof(struct parser_gpu_ctx, preserve_key),
      "Keep original Key_Name field in the parsed result." },
    { FLB_CONFIG_MAP_BOOL, "Reserve_Data", "false", 0, FLB_TRUE, offsetof(struct parser_gpu_ctx, reserve_data),
      "Keep all other original fields in the parsed result." },
    { FLB_CONFIG_MAP_DEPRECATED, "Unescape_key", NULL, 0, FLB_FALSE, 0, "(deprecated)" },
    {0}
};

struct flb_filter_plugin filter_parser_gpu_plugin = {
    .name         = "parser" FLBGPU_PLUGIN_SUFFIX,
    .description  = "Parse events (MI355X)",
    .cb_init      = cb_parser_gpu_init,
    .cb_filter    = cb_gpu_filter,
    .cb_exit      = cb_parser_gpu_exit,
    .config_map   = parser_config_map,
    .flags        = 0
};

/* flb_parser_do() on the GPU path: same signature and results as the reference's
 * (include/fluent-bit/flb_parser.h:149, src/flb_parser.c:1784: last byte consumed or -1, *out_buf a msgpack map
 * released with flb_free, *out_time the parsed time or zero).  The GPU twin of a struct flb_parser is created on
 * first use and kept in a small table keyed by the parser's address. */
#include <fluent-bit/flb_time.h>
#include <pthread.h>
/*
 * The twins are keyed by the parser's CONFIGURATION (name, regex, time settings, types), not by the address of the
 * struct flb_parser: after a hot reload a new parser may live where a destroyed one did.  The table is guarded by a
 * mutex, every twin by its own (flb_parser_do is re-entrant in the reference; a flbgpu_parser owns one stream and one
 * set of device buffers), and a full table evicts its least recently used idle entry instead of failing.
 */
#define MAX_TWINS 64
struct parser_twin {
    char *sig;
    flbgpu_parser *g;
    pthread_mutex_t lock;
    unsigned long stamp;
    int users;
};
static struct parser_twin twins[MAX_TWINS];
static int n_twins = 0;
static unsigned long twin_clock = 0;
static pthread_mutex_t twins_mu = PTHREAD_MUTEX_INITIALIZER;

/* The signature (every option, as text) is what makes two parsers the same twin -- built with malloc + snprintf.  A call for the parser
 * a thread used last skips it: the parser's address, a fingerprint of the same options computed without allocating (FNV-1a over the
 * strings, the numbers, the Types entries and the decoder rules) and the generation of the twin table (bumped whenever a slot is given
 * to another parser) say the slot it found then is still the one. */
static __thread struct { const struct flb_parser *p; uint64_t fp; unsigned gen; int slot; } twin_hit = { NULL, 0, 0, -1 };
static unsigned twin_gen = 1;

static uint64_t fnv_str(uint64_t h, const char *s)
{
    if (!s) {
        return (h ^ 0xfe) * 0x100000001b3ULL;
    }
    while (*s) {
        h = (h ^ (unsigned char) *s++) * 0x100000001b3ULL;
    }
    return (h ^ 0xff) * 0x100000001b3ULL;
}

static uint64_t fnv_int(uint64_t h, long v)
{
    int i;

    for (i = 0; i < 8; i++) {
        h = (h ^ (unsigned char) (v >> (8 * i))) * 0x100000001b3ULL;
    }
    return h;
}

static uint64_t twin_fingerprint(struct flb_parser *parser)
{
    int i;
    uint64_t h = 0xcbf29ce484222325ULL;
    struct mk_list *head;
    struct mk_list *r_head;
    struct flb_parser_dec *dec;
    struct flb_parser_dec_rule *rule;

    h = fnv_str(h, parser->name);
    h = fnv_str(h, parser->p_regex);
    h = fnv_str(h, parser->time_fmt_full);
    h = fnv_str(h, parser->time_key);
    h = fnv_int(h, parser->skip_empty);
    h = fnv_int(h, parser->time_offset);
    h = fnv_int(h, parser->time_keep);
    h = fnv_int(h, parser->time_strict);
    for (i = 0; i < parser->types_len; i++) {
        h = fnv_str(h, parser->types[i].key);
        h = fnv_int(h, parser->types[i].type);
    }
    if (parser->decoders) {
        mk_list_foreach(head, parser->decoders) {
            dec = mk_list_entry(head, struct flb_parser_dec, _head);
            mk_list_foreach(r_head, &dec->rules) {
                rule = mk_list_entry(r_head, struct flb_parser_dec_rule, _head);
                h = fnv_str(h, dec->key);
                h = fnv_int(h, rule->type * 10000 + rule->backend * 100 + rule->action);
            }
        }
    }
    return h;
}

static char *twin_signature(struct flb_parser *parser, const char *types)
{
    size_t len;
    char *sig;

    size_t at;
    struct mk_list *head;
    struct mk_list *r_head;
    struct flb_parser_dec *dec;
    struct flb_parser_dec_rule *rule;

    len = 96 + (parser->name ? strlen(parser->name) : 0) + (parser->p_regex ? strlen(parser->p_regex) : 0) +
          (parser->time_fmt_full ? strlen(parser->time_fmt_full) : 0) + (parser->time_key ? strlen(parser->time_key) : 0) +
          (types ? strlen(types) : 0);
    if (parser->decoders) {
        mk_list_foreach(head, parser->decoders) {
            dec = mk_list_entry(head, struct flb_parser_dec, _head);
            mk_list_foreach(r_head, &dec->rules) {
                len += strlen(dec->key) + 16;
            }
        }
    }
    sig = flb_malloc(len);
    if (!sig) {
        return NULL;
    }
    snprintf(sig, len, "%s\x01%s\x01%d\x01%s\x01%s\x01%d\x01%d\x01%d\x01%s",
             parser->name ? parser->name : "", parser->p_regex ? parser->p_regex : "", parser->skip_empty,
             parser->time_fmt_full ? parser->time_fmt_full : "\x02", parser->time_key ? parser->time_key : "\x02",
             parser->time_offset, parser->time_keep, parser->time_strict, types ? types : "");
    if (parser->decoders) {
        mk_list_foreach(head, parser->decoders) {
            dec = mk_list_entry(head, struct flb_parser_dec, _head);
            mk_list_foreach(r_head, &dec->rules) {
                rule = mk_list_entry(r_head, struct flb_parser_dec_rule, _head);
                at = strlen(sig);
                snprintf(sig + at, len - at, "\x01%s:%d%d%d", dec->key, rule->type, rule->backend, rule->action);
            }
        }
    }
    return sig;
}

int flb_parser_do_gpu(struct flb_parser *parser, const char *buf, size_t length,
                      void **out_buf, size_t *out_size, struct flb_time *out_time)
{
    int i;
    int ret;
    int slot = -1;
    int64_t sec = 0;
    int64_t nsec = 0;
    char off[16];
    char *types = NULL;
    char *sig = NULL;
    uint64_t fp;
    flbgpu_parser *g = NULL;

    if (parser->type != FLB_PARSER_REGEX || parser->time_zone != NULL || parser->time_system_timezone) {
        return -1;
    }
    fp = twin_fingerprint(parser);
    pthread_mutex_lock(&twins_mu);
    if (twin_hit.p == parser && twin_hit.fp == fp && twin_hit.gen == twin_gen && twin_hit.slot >= 0 && twin_hit.slot < n_twins) {
        slot = twin_hit.slot;
    }
    else {
        pthread_mutex_unlock(&twins_mu);
        types = types_to_str(parser);
        sig = twin_signature(parser, types);
        if (!sig) {
            flb_free(types);
            return -1;
        }
        pthread_mutex_lock(&twins_mu);
        for (i = 0; i < n_twins; i++) {
            if (twins[i].sig && strcmp(twins[i].sig, sig) == 0) {
                slot = i;
                break;
            }
        }
    }
    if (slot < 0) {
        if (flbgpu_init(0) != 0) {
            goto fail_locked;
        }
        snprintf(off, sizeof(off), "%c%02d%02d", parser->time_offset < 0 ? '-' : '+',
                 abs(parser->time_offset) / 3600, (abs(parser->time_offset) / 60) % 60);
        g = flbgpu_parser_create(parser->name, parser->p_regex, parser->skip_empty, parser->time_fmt_full,
                                 parser->time_key, parser->time_offset ? off : NULL, parser->time_keep,
                                 parser->time_strict, types);
        if (!g) {
            goto fail_locked;
        }
        if (add_decoders(g, parser) != 0) {
            flbgpu_parser_destroy(g);
            goto fail_locked;
        }
        if (n_twins < MAX_TWINS) {
            slot = n_twins++;
            pthread_mutex_init(&twins[slot].lock, NULL);
        }
        else {
            /* the least recently used twin nobody is inside of */
            for (i = 0; i < MAX_TWINS; i++) {
                if (twins[i].users == 0 && (slot < 0 || twins[i].stamp < twins[slot].stamp)) {
                    slot = i;
                }
            }
            if (slot < 0) {
                flbgpu_parser_destroy(g);
                goto fail_locked;
            }
            flbgpu_parser_destroy(twins[slot].g);
            flb_free(twins[slot].sig);
            twin_gen++;                                /* (what a thread remembers of this slot is another parser's now) */
        }
        twins[slot].g = g;
        twins[slot].sig = sig;
        twins[slot].users = 0;
        sig = NULL;
    }
    twins[slot].users++;
    twins[slot].stamp = ++twin_clock;
    g = twins[slot].g;
    twin_hit.p = parser; twin_hit.fp = fp; twin_hit.gen = twin_gen; twin_hit.slot = slot;
    pthread_mutex_unlock(&twins_mu);
    flb_free(types);
    flb_free(sig);

    pthread_mutex_lock(&twins[slot].lock);
    ret = flbgpu_parser_do(g, buf, length, out_buf, out_size, &sec, &nsec);
    pthread_mutex_unlock(&twins[slot].lock);

    pthread_mutex_lock(&twins_mu);
    twins[slot].users--;
    pthread_mutex_unlock(&twins_mu);
    if (ret >= 0 && out_time) {
        flb_time_set(out_time, (time_t) sec, (long) nsec);
    }
    return ret;

fail_locked:
    pthread_mutex_unlock(&twins_mu);
    flb_free(types);
    flb_free(sig);
    return -1;
}
#endif /* FLBGPU_WITH_PARSER */

/* ------------------------------------------------------------------ log_to_metrics */
#ifdef FLBGPU_WITH_L2M
/*
 * The device keeps the series state (flbgpu_filter_l2m_create / flbgpu_filter_run); this shim keeps
 * what the reference plugin keeps around it (plugins/filter_log_to_metrics/log_to_metrics.c:655-968,
 * 610-652): the cmetrics context with one counter / gauge / histogram, the hidden emitter input and
 * the optional flush timer.  Publishing = copy the finalized series into the cmetrics context with
 * the *_set calls and append it to the emitter.
 */
#include <fluent-bit/flb_input.h>
#include <fluent-bit/flb_input_metric.h>
#include <fluent-bit/flb_scheduler.h>
#include <fluent-bit/flb_storage.h>
#include <fluent-bit/flb_sds.h>
#include <cmetrics/cmetrics.h>
#include <cmetrics/cmt_counter.h>
#include <cmetrics/cmt_gauge.h>
#include <cmetrics/cmt_histogram.h>
#include <cfl/cfl_time.h>

struct l2m_gpu_ctx {
    flbgpu_filter *f;                 /* first member: shared with cb_gpu_filter's view */
    struct flb_filter_instance *ins;
    struct gpu_watch w;
    int mode, label_count, nbuckets, row_words;
    int sum_reference;                /* sum_order reference (the default): the histogram sum is cmetrics' sequential sum */
    struct cmt *cmt;
    struct cmt_counter *c;
    struct cmt_gauge *g;
    struct cmt_histogram *h;
    struct flb_input_instance *emitter;
    struct flb_sched_timer *timer;
    int timer_mode;
    int new_data;
    /* config map targets */
    flb_sds_t mode_name, value_field, metric_name, metric_namespace, metric_subsystem, metric_description;
    flb_sds_t tag, emitter_name, sum_order;
    size_t emitter_mem_buf_limit;
    int kubernetes_mode, discard_logs;
    int flush_interval_sec, flush_interval_nsec;
};

/* device state -> cmetrics context -> emitter */
static int l2m_gpu_publish(struct l2m_gpu_ctx *ctx)
{
    int64_t n;
    int64_t s;
    int i;
    size_t need = 0;
    size_t kcap = 1 << 16;
    uint64_t cap = 1024;
    uint64_t *rows = NULL;
    uint64_t *koff = NULL;
    uint64_t *buckets = NULL;
    char *keys = NULL;
    char **labels = NULL;
    double *seq = NULL;
    uint64_t ts = cfl_time_now();

    for (;;) {
        rows = flb_malloc(cap * ctx->row_words * sizeof(uint64_t));
        koff = flb_malloc((cap + 1) * sizeof(uint64_t));
        keys = flb_malloc(kcap);
        if (!rows || !koff || !keys) {
            flb_errno();
            flb_free(rows); flb_free(koff); flb_free(keys);
            return -1;
        }
        n = flbgpu_l2m_export(ctx->f, cap, rows, koff, keys, kcap, &need);
        if (n >= 0) {
            break;
        }
        flb_free(rows); flb_free(koff); flb_free(keys);
        if (n == -1) {
            flb_plg_error(ctx->ins, "%s", flbgpu_last_error());
            return -1;
        }
        cap = (uint64_t) (-n - 2) + 16;
        kcap = need + 16;
    }
    labels = flb_calloc(ctx->label_count ? ctx->label_count : 1, sizeof(char *));
    buckets = flb_calloc(ctx->nbuckets + 1, sizeof(uint64_t));
    if (ctx->mode == 2 && ctx->sum_reference && n > 0) {
        /* sum_order reference: the sums exactly as cmt_metric_hist_sum_add builds them (one f64 addition per observation in record
         * order, lib/cmetrics/src/cmt_metric_histogram.c:124-137), in flbgpu_l2m_export's series order */
        seq = flb_malloc(n * sizeof(double));
        if (!seq || flbgpu_l2m_seq_sums(ctx->f, (uint64_t) n, seq) != n) {
            flb_plg_error(ctx->ins, "sum_order reference: %s", flbgpu_last_error());
            flb_free(seq); flb_free(labels); flb_free(buckets); flb_free(rows); flb_free(koff); flb_free(keys);
            return -1;
        }
    }
    for (s = 0; s < n && labels && buckets; s++) {
        double value;
        double sum;
        uint64_t count;
        char *p = keys + koff[s];

        for (i = 0; i < ctx->label_count; i++) {      /* NUL-terminated label values back to back */
            labels[i] = p;
            p += strlen(p) + 1;
        }
        flbgpu_l2m_finalize_row(ctx->mode, ctx->nbuckets, rows + s * ctx->row_words, &value, buckets, &count, &sum);
        if (seq) {
            sum = seq[s];
        }
        if (ctx->mode == 0) {
            cmt_counter_set(ctx->c, ts, value, ctx->label_count, labels);
        }
        else if (ctx->mode == 1) {
            cmt_gauge_set(ctx->g, ts, value, ctx->label_count, labels);
        }
        else {
            cmt_histogram_set_default(ctx->h, ts, buckets, sum, count, ctx->label_count, labels);
        }
    }
    flb_free(seq); flb_free(labels); flb_free(buckets); flb_free(rows); flb_free(koff); flb_free(keys);
    return flb_input_metrics_append(ctx->emitter, ctx->tag, flb_sds_len(ctx->tag), ctx->cmt);
}

static void cb_l2m_gpu_timer(struct flb_config *config, void *data)
{
    struct l2m_gpu_ctx *ctx = data;
    (void) config;
    if (ctx->new_data && l2m_gpu_publish(ctx) == 0) {
        ctx->new_data = FLB_FALSE;
    }
}

static int cb_l2m_gpu_filter(const void *data, size_t bytes, const char *tag, int tag_len,
                             void **out_buf, size_t *out_size,
                             struct flb_filter_instance *f_ins, struct flb_input_instance *i_ins,
                             void *context, struct flb_config *config)
{
    struct l2m_gpu_ctx *ctx = context;
    int ret;
    (void) tag; (void) tag_len; (void) f_ins; (void) i_ins; (void) config;

    ret = flbgpu_filter_run(ctx->f, data, bytes, out_buf, out_size);
    gpu_watch_after_run(ctx->ins, ctx->f, &ctx->w);
    if (ctx->timer_mode) {
        ctx->new_data = FLB_TRUE;
    }
    else if (l2m_gpu_publish(ctx) != 0) {
        flb_plg_error(ctx->ins, "could not append metrics");
    }
    return ret;
}

static int cb_l2m_gpu_exit(void *data, struct flb_config *config)
{
    struct l2m_gpu_ctx *ctx = data;
    (void) config;
    if (!ctx) {
        return 0;
    }
    if (ctx->timer) {
        flb_sched_timer_cb_destroy(ctx->timer);
    }
    if (ctx->cmt) {
        cmt_destroy(ctx->cmt);
    }
    flbgpu_filter_destroy(ctx->f);
    flb_free(ctx);
    return 0;
}

static int cb_l2m_gpu_init(struct flb_filter_instance *f_ins, struct flb_config *config, void *data)
{
    int n = 0;
    int i = 0;
    int ms;
    const char **keys;
    const char **vals;
    char **label_keys;
    double *bounds;
    const char *alias;
    char alias_buf[256];
    char limit_buf[32];
    struct mk_list *head;
    struct flb_kv *kv;
    struct l2m_gpu_ctx *ctx;
    struct flb_sched *sched;
    (void) data;

    if (ensure_gpu(f_ins) != 0) {
        return -1;
    }
    ctx = flb_calloc(1, sizeof(struct l2m_gpu_ctx));
    if (!ctx) {
        flb_errno();
        return -1;
    }
    ctx->ins = f_ins;
    if (flb_filter_config_map_set(f_ins, ctx) < 0) {
        flb_free(ctx);
        return -1;
    }
    if (!ctx->tag || !ctx->metric_name || !ctx->metric_description) {
        flb_plg_error(f_ins, "tag, metric_name and metric_description must be set");
        flb_free(ctx);
        return -1;
    }
    /* the multi-valued properties in configuration order (set_rules / set_labels / set_buckets) */
    mk_list_foreach(head, &f_ins->properties) {
        n++;
    }
    keys = flb_calloc(n ? n : 1, sizeof(char *));
    vals = flb_calloc(n ? n : 1, sizeof(char *));
    mk_list_foreach(head, &f_ins->properties) {
        kv = mk_list_entry(head, struct flb_kv, _head);
        keys[i] = kv->key;
        vals[i] = kv->val;
        i++;
    }
    ctx->f = flbgpu_filter_l2m_create(ctx->mode_name, n, keys, vals, ctx->kubernetes_mode, ctx->value_field,
                                      ctx->discard_logs);
    flb_free(keys);
    flb_free(vals);
    if (!ctx->f) {
        flb_plg_error(f_ins, "%s", flbgpu_last_error());
        flb_free(ctx);
        return -1;
    }
    gpu_note_host_rules(f_ins, ctx->f);
    flbgpu_l2m_info(ctx->f, &ctx->mode, &ctx->label_count, &ctx->nbuckets, &ctx->row_words);
    /* sum_order: "reference" (default) = the histogram sum is the reference plugin's own bits (sequential f64 additions in record
     * order); "exact" = the exact sum of the observations rounded once (order-independent: what a sharded multi-GPU run can merge) */
    if (!ctx->sum_order || strcasecmp(ctx->sum_order, "reference") == 0) {
        ctx->sum_reference = FLB_TRUE;
    }
    else if (strcasecmp(ctx->sum_order, "exact") == 0) {
        ctx->sum_reference = FLB_FALSE;
    }
    else {
        flb_plg_error(f_ins, "sum_order must be 'reference' or 'exact'");
        flbgpu_filter_destroy(ctx->f);
        flb_free(ctx);
        return -1;
    }
    if (ctx->mode == 2 && ctx->sum_reference && flbgpu_l2m_set_sum_order(ctx->f, 1) != 0) {
        flb_plg_error(f_ins, "%s", flbgpu_last_error());
        flbgpu_filter_destroy(ctx->f);
        flb_free(ctx);
        return -1;
    }

    /* cmetrics context (:825-850); an empty subsystem defaults to the mode name (:769-776) */
    label_keys = flb_calloc(ctx->label_count ? ctx->label_count : 1, sizeof(char *));
    for (i = 0; i < ctx->label_count; i++) {
        label_keys[i] = (char *) flbgpu_l2m_label_key(ctx->f, i);
    }
    ctx->cmt = cmt_create();
    if (!ctx->metric_subsystem || flb_sds_len(ctx->metric_subsystem) == 0) {
        ctx->metric_subsystem = ctx->mode_name;
    }
    if (ctx->mode == 0) {
        ctx->c = cmt_counter_create(ctx->cmt, ctx->metric_namespace, ctx->metric_subsystem, ctx->metric_name,
                                    ctx->metric_description, ctx->label_count, label_keys);
    }
    else if (ctx->mode == 1) {
        ctx->g = cmt_gauge_create(ctx->cmt, ctx->metric_namespace, ctx->metric_subsystem, ctx->metric_name,
                                  ctx->metric_description, ctx->label_count, label_keys);
    }
    else {
        bounds = flb_calloc(ctx->nbuckets ? ctx->nbuckets : 1, sizeof(double));
        flbgpu_l2m_bounds(ctx->f, bounds);
        ctx->h = cmt_histogram_create(ctx->cmt, ctx->metric_namespace, ctx->metric_subsystem, ctx->metric_name,
                                      ctx->metric_description,
                                      cmt_histogram_buckets_create_size(bounds, ctx->nbuckets),
                                      ctx->label_count, label_keys);
        flb_free(bounds);
    }
    flb_free(label_keys);

    /* hidden emitter input (:852-930) */
    if (ctx->emitter_name && flb_sds_len(ctx->emitter_name) > 0) {
        alias = ctx->emitter_name;
    }
    else {
        snprintf(alias_buf, sizeof(alias_buf) - 1, "emitter_for_%s", flb_filter_name(f_ins));
        alias = alias_buf;
    }
    if (flb_input_name_exists(alias, config)) {
        flb_plg_error(f_ins, "emitter_name '%s' already exists", alias);
        cb_l2m_gpu_exit(ctx, config);
        return -1;
    }
    ctx->emitter = flb_input_new(config, "emitter", NULL, FLB_FALSE);
    if (!ctx->emitter ||
        flb_input_set_property(ctx->emitter, "alias", alias) == -1 ||
        flb_input_set_property(ctx->emitter, "storage.type", "memory") == -1) {
        flb_plg_error(f_ins, "cannot create metrics emitter instance");
        cb_l2m_gpu_exit(ctx, config);
        return -1;
    }
    if (ctx->emitter_mem_buf_limit > 0) {
        /* the reference writes input_ins->mem_buf_limit (:913-915): the same value through the property, so that no struct offset
         * behind an #ifdef of flb_input.h (FLB_HAVE_CHUNK_TRACE) is touched from a separately built object */
        snprintf(limit_buf, sizeof(limit_buf), "%zu", ctx->emitter_mem_buf_limit);
        if (flb_input_set_property(ctx->emitter, "mem_buf_limit", limit_buf) == -1) {
            flb_plg_error(f_ins, "cannot set the emitter's mem_buf_limit");
            cb_l2m_gpu_exit(ctx, config);
            return -1;
        }
    }
    if (flb_input_instance_init(ctx->emitter, config) == -1 ||
        flb_storage_input_create(config->cio, ctx->emitter) == -1) {
        flb_plg_error(f_ins, "cannot initialize metrics emitter instance");
        cb_l2m_gpu_exit(ctx, config);
        return -1;
    }

    /* flush timer (:933-966): both intervals 0 => publish after every call */
    ms = ctx->flush_interval_sec * 1000 + ctx->flush_interval_nsec / 1000000;
    if (ms > 0) {
        sched = flb_sched_ctx_get();
        if (!sched || flb_sched_timer_cb_create(sched, FLB_SCHED_TIMER_CB_PERM, ms, cb_l2m_gpu_timer, ctx,
                                                &ctx->timer) < 0) {
            flb_plg_error(f_ins, "could not create timer callback");
            cb_l2m_gpu_exit(ctx, config);
            return -1;
        }
        ctx->timer_mode = FLB_TRUE;
    }
    flb_filter_set_context(f_ins, ctx);
    return 0;
}

static struct flb_config_map l2m_config_map[] = {
    { FLB_CONFIG_MAP_STR, "regex", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0, "Optional filter: KEY REGEX must match." },
    { FLB_CONFIG_MAP_STR, "exclude", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0, "Optional filter: KEY REGEX must not match." },
    { FLB_CONFIG_MAP_STR, "metric_mode", "counter", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, mode_name), "counter, gauge or histogram." },
    { FLB_CONFIG_MAP_STR, "value_field", NULL, 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, value_field), "Numeric field for gauge / histogram." },
    { FLB_CONFIG_MAP_STR, "metric_name", "a", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, metric_name), "Name of the metric." },
    { FLB_CONFIG_MAP_STR, "metric_namespace", "log_metric", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, metric_namespace), "Namespace of the metric." },
    { FLB_CONFIG_MAP_STR, "metric_subsystem", NULL, 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, metric_subsystem), "Subsystem of the metric." },
    { FLB_CONFIG_MAP_STR, "metric_description", NULL, 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, metric_description), "Help text for metric." },
    { FLB_CONFIG_MAP_BOOL, "kubernetes_mode", "false", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, kubernetes_mode), "Enable kubernetes log metric fields." },
    { FLB_CONFIG_MAP_STR, "add_label", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0, "Add a label: NAME ACCESSOR." },
    { FLB_CONFIG_MAP_STR, "label_field", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0, "Message field to include as a label." },
    { FLB_CONFIG_MAP_STR, "bucket", NULL, FLB_CONFIG_MAP_MULT, FLB_FALSE, 0, "Histogram bucket upper bound." },
    { FLB_CONFIG_MAP_STR, "tag", NULL, 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, tag), "Metric Tag." },
    { FLB_CONFIG_MAP_STR, "emitter_name", NULL, 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, emitter_name), "Name of the emitter." },
    { FLB_CONFIG_MAP_SIZE, "emitter_mem_buf_limit", "10M", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, emitter_mem_buf_limit), "Emitter buffer limit." },
    { FLB_CONFIG_MAP_INT, "flush_interval_sec", "0", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, flush_interval_sec), "Timer interval (s); 0/0 = emit immediately." },
    { FLB_CONFIG_MAP_INT, "flush_interval_nsec", "0", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, flush_interval_nsec), "Timer interval (ns part)." },
    { FLB_CONFIG_MAP_BOOL, "discard_logs", "false", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, discard_logs), "Drop the logs after processing." },
    { FLB_CONFIG_MAP_STR, "sum_order", "reference", 0, FLB_TRUE, offsetof(struct l2m_gpu_ctx, sum_order),
      "Histogram sum: 'reference' = cmetrics' sequential f64 sum, bit for bit (default); 'exact' = the exact sum rounded once." },
    {0}
};

struct flb_filter_plugin filter_log_to_metrics_gpu_plugin = {
    .name         = "log_to_metrics" FLBGPU_PLUGIN_SUFFIX,
    .description  = "generate log derived metrics (MI355X)",
    .cb_init      = cb_l2m_gpu_init,
    .cb_filter    = cb_l2m_gpu_filter,
    .cb_exit      = cb_l2m_gpu_exit,
    .config_map   = l2m_config_map,
    .flags        = 0
};
#endif /* FLBGPU_WITH_L2M */
