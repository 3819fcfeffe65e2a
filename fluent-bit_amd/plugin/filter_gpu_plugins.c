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

static int gpu_ready = 0;

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
};

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
    flb_filter_set_context(f_ins, ctx);
    return 0;
}

static int cb_gpu_filter(const void *data, size_t bytes, const char *tag, int tag_len,
                         void **out_buf, size_t *out_size,
                         struct flb_filter_instance *f_ins, struct flb_input_instance *i_ins,
                         void *context, struct flb_config *config)
{
    /* grep_gpu_ctx and parser_gpu_ctx share their first member */
    struct grep_gpu_ctx *ctx = context;
    (void) tag;
    (void) tag_len;
    (void) f_ins;
    (void) i_ins;
    (void) config;
    /* FLBGPU_FILTER_MODIFIED/NOTOUCH == FLB_FILTER_MODIFIED/NOTOUCH; the output buffer is
     * malloc'd, the engine releases it with flb_free (src/flb_filter.c:235-237) */
    return flbgpu_filter_run(ctx->f, data, bytes, out_buf, out_size);
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

/* ------------------------------------------------------------------ parser */
#define MAX_GPU_PARSERS 16

struct parser_gpu_ctx {
    flbgpu_filter *f;                       /* must stay first (cb_gpu_filter) */
    struct flb_filter_instance *ins;
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

static int cb_parser_gpu_init(struct flb_filter_instance *f_ins, struct flb_config *config, void *data)
{
    char *types;
    char off[16];
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
        if (p->type != FLB_PARSER_REGEX || p->decoders != NULL || p->time_zone != NULL ||
            p->time_system_timezone) {
            flb_plg_error(f_ins, "parser '%s': only Format regex without decoders/time zones "
                          "is on the GPU path", kv->val);
            goto error;
        }
        if (ctx->n_parsers >= MAX_GPU_PARSERS) {
            goto error;
        }
        types = types_to_str(p);
        snprintf(off, sizeof(off), "%c%02d%02d", p->time_offset < 0 ? '-' : '+',
                 abs(p->time_offset) / 3600, (abs(p->time_offset) / 60) % 60);
        ctx->parsers[ctx->n_parsers] =
            flbgpu_parser_create(p->name, p->p_regex, p->skip_empty, p->time_fmt_full, p->time_key,
                                 p->time_offset ? off : NULL, p->time_keep, p->time_strict, types);
        flb_free(types);
        if (!ctx->parsers[ctx->n_parsers]) {
            flb_plg_error(f_ins, "%s", flbgpu_last_error());
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
