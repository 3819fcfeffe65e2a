/*
 * oracle/oflb.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (oracle) of the reference filter hot path, one function per reference
 * function, each citing the reference lines it follows:
 *
 *   oflb_regex_create     src/flb_regex.c:60-152   (check_option + str_to_regex: /pat/imx)
 *   oflb_parser_create    src/flb_parser.c:805-1049 (time format analysis, %L split, types)
 *   oflb_parser_do        src/flb_parser_regex.c:44-227 (cb_results + flb_parser_regex_do)
 *   time lookup           src/flb_parser.c:1876-2065, include/fluent-bit/flb_parser.h:80-94
 *   typecast              src/flb_parser.c:2067-2164
 *   oflb_grep_*           plugins/filter_grep/grep.c:56-392, src/flb_ra_key.c:108-434
 *   oflb_l2m_*            plugins/filter_log_to_metrics/log_to_metrics.c:216-343,355-595,970-1156,
 *                         lib/cmetrics/src/cmt_histogram.c:328-361
 *   oflb_fparser_*        plugins/filter_parser/filter_parser.c:174-442,
 *                         src/flb_pack.c:1664-1738 (flb_msgpack_expand_map),
 *                         src/flb_log_event_encoder.c:172-363
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdio.h>
#include <time.h>
#include <limits.h>
#include "omp.h"
#include "orx.h"
#include "otime.h"

#define FLB_FILTER_MODIFIED 1   /* include/fluent-bit/flb_filter.h:41-42 */
#define FLB_FILTER_NOTOUCH  2

/* ------------------------------------------------------------------ flb_regex */
typedef struct oflb_regex { orx_t *rx; } oflb_regex;

/* src/flb_regex.c:60-114 check_option(); returns option bits or -1 for ONIG_OPTION_DEFAULT */
static int check_option(const char *start, const char *end, const char **new_end)
{
    const char *chr;
    int option = 0;
    *new_end = NULL;
    if (start[0] != '/') return -1;
    chr = strrchr(start, '/');
    if (!chr) return -1;
    if (chr == start || chr == end) return -1;
    *new_end = chr;
    chr++;
    while (chr != end && *chr != '\0') {
        switch (*chr) {
        case 'm': option |= ORX_OPT_MULTILINE; break;
        case 'i': option |= ORX_OPT_IGNORECASE; break;
        case 'o': break;
        case 'x': option |= ORX_OPT_EXTEND; break;
        default:
            *new_end = NULL;
            return -1;
        }
        chr++;
    }
    if (option == 0) { *new_end = NULL; return -1; }
    return option;
}

oflb_regex *oflb_regex_create(const char *pattern)
{
    size_t len = strlen(pattern);
    const char *start = pattern, *end = pattern + len, *new_end = NULL;
    int option = check_option(start, end, &new_end);
    oflb_regex *r;
    orx_t *rx;
    if (len > 1 && pattern[0] == '/' && pattern[len - 1] == '/') { start++; end--; }
    if (new_end != NULL) { start++; end = new_end; }
    /* ONIG_OPTION_DEFAULT == ONIG_OPTION_NONE */
    rx = orx_compile(start, (int) (end - start), option < 0 ? 0 : (unsigned) option, NULL, 0);
    if (!rx) return NULL;
    r = calloc(1, sizeof(*r));
    r->rx = rx;
    return r;
}

void oflb_regex_destroy(oflb_regex *r) { if (r) { orx_free(r->rx); free(r); } }

/* src/flb_regex.c:270-291 */
int oflb_regex_match(oflb_regex *r, const char *s, size_t len) { return orx_match(r->rx, s, (int) len); }

/* ------------------------------------------------------------------ parser */
enum { T_INT = 1, T_FLOAT, T_BOOL, T_STRING, T_HEX };
struct ptype { char *key; int key_len; int type; };

enum { KV_NONE = 0, KV_LOGFMT = 1, KV_LTSV = 2 };
typedef struct oflb_parser {
    int is_json;             /* Format json (src/flb_parser_json.c) instead of Format regex */
    int kv_format;           /* Format logfmt / ltsv (src/flb_parser_logfmt.c, src/flb_parser_ltsv.c) */
    int no_bare_keys;        /* Logfmt_No_Bare_Keys */
    oflb_regex *regex;
    int skip_empty;
    char *time_fmt;          /* cut at %L */
    char *time_fmt_year;     /* "%Y " + fmt, cut at %L */
    char *time_frac_secs;    /* format text after %L, or NULL */
    char *time_key;
    int time_offset;
    int time_keep;
    int time_strict;
    int time_with_year;
    int time_with_tz;
    struct ptype *types;
    int types_len;
    struct odec *decs;       /* Decode_Field / Decode_Field_As (src/flb_parser_decoder.c), one entry per key */
    int ndecs;
    /* Time_Zone (struct tzif, src/flb_parser.c:217-229) / Time_System_Timezone */
    int has_zone, system_tz;
    int tz_timecnt, tz_typecnt, tz_default;
    int64_t *tz_trans;
    unsigned char *tz_ttype;
    int32_t *tz_gmtoff;
} oflb_parser;

/* include/fluent-bit/flb_parser_decoder.h:28-59 */
enum { DEC_DEFAULT = 0, DEC_AS = 1 };
enum { DEC_JSON = 0, DEC_ESCAPED = 1, DEC_ESCAPED_UTF8 = 2, DEC_MYSQL_QUOTED = 3 };
enum { ACT_NONE = 0, ACT_TRY_NEXT = 1, ACT_DO_NEXT = 2 };
struct odec_rule { int type, backend, action; };
struct odec { char *key; size_t key_len; int add_extra_keys; struct odec_rule *rules; int nrules; };

/* src/flb_parser.c:1806-1870 flb_parser_tzone_offset */
static int tzone_offset(const char *str, int len, int *tmdiff)
{
    int neg;
    long hour, min;
    const char *end, *p = str;
    if (*p == 'Z') { *tmdiff = 0; return 0; }
    if (*p != '+' && *p != '-') { *tmdiff = 0; return -1; }
    if (len < 4) { *tmdiff = 0; return -1; }
    neg = (*p++ == '-');
    end = str + len;
    hour = ((p[0] - '0') * 10) + (p[1] - '0');
    if (end - p == 5 && p[2] == ':') {
        if (len < 5) { *tmdiff = 0; return -1; }
        min = ((p[3] - '0') * 10) + (p[4] - '0');
    }
    else min = ((p[2] - '0') * 10) + (p[3] - '0');
    if (hour < 0 || hour > 59 || min < 0 || min > 59) return -1;
    *tmdiff = (int) ((hour * 3600) + (min * 60));
    if (neg) *tmdiff = -*tmdiff;
    return 0;
}

/* src/flb_parser.c:1130-1182 proc_types_str */
static int proc_types(const char *types_str, struct ptype **out)
{
    /* flb_utils_split(str, ' ', 256): tokens separated by single spaces, leading ones skipped */
    int n = 0, cap = 8;
    struct ptype *t = calloc(cap, sizeof(*t));
    const char *p = types_str;
    while (*p) {
        const char *q, *colon;
        while (*p == ' ') p++;
        if (!*p) break;
        q = strchr(p, ' ');
        if (!q) q = p + strlen(p);
        if (n == cap) { cap *= 2; t = realloc(t, cap * sizeof(*t)); }
        t[n].key = NULL; t[n].type = T_STRING; t[n].key_len = 0;
        colon = memchr(p, ':', q - p);
        if (colon) {
            size_t tl = q - (colon + 1);
            const char *ts = colon + 1;
            t[n].key = strndup(p, colon - p);
            t[n].key_len = (int) (colon - p);
            if (tl == 7 && !strncasecmp(ts, "integer", 7)) t[n].type = T_INT;
            else if (tl == 4 && !strncasecmp(ts, "bool", 4)) t[n].type = T_BOOL;
            else if (tl == 5 && !strncasecmp(ts, "float", 5)) t[n].type = T_FLOAT;
            else if (tl == 3 && !strncasecmp(ts, "hex", 3)) t[n].type = T_HEX;
            else t[n].type = T_STRING;
        }
        n++;
        p = *q ? q + 1 : q;
    }
    *out = t;
    return n;
}

oflb_parser *oflb_parser_create(const char *regex, int skip_empty, const char *time_fmt,
                                const char *time_key, const char *time_offset, int time_keep,
                                int time_strict, const char *types_str)
{
    oflb_parser *p = calloc(1, sizeof(*p));
    if (regex == NULL) p->is_json = 1;                   /* Format json: no regex */
    else {
        p->regex = oflb_regex_create(regex);
        if (!p->regex) { free(p); return NULL; }
    }
    p->skip_empty = skip_empty;
    if (time_fmt && time_fmt[0]) {
        int is_epoch = 0;
        char *timeptr, *tmp;
        p->time_fmt = strdup(time_fmt);
        if (strstr(p->time_fmt, "%Y") || strstr(p->time_fmt, "%y")) p->time_with_year = 1;
        else if (strstr(p->time_fmt, "%s")) { is_epoch = 1; p->time_with_year = 1; }
        else {
            size_t size = strlen(p->time_fmt);
            p->time_with_year = 0;
            p->time_fmt_year = malloc(size + 4);
            memcpy(p->time_fmt_year, "%Y ", 3);
            memcpy(p->time_fmt_year + 3, p->time_fmt, size + 1);
        }
        if (strstr(p->time_fmt, "%z") || strstr(p->time_fmt, "%Z") || strstr(p->time_fmt, "%SZ")
            || strstr(p->time_fmt, "%S.%LZ")) p->time_with_tz = 1;
        timeptr = (is_epoch || p->time_with_year) ? p->time_fmt : p->time_fmt_year;
        tmp = strstr(timeptr, "%L");
        if (tmp) { tmp[0] = '\0'; tmp[1] = '\0'; p->time_frac_secs = tmp + 2; }
        if (time_offset && time_offset[0]) {
            int diff = 0;
            if (tzone_offset(time_offset, (int) strlen(time_offset), &diff) == -1) {
                oflb_regex_destroy(p->regex); free(p); return NULL;
            }
            p->time_offset = diff;
        }
    }
    if (time_key && time_key[0]) p->time_key = strdup(time_key);
    p->time_keep = time_keep;
    p->time_strict = time_strict;
    if (types_str && types_str[0]) p->types_len = proc_types(types_str, &p->types);
    return p;
}

/* Format logfmt (1) / ltsv (2): flb_parser_create(name, "logfmt" | "ltsv", NULL, ...) */
oflb_parser *oflb_parser_create_kv2(int kv_format, const char *time_fmt, const char *time_key, const char *time_offset,
                                    int time_keep, int time_strict, int no_bare_keys, const char *types_str)
{
    oflb_parser *p = oflb_parser_create(NULL, 0, time_fmt, time_key, time_offset, time_keep, time_strict, types_str);
    if (!p) return NULL;
    p->is_json = 0;
    p->kv_format = kv_format;
    p->no_bare_keys = no_bare_keys;
    return p;
}

oflb_parser *oflb_parser_create_kv(int kv_format, const char *time_fmt, const char *time_key, const char *time_offset,
                                   int time_keep, int time_strict, int no_bare_keys)
{
    oflb_parser *p = oflb_parser_create(NULL, 0, time_fmt, time_key, time_offset, time_keep, time_strict, NULL);
    if (!p) return NULL;
    p->is_json = 0;
    p->kv_format = kv_format;
    p->no_bare_keys = no_bare_keys;
    return p;
}

void oflb_parser_destroy(oflb_parser *p)
{
    int i;
    if (!p) return;
    oflb_regex_destroy(p->regex);
    free(p->time_fmt); free(p->time_fmt_year); free(p->time_key);
    for (i = 0; i < p->types_len; i++) free(p->types[i].key);
    free(p->types);
    for (i = 0; i < p->ndecs; i++) { free(p->decs[i].key); free(p->decs[i].rules); }
    free(p->decs);
    free(p);
}

/* One `Decode_Field[_As] <backend> <field> [action]` line, in configuration order: flb_parser_decoder_list_create
 * (src/flb_parser_decoder.c:603-745) with get_decoder_key_context (:555-601).  -1: unknown backend. */
int oflb_parser_add_decoder(oflb_parser *p, int as, const char *backend, const char *field, const char *action)
{
    struct odec *d = NULL;
    struct odec_rule r;
    int i;
    if (!strcasecmp(backend, "json")) r.backend = DEC_JSON;
    else if (!strcasecmp(backend, "escaped")) r.backend = DEC_ESCAPED;
    else if (!strcasecmp(backend, "escaped_utf8")) r.backend = DEC_ESCAPED_UTF8;
    else if (!strcasecmp(backend, "mysql_quoted")) r.backend = DEC_MYSQL_QUOTED;
    else return -1;
    r.type = as ? DEC_AS : DEC_DEFAULT;
    r.action = ACT_NONE;
    if (action && action[0]) {
        if (!strcasecmp(action, "try_next")) r.action = ACT_TRY_NEXT;
        else if (!strcasecmp(action, "do_next")) r.action = ACT_DO_NEXT;
    }
    for (i = 0; i < p->ndecs; i++)
        if (p->decs[i].key_len == strlen(field) && memcmp(p->decs[i].key, field, p->decs[i].key_len) == 0) { d = &p->decs[i]; break; }
    if (!d) {
        p->decs = realloc(p->decs, sizeof(*p->decs) * (size_t) (p->ndecs + 1));
        d = &p->decs[p->ndecs++];
        memset(d, 0, sizeof(*d));
        d->key = strdup(field);
        d->key_len = strlen(field);
    }
    if (r.type == DEC_DEFAULT) d->add_extra_keys = 1;
    d->rules = realloc(d->rules, sizeof(*d->rules) * (size_t) (d->nrules + 1));
    d->rules[d->nrules++] = r;
    return 0;
}

/* src/flb_parser.c:1876-1897 */
static int parse_subseconds(const char *str, int len, double *subsec)
{
    char buf[16];
    char *end;
    int consumed, digits = 9;
    if (len < digits) digits = len;
    memcpy(buf, "0.", 2);
    memcpy(buf + 2, str, digits);
    buf[digits + 2] = '\0';
    *subsec = strtod(buf, &end);
    consumed = (int) (end - buf - 2);
    if (consumed <= 0) return -1;
    return consumed;
}

/* test hook: pins the time(NULL) of year-less formats (the filters call the lookup with now == 0) */
static time_t g_now_override = 0;
void oflb_set_time_now(int64_t now) { g_now_override = (time_t) now; }

/* src/flb_parser.c:1899-2065 (time_zone / system timezone branches not restated) */
static int time_lookup(const char *time_str, size_t tsize, time_t now, oflb_parser *parser,
                       struct otm *tm, double *ns)
{
    int ret, time_len = (int) tsize;
    const char *p, *time_ptr = time_str;
    char tmp[64];
    char *buf = tmp, *time_buf = NULL;
    size_t buf_size = sizeof(tmp);
    *ns = 0;
    if (tsize > sizeof(tmp) - 1) {
        buf_size = tsize + 8;
        time_buf = malloc(buf_size);
        buf = time_buf;
    }
    if (!parser->time_with_year) {
        time_t time_now;
        struct tm tmy;
        char *fmt;
        if (time_len + 6 >= (int) buf_size) { free(time_buf); return -1; }
        time_now = now <= 0 ? (g_now_override > 0 ? g_now_override : time(NULL)) : now;
        gmtime_r(&time_now, &tmy);
        tm->tm.tm_mon = tmy.tm_mon;
        tm->tm.tm_mday = tmy.tm_mday;
        fmt = buf;
        sprintf(fmt, "%04d", tmy.tm_year + 1900);
        fmt += 4;
        *fmt++ = ' ';
        memcpy(fmt, time_ptr, time_len);
        fmt += time_len;
        *fmt++ = '\0';
        time_ptr = buf;
        time_len = (int) strlen(buf);
        p = o_strptime(time_ptr, parser->time_fmt_year, tm);
    }
    else {
        if (time_len >= (int) buf_size) { free(time_buf); return -1; }
        memcpy(buf, time_ptr, time_len);
        buf[time_len] = '\0';
        time_ptr = buf;
        time_len = (int) strlen(buf);
        p = o_strptime(time_ptr, parser->time_fmt, tm);
    }
    if (p == NULL) {
        free(time_buf);
        return parser->time_strict ? -1 : 0;
    }
    if (parser->time_frac_secs) {
        ret = parse_subseconds(p, time_len - (int) (p - time_ptr), ns);
        if (ret < 0) { free(time_buf); return parser->time_strict ? -1 : 0; }
        p += ret;
        p = o_strptime(p, parser->time_frac_secs, tm);
        if (p == NULL) { free(time_buf); return parser->time_strict ? -1 : 0; }
    }
    if (!parser->time_with_tz) tm->gmtoff = parser->time_offset;
    free(time_buf);
    return 0;
}

/* src/flb_parser.c:539-558 tzif_type_at_utc, said the long way round: the type of the LAST transition at or before `utc`
 * (the reference bisects; the transitions of a TZif file ascend, so the last one that is <= utc is the same entry) */
static int zone_type_at(const oflb_parser *p, int64_t utc)
{
    int i, at = -1;
    for (i = 0; i < p->tz_timecnt; i++) {
        if (p->tz_trans[i] <= utc) at = i;
        else break;
    }
    if (at < 0) return p->tz_default;
    return p->tz_ttype[at];
}

/* src/flb_parser.c:560-590 tzif_tm2time */
static time_t zone_tm2time(const oflb_parser *p, const struct otm *src)
{
    struct tm tmp = src->tm;
    int64_t local_epoch, cand;
    int i, ty;
    tmp.tm_isdst = 0;
    local_epoch = (int64_t) timegm(&tmp);
    for (i = 0; i < p->tz_typecnt; i++) {
        cand = local_epoch - (int64_t) p->tz_gmtoff[i];
        ty = zone_type_at(p, cand);
        if (ty >= 0 && ty < p->tz_typecnt && p->tz_gmtoff[ty] == p->tz_gmtoff[i]) return (time_t) cand;
    }
    ty = zone_type_at(p, local_epoch);
    if (ty < 0 || ty >= p->tz_typecnt) return (time_t) -1;
    return (time_t) (local_epoch - (int64_t) p->tz_gmtoff[ty]);
}

/* src/flb_parser.c:685-696 flb_parser_tm2time_parser over include/fluent-bit/flb_parser.h:80-94 flb_parser_tm2time */
static time_t tm2time(const oflb_parser *parser, const struct otm *src)
{
    struct tm tmp = src->tm;
    if (parser && parser->has_zone && !parser->time_with_tz) return zone_tm2time(parser, src);
    if (parser && parser->system_tz) { tmp.tm_isdst = -1; return mktime(&tmp); }
    return timegm(&tmp) - src->gmtoff;
}

static uint32_t be32(const unsigned char *b) { return ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | b[3]; }

/* The `time_zone` / `time_system_timezone` arguments of flb_parser_create_with_time_zone (src/flb_parser.c:986-1022) on a parser made by
 * oflb_parser_create*: the zone's file read as tzif_load :452-537 + tzif_parse_data :359-450 read it.  -1 where the reference
 * returns NULL (no Time_Format :894, both at once :989, with Time_Offset :995 -- `had_offset` says the create call was given
 * one --, no such file :1003, a file that does not parse :1015). */
int oflb_parser_set_time_zone(oflb_parser *p, const char *zone, int system_tz, int had_offset)
{
    const char *tzdir = getenv("TZDIR");
    char path[4096];
    FILE *fp;
    unsigned char *buf, *h;
    long fsz;
    size_t size, off;
    int time_size = 4, i;
    uint32_t timecnt, typecnt;
    if (system_tz) { p->system_tz = 1; p->time_offset = 0; }
    if (!zone || !zone[0]) return 0;
    if (!p->time_fmt || system_tz || had_offset) return -1;
    {
        /* validate_time_zone :606-638: a name outside the built-in index (src/flb_time_tz.c) is refused before its file is looked for */
        static const char *const known[] = {
#include "tz_names.inc"
        };
        int found = 0;
        for (i = 0; i < (int) (sizeof(known) / sizeof(known[0])); i++) if (!strcmp(known[i], zone)) found = 1;
        if (!found) return -1;
    }
    if (!tzdir || !tzdir[0]) tzdir = "/usr/share/zoneinfo";
    if (snprintf(path, sizeof(path), "%s/%s", tzdir, zone) >= (int) sizeof(path)) return -1;
    fp = fopen(path, "rb");
    if (!fp) return -1;
    fseek(fp, 0, SEEK_END); fsz = ftell(fp); rewind(fp);
    if (fsz <= 0) { fclose(fp); return -1; }
    buf = malloc((size_t) fsz);
    if (fread(buf, 1, (size_t) fsz, fp) != (size_t) fsz) { fclose(fp); free(buf); return -1; }
    fclose(fp);
    if (fsz < 44 || memcmp(buf, "TZif", 4) != 0) { free(buf); return -1; }
    h = buf; size = (size_t) fsz;
    if (buf[4] == '2' || buf[4] == '3' || buf[4] == '4') {
        /* tzif_data_size :318-345: the length of the 32-bit block, whose 64-bit twin follows it */
        size_t block = (size_t) be32(buf + 32) * 4 + be32(buf + 32) + (size_t) be32(buf + 36) * 6 + be32(buf + 40) +
                       (size_t) be32(buf + 28) * 8 + be32(buf + 24) + be32(buf + 20);
        if (44 + block + 44 > (size_t) fsz || memcmp(buf + 44 + block, "TZif", 4) != 0) { free(buf); return -1; }
        h = buf + 44 + block; size = (size_t) fsz - 44 - block; time_size = 8;
    }
    timecnt = be32(h + 32); typecnt = be32(h + 36);
    if (typecnt == 0 || timecnt > 0x7fffffffu || typecnt > 0x7fffffffu ||
        44 + (size_t) timecnt * time_size + timecnt + (size_t) typecnt * 6 > size) { free(buf); return -1; }
    p->tz_trans = calloc(timecnt + 1, sizeof(int64_t)); p->tz_ttype = calloc(timecnt + 1, 1); p->tz_gmtoff = calloc(typecnt, sizeof(int32_t));
    off = 44;
    for (i = 0; i < (int) timecnt; i++, off += time_size)
        p->tz_trans[i] = time_size == 8 ? (int64_t) (((uint64_t) be32(h + off) << 32) | be32(h + off + 4)) : (int64_t) (int32_t) be32(h + off);
    memcpy(p->tz_ttype, h + off, timecnt);
    off += timecnt;
    for (i = 0; i < (int) timecnt; i++) if (p->tz_ttype[i] >= typecnt) { free(buf); return -1; }
    p->tz_default = -1;
    for (i = 0; i < (int) typecnt; i++, off += 6) {
        p->tz_gmtoff[i] = (int32_t) be32(h + off);
        if (p->tz_default < 0 && h[off + 4] == 0) p->tz_default = i;
    }
    if (p->tz_default < 0) p->tz_default = 0;
    p->tz_timecnt = (int) timecnt; p->tz_typecnt = (int) typecnt; p->has_zone = 1;
    free(buf);
    return 0;
}

/* src/flb_parser.c:2067-2164 */
static void typecast(oflb_parser *parser, const char *key, int key_len, const char *val, int val_len,
                     omp_buf *pck)
{
    int i, casted = 0, error = 0;
    char *tmp_str;
    for (i = 0; i < parser->types_len; i++) {
        struct ptype *t = &parser->types[i];
        if (t->key != NULL && key_len == t->key_len && !strncmp(key, t->key, key_len)) {
            casted = 1;
            omp_pack_str_with_body(pck, key, key_len);
            switch (t->type) {
            case T_INT:
                tmp_str = strndup(val, val_len);
                omp_pack_int64(pck, atoll(tmp_str));
                free(tmp_str);
                break;
            case T_HEX:
                tmp_str = strndup(val, val_len);
                omp_pack_uint64(pck, strtoull(tmp_str, NULL, 16));
                free(tmp_str);
                break;
            case T_FLOAT:
                tmp_str = strndup(val, val_len);
                omp_pack_double(pck, atof(tmp_str));
                free(tmp_str);
                break;
            case T_BOOL:
                if (val_len >= 4 && !strncasecmp(val, "true", 4)) omp_pack_bool(pck, 1);
                else if (val_len >= 5 && !strncasecmp(val, "false", 5)) omp_pack_bool(pck, 0);
                else error = 1;
                break;
            case T_STRING:
                omp_pack_str_with_body(pck, val, val_len);
                break;
            default:
                error = 1;
            }
            if (error) omp_pack_str_with_body(pck, val, val_len);
            break;
        }
    }
    if (!casted) {
        omp_pack_str_with_body(pck, key, key_len);
        omp_pack_str_with_body(pck, val, val_len);
    }
}

/*
 * src/flb_parser_regex.c:114-227 flb_parser_regex_do (+ cb_results :44-112, and the group
 * iteration of src/flb_regex.c:28-58,182-231,294-313).  Returns last byte consumed or -1.
 * *out is malloc'd.
 */
int ojson_pack(const char *js, size_t len, char **buffer, size_t *size, int *root_type, int *records, size_t *consumed);

/*
 * flb_parser_json_do: src/flb_parser_json.c:28-247.  JSON text -> msgpack (exactly one value, a map),
 * then the time key: the FIRST key equal to time_key (default "time") whose value is a string goes
 * through the time lookup; on success the pair is dropped unless time_keep, on failure the map is
 * left whole and the time is 0 (+ whatever fraction was parsed).  Decoders are not restated.
 * Returns the bytes consumed or -1.
 */
static int decoder_do(oflb_parser *parser, const char *in_buf, size_t in_size, char **out_buf, size_t *out_size);
static int oflb_parser_json_do(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size,
                               int64_t *out_sec, int64_t *out_nsec)
{
    char *mp = NULL;
    size_t mp_size = 0, consumed = 0, off = 0;
    int root_type = 0, records = 0, i, skip, map_size;
    omp_arena arena;
    omp_obj map;
    const char *time_key = parser->time_key ? parser->time_key : "time";
    int slen = (int) strlen(time_key);
    const omp_obj *k = NULL, *v = NULL;
    double tmfrac = 0;
    struct otm tm;
    time_t time_lookup_v;
    omp_buf nb;

    *out_sec = 0; *out_nsec = 0;
    if (ojson_pack(buf, length, &mp, &mp_size, &root_type, &records, &consumed) != 0) return -1;
    if (records != 1) { free(mp); return -1; }
    omp_arena_init(&arena);
    if (omp_unpack_next(&arena, &map, mp, mp_size, &off) != OMP_UNPACK_SUCCESS || map.type != OMP_MAP) {
        free(mp); omp_arena_free(&arena);
        return -1;
    }
    if (parser->ndecs > 0) {
        /* src/flb_parser_json.c:100-114: the decoders see the map before the time key is looked up in it */
        char *d = NULL;
        size_t dn = 0;
        if (decoder_do(parser, mp, mp_size, &d, &dn) == 0) {
            free(mp);
            mp = d; mp_size = dn;
            omp_arena_free(&arena);
            omp_arena_init(&arena);
            off = 0;
            omp_unpack_next(&arena, &map, mp, mp_size, &off);
        }
    }
    *out = mp; *out_size = mp_size;
    if (!parser->time_fmt) { omp_arena_free(&arena); return (int) consumed; }
    map_size = (int) map.via.map.size;
    skip = map_size;
    for (i = 0; i < map_size; i++) {
        k = &map.via.map.ptr[i].key;
        v = &map.via.map.ptr[i].val;
        if ((int) k->via.str.size != slen) { k = NULL; v = NULL; continue; }
        if (strncmp(k->via.str.ptr, time_key, k->via.str.size) == 0) {
            skip = parser->time_keep ? -1 : i;
            break;
        }
        k = NULL; v = NULL;
    }
    if (i >= map_size || !k || !v || v->type != OMP_STR) { omp_arena_free(&arena); return (int) consumed; }
    memset(&tm, 0, sizeof(tm));
    if (time_lookup(v->via.str.ptr, v->via.str.size, 0, parser, &tm, &tmfrac) == -1) {
        time_lookup_v = 0;
        skip = map_size;
    }
    else time_lookup_v = tm2time(parser, &tm);
    omp_buf_init(&nb);
    omp_pack_map(&nb, (!parser->time_keep && skip < map_size) ? map_size - 1 : map_size);
    for (i = 0; i < map_size; i++) {
        if (i == skip) continue;
        omp_pack_object(&nb, &map.via.map.ptr[i].key);
        omp_pack_object(&nb, &map.via.map.ptr[i].val);
    }
    free(mp);
    omp_arena_free(&arena);
    *out = nb.data; *out_size = nb.size;
    *out_sec = (int64_t) time_lookup_v;
    *out_nsec = (int64_t) (long) (tmfrac * 1000000000);
    return (int) consumed;
}


/* ------------------------------------------------------------------ Format logfmt / ltsv
 * flb_unescape_string_utf8 (src/flb_unescape.c:186-277) with u8_read_escape_sequence (:78-184) and
 * u8_wc_toutf8 (:40-64), restated.  `char` is signed there (x86-64): a byte >= 0x80 that is not part
 * of an escape becomes a code point >= 0xffffff80, which u8_wc_toutf8 refuses (0 bytes) and the byte
 * is copied as it is.  Returns the bytes written; out holds sz + 1 bytes.
 */
static int kv_is_oct(int c) { return c >= '0' && c <= '7'; }
static int kv_is_hex(int c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'F') || (c >= 'a' && c <= 'f'); }
static uint32_t kv_hexval(const char *d, int n)
{
    uint32_t v = 0;
    int i;
    for (i = 0; i < n; i++) {
        int c = (unsigned char) d[i];
        v = v * 16 + (uint32_t) (c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
    }
    return v;
}
static int kv_wc_toutf8(char *dest, uint32_t ch)
{
    if (ch < 0x80) { dest[0] = (char) ch; return 1; }
    if (ch < 0x800) { dest[0] = (char) ((ch >> 6) | 0xC0); dest[1] = (char) ((ch & 0x3F) | 0x80); return 2; }
    if (ch < 0x10000) {
        dest[0] = (char) ((ch >> 12) | 0xE0); dest[1] = (char) (((ch >> 6) & 0x3F) | 0x80); dest[2] = (char) ((ch & 0x3F) | 0x80);
        return 3;
    }
    if (ch < 0x110000) {
        dest[0] = (char) ((ch >> 18) | 0xF0); dest[1] = (char) (((ch >> 12) & 0x3F) | 0x80);
        dest[2] = (char) (((ch >> 6) & 0x3F) | 0x80); dest[3] = (char) ((ch & 0x3F) | 0x80);
        return 4;
    }
    return 0;
}
/* str points behind the backslash; returns the characters consumed from there */
static int kv_read_escape(const char *str, int size, uint32_t *dest)
{
    uint32_t ch = (uint32_t) (signed char) str[0];      /* the literal character */
    int i = 1, dno = 0;
    switch (str[0]) {
    case 'n': ch = '\n'; break;
    case 't': ch = '\t'; break;
    case 'r': ch = '\r'; break;
    case 'b': ch = '\b'; break;
    case 'f': ch = '\f'; break;
    case 'v': ch = '\v'; break;
    case 'a': ch = '\a'; break;
    default:
        if (kv_is_oct(str[0])) {
            uint32_t v = 0;
            i = 0;
            do { v = v * 8 + (uint32_t) (str[i++] - '0'); dno++; } while (i < size && kv_is_oct(str[i]) && dno < 3);
            ch = v;
        }
        else if (str[0] == 'x') {
            while (i < size && kv_is_hex(str[i]) && dno < 2) { i++; dno++; }
            if (dno > 0) ch = kv_hexval(str + 1, dno);
        }
        else if (str[0] == 'u') {
            while (i < size && kv_is_hex(str[i]) && dno < 4) { i++; dno++; }
            if (dno != 4 && dno > 0) { ch = 0xFFFD; break; }          /* incomplete */
            ch = kv_hexval(str + 1, dno);                            /* (no digit at all: strtol("") = 0) */
            if (ch >= 0xDC00 && ch <= 0xDFFF) ch = 0xFFFD;           /* low surrogate first */
            else if (ch >= 0xD800 && ch <= 0xDBFF) {
                if (i + 2 < size && str[i] == '\\' && str[i + 1] == 'u') {
                    int ls;
                    uint32_t low;
                    dno = 0;
                    i += 2;
                    ls = i;
                    while (i < size && kv_is_hex(str[i]) && dno < 4) { i++; dno++; }
                    if (dno != 4 && dno > 0) { ch = 0xFFFD; break; }
                    low = kv_hexval(str + ls, dno);
                    if (low >= 0xDC00 && low <= 0xDFFF) ch = 0x10000 + (((ch - 0xD800) << 10) | (low - 0xDC00));
                    else ch = 0xFFFD;
                }
                else ch = 0xFFFD;
            }
        }
        else if (str[0] == 'U') {
            while (i < size && kv_is_hex(str[i]) && dno < 8) { i++; dno++; }
            if (dno > 0) ch = kv_hexval(str + 1, dno);               /* strtol on <= 8 hex digits fits a long */
        }
    }
    *dest = ch;
    return i;
}
static int kv_unescape_utf8(const char *in_buf, int sz, char *out_buf)
{
    const char *end = in_buf + sz;
    int count_out = 0, count_in = 0;
    while (in_buf < end && *in_buf && count_in < sz) {
        const char *next = in_buf + 1;
        uint32_t ch;
        int esc_in, esc_out;
        char temp[4];
        if (next < end && *in_buf == '\\') {
            esc_in = 2;
            switch (*next) {
            case '"': ch = '"'; break;
            case '\'': ch = '\''; break;
            case '\\': ch = '\\'; break;
            case '/': ch = '/'; break;
            case 'n': ch = '\n'; break;
            case 'b': ch = '\b'; break;
            case 't': ch = '\t'; break;
            case 'f': ch = '\f'; break;
            case 'r': ch = '\r'; break;
            default: esc_in = kv_read_escape(next, (int) (end - next), &ch) + 1;
            }
        }
        else { ch = (uint32_t) (signed char) *in_buf; esc_in = 1; }
        in_buf += esc_in;
        count_in += esc_in;
        esc_out = kv_wc_toutf8(temp, ch);
        if (esc_out > sz - count_out) break;
        if (esc_out == 0) { out_buf[count_out] = (char) ch; esc_out = 1; }
        else memcpy(out_buf + count_out, temp, (size_t) esc_out);
        count_out += esc_out;
    }
    out_buf[count_out] = 0;
    return count_out;
}
/* exported for the pin against the reference's own flb_unescape.c (oracle/_ref/libunescape_ref.so) */
int oflb_unescape_utf8(const char *in_buf, int sz, char *out_buf) { return kv_unescape_utf8(in_buf, sz, out_buf); }

static int kv_ident(int c) { return c > ' ' && c != '=' && c != '"'; }                    /* flb_parser_logfmt.c:44-61 */
static int kv_label(int c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_' || c == '.' || c == '-'; }
static int kv_field(int c) { return c != 0 && c != '\t' && c != '\n' && c != '\r'; }      /* flb_parser_ltsv.c:43-79 */

/*
 * logfmt_parser / ltsv_parser (src/flb_parser_logfmt.c:63-240, src/flb_parser_ltsv.c:82-193): one walk
 * that either counts the pairs that will be packed (pck == NULL) or packs them.  Returns -1 when a
 * time value does not parse (strict), -2 for a bare key under Logfmt_No_Bare_Keys, else the bytes
 * consumed.
 */
static int kv_walk(oflb_parser *parser, const char *in_buf, size_t in_size, omp_buf *pck, const char *time_key, size_t time_key_len,
                   time_t *time_out, double *tmfrac, size_t *count)
{
    const unsigned char *c = (const unsigned char *) in_buf, *end = c + in_size;
    struct otm tm;
    memset(&tm, 0, sizeof(tm));
    while (c < end) {
        const unsigned char *key, *value = NULL;
        size_t key_len, value_len = 0;
        int value_set = 0, value_str = 0, value_escape = 0;
        if (parser->kv_format == KV_LOGFMT) {
            while (c < end && !kv_ident(*c)) c++;
            if (c == end) break;
            key = c;
            while (c < end && kv_ident(*c)) c++;
            key_len = (size_t) (c - key);
            if (c < end && *c == '=') {
                value_set = 1;
                c++;
                if (c < end) {
                    if (*c == '"') {
                        c++;
                        value = c;
                        value_str = 1;
                        while (c < end) {
                            if (*c != '\\' && *c != '"') c++;
                            else if (*c == '\\') {
                                value_escape = 1;
                                c++;
                                if (c == end) break;
                                c++;
                            }
                            else break;
                        }
                        value_len = (size_t) (c - value);
                        if (c < end && *c == '"') c++;
                    }
                    else {
                        value = c;
                        while (c < end && kv_ident(*c)) c++;
                        value_len = (size_t) (c - value);
                    }
                }
            }
        }
        else {
            key = c;
            while (c < end && kv_label(*c)) c++;
            key_len = (size_t) (c - key);
            if (c == end) break;
            if (*c != ':') break;
            c++;
            value = c;
            while (c < end && kv_field(*c)) c++;
            value_len = (size_t) (c - value);
        }
        if (key_len > 0) {
            int time_found = 0;
            if (parser->kv_format == KV_LOGFMT && parser->no_bare_keys && value_len == 0 && !value_set) return -2;
            if (parser->time_fmt && key_len == time_key_len && value_len > 0 && !strncmp((const char *) key, time_key, key_len)) {
                if (pck) {
                    if (time_lookup((const char *) value, value_len, 0, parser, &tm, tmfrac) == -1) return -1;
                    *time_out = tm2time(parser, &tm);
                }
                time_found = 1;
            }
            if (!time_found || parser->time_keep) {
                if (!pck) (*count)++;
                /* with Types every pair goes through flb_parser_typecast on the RAW value text: no "true" for an
                 * empty value, no unescaping (src/flb_parser_logfmt.c:176-182, src/flb_parser_ltsv.c:149-155) */
                else if (parser->types_len != 0) typecast(parser, (const char *) key, (int) key_len, (const char *) value, (int) value_len, pck);
                else {
                    omp_pack_str(pck, key_len);
                    omp_buf_write(pck, key, key_len);
                    if (parser->kv_format == KV_LOGFMT && value_len == 0) {
                        if (value_str) omp_pack_str(pck, 0);
                        else omp_pack_bool(pck, 1);
                    }
                    else if (value_escape) {
                        char *tmp = malloc(value_len + 1);
                        size_t n;
                        tmp[0] = 0;
                        kv_unescape_utf8((const char *) value, (int) value_len, tmp);
                        n = strlen(tmp);
                        omp_pack_str(pck, n);
                        omp_buf_write(pck, tmp, n);
                        free(tmp);
                    }
                    else {
                        omp_pack_str(pck, value_len);
                        omp_buf_write(pck, value, value_len);
                    }
                }
            }
        }
        if (c == end) break;
        if (parser->kv_format == KV_LTSV) {
            if (*c == '\t') c++;
            if (c == end) break;
        }
        if (*c == '\r') {
            c++;
            if (c == end) break;
            if (*c == '\n') c++;
            break;
        }
        if (*c == '\n') { c++; break; }
    }
    return (int) ((const char *) c - in_buf);
}

/* flb_parser_logfmt_do / flb_parser_ltsv_do (src/flb_parser_logfmt.c:242-325, src/flb_parser_ltsv.c:195-268);
 * Types (flb_parser_typecast per pair) and decoders are not restated. */
static int oflb_parser_kv_do(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size,
                             int64_t *out_sec, int64_t *out_nsec)
{
    const char *time_key = parser->time_key ? parser->time_key : "time";
    size_t time_key_len = strlen(time_key), map_size = 0;
    time_t tl = 0;
    double tmfrac = 0;
    omp_buf pck;
    int last;
    *out_sec = 0; *out_nsec = 0;
    if (kv_walk(parser, buf, length, NULL, time_key, time_key_len, &tl, &tmfrac, &map_size) == -2) return -1;
    if (map_size == 0) return -1;
    omp_buf_init(&pck);
    omp_pack_map(&pck, map_size);
    last = kv_walk(parser, buf, length, &pck, time_key, time_key_len, &tl, &tmfrac, &map_size);
    if (last < 0) { free(pck.data); return -1; }
    *out = pck.data; *out_size = pck.size;
    *out_sec = (int64_t) tl;
    *out_nsec = (int64_t) (long) (tmfrac * 1000000000);
    return last;
}

/* ------------------------------------------------------------------ Decode_Field / Decode_Field_As
 * src/flb_parser_decoder.c.  The string backends: flb_unescape_string (src/flb_unescape.c:278-335), flb_unescape_string_utf8
 * (above), flb_mysql_unquote_string (:338-388); the json backend: decode_json (:39-83) over flb_pack_json_recs. */
static int unescape_plain(const char *buf, int buf_len, char *p)
{
    int i = 0, j = 0;
    while (i < buf_len) {
        if (buf[i] == '\\') {
            if (i + 1 < buf_len) {
                char n = buf[i + 1];
                if (n == 'n') { p[j++] = '\n'; i++; }
                else if (n == 'a') { p[j++] = '\a'; i++; }
                else if (n == 'b') { p[j++] = '\b'; i++; }
                else if (n == 't') { p[j++] = '\t'; i++; }
                else if (n == 'v') { p[j++] = '\v'; i++; }
                else if (n == 'f') { p[j++] = '\f'; i++; }
                else if (n == 'r') { p[j++] = '\r'; i++; }
                else if (n == '\\') { p[j++] = '\\'; i++; }
                i++;
                continue;
            }
            else i++;                    /* (a trailing backslash: the byte behind the text is copied -- the NUL of the sds) */
        }
        p[j++] = buf[i++];
    }
    p[j] = '\0';
    return j;
}
static int mysql_unquote(const char *buf, int buf_len, char *p)
{
    int i = 0, j = 0;
    char n;
    while (i < buf_len) {
        if ((n = buf[i++]) != '\\') p[j++] = n;
        else if (i >= buf_len) p[j++] = n;
        else {
            n = buf[i++];
            switch (n) {
            case 'n': p[j++] = '\n'; break;
            case 'r': p[j++] = '\r'; break;
            case 't': p[j++] = '\t'; break;
            case '\\': p[j++] = '\\'; break;
            case '\'': p[j++] = '\''; break;
            case '\"': p[j++] = '\"'; break;
            case '0': p[j++] = 0; break;
            case 'Z': p[j++] = 0x1a; break;
            default: p[j++] = '\\'; p[j++] = n; break;
            }
        }
    }
    p[j] = '\0';
    return j;
}
/* exported for the pin against the reference's own flb_unescape.c (oracle/_ref/libunescape_ref.so) */
int oflb_unescape_plain(const char *in_buf, int sz, char *out_buf) { return unescape_plain(in_buf, sz, out_buf); }
int oflb_mysql_unquote(const char *in_buf, int sz, char *out_buf) { return mysql_unquote(in_buf, sz, out_buf); }

/* one backend on `in` (NUL-terminated copy of the current content); 0 and *o / *on / *otype (0 string, 1 object), or -1 */
static int dec_backend(int backend, const char *in, size_t in_size, char **o, size_t *on, int *otype)
{
    char *b;
    if (backend == DEC_JSON) {
        const char *p = in;
        int root_type = 0, records = 0;
        size_t consumed = 0, size = 0;
        char *mp = NULL;
        while (*p == ' ') p++;
        if (p[0] != '{' && p[0] != '[') return -1;
        if (ojson_pack(p, in_size - (size_t) (p - in), &mp, &size, &root_type, &records, &consumed) != 0) return -1;
        if (records != 1) { free(mp); return -1; }
        if (root_type != 1 /* JSMN_OBJECT */) { free(mp); return -1; }
        *o = mp; *on = size; *otype = 1;
        return 0;
    }
    b = malloc(in_size * 2 + 16);
    *otype = 0;
    if (backend == DEC_ESCAPED) *on = (size_t) unescape_plain(in, (int) in_size, b);
    else if (backend == DEC_ESCAPED_UTF8) *on = (size_t) kv_unescape_utf8(in, (int) in_size, b);
    else {
        /* decode_mysql_quoted (:114-147) */
        if (in_size < 2) { b[0] = in[0]; b[1] = 0; *on = in_size; }
        else if ((in[0] == '\'' && in[in_size - 1] == '\'') || (in[0] == '"' && in[in_size - 1] == '"'))
            *on = (size_t) mysql_unquote(in + 1, (int) in_size - 2, b);
        else { memcpy(b, in, in_size); b[in_size] = 0; *on = in_size; }
    }
    *o = b;
    return 0;
}

/* flb_parser_decoder_do (src/flb_parser_decoder.c:215-550) + merge_record_and_extra_keys (:149-209) */
static int decoder_do(oflb_parser *parser, const char *in_buf, size_t in_size, char **out_buf, size_t *out_size)
{
    omp_arena arena;
    omp_obj map;
    size_t off = 0;
    int matched = -1, extra_keys = 0, q;
    uint32_t i;
    omp_buf pck, extra;
    char *in_sds = NULL, *out_sds = NULL, *data = NULL;
    size_t in_len = 0, out_len = 0, data_len = 0;
    omp_arena_init(&arena);
    if (omp_unpack_next(&arena, &map, in_buf, in_size, &off) != OMP_UNPACK_SUCCESS || map.type != OMP_MAP) { omp_arena_free(&arena); return -1; }
    for (i = 0; i < map.via.map.size && matched < 0; i++) {
        const omp_obj *k = &map.via.map.ptr[i].key;
        if (k->type != OMP_STR) continue;
        for (q = 0; q < parser->ndecs; q++)
            if (parser->decs[q].key_len == k->via.str.size && memcmp(parser->decs[q].key, k->via.str.ptr, k->via.str.size) == 0) { matched = (int) i; break; }
    }
    if (matched == -1) { omp_arena_free(&arena); return -1; }
    omp_buf_init(&pck);
    memset(&extra, 0, sizeof(extra));
    omp_pack_map(&pck, map.via.map.size);
    for (i = 0; i < map.via.map.size; i++) {
        const omp_obj *k = &map.via.map.ptr[i].key, *v = &map.via.map.ptr[i].val;
        struct odec *dec = NULL;
        int is_decoded = 0, is_decoded_as = 0, in_type = 0, out_type = 0, ri;
        if ((int) i < matched || k->type != OMP_STR || v->type != OMP_STR) { omp_pack_object(&pck, k); omp_pack_object(&pck, v); continue; }
        for (q = 0; q < parser->ndecs; q++)
            if (parser->decs[q].key_len == k->via.str.size && memcmp(parser->decs[q].key, k->via.str.ptr, k->via.str.size) == 0) { dec = &parser->decs[q]; break; }
        if (!dec) { omp_pack_object(&pck, k); omp_pack_object(&pck, v); continue; }
        free(data);
        data = malloc(v->via.str.size + 1);
        memcpy(data, v->via.str.ptr, v->via.str.size);
        data[v->via.str.size] = 0;
        data_len = v->via.str.size;
        if (dec->add_extra_keys) {
            if (extra_keys) free(extra.data);
            extra_keys = 1;
            omp_buf_init(&extra);
        }
        for (ri = 0; ri < dec->nrules; ri++) {
            const struct odec_rule *rule = &dec->rules[ri];
            char *dbuf = NULL;
            size_t dsize = 0;
            int dtype = 0;
            if (rule->type == DEC_DEFAULT && rule->action == ACT_DO_NEXT && is_decoded) continue;
            if (is_decoded_as && in_type != 0) continue;
            if (dec_backend(rule->backend, data, data_len, &dbuf, &dsize, &dtype) == -1) {
                if (rule->action == ACT_TRY_NEXT || rule->action == ACT_DO_NEXT) continue;
                break;
            }
            if (rule->type == DEC_AS) {
                free(in_sds);
                in_sds = malloc(dsize + 1); memcpy(in_sds, dbuf, dsize); in_len = dsize;
                free(data);
                data = malloc(dsize + 1); memcpy(data, dbuf, dsize); data[dsize] = 0; data_len = dsize;
                in_type = dtype;
                is_decoded_as = 1;
            }
            else {
                free(out_sds);
                out_sds = malloc(dsize + 1); memcpy(out_sds, dbuf, dsize); out_len = dsize;
                out_type = dtype;
                is_decoded = 1;
            }
            free(dbuf);
            if (rule->action == ACT_DO_NEXT) continue;
            break;
        }
        omp_pack_object(&pck, k);
        if (is_decoded_as) {
            if (in_type == 0) omp_pack_str_with_body(&pck, in_sds, in_len);
            else omp_buf_write(&pck, in_sds, in_len);
        }
        else omp_pack_object(&pck, v);
        if (is_decoded && out_type == 1) omp_buf_write(&extra, out_sds, out_len);     /* (a string: "not allowed", logged only) */
    }
    free(in_sds); free(out_sds); free(data);
    *out_buf = pck.data; *out_size = pck.size;
    if (extra_keys) {
        omp_arena a2;
        omp_obj in_map, ex_map;
        size_t o1 = 0, o2 = 0;
        omp_arena_init(&a2);
        if (omp_unpack_next(&a2, &ex_map, extra.data, extra.size, &o2) == OMP_UNPACK_SUCCESS &&
            omp_unpack_next(&a2, &in_map, pck.data, pck.size, &o1) == OMP_UNPACK_SUCCESS) {
            omp_buf m;
            omp_buf_init(&m);
            omp_pack_map(&m, (size_t) in_map.via.map.size + ex_map.via.map.size);
            for (i = 0; i < in_map.via.map.size; i++) { omp_pack_object(&m, &in_map.via.map.ptr[i].key); omp_pack_object(&m, &in_map.via.map.ptr[i].val); }
            for (i = 0; i < ex_map.via.map.size; i++) { omp_pack_object(&m, &ex_map.via.map.ptr[i].key); omp_pack_object(&m, &ex_map.via.map.ptr[i].val); }
            free(pck.data);
            *out_buf = m.data; *out_size = m.size;
        }
        omp_arena_free(&a2);
        free(extra.data);
    }
    omp_arena_free(&arena);
    return 0;
}

static int parser_do_inner(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size,
                           int64_t *out_sec, int64_t *out_nsec);

/* the decoders run on the map a regex / logfmt / ltsv parser packed, after its time was taken (src/flb_parser_regex.c:210-221,
 * flb_parser_logfmt.c:314-324, flb_parser_ltsv.c:257-267); Format json applies them before its time lookup (oflb_parser_json_do) */
int oflb_parser_do(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size,
                   int64_t *out_sec, int64_t *out_nsec)
{
    int ret = parser_do_inner(parser, buf, length, out, out_size, out_sec, out_nsec);
    if (ret >= 0 && parser->ndecs > 0 && !parser->is_json) {
        char *d = NULL;
        size_t dn = 0;
        if (decoder_do(parser, *out, *out_size, &d, &dn) == 0) { free(*out); *out = d; *out_size = dn; }
    }
    return ret;
}

static int parser_do_inner(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size,
                           int64_t *out_sec, int64_t *out_nsec)
{
    if (parser->kv_format) return oflb_parser_kv_do(parser, buf, length, out, out_size, out_sec, out_nsec);
    if (parser->is_json) return oflb_parser_json_do(parser, buf, length, out, out_size, out_sec, out_nsec);
    int beg[ORX_MAX_GROUPS], end[ORX_MAX_GROUPS];
    int nregs, n, i, k, last_pos = -1, num_skipped = 0;
    time_t time_lookup_v = 0;
    double time_frac = 0;
    omp_buf pck;
    orx_t *rx;

    rx = parser->regex->rx;
    nregs = orx_search(rx, buf, (int) length, beg, end, ORX_MAX_GROUPS);
    if (nregs < 0) return -1;
    n = nregs - 1;                           /* flb_regex_do returns num_regs - 1 */
    if (n <= 0) return -1;                   /* flb_parser_regex_do: if (n <= 0) return -1 */

    omp_buf_init(&pck);
    omp_pack_map(&pck, n);

    for (i = 0; i < orx_num_names(rx); i++) {
        const char *name = orx_name(rx, i);
        int len = (int) strlen(name);
        for (k = 0; k < orx_name_ngroups(rx, i); k++) {
            int gn = orx_name_group(rx, i, k);
            const char *value = buf + beg[gn];
            size_t vlen = (size_t) (end[gn] - beg[gn]);
            int done = 0;
            /* cb_onig_named: last_pos updated after the callback when end >= 0 */
            /* ---- cb_results */
            if (vlen == 0 && parser->skip_empty) { num_skipped++; done = 1; }
            if (!done && parser->time_fmt) {
                const char *time_key = parser->time_key ? parser->time_key : "time";
                if (strcmp(name, time_key) == 0) {
                    struct otm tm;
                    double frac = 0;
                    memset(&tm, 0, sizeof(tm));
                    if (time_lookup(value, vlen, 0, parser, &tm, &frac) == -1) {
                        num_skipped++;
                        done = 1;
                    }
                    else {
                        time_frac = frac;
                        time_lookup_v = tm2time(parser, &tm);
                        if (!parser->time_keep) { num_skipped++; done = 1; }
                    }
                }
            }
            if (!done) {
                if (parser->types_len != 0) typecast(parser, name, len, value, (int) vlen, &pck);
                else {
                    omp_pack_str_with_body(&pck, name, len);
                    omp_pack_str_with_body(&pck, value, vlen);
                }
            }
            if (end[gn] >= 0) last_pos = end[gn];
        }
    }
    if (last_pos == -1) { omp_buf_free(&pck); return -1; }

    if (num_skipped > 0) {
        /* header patched in place, width kept (src/flb_parser_regex.c:182-199) */
        int arr_size = n - num_skipped;
        unsigned char *t = (unsigned char *) pck.data;
        unsigned char h = t[0];
        if (h >> 4 == 0x8) t[0] = (unsigned char) ((0x8 << 4) | (unsigned char) arr_size);
        else if (h == 0xde) { t[1] = (unsigned char) (arr_size >> 8); t[2] = (unsigned char) arr_size; }
        else if (h == 0xdf) { t[1] = (unsigned char) (arr_size >> 24); t[2] = (unsigned char) (arr_size >> 16);
                              t[3] = (unsigned char) (arr_size >> 8); t[4] = (unsigned char) arr_size; }
    }
    *out = pck.data;
    *out_size = pck.size;
    *out_sec = (int64_t) time_lookup_v;
    *out_nsec = (int64_t) (long) (time_frac * 1000000000);
    return last_pos;
}

/* ------------------------------------------------------------------ record accessor (subset) */
struct ra_sub { int is_index; int index; char *str; int len; };
typedef struct ora {
    char *key; int key_len;       /* NULL key => never matches ($TAG, $0.. forms) */
    struct ra_sub *subs; int nsubs;
} ora;

/* grammar: src/record_accessor/ra.l:54-67, ra.y:60-99 */
static ora *ora_create(const char *pat)
{
    ora *ra = calloc(1, sizeof(*ra));
    const char *p = pat, *q;
    if (*p != '$') { free(ra); return NULL; }
    p++;
    if (!((*p >= 'A' && *p <= 'Z') || (*p >= 'a' && *p <= 'z') || *p == '_')) { free(ra); return NULL; }
    q = p;
    while ((*q >= 'A' && *q <= 'Z') || (*q >= 'a' && *q <= 'z') || (*q >= '0' && *q <= '9') || *q == '_'
           || *q == '.' || *q == '-' || *q == '/') q++;
    ra->key = strndup(p, q - p);
    ra->key_len = (int) (q - p);
    p = q;
    while (*p == '[') {
        struct ra_sub s;
        memset(&s, 0, sizeof(s));
        p++;
        if (*p == '\'') {
            char *o;
            p++;
            s.str = malloc(strlen(p) + 1);
            o = s.str;
            for (;;) {
                if (*p == '\0') goto bad;
                if (*p == '\'') {
                    if (p[1] == '\'') { *o++ = '\''; p += 2; continue; }
                    p++;
                    break;
                }
                *o++ = *p++;
            }
            *o = 0;
            s.len = (int) (o - s.str);
        }
        else if (*p >= '0' && *p <= '9') {
            s.is_index = 1;
            s.index = atoi(p);
            while (*p >= '0' && *p <= '9') p++;
        }
        else goto bad;
        if (*p != ']') { free(s.str); goto bad; }
        p++;
        ra->subs = realloc(ra->subs, sizeof(s) * (ra->nsubs + 1));
        ra->subs[ra->nsubs++] = s;
    }
    if (*p != '\0') goto bad;
    return ra;
bad:
    free(ra->key); free(ra->subs); free(ra);
    return NULL;
}

static void ora_destroy(ora *ra)
{
    int i;
    if (!ra) return;
    for (i = 0; i < ra->nsubs; i++) free(ra->subs[i].str);
    free(ra->subs); free(ra->key); free(ra);
}

/* src/flb_ra_key.c:108-135: scans the map BACKWARDS, STR keys only */
static int ra_key_val_id(const char *ckey, int klen, const omp_obj *map)
{
    int i;
    if (map->type != OMP_MAP) return -1;
    for (i = (int) map->via.map.size - 1; i >= 0; i--) {
        const omp_obj *key = &map->via.map.ptr[i].key;
        if (key->type != OMP_STR) continue;
        if ((int) key->via.str.size != klen || memcmp(key->via.str.ptr, ckey, klen) != 0) continue;
        return i;
    }
    return -1;
}

/* src/flb_ra_key.c:151-236 */
static int subkey_to_object(const omp_obj *map, const ora *ra, const omp_obj **out_val)
{
    int i, levels = ra->nsubs, matched = 0, s;
    const omp_obj *val = NULL;
    omp_obj cur;
    if (levels == 0) return -1;
    cur = *map;
    for (s = 0; s < ra->nsubs; s++) {
        const struct ra_sub *e = &ra->subs[s];
        if (e->is_index) {
            if (cur.type != OMP_ARRAY) return -1;
            if (e->index == INT_MAX || (uint32_t) e->index >= cur.via.array.size) return -1;
            val = &cur.via.array.ptr[e->index];
            cur = *val;
            matched++;
            if (levels == matched) break;
            continue;
        }
        if (cur.type != OMP_MAP) break;
        i = ra_key_val_id(e->str, e->len, &cur);
        if (i == -1) continue;
        val = &cur.via.map.ptr[i].val;
        cur = *val;
        matched++;
        if (levels == matched) break;
    }
    if (matched == 0 || (matched > 0 && levels != matched)) return -1;
    *out_val = val;
    return 0;
}

/* src/flb_ra_key.c:374-434 with result == NULL, via src/flb_record_accessor.c:753-765 */
static int ra_regex_match(const ora *ra, const omp_obj *map, oflb_regex *regex)
{
    int i;
    const omp_obj *val, *out_val;
    if (ra->key == NULL) return -1;
    i = ra_key_val_id(ra->key, ra->key_len, map);
    if (i == -1) return -1;
    val = &map->via.map.ptr[i].val;
    if ((val->type == OMP_MAP || val->type == OMP_ARRAY) && ra->nsubs > 0) {
        if (subkey_to_object(val, ra, &out_val) == 0) {
            if (out_val->type != OMP_STR) return -1;
            return oflb_regex_match(regex, out_val->via.str.ptr, out_val->via.str.size);
        }
        return -1;
    }
    if (val->type != OMP_STR) return -1;
    return oflb_regex_match(regex, val->via.str.ptr, val->via.str.size);
}

/* flb_ra_get_value_object for the filter_parser '$key' form (src/flb_record_accessor.c:803-814,
 * src/flb_ra_key.c:238-300): value of key (+subkeys) or NULL */
static const omp_obj *ra_get_value(const ora *ra, const omp_obj *map)
{
    int i;
    const omp_obj *val, *out_val;
    if (ra->key == NULL) return NULL;
    i = ra_key_val_id(ra->key, ra->key_len, map);
    if (i == -1) return NULL;
    val = &map->via.map.ptr[i].val;
    if ((val->type == OMP_MAP || val->type == OMP_ARRAY) && ra->nsubs > 0) {
        if (subkey_to_object(val, ra, &out_val) == 0) return out_val;
        return NULL;
    }
    return val;
}

/* ------------------------------------------------------------------ filter_grep */
enum { GREP_REGEX = 1, GREP_EXCLUDE = 2 };
enum { OP_LEGACY = 0, OP_OR = 1, OP_AND = 2 };

struct grep_rule { int type; ora *ra; oflb_regex *regex; };
typedef struct oflb_grep { int logical_op; struct grep_rule *rules; int nrules; } oflb_grep;

void oflb_grep_destroy(oflb_grep *g)
{
    int i;
    if (!g) return;
    for (i = 0; i < g->nrules; i++) { ora_destroy(g->rules[i].ra); oflb_regex_destroy(g->rules[i].regex); }
    free(g->rules); free(g);
}

/* plugins/filter_grep/grep.c:56-164 set_rules + :196-248 cb_grep_init.
 * kinds[i] is the property key ("regex"/"exclude", case-insensitive), vals[i] "<key> <regex>" */
oflb_grep *oflb_grep_create(int n, const char **kinds, const char **vals, const char *logical_op)
{
    oflb_grep *g = calloc(1, sizeof(*g));
    int i, first_rule = 0;
    g->logical_op = OP_LEGACY;
    if (logical_op) {
        size_t len = strlen(logical_op);
        if (len == 3 && strncasecmp("AND", logical_op, 3) == 0) g->logical_op = OP_AND;
        else if (len == 2 && strncasecmp("OR", logical_op, 2) == 0) g->logical_op = OP_OR;
    }
    g->rules = calloc(n ? n : 1, sizeof(struct grep_rule));
    for (i = 0; i < n; i++) {
        struct grep_rule *r = &g->rules[g->nrules];
        const char *v = vals[i], *sp;
        char *field, *rafield;
        int flen;
        if (strcasecmp(kinds[i], "regex") == 0) r->type = GREP_REGEX;
        else if (strcasecmp(kinds[i], "exclude") == 0) r->type = GREP_EXCLUDE;
        else continue;
        if (g->logical_op != OP_LEGACY && first_rule != 0 && first_rule != r->type) goto fail;
        first_rule = r->type;
        /* flb_utils_split(val, ' ', 1): src/flb_utils.c:386-462 */
        while (*v == ' ') v++;
        sp = strchr(v, ' ');
        if (!sp || sp == v || sp[1] == '\0') goto fail;      /* need exactly 2 tokens */
        flen = (int) (sp - v);
        field = strndup(v, flen);
        if (field[0] == '$') rafield = strdup(field);
        else { rafield = malloc(flen + 2); rafield[0] = '$'; memcpy(rafield + 1, field, flen + 1); }
        free(field);
        r->ra = ora_create(rafield);
        free(rafield);
        if (!r->ra) goto fail;
        r->regex = oflb_regex_create(sp + 1);
        if (!r->regex) { ora_destroy(r->ra); r->ra = NULL; goto fail; }
        g->nrules++;
    }
    return g;
fail:
    oflb_grep_destroy(g);
    return NULL;
}

#define GREP_RET_KEEP 0
#define GREP_RET_EXCLUDE 1

/* plugins/filter_grep/grep.c:167-194 */
static int grep_filter_data(const omp_obj *map, oflb_grep *ctx)
{
    int i;
    for (i = 0; i < ctx->nrules; i++) {
        struct grep_rule *rule = &ctx->rules[i];
        int ret = ra_regex_match(rule->ra, map, rule->regex);
        if (ret <= 0) {
            if (rule->type == GREP_REGEX) return GREP_RET_EXCLUDE;
        }
        else {
            if (rule->type == GREP_EXCLUDE) return GREP_RET_EXCLUDE;
            return GREP_RET_KEEP;
        }
    }
    return GREP_RET_KEEP;
}

/* plugins/filter_grep/grep.c:250-284 */
static int grep_filter_data_and_or(const omp_obj *map, oflb_grep *ctx)
{
    int i, found = 0;
    struct grep_rule *rule = NULL;
    for (i = 0; i < ctx->nrules; i++) {
        int ra_ret;
        found = 0;
        rule = &ctx->rules[i];
        ra_ret = ra_regex_match(rule->ra, map, rule->regex);
        if (ra_ret > 0) found = 1;
        if (ctx->logical_op == OP_OR && found) break;
        if (ctx->logical_op == OP_AND && !found) break;
    }
    if (rule == NULL) return GREP_RET_KEEP;     /* no rules: the reference would deref an
                                                   uninitialised pointer; not reachable via config */
    if (rule->type == GREP_REGEX) return found ? GREP_RET_KEEP : GREP_RET_EXCLUDE;
    return found ? GREP_RET_EXCLUDE : GREP_RET_KEEP;
}

/* plugins/filter_grep/grep.c:286-392.  *out malloc'd on MODIFIED. */
int oflb_grep_filter(oflb_grep *ctx, const char *data, size_t bytes, char **out, size_t *out_size)
{
    oev_decoder dec;
    oev_event ev;
    omp_buf enc;
    int ret, old_size = 0, new_size = 0;
    oev_decoder_init(&dec, data, bytes);
    omp_buf_init(&enc);
    while ((ret = oev_decoder_next(&dec, &ev)) == OEV_SUCCESS) {
        old_size++;
        if (ctx->logical_op == OP_LEGACY) ret = grep_filter_data(ev.body, ctx);
        else ret = grep_filter_data_and_or(ev.body, ctx);
        if (ret == GREP_RET_KEEP) {
            omp_buf_write(&enc, ev.record_base, ev.record_length);
            new_size++;
        }
    }
    if (ret == OEV_ERR_INSUFFICIENT_DATA && dec.off == bytes) ret = 0;
    oev_decoder_destroy(&dec);
    if (old_size == new_size) { omp_buf_free(&enc); return FLB_FILTER_NOTOUCH; }
    if (ret == 0) {
        *out = enc.data;
        *out_size = enc.size;
        return FLB_FILTER_MODIFIED;
    }
    omp_buf_free(&enc);
    return FLB_FILTER_NOTOUCH;
}

/* ------------------------------------------------------------------ filter_parser */
typedef struct oflb_fparser {
    char *key_name; int key_name_len;
    ora *ra_key;
    int reserve_data, preserve_key;
    oflb_parser **parsers; int nparsers;
} oflb_fparser;

/* plugins/filter_parser/filter_parser.c:96-149; parsers are borrowed */
oflb_fparser *oflb_fparser_create(const char *key_name, int reserve_data, int preserve_key,
                                  int nparsers, oflb_parser **parsers)
{
    oflb_fparser *f;
    if (!key_name || nparsers == 0) return NULL;
    f = calloc(1, sizeof(*f));
    f->key_name = strdup(key_name);
    f->key_name_len = (int) strlen(key_name);
    f->reserve_data = reserve_data;
    f->preserve_key = preserve_key;
    if (key_name[0] == '$') {
        f->ra_key = ora_create(key_name);
        if (!f->ra_key) { free(f->key_name); free(f); return NULL; }
    }
    f->parsers = malloc(sizeof(*parsers) * nparsers);
    memcpy(f->parsers, parsers, sizeof(*parsers) * nparsers);
    f->nparsers = nparsers;
    return f;
}

void oflb_fparser_destroy(oflb_fparser *f)
{
    if (!f) return;
    ora_destroy(f->ra_key); free(f->key_name); free(f->parsers); free(f);
}

static int obj2char(const omp_obj *o, const char **s, int *n)
{
    if (o->type == OMP_STR || o->type == OMP_BIN) { *s = o->via.str.ptr; *n = (int) o->via.str.size; return 0; }
    return -1;
}

/* flb_time_is_valid_eventtime: include/fluent-bit/flb_time.h:88-97 */
static int valid_eventtime(const oev_time *t)
{
    if (t->sec < 0 || (uint64_t) t->sec > 0xffffffffULL || t->nsec < 0 || t->nsec >= 1000000000L) return 0;
    return 1;
}

/* plugins/filter_parser/filter_parser.c:174-442 */
int oflb_fparser_filter(oflb_fparser *ctx, const char *data, size_t bytes, char **out, size_t *out_size)
{
    oev_decoder dec;
    oev_event ev;
    omp_buf enc;
    int ret, parse_ret = -1;
    omp_arena tmp_arena;

    oev_decoder_init(&dec, data, bytes);
    omp_buf_init(&enc);
    omp_arena_init(&tmp_arena);

    while ((ret = oev_decoder_next(&dec, &ev)) == OEV_SUCCESS) {
        char *out_buf = NULL;
        size_t out_sz = 0;
        oev_time tm = ev.ts;
        const omp_obj *obj = ev.body;
        int map_num = (int) obj->via.map.size, i, p;
        const omp_kv **append_arr = NULL;
        size_t append_arr_len;
        int encoder_ok = 1;
        const char *val_str; int val_len;
        const char *key_str; int key_len;

        append_arr_len = ctx->reserve_data ? (size_t) map_num : 0;
        if (ctx->preserve_key && !ctx->reserve_data) append_arr_len = 1;
        if (append_arr_len > 0) {
            append_arr = calloc(append_arr_len, sizeof(*append_arr));
            if (ctx->reserve_data) for (i = 0; i < map_num; i++) append_arr[i] = &obj->via.map.ptr[i];
        }

        if (ctx->ra_key) {
            const omp_obj *rval = ra_get_value(ctx->ra_key, obj);
            if (rval && obj2char(rval, &val_str, &val_len) == 0) {
                for (p = 0; p < ctx->nparsers; p++) {
                    int64_t ps = 0, pn = 0;
                    char *ob = NULL; size_t os = 0;
                    parse_ret = oflb_parser_do(ctx->parsers[p], val_str, val_len, &ob, &os, &ps, &pn);
                    if (parse_ret >= 0) {
                        free(out_buf);           /* (the reference leaks here; bytes are the same) */
                        out_buf = ob; out_sz = os;
                        if ((uint64_t) ps * 1000000000ULL + (uint64_t) pn != 0) { tm.sec = ps; tm.nsec = pn; }
                        break;
                    }
                }
            }
        }
        else {
            for (i = 0; i < map_num; i++) {
                const omp_kv *kv = &obj->via.map.ptr[i];
                if (obj2char(&kv->key, &key_str, &key_len) < 0) continue;
                if (key_len == ctx->key_name_len && !strncmp(key_str, ctx->key_name, key_len)) {
                    if (obj2char(&kv->val, &val_str, &val_len) < 0) continue;
                    for (p = 0; p < ctx->nparsers; p++) {
                        int64_t ps = 0, pn = 0;
                        char *ob = NULL; size_t os = 0;
                        parse_ret = oflb_parser_do(ctx->parsers[p], val_str, val_len, &ob, &os, &ps, &pn);
                        if (parse_ret >= 0) {
                            free(out_buf);
                            out_buf = ob; out_sz = os;
                            if ((uint64_t) ps * 1000000000ULL + (uint64_t) pn != 0) { tm.sec = ps; tm.nsec = pn; }
                            if (append_arr != NULL) {
                                if (!ctx->preserve_key) append_arr[i] = NULL;
                                else if (!ctx->reserve_data) append_arr[0] = kv;
                            }
                            break;
                        }
                    }
                }
            }
        }

        /* encoder: begin_record / set_timestamp / set_metadata_from_msgpack_object */
        /* src/flb_log_event_encoder.c:345-363: outside the EventTime range the record fails -- except for the two values of the group
         * markers (-1 s, -2 s, no nanoseconds), which set_timestamp takes from anyone */
        if (!valid_eventtime(&tm) && !(tm.nsec == 0 && (tm.sec == -1 || tm.sec == -2))) encoder_ok = 0;

        if (out_buf != NULL && parse_ret >= 0) {
            if (append_arr != NULL && append_arr_len > 0) {
                size_t valid = 0, j;
                for (j = 0; j < append_arr_len; j++) if (append_arr[j] != NULL) valid++;
                if (valid > 0) {
                    /* flb_msgpack_expand_map: src/flb_pack.c:1664-1738 */
                    omp_obj m;
                    size_t off = 0;
                    omp_buf nb;
                    uint32_t q;
                    omp_arena_reset(&tmp_arena);
                    if (omp_unpack_next(&tmp_arena, &m, out_buf, out_sz, &off) != OMP_UNPACK_SUCCESS || m.type != OMP_MAP) {
                        free(out_buf); free(append_arr);
                        oev_decoder_destroy(&dec); omp_buf_free(&enc); omp_arena_free(&tmp_arena);
                        return FLB_FILTER_NOTOUCH;
                    }
                    omp_buf_init(&nb);
                    omp_pack_map(&nb, m.via.map.size + valid);
                    for (q = 0; q < m.via.map.size; q++) {
                        omp_pack_object(&nb, &m.via.map.ptr[q].key);
                        omp_pack_object(&nb, &m.via.map.ptr[q].val);
                    }
                    for (j = 0; j < append_arr_len; j++) {
                        if (append_arr[j] == NULL) continue;
                        omp_pack_object(&nb, &append_arr[j]->key);
                        omp_pack_object(&nb, &append_arr[j]->val);
                    }
                    free(out_buf);
                    out_buf = nb.data; out_sz = nb.size;
                }
            }
            if (encoder_ok) {
                /* emit_record direct path: src/flb_log_event_encoder.c:195-217 */
                unsigned char hdr[12];
                uint32_t s = (uint32_t) tm.sec, ns = (uint32_t) tm.nsec;
                hdr[0] = 0x92; hdr[1] = 0x92; hdr[2] = 0xd7; hdr[3] = 0x00;
                hdr[4] = s >> 24; hdr[5] = s >> 16; hdr[6] = s >> 8; hdr[7] = s;
                hdr[8] = ns >> 24; hdr[9] = ns >> 16; hdr[10] = ns >> 8; hdr[11] = ns;
                omp_buf_write(&enc, hdr, 12);
                omp_pack_object(&enc, ev.metadata);
                omp_buf_write(&enc, out_buf, out_sz);
            }
            free(out_buf);
        }
        else {
            free(out_buf);     /* reference leaks when a later duplicate key fails after a success */
            if (encoder_ok) {
                unsigned char hdr[12];
                uint32_t s = (uint32_t) tm.sec, ns = (uint32_t) tm.nsec;
                hdr[0] = 0x92; hdr[1] = 0x92; hdr[2] = 0xd7; hdr[3] = 0x00;
                hdr[4] = s >> 24; hdr[5] = s >> 16; hdr[6] = s >> 8; hdr[7] = s;
                hdr[8] = ns >> 24; hdr[9] = ns >> 16; hdr[10] = ns >> 8; hdr[11] = ns;
                omp_buf_write(&enc, hdr, 12);
                omp_pack_object(&enc, ev.metadata);
                omp_pack_object(&enc, ev.body);
            }
        }
        free(append_arr);
    }

    oev_decoder_destroy(&dec);
    omp_arena_free(&tmp_arena);
    if (enc.size > 0) {
        *out = enc.data;
        *out_size = enc.size;
        return FLB_FILTER_MODIFIED;
    }
    omp_buf_free(&enc);
    return FLB_FILTER_NOTOUCH;
}

/* ------------------------------------------------------------------ helpers for tests/bench */

/* ------------------------------------------------------------------ filter_log_to_metrics */
/*
 * plugins/filter_log_to_metrics/log_to_metrics.c restated:
 *   set_rules            :216-312   rule field handed VERBATIM to flb_ra_create (no '$' prefixing)
 *   grep_filter_data     :315-343   legacy rule evaluation
 *   set_labels           :355-497   kubernetes labels first, then label_field / add_label in order
 *   set_buckets          :540-595   strtod each "bucket", sort ascending
 *   cb_log_to_metrics_filter :970-1156
 * and the cmetrics arithmetic it drives:
 *   cmt_counter_inc      lib/cmetrics/src/cmt_counter.c:100-116  (+1.0 on an f64)
 *   cmt_gauge_set        lib/cmetrics/src/cmt_gauge.c
 *   cmt_histogram_observe lib/cmetrics/src/cmt_histogram.c:328-361, default buckets :89-95
 *   series identity      lib/cmetrics/src/cmt_map.c:377-452 (one series per label-value tuple, kept in
 *                        insertion order; the 64-bit label hash is treated as collision free)
 */
#define L2M_COUNTER 0
#define L2M_GAUGE 1
#define L2M_HISTOGRAM 2
#define L2M_MAX_LABEL_LENGTH 253      /* log_to_metrics.h:48 */
#define L2M_MAX_LABEL_COUNT 128       /* log_to_metrics.h:50 */

struct l2m_series {
    char **labels;
    double value;                 /* counter / gauge */
    uint64_t *buckets;            /* [nb + 1] cumulative, last = +Inf */
    uint64_t count;
    double sum;
};

typedef struct oflb_l2m {
    int mode;
    int discard_logs;
    struct grep_rule *rules; int nrules;
    int label_count;
    char **label_keys;
    ora **label_ras;              /* NULL entry: label stays empty */
    ora *value_ra;
    int nb; double *bounds;
    struct l2m_series *series; int nseries, cap;
    int static_set;               /* label_count == 0: the map's static metric has been touched */
} oflb_l2m;

/* first parser entry of flb_ra_create(str): src/flb_record_accessor.c:74-232.  Text before the first
 * '$' (or the whole string) is a STRING part whose name doubles as a top-level key
 * (src/record_accessor/flb_ra_parser.c:224-249); '$TAG' / '$0' entries carry no key. */
static ora *ora_create_first_part(const char *str, int *null_ok)
{
    const char *d = strchr(str, '$');
    ora *ra;
    *null_ok = 0;
    if (str[0] != '$') {
        size_t n = d ? (size_t) (d - str) : strlen(str);
        if (n == 0) return NULL;
        ra = calloc(1, sizeof(*ra));
        ra->key = strndup(str, n);
        ra->key_len = (int) n;
        return ra;
    }
    if (str[1] == '\0') return NULL;
    if ((str[1] >= '0' && str[1] <= '9') || strncmp(str + 1, "TAG", 3) == 0) {
        *null_ok = 1;                         /* rp->key == NULL: lookups fail at run time */
        return calloc(1, sizeof(*ra));
    }
    {
        /* segment end: '.', ' ', ',', '"' outside quotes (:175-186) */
        int quote = 0;
        size_t end;
        char *seg;
        for (end = 1; str[end]; end++) {
            if (str[end] == '\'') quote++;
            else if (str[end] == '.' && (quote & 1)) continue;
            else if (str[end] == '.' || str[end] == ' ' || str[end] == ',' || str[end] == '"') break;
        }
        seg = strndup(str, end);
        ra = ora_create(seg);
        free(seg);
        return ra;
    }
}

void oflb_l2m_destroy(oflb_l2m *c)
{
    int i, j;
    if (!c) return;
    for (i = 0; i < c->nrules; i++) { ora_destroy(c->rules[i].ra); oflb_regex_destroy(c->rules[i].regex); }
    free(c->rules);
    for (i = 0; i < c->label_count; i++) { free(c->label_keys[i]); ora_destroy(c->label_ras[i]); }
    free(c->label_keys); free(c->label_ras);
    ora_destroy(c->value_ra);
    free(c->bounds);
    for (i = 0; i < c->nseries; i++) {
        for (j = 0; j < c->label_count; j++) free(c->series[i].labels[j]);
        free(c->series[i].labels); free(c->series[i].buckets);
    }
    free(c->series);
    free(c);
}

/* props: the filter's properties in configuration order as (key, value) pairs; the keys that matter
 * are regex / exclude / label_field / add_label / bucket (all matched with strcasecmp, like the
 * mk_list_foreach loops of set_rules/set_labels/set_buckets). */
oflb_l2m *oflb_l2m_create(const char *mode, int nprops, const char **keys, const char **vals,
                          int kubernetes_mode, const char *value_field, int discard_logs)
{
    static const char *k8s[5] = { "namespace_name", "pod_name", "container_name", "docker_id", "pod_id" };
    oflb_l2m *c = calloc(1, sizeof(*c));
    int i, n, null_ok;
    c->discard_logs = discard_logs;
    /* :731-749 */
    if (mode == NULL || strcasecmp(mode, "counter") == 0) c->mode = L2M_COUNTER;
    else if (strcasecmp(mode, "gauge") == 0) c->mode = L2M_GAUGE;
    else if (strcasecmp(mode, "histogram") == 0) c->mode = L2M_HISTOGRAM;
    else goto fail;
    /* set_rules */
    c->rules = calloc(nprops ? nprops : 1, sizeof(struct grep_rule));
    for (i = 0; i < nprops; i++) {
        struct grep_rule *r = &c->rules[c->nrules];
        const char *v = vals[i], *sp;
        char *field;
        if (strcasecmp(keys[i], "regex") == 0) r->type = GREP_REGEX;
        else if (strcasecmp(keys[i], "exclude") == 0) r->type = GREP_EXCLUDE;
        else continue;
        while (*v == ' ') v++;
        sp = strchr(v, ' ');
        if (!sp || sp == v || sp[1] == '\0') goto fail;
        field = strndup(v, sp - v);
        r->ra = ora_create_first_part(field, &null_ok);
        free(field);
        if (!r->ra) goto fail;
        r->regex = oflb_regex_create(sp + 1);
        if (!r->regex) { ora_destroy(r->ra); r->ra = NULL; goto fail; }
        c->nrules++;
    }
    /* set_labels */
    n = kubernetes_mode ? 5 : 0;
    for (i = 0; i < nprops; i++)
        if (strcasecmp(keys[i], "label_field") == 0 || strcasecmp(keys[i], "add_label") == 0) n++;
    if (n > L2M_MAX_LABEL_COUNT) goto fail;
    c->label_keys = calloc(n ? n : 1, sizeof(char *));
    c->label_ras = calloc(n ? n : 1, sizeof(ora *));
    if (kubernetes_mode) {
        for (i = 0; i < 5; i++) {
            char fmt[64];
            snprintf(fmt, sizeof(fmt), "$kubernetes['%s']", k8s[i]);
            c->label_keys[c->label_count] = strdup(k8s[i]);
            c->label_ras[c->label_count++] = ora_create_first_part(fmt, &null_ok);
        }
    }
    for (i = 0; i < nprops; i++) {
        if (strcasecmp(keys[i], "label_field") == 0) {
            c->label_keys[c->label_count] = strdup(vals[i]);
            c->label_ras[c->label_count++] = ora_create_first_part(vals[i], &null_ok);
        }
        else if (strcasecmp(keys[i], "add_label") == 0) {
            const char *v = vals[i], *sp;
            while (*v == ' ') v++;
            sp = strchr(v, ' ');
            if (!sp || sp == v || sp[1] == '\0') goto fail;
            c->label_keys[c->label_count] = strndup(v, sp - v);
            c->label_ras[c->label_count++] = ora_create_first_part(sp + 1, &null_ok);
        }
    }
    /* value_field: only for gauge / histogram (:794-808) */
    if (c->mode > 0) {
        if (!value_field || !*value_field) goto fail;
        c->value_ra = ora_create_first_part(value_field, &null_ok);
        if (!c->value_ra) goto fail;
    }
    /* set_buckets + defaults (:811-822) */
    if (c->mode == L2M_HISTOGRAM) {
        int nbk = 0, j;
        for (i = 0; i < nprops; i++) if (strcasecmp(keys[i], "bucket") == 0) nbk++;
        if (nbk == 0) {
            static const double def[11] = { 0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0 };
            c->nb = 11;
            c->bounds = malloc(sizeof(def));
            memcpy(c->bounds, def, sizeof(def));
        }
        else {
            c->bounds = calloc(nbk, sizeof(double));
            for (i = 0; i < nprops; i++) {
                char *end;
                if (strcasecmp(keys[i], "bucket") != 0) continue;
                c->bounds[c->nb] = strtod(vals[i], &end);
                if (end == vals[i]) goto fail;
                c->nb++;
            }
            for (i = 0; i < c->nb - 1; i++)
                for (j = 0; j < c->nb - i - 1; j++)
                    if (c->bounds[j] > c->bounds[j + 1]) { double t = c->bounds[j]; c->bounds[j] = c->bounds[j + 1]; c->bounds[j + 1] = t; }
        }
    }
    return c;
fail:
    oflb_l2m_destroy(c);
    return NULL;
}

static struct l2m_series *l2m_get_series(oflb_l2m *c, char **vals)
{
    int i, j;
    struct l2m_series *s;
    for (i = 0; i < c->nseries; i++) {
        for (j = 0; j < c->label_count; j++) if (strcmp(c->series[i].labels[j], vals[j]) != 0) break;
        if (j == c->label_count) return &c->series[i];
    }
    if (c->nseries == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->series = realloc(c->series, c->cap * sizeof(*s)); }
    s = &c->series[c->nseries++];
    memset(s, 0, sizeof(*s));
    s->labels = calloc(c->label_count ? c->label_count : 1, sizeof(char *));
    for (j = 0; j < c->label_count; j++) s->labels[j] = strdup(vals[j]);
    if (c->mode == L2M_HISTOGRAM) s->buckets = calloc(c->nb + 1, sizeof(uint64_t));
    return s;
}

/* value of an accessor as the callback sees it: 0 none, 1 string, 2 float, 3 int, 4 other */
static int l2m_value(const ora *ra, const omp_obj *map, char **str, double *f64, int64_t *i64)
{
    const omp_obj *o;
    if (!ra || map->type != OMP_MAP) return 0;
    o = ra_get_value(ra, map);
    if (!o) return 0;
    switch (o->type) {
    case OMP_STR:
        *str = malloc(o->via.str.size + 1);             /* flb_sds_create_len: NUL terminated copy */
        memcpy(*str, o->via.str.ptr, o->via.str.size);
        (*str)[o->via.str.size] = 0;
        return 1;
    case OMP_F64: case OMP_F32: *f64 = o->via.f64; return 2;
    case OMP_POS: case OMP_NEG: *i64 = o->via.i64; return 3;
    case OMP_BOOL: case OMP_MAP: case OMP_BIN: case OMP_NIL: return 4;
    default: return 0;                                   /* array / ext: msgpack_object_to_ra_value == -1 */
    }
}

int oflb_l2m_filter(oflb_l2m *c, const char *data, size_t bytes)
{
    omp_arena arena;
    omp_obj root;
    size_t off = 0;
    double gauge_value = 0, histogram_value = 0;
    char **vals = calloc(c->label_count ? c->label_count : 1, sizeof(char *));
    char *buf = calloc(c->label_count ? c->label_count : 1, L2M_MAX_LABEL_LENGTH);
    int i;
    for (i = 0; i < c->label_count; i++) vals[i] = buf + (size_t) i * L2M_MAX_LABEL_LENGTH;
    omp_arena_init(&arena);
    for (;;) {
        const omp_obj *map;
        omp_obj nomap;
        int keep = 1;
        omp_arena_reset(&arena);
        if (omp_unpack_next(&arena, &root, data, bytes, &off) != OMP_UNPACK_SUCCESS) break;
        if (root.type != OMP_ARRAY) continue;
        /* map = root.via.array.ptr[1]; an array shorter than 2 is read out of bounds by the
         * reference -- restated as "not a map" (every accessor then fails) */
        if (root.via.array.size >= 2) map = &root.via.array.ptr[1];
        else { memset(&nomap, 0, sizeof(nomap)); nomap.type = OMP_NIL; map = &nomap; }
        /* grep_filter_data :315-343 */
        for (i = 0; i < c->nrules; i++) {
            struct grep_rule *r = &c->rules[i];
            int ret = map->type == OMP_MAP ? ra_regex_match(r->ra, map, r->regex) : -1;
            if (ret <= 0) { if (r->type == GREP_REGEX) { keep = 0; break; } }
            else { keep = (r->type != GREP_EXCLUDE); break; }
        }
        if (!keep) continue;
        for (i = 0; i < c->label_count; i++) {
            char *str = NULL; double f = 0; int64_t iv = 0;
            int t;
            vals[i][0] = '\0';
            t = l2m_value(c->label_ras[i], map, &str, &f, &iv);
            if (t == 1) { snprintf(vals[i], L2M_MAX_LABEL_LENGTH - 1, "%s", str); free(str); }
            else if (t == 2) snprintf(vals[i], L2M_MAX_LABEL_LENGTH - 1, "%f", f);
            else if (t == 3) snprintf(vals[i], L2M_MAX_LABEL_LENGTH - 1, "%ld", (long) iv);
        }
        if (c->mode == L2M_COUNTER) {
            struct l2m_series *s = l2m_get_series(c, vals);
            s->value += 1.0;
            c->static_set = 1;
        }
        else {
            char *str = NULL; double f = 0; int64_t iv = 0;
            double *slot = c->mode == L2M_GAUGE ? &gauge_value : &histogram_value;
            int t = l2m_value(c->value_ra, map, &str, &f, &iv);
            struct l2m_series *s;
            if (t == 0 || t == 4) continue;             /* missing / cannot convert: no update */
            if (t == 1) { sscanf(str, "%lf", slot); free(str); }   /* failure keeps the previous value */
            else if (t == 2) *slot = f;
            else *slot = (double) iv;
            s = l2m_get_series(c, vals);
            c->static_set = 1;
            if (c->mode == L2M_GAUGE) s->value = *slot;
            else {
                /* cmt_histogram_observe */
                int b;
                for (b = c->nb - 1; b >= 0; b--) {
                    if (*slot > c->bounds[b]) break;
                    s->buckets[b]++;
                }
                s->buckets[c->nb]++;
                s->count++;
                s->sum += *slot;
            }
        }
    }
    omp_arena_free(&arena);
    free(buf); free(vals);
    return c->discard_logs ? FLB_FILTER_MODIFIED : FLB_FILTER_NOTOUCH;
}

int oflb_l2m_info(oflb_l2m *c, int *label_count, int *nbuckets, double *bounds)
{
    int i;
    *label_count = c->label_count;
    *nbuckets = c->nb;
    if (bounds) for (i = 0; i < c->nb; i++) bounds[i] = c->bounds[i];
    return c->nseries;
}
const char *oflb_l2m_label_key(oflb_l2m *c, int i) { return c->label_keys[i]; }
const char *oflb_l2m_series_label(oflb_l2m *c, int s, int i) { return c->series[s].labels[i]; }
int oflb_l2m_series_get(oflb_l2m *c, int s, double *value, uint64_t *buckets, uint64_t *count, double *sum)
{
    int i;
    *value = c->series[s].value;
    *count = c->series[s].count;
    *sum = c->series[s].sum;
    if (buckets && c->series[s].buckets) for (i = 0; i <= c->nb; i++) buckets[i] = c->series[s].buckets[i];
    return 0;
}

void oflb_free(void *p) { free(p); }

int oflb_count_records(const char *data, size_t bytes) { return oev_count_records(data, bytes); }

/* canonical re-pack of one object (msgpack_pack_object) -- used by parity tests */
int oflb_repack(const char *data, size_t bytes, char **out, size_t *out_size)
{
    omp_arena a;
    omp_obj o;
    omp_buf b;
    size_t off = 0;
    omp_arena_init(&a);
    omp_buf_init(&b);
    while (omp_unpack_next(&a, &o, data, bytes, &off) == OMP_UNPACK_SUCCESS) omp_pack_object(&b, &o);
    omp_arena_free(&a);
    *out = b.data; *out_size = b.size;
    return (int) off;
}

/* timing loops (clock_gettime(CLOCK_MONOTONIC), as benchmarks/pack_json.c:138-166) */
double oflb_bench_fparser(oflb_fparser *ctx, const char *data, size_t bytes, int iters, size_t *out_bytes)
{
    struct timespec t0, t1;
    int i;
    size_t total = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < iters; i++) {
        char *o = NULL; size_t os = 0;
        if (oflb_fparser_filter(ctx, data, bytes, &o, &os) == FLB_FILTER_MODIFIED) { total += os; free(o); }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *out_bytes = total;
    return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

double oflb_bench_grep(oflb_grep *ctx, const char *data, size_t bytes, int iters, size_t *out_bytes)
{
    struct timespec t0, t1;
    int i;
    size_t total = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < iters; i++) {
        char *o = NULL; size_t os = 0;
        if (oflb_grep_filter(ctx, data, bytes, &o, &os) == FLB_FILTER_MODIFIED) { total += os; free(o); }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *out_bytes = total;
    return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/* test hook: flb_parser_time_lookup + flb_parser_tm2time as driven by
 * tests/internal/parser.c:228-290 (test_parser_time_lookup).  toff_enc < 0 keeps the parser's
 * own offset, else (toff_enc - 2^20) replaces it for this call. */
int oflb_time_lookup(oflb_parser *p, const char *s, size_t len, int64_t now, int toff_enc,
                     int64_t *sec, double *frac)
{
    struct otm tm;
    int saved = p->time_offset, r;
    memset(&tm, 0, sizeof(tm));
    if (toff_enc >= 0) p->time_offset = toff_enc - (1 << 20);
    r = time_lookup(s, len, (time_t) now, p, &tm, frac);
    p->time_offset = saved;
    if (r == 0) *sec = (int64_t) tm2time(p, &tm);
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * in_tail: the buffer of a tailed file cut into lines, every line one log event
 * plugins/in_tail/tail_file.c:689-1040 (process_content, the plain path: no multiline, no parser, no docker mode, no
 * truncate_long_lines, no encoding conversion) + :552-604 (flb_tail_file_pack_line) + the encoder's layout for a record built
 * with begin_record / append_body_values (src/flb_log_event_encoder.c:195-217, src/flb_mp.c:591-640: map32 headers for metadata
 * and body).  The reference stamps every record with "now" (flb_time_get); here the time is a parameter.
 * Returns the number of lines; *processed = bytes consumed (what stays in the file's buffer starts there). */
int oflb_tail_process(const char *buf, size_t len, const char *key, const char *path_key, const char *path, const char *offset_key,
                      uint64_t stream_offset, int skip_empty_lines, uint32_t sec, uint32_t nsec, char **out, size_t *out_size, uint64_t *processed_out)
{
    omp_buf b;
    const char *d = buf, *end = buf + len, *nl;
    uint64_t processed = 0;
    int lines = 0;
    uint32_t nbody = 1u + (path_key ? 1u : 0u) + (offset_key ? 1u : 0u);
    omp_buf_init(&b);
    while (d < end && *d == '\0') { d++; processed++; }                                  /* :783-786 */
    while (d < end && (nl = memchr(d, '\n', (size_t) (end - d)))) {                       /* :840 */
        size_t ll = (size_t) (nl - d), line_len;
        int crlf = 0;
        uint8_t h[22];
        if (skip_empty_lines) {                                                          /* :863-874 */
            if (ll == 0) { d++; processed++; continue; }
            else if (ll == 1 && d[0] == '\r') { d += 2; processed += 2; continue; }
        }
        if (ll >= 2) crlf = (d[ll - 1] == '\r');                                         /* :877-884 */
        line_len = ll - (size_t) crlf;
        /* 92 92 d7 00 <sec> <nsec> | df 00 00 00 00 | df 00 00 00 nn */
        h[0] = 0x92; h[1] = 0x92; h[2] = 0xd7; h[3] = 0x00;
        h[4] = (uint8_t) (sec >> 24); h[5] = (uint8_t) (sec >> 16); h[6] = (uint8_t) (sec >> 8); h[7] = (uint8_t) sec;
        h[8] = (uint8_t) (nsec >> 24); h[9] = (uint8_t) (nsec >> 16); h[10] = (uint8_t) (nsec >> 8); h[11] = (uint8_t) nsec;
        h[12] = 0xdf; h[13] = h[14] = h[15] = h[16] = 0;
        h[17] = 0xdf; h[18] = h[19] = h[20] = 0; h[21] = (uint8_t) nbody;
        omp_buf_write(&b, h, 22);
        if (path_key) { omp_pack_str_with_body(&b, path_key, strlen(path_key)); omp_pack_str_with_body(&b, path, strlen(path)); }      /* :565-573 */
        if (offset_key) { omp_pack_str_with_body(&b, offset_key, strlen(offset_key)); omp_pack_uint64(&b, stream_offset + processed); }  /* :576-587 */
        omp_pack_str_with_body(&b, key, strlen(key));                                   /* :589-595 */
        omp_pack_str_with_body(&b, d, line_len);
        d += ll + 1; processed += ll + 1; lines++;                                      /* :985-992 */
    }
    *out = b.data; *out_size = b.size; *processed_out = processed;
    return lines;
}
