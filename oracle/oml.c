/*
 * oracle/oml.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's multiline core for the path in_tail drives
 * (text lines, ONE multiline parser, no sub-parser): the regex rule state machine, the endswith / equal types, the stream
 * group buffer with its separator / truncation rules, and the record a flush produces.  Never linked into the product; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Follows, line by line in arrival order:
 *   src/multiline/flb_ml.c:685-762   flb_ml_append_text (try the parser; a line nobody takes flushes the stream and leaves alone)
 *   src/multiline/flb_ml.c:197-364   package_content (REGEX / ENDSWITH / EQ; when the time of the group is registered)
 *   src/multiline/flb_ml.c:179-195   breakline_prepare
 *   src/multiline/flb_ml_rule.c:48-118,199-243  flb_ml_rule_create, set_to_state_map (rule order kept)
 *   src/multiline/flb_ml_rule.c:245-277 try_flushing_buffer, :301-327 try_start_state, :329-436 flb_ml_rule_process
 *   src/multiline/flb_ml_group.c:87-122 flb_ml_group_cat (buffer_limit, truncation)
 *   src/multiline/flb_ml.c:1590-1790 flb_ml_flush_stream_group (text mode: {key_content | "log": buffer}, metadata marker of a truncated
 *                                    group, what a flush resets and what it does not: rule_to_state and mp_time stay)
 *   src/multiline/flb_ml_parser_{java,go,python,ruby}.c  the rule tables of the built-in regex parsers (oml_builtin)
 *   plugins/in_tail/tail_file.c:783-786,840-898,985-992  the line loop in front (oml_tail_chunk)
 * Pinned on the reference itself: tests/test_multiline_oracle.py runs the same frames through oracle/_ref/ref_filters kind 5 (the
 * reference's own src/multiline/*.c compiled in place) and wants identical bytes, and checks the vectors of tests/internal/multiline.c.
 *
 *   src/multiline/flb_ml.c:505-532 ml_append_try_parser_type_text, :403-503 process_append (TYPE_MAP), :366-401 get_key_id,
 *   src/multiline/flb_ml_stream.c:91-133 flb_ml_stream_group_get, flb_ml_parser_cri.c / _docker.c   a sub-parser in front of an endswith /
 *                                    equal parser: the line is parsed first, key_content / key_pattern / key_group come from its map, one
 *                                    buffer per key_group value, the first line's map re-packed with the concatenation at the flush
 * Not restated (the product refuses the same configurations): a sub-parser in front of a REGEX parser, several parsers in one context
 * (flb_ml_append_text's LRU), flb_ml_append_object.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "omp.h"

typedef struct oflb_regex oflb_regex;
typedef struct oflb_parser oflb_parser;
oflb_parser *oflb_parser_create(const char *regex, int skip_empty, const char *time_fmt, const char *time_key, const char *time_offset, int time_keep,
                                int time_strict, const char *types_str);
void oflb_parser_destroy(oflb_parser *p);
int oflb_parser_do(oflb_parser *parser, const char *buf, size_t length, char **out, size_t *out_size, int64_t *out_sec, int64_t *out_nsec);
oflb_regex *oflb_regex_create(const char *pattern);
void oflb_regex_destroy(oflb_regex *r);
int oflb_regex_match(oflb_regex *r, const char *s, size_t len);

enum { OML_REGEX = 0, OML_ENDSWITH = 1, OML_EQ = 2 };      /* flb_ml.h FLB_ML_REGEX / ENDSWITH / EQ */
#define OML_MAX_RULES 64
#define OML_MAX_STATES 16
#define OML_MAX_GROUPS 6                   /* FLB_ML_MAX_GROUPS */

struct oml_rule {
    char *from[OML_MAX_STATES];
    int nfrom, start_state;
    char *to_state;
    oflb_regex *rx;
    int map[OML_MAX_RULES], nmap;                           /* to_state_map: rule indexes, in rule order */
};

typedef struct oml {
    int type, negate;
    char *match_str, *key_content;
    size_t buffer_limit;
    struct oml_rule rules[OML_MAX_RULES];
    int nrules;
    /* the stream groups: [0] "_default", then one per key_group value in order of appearance (flb_ml_stream.c:91-133, at most 6) */
    struct oml_group {
        char *name; size_t name_len;
        char *buf; size_t len, cap;                          /* flb_ml_stream_group.buf */
        char *map; size_t map_len;                           /* mp_sbuf: the first line's map (sub-parser path) */
        int64_t t_sec, t_nsec;                               /* mp_time */
        int truncated;
    } groups[OML_MAX_GROUPS], *g;                            /* g: the group at work */
    int ngroups;
    int rule_to_state;                                       /* -1: none (regex parsers use the default group only) */
    int64_t now_sec, now_nsec;                               /* "now" for a group that never saw a time */
    oflb_parser *sub;                                        /* the parser in front (cri: regex, docker: json) */
    char *key_group, *key_pattern;
    /* in_tail's file buffer */
    char *pend; size_t pend_len;
    /* flushed records */
    char *out; size_t out_len, out_cap; int records, truncations;
    /* a list of parsers on one stream (flb_ml.c:671-760, in_tail's `multiline.parser a, b`): the head holds the others in order; what
     * any of them flushes lands in the head's output */
    struct oml *chain[7]; int nchain; int lru;               /* lru: index into {head, chain...}, -1 none (flb_ml_group.lru_parser) */
    struct oml *sink;
} oml;

static void cat_raw(oml *m, const char *d, size_t n)        /* flb_sds_cat_safe */
{
    if (m->g->len + n + 1 > m->g->cap) { m->g->cap = (m->g->len + n + 1) * 2 + 64; m->g->buf = realloc(m->g->buf, m->g->cap); }
    memcpy(m->g->buf + m->g->len, d, n);
    m->g->len += n;
}

static void out_put(oml *m, const void *d, size_t n)
{
    if (m->sink) m = m->sink;
    if (m->out_len + n > m->out_cap) { m->out_cap = (m->out_len + n) * 2 + 4096; m->out = realloc(m->out, m->out_cap); }
    memcpy(m->out + m->out_len, d, n);
    m->out_len += n;
}
static void out_u8(oml *m, unsigned v) { unsigned char c = (unsigned char) v; out_put(m, &c, 1); }
static void out_be(oml *m, uint64_t v, int n) { int i; for (i = n - 1; i >= 0; i--) out_u8(m, (unsigned) (v >> (8 * i)) & 255); }
static void out_str_hdr(oml *m, size_t n)                   /* msgpack_pack_str */
{
    if (n < 32) out_u8(m, 0xa0 | (unsigned) n);
    else if (n < 256) { out_u8(m, 0xd9); out_u8(m, (unsigned) n); }
    else if (n < 65536) { out_u8(m, 0xda); out_be(m, n, 2); }
    else { out_u8(m, 0xdb); out_be(m, n, 4); }
}

oml *oml_create(int type, const char *match_str, int negate, const char *key_content, int64_t buffer_limit)
{
    oml *m = calloc(1, sizeof(*m));
    m->type = type; m->negate = negate ? 1 : 0;
    m->match_str = match_str ? strdup(match_str) : NULL;
    m->key_content = key_content && key_content[0] ? strdup(key_content) : NULL;
    m->buffer_limit = buffer_limit >= 0 ? (size_t) buffer_limit : 2 * 1024 * 1024;   /* flb_ml.c:896-902, FLB_ML_BUFFER_LIMIT_DEFAULT */
    m->rule_to_state = -1;
    m->lru = -1;
    m->ngroups = 1; m->groups[0].name = strdup("_default"); m->groups[0].name_len = 8;
    m->g = &m->groups[0];
    return m;
}

void oml_destroy(oml *m)
{
    int i, k;
    if (!m) return;
    for (i = 0; i < m->nrules; i++) {
        for (k = 0; k < m->rules[i].nfrom; k++) free(m->rules[i].from[k]);
        free(m->rules[i].to_state);
        oflb_regex_destroy(m->rules[i].rx);
    }
    for (i = 0; i < m->ngroups; i++) { free(m->groups[i].name); free(m->groups[i].buf); free(m->groups[i].map); }
    if (m->sub) oflb_parser_destroy(m->sub);
    free(m->match_str); free(m->key_content); free(m->key_group); free(m->key_pattern); free(m->pend); free(m->out); free(m);
}

void oml_set_now(oml *m, int64_t sec, int64_t nsec) { m->now_sec = sec; m->now_nsec = nsec; }

/* flb_ml_rule.c:48-118 + flb_slist_split_string(',') with its space trimming */
int oml_add_rule(oml *m, const char *from_states, const char *regex, const char *to_state)
{
    struct oml_rule *r;
    const char *p = from_states;
    if (m->nrules >= OML_MAX_RULES) return -1;
    r = &m->rules[m->nrules];
    memset(r, 0, sizeof(*r));
    while (*p) {
        const char *e = strchr(p, ','), *a, *b;
        if (!e) e = p + strlen(p);
        a = p; b = e;
        while (a < b && *a == ' ') a++;
        while (b > a && b[-1] == ' ') b--;
        if (b > a) {
            if (r->nfrom >= OML_MAX_STATES) return -1;
            r->from[r->nfrom] = strndup(a, (size_t) (b - a));
            if (!strcmp(r->from[r->nfrom], "start_state")) r->start_state = 1;
            r->nfrom++;
        }
        p = *e ? e + 1 : e;
    }
    if (r->nfrom == 0) return -1;                                    /* :73-77 */
    if (!r->start_state && m->nrules == 0) return -1;                /* :83-87 the first rule must hold a start_state */
    r->rx = oflb_regex_create(regex);
    if (!r->rx) return -1;
    r->to_state = to_state && to_state[0] ? strdup(to_state) : NULL;
    m->nrules++;
    return 0;
}

/* flb_ml_rule.c:199-243 set_to_state_map for every rule (flb_ml_rule_init :279-299) */
int oml_init(oml *m)
{
    int i, j, k;
    for (i = 0; i < m->nrules; i++) {
        struct oml_rule *r = &m->rules[i];
        int exists = 0;
        r->nmap = 0;
        if (!r->to_state) continue;
        for (j = 0; j < m->nrules && !exists; j++)
            for (k = 0; k < m->rules[j].nfrom; k++) if (!strcmp(m->rules[j].from[k], r->to_state)) { exists = 1; break; }
        if (!exists) return -1;                                       /* "to_state='%s' is not registered" */
        for (j = 0; j < m->nrules; j++)
            for (k = 0; k < m->rules[j].nfrom; k++)
                if (!strcmp(m->rules[j].from[k], r->to_state)) { r->map[r->nmap++] = j; break; }
    }
    return 0;
}

/* the rule tables of src/multiline/flb_ml_parser_java.c:59-128, _go.c:59-125, _python.c:60-83, _ruby.c:59-71 */
int oml_builtin(oml *m, const char *name)
{
    static const char *java[][3] = {
        {"start_state, java_start_exception", "/(.)(?:Exception|Error|Throwable|V8 errors stack trace)[:\\r\\n]/", "java_after_exception"},
        {"java_after_exception", "/^[\\t ]*nested exception is:[\\t ]*/", "java_start_exception"},
        {"java_after_exception", "/^[\\r\\n]*$/", "java_after_exception"},
        {"java_after_exception, java", "/^[\\t ]+(?:eval )?at /", "java"},
        {"java_after_exception, java", "/^[\\t ]+--- End of inner exception stack trace ---$/", "java"},
        {"java_after_exception, java", "/^--- End of stack trace from previous (?x:)location where exception was thrown ---$/", "java"},
        {"java_after_exception, java", "/^[\\t ]*(?:Caused by|Suppressed):/", "java_after_exception"},
        {"java_after_exception, java", "/^[\\t ]*... \\d+ (?:more|common frames omitted)/", "java"}, {0, 0, 0}};
    static const char *go[][3] = {
        {"start_state", "/\\bpanic: /", "go_after_panic"},
        {"start_state", "/http: panic serving/", "go_goroutine"},
        {"go_after_panic", "/^$/", "go_goroutine"},
        {"go_after_panic, go_after_signal, go_frame_1", "/^$/", "go_goroutine"},
        {"go_after_panic", "/^\\[signal /", "go_after_signal"},
        {"go_goroutine", "/^goroutine \\d+ \\[[^\\]]+\\]:$/", "go_frame_1"},
        {"go_frame_1", "/^(?:[^\\s.:]+\\.)*[^\\s.():]+\\(|^created by /", "go_frame_2"},
        {"go_frame_2", "/^\\s/", "go_frame_1"}, {0, 0, 0}};
    static const char *python[][3] = {
        {"start_state", "/^Traceback \\(most recent call last\\):$/", "python"},
        {"python", "/^[\\t ]+File /", "python_code"},
        {"python_code", "/[^\\t ]/", "python"},
        {"python", "/^(?:[^\\s.():]+\\.)*[^\\s.():]+:/", "start_state"}, {0, 0, 0}};
    static const char *ruby[][3] = {
        {"start_state, ruby_start_exception", "/^.+:\\d+:in\\s+.*/", "ruby_after_exception"},
        {"ruby_after_exception, ruby", "/^\\s+from\\s+.*:\\d+:in\\s+.*/", "ruby"}, {0, 0, 0}};
    const char *(*t)[3] = !strcmp(name, "java") ? java : !strcmp(name, "go") ? go : !strcmp(name, "python") ? python : !strcmp(name, "ruby") ? ruby : NULL;
    int i;
    if (!t) return -1;
    for (i = 0; t[i][0]; i++) if (oml_add_rule(m, t[i][0], t[i][1], t[i][2]) != 0) return -1;
    return oml_init(m);
}

/* flb_ml_group.c:87-122: 0 ok, 1 truncated */
static int group_cat(oml *m, const char *d, size_t n)
{
    int status = 0;
    if (m->buffer_limit > 0) {
        size_t avail;
        if (m->g->len >= m->buffer_limit) { m->g->truncated = 1; return 1; }
        avail = m->buffer_limit - m->g->len;
        if (n > avail) { n = avail; m->g->truncated = 1; status = 1; }
    }
    if (n) cat_raw(m, d, n);
    return status;
}

/* flb_ml.c:1590-1790, text mode (the group holds no first-line map) */
/* the time a flush stamps (:1619-1624) and whether the encoder takes it (flb_log_event_encoder_set_timestamp:
 * src/flb_log_event_encoder.c:345-363 -- a parsed time that is no time, e.g. tm2time of an all-zero struct tm, is refused) */
static int flush_time_ok(const oml *m)
{
    int64_t sec = m->g->t_sec, nsec = m->g->t_nsec;
    if (sec == 0 && nsec == 0) { sec = m->now_sec; nsec = m->now_nsec; }
    return sec >= 0 && sec <= 0xFFFFFFFFll && nsec >= 0 && nsec < 1000000000ll;
}

static void out_record_head(oml *m)
{
    int64_t sec = m->g->t_sec, nsec = m->g->t_nsec;
    if (sec == 0 && nsec == 0) { sec = m->now_sec; nsec = m->now_nsec; }                           /* :1619-1624 */
    out_u8(m, 0x92); out_u8(m, 0x92); out_u8(m, 0xd7); out_u8(m, 0x00);                             /* [[ext 0 (sec, nsec), metadata], body] */
    out_be(m, (uint64_t) sec, 4); out_be(m, (uint64_t) nsec, 4);
    out_u8(m, 0xdf); out_be(m, m->g->truncated ? 1 : 0, 4);                                        /* the encoder's map32 */
    if (m->g->truncated) { out_str_hdr(m, 19); out_put(m, "multiline_truncated", 19); out_u8(m, 0xc3); }   /* :1733-1738 */
}

/* flb_ml.c:1590-1790 on the group at work */
static void flush_group(oml *m)
{
    if (!m->key_content && m->g->len > 0 && m->g->buf[m->g->len - 1] != '\n') cat_raw(m, "\n", 1);      /* breakline_prepare :179-195 */
    if (m->g->map_len > 0) {
        /* :1626-1690 the first line's map, the value of key_content replaced by the buffer (an empty buffer: the map as it is) */
        omp_arena ar;
        omp_obj map;
        size_t off = 0;
        omp_buf body;
        omp_arena_init(&ar);
        omp_buf_init(&body);
        if (omp_unpack_next(&ar, &map, m->g->map, m->g->map_len, &off) == OMP_UNPACK_SUCCESS && map.type == OMP_MAP) {
            if (m->g->len > 0) {
                const size_t klen = m->key_content ? strlen(m->key_content) : 0;
                uint32_t i;
                omp_pack_map(&body, map.via.map.size);
                for (i = 0; i < map.via.map.size; i++) {
                    const omp_obj *k = &map.via.map.ptr[i].key, *v = &map.via.map.ptr[i].val;
                    omp_pack_object(&body, k);
                    if (k->type == OMP_STR && m->key_content && k->via.str.size == klen && !strncmp(k->via.str.ptr, m->key_content, klen))
                        omp_pack_str_with_body(&body, m->g->buf, m->g->len);
                    else omp_pack_object(&body, v);
                }
            }
            else omp_pack_object(&body, &map);
            if (!flush_time_ok(m)) {
                /* "[multiline] error packing event": the function returns before it empties the buffer (:1744-1752) -- the first-line
                 * map is gone (:1687), the bytes stay and the next line of the group becomes its first line */
                omp_buf_free(&body);
                omp_arena_free(&ar);
                m->g->map_len = 0;
                return;
            }
            out_record_head(m);
            out_put(m, body.data, body.size);
            (m->sink ? m->sink : m)->records++;
        }
        omp_buf_free(&body);
        omp_arena_free(&ar);
        m->g->map_len = 0;
    }
    else if (m->g->len > 0) {
        const char *key = m->key_content ? m->key_content : "log";
        if (!flush_time_ok(m)) return;
        out_record_head(m);
        out_u8(m, 0x81); out_str_hdr(m, strlen(key)); out_put(m, key, strlen(key));
        out_str_hdr(m, m->g->len); out_put(m, m->g->buf, m->g->len);
        (m->sink ? m->sink : m)->records++;
    }
    m->g->len = 0;
    m->g->truncated = 0;
}

static void register_time(oml *m, int64_t sec, int64_t nsec) { m->g->t_sec = sec; m->g->t_nsec = nsec; }     /* flb_ml_register_context, no map */

/* flb_ml_rule.c:329-436: 0 processed, 1 truncated, -1 no rule takes the line */
static int rule_process(oml *m, const char *d, size_t n, int64_t sec, int64_t nsec)
{
    int rule = -1, i;
    if (m->rule_to_state >= 0) {
        const struct oml_rule *cur = &m->rules[m->rule_to_state];
        for (i = 0; i < cur->nmap; i++) {
            const struct oml_rule *c = &m->rules[cur->map[i]];
            if (c->start_state) continue;
            if (oflb_regex_match(c->rx, d, n)) {
                if (m->g->len >= 1 && m->g->buf[m->g->len - 1] != '\n') cat_raw(m, "\n", 1);
                if (n == 0) cat_raw(m, "\n", 1);
                else if (group_cat(m, d, n) == 1) {
                    flush_group(m);                          /* "Buffer is full. Flush immediately to send the truncated record." */
                    m->rule_to_state = -1;
                    return 1;
                }
                rule = cur->map[i];
                break;
            }
        }
    }
    if (rule < 0) {
        for (i = 0; i < m->nrules; i++)                       /* try_start_state */
            if (m->rules[i].start_state && oflb_regex_match(m->rules[i].rx, d, n)) { rule = i; break; }
        if (rule >= 0) {
            if (m->g->len > 0) flush_group(m);
            m->rule_to_state = rule;
            if (group_cat(m, d, n) == 1) return 1;
            register_time(m, sec, nsec);
        }
    }
    if (rule >= 0) {
        const struct oml_rule *r = &m->rules[rule];
        int next_start = 0;
        m->rule_to_state = rule;
        for (i = 0; i < r->nmap; i++) if (m->rules[r->map[i]].start_state) { next_start = 1; break; }      /* try_flushing_buffer */
        if (next_start && m->g->len > 0) flush_group(m);
        return 0;
    }
    return -1;
}

static int match_negate(const oml *m, int matched) { return m->negate ? !matched : matched; }

/* flb_ml.c:366-401 get_key_id: the first entry whose key is the string `name` and whose value is a string */
static const omp_obj *map_str_value(const omp_obj *map, const char *name)
{
    uint32_t i;
    const size_t len = name ? strlen(name) : 0;
    if (!name) return NULL;
    for (i = 0; i < map->via.map.size; i++) {
        const omp_obj *k = &map->via.map.ptr[i].key, *v = &map->via.map.ptr[i].val;
        if (k->type != OMP_STR || v->type != OMP_STR) continue;
        if (k->via.str.size != len) continue;
        if (!strncmp(k->via.str.ptr, name, len)) return v;
    }
    return NULL;
}

/* the parser in front of an ENDSWITH / EQ parser: the cri regex, docker's json (flb_ml_parser_cri.c, _docker.c: time_keep on) */
int oml_set_subparser(oml *m, const char *regex, const char *time_fmt, const char *time_key, int skip_empty, const char *key_group, const char *key_pattern)
{
    if (m->type == OML_REGEX) return -1;
    m->sub = oflb_parser_create(regex && regex[0] ? regex : NULL, skip_empty, time_fmt, time_key, NULL, 1, 0, NULL);
    if (!m->sub) return -1;
    m->key_group = key_group && key_group[0] ? strdup(key_group) : NULL;
    m->key_pattern = key_pattern && key_pattern[0] ? strdup(key_pattern) : NULL;
    return 0;
}

/* ml_append_try_parser_type_text + process_append (TYPE_MAP) + package_content for ENDSWITH / EQ: 0 processed, -1 nobody takes the line */
static int process_with_subparser(oml *m, int64_t sec, int64_t nsec, const char *d, size_t n)
{
    char *pm = NULL;
    size_t pn = 0, off = 0;
    int64_t psec = 0, pnsec = 0;
    omp_arena ar;
    omp_obj map;
    const omp_obj *vc, *vp, *vg, *val;
    int ret = -1, rule_match, i;
    if (oflb_parser_do(m->sub, d, n, &pm, &pn, &psec, &pnsec) < 0) return -1;                    /* :518-531 */
    if (psec == 0 && pnsec == 0) { psec = sec; pnsec = nsec; }                                      /* :521-523, :642-649 */
    omp_arena_init(&ar);
    if (omp_unpack_next(&ar, &map, pm, pn, &off) != OMP_UNPACK_SUCCESS || map.type != OMP_MAP) goto done;
    vc = map_str_value(&map, m->key_content);                                                      /* :455-466 */
    if (!vc) goto done;
    vp = map_str_value(&map, m->key_pattern);
    vg = map_str_value(&map, m->key_group);
    /* flb_ml_stream_group_get (flb_ml_stream.c:91-133) */
    m->g = &m->groups[0];
    if (m->key_group && vg) {
        for (i = 0; i < m->ngroups; i++)
            if (m->groups[i].name_len == vg->via.str.size && !memcmp(m->groups[i].name, vg->via.str.ptr, vg->via.str.size)) break;
        if (i == m->ngroups) {
            if (m->ngroups >= OML_MAX_GROUPS) { m->g = &m->groups[0]; goto done; }                   /* (the reference walks into a NULL group here) */
            m->groups[i].name = malloc(vg->via.str.size + 1);
            memcpy(m->groups[i].name, vg->via.str.ptr, vg->via.str.size);
            m->groups[i].name_len = vg->via.str.size;
            m->ngroups++;
        }
        m->g = &m->groups[i];
    }
    val = vp ? vp : vc;
    if (m->type == OML_ENDSWITH) {
        const size_t len = m->match_str ? strlen(m->match_str) : 0;
        if (len > val->via.str.size) goto done;                                                    /* processed stays FLB_FALSE: -1 on this path (:498-500) */
        rule_match = match_negate(m, memcmp(val->via.str.ptr + (val->via.str.size - len), m->match_str, len) == 0);
    }
    else {
        const size_t len = m->match_str ? strlen(m->match_str) : 0;
        rule_match = match_negate(m, val->via.str.size == len && memcmp(val->via.str.ptr, m->match_str, len) == 0);
    }
    if (m->g->map_len == 0) {                                                                      /* flb_ml_register_context: the time and the map of the first line */
        omp_buf fb;
        omp_buf_init(&fb);
        omp_pack_object(&fb, &map);
        free(m->g->map);
        m->g->map = fb.data; m->g->map_len = fb.size;
        m->g->t_sec = psec; m->g->t_nsec = pnsec;
    }
    if (!m->key_content && m->g->len > 0 && m->g->buf[m->g->len - 1] != '\n') cat_raw(m, "\n", 1);      /* breakline_prepare */
    cat_raw(m, vc->via.str.ptr, vc->via.str.size);
    if (rule_match) flush_group(m);
    ret = 0;
done:
    omp_arena_free(&ar);
    free(pm);
    m->g = &m->groups[0];
    return ret;
}

/* flb_ml_append_text with one parser instance; returns 1 when the line truncated a buffer */
/* ml_append_try_parser (flb_ml.c:591-669): >= 0 the parser took the line (1: truncated), -1 it did not */
static int try_parser(oml *m, int64_t sec, int64_t nsec, const char *d, size_t n)
{
    int ret = -1;
    m->g = &m->groups[0];
    if (m->sub) ret = process_with_subparser(m, sec, nsec, d, n);
    else if (m->type == OML_REGEX) {
        ret = rule_process(m, d, n, sec, nsec);
        if (ret == 0) register_time(m, sec, nsec);           /* package_content :263-265 (the text path's first-line map is always empty) */
    }
    else if (m->type == OML_ENDSWITH) {
        const size_t len = m->match_str ? strlen(m->match_str) : 0;
        ret = 0;                                              /* a line shorter than the string: nobody says -1 (:266 processed stays false) */
        if (len <= n) {
            const int rule_match = match_negate(m, memcmp(d + (n - len), m->match_str, len) == 0);
            register_time(m, sec, nsec);
            if (!m->key_content && m->g->len > 0 && m->g->buf[m->g->len - 1] != '\n') cat_raw(m, "\n", 1);
            cat_raw(m, d, n);
            if (rule_match) flush_group(m);
        }
    }
    else {
        const size_t len = m->match_str ? strlen(m->match_str) : 0;
        const int rule_match = match_negate(m, n == len && memcmp(d, m->match_str, n) == 0);
        ret = 0;
        register_time(m, sec, nsec);
        if (!m->key_content && m->g->len > 0 && m->g->buf[m->g->len - 1] != '\n') cat_raw(m, "\n", 1);
        cat_raw(m, d, n);
        if (rule_match) flush_group(m);
    }
    return ret;
}

static void flush_all_groups(oml *m)
{
    int i;
    for (i = 0; i < m->ngroups; i++) { m->g = &m->groups[i]; flush_group(m); }
    m->g = &m->groups[0];
}

/* another parser behind the head's (in_tail: one flb_ml_parser_instance_create per name of `multiline.parser`) */
int oml_chain_add(oml *head, oml *next)
{
    if (head->nchain >= 7 || next->sink || next->nchain) return -1;
    head->chain[head->nchain++] = next;
    next->sink = head;
    return 0;
}

int oml_append_text(oml *m, int64_t sec, int64_t nsec, const char *d, size_t n)
{
    int ret = -1, truncated = 0, i;
    const int np = 1 + m->nchain;
    /* flb_ml_append_text :686-728: the parser that took the stream's last line first, then the others in order */
    if (m->lru >= 0) {
        oml *p = m->lru == 0 ? m : m->chain[m->lru - 1];
        ret = try_parser(p, sec, nsec, d, n);
    }
    for (i = 0; ret < 0 && i < np; i++) {
        oml *p = i == 0 ? m : m->chain[i - 1];
        if (i == m->lru) continue;
        ret = try_parser(p, sec, nsec, d, n);
        if (ret >= 0) m->lru = i;
    }
    if (ret == 1) truncated = 1;
    if (ret < 0) {
        /* :729-757: "A non-matching line breaks any multiline sequence" (every parser's groups of the stream, in the order they were
         * created), then the line alone through the FIRST parser's default group */
        for (i = 0; i < np; i++) flush_all_groups(i == 0 ? m : m->chain[i - 1]);
        m->g = &m->groups[0];
        register_time(m, sec, nsec);
        if (group_cat(m, d, n) == 1) truncated = 1;
        flush_group(m);
    }
    if (truncated) m->truncations++;
    return truncated;
}

void oml_flush_pending(oml *m)                               /* flb_ml_flush_pending(_now): the timer's forced flush, every parser's every group */
{
    int i;
    flush_all_groups(m);
    for (i = 0; i < m->nchain; i++) flush_all_groups(m->chain[i]);
}

/* plugins/in_tail/tail_file.c process_content: what one read appends to the file's buffer */
void oml_tail_chunk(oml *m, const char *text, size_t bytes, int skip_empty, int64_t sec, int64_t nsec)
{
    char *d, *end, *nl;
    m->pend = realloc(m->pend, m->pend_len + bytes + 1);
    memcpy(m->pend + m->pend_len, text, bytes);
    m->pend_len += bytes;
    d = m->pend; end = m->pend + m->pend_len;
    while (d < end && *d == '\0') d++;                                     /* :783-786 */
    while (d < end && (nl = memchr(d, '\n', (size_t) (end - d)))) {       /* :840 */
        size_t ll = (size_t) (nl - d);
        int crlf = 0;
        if (skip_empty) {                                                  /* :863-874 */
            if (ll == 0) { d++; continue; }
            else if (ll == 1 && d[0] == '\r') { d += 2; continue; }
        }
        if (ll >= 2) crlf = d[ll - 1] == '\r';                             /* :877-884 */
        oml_append_text(m, sec, nsec, d, ll - (size_t) crlf);
        d += ll + 1;
    }
    m->pend_len = (size_t) (end - d);
    memmove(m->pend, d, m->pend_len);
}

/* the records flushed so far (the caller copies); resets the output */
size_t oml_output(oml *m, const char **out, int *records, int *truncations)
{
    size_t n = m->out_len;
    *out = m->out; *records = m->records; *truncations = m->truncations;
    m->out_len = 0; m->records = 0; m->truncations = 0;
    return n;
}

/* what the stream carries between calls: rule_to_state (-1 none), buffered bytes, unconsumed tail of the file buffer */
void oml_state(const oml *m, int *rule_to_state, size_t *buffered, size_t *pending)
{
    int i;
    int k;
    *rule_to_state = m->rule_to_state; *buffered = 0; *pending = m->pend_len;
    for (i = 0; i < m->ngroups; i++) *buffered += m->groups[i].len;
    for (k = 0; k < m->nchain; k++) for (i = 0; i < m->chain[k]->ngroups; i++) *buffered += m->chain[k]->groups[i].len;
}
