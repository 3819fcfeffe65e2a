/* ref_sp_shim.c -- TEST INFRASTRUCTURE.  Drives the REAL stream processor of the reference -- src/stream_processor/flb_sp.c
 * (flb_sp_task_create, sp_process_data_aggr :1435, sp_process_aggregate_data :1280, package_results :1161, flb_sp_window_prune),
 * flb_sp_aggregate_func.c, flb_sp_groupby.c, flb_sp_key.c, flb_sp_window.c, flb_sp_func_*.c, parser/flb_sp_parser.c and
 * lib/rbtree -- compiled from where they lie (oracle/Makefile, _ref/ref_sp: an executable for the reason given at ref_filters).
 *
 * What is NOT the reference's: (1) the two files flex / bison generate from parser/sql.l / sql.y (neither tool is in this image):
 * the grammar is written out by hand below -- same token rules (caseless keywords, INTEGER through atoi into an int, FLOATING
 * through atof into a *float*, ''-escaped strings, identifiers [_A-Za-z][A-Za-z0-9_.]*), same builder calls in the same order
 * (an alias / sub-key list is handed over before flb_sp_cmd_key_add, as the mid-rule reductions do), and bison's default
 * conflict resolution for the precedence-less condition rules: shift, i.e. AND / OR associate to the right and NOT takes
 * everything after it; (2) the engine around a task: the two drivers are the ones tests/internal/stream_processor.c:60-150
 * defines for itself (flb_sp_do_test / flb_sp_fd_event_test), the timer fd is a constant, flb_time_get() is pinned
 * (-Wl,--wrap) so that the packaged records are reproducible.
 *
 * Protocol (stdin -> stdout, binary): ops until EOF
 *   op 1: u32 str_conv, u32 sec, u32 nsec, str sql      -> i32 0 / -1 (task not created)
 *   op 2: u64 len, chunk bytes                         -> i32 ret (window.records), u64 out_len, out (DEFAULT window: packaged now)
 *   op 3: (timer of the window fires)                  -> i32 0, u64 out_len, out
 *   op 4: destroy the task                             -> i32 0
 *   op 5: (the hop timer of a HOPPING window fires)    -> i32 ret of sp_process_hopping_slot (flb_sp_fd_event, the window.fd_hop
 *         branch, src/stream_processor/flb_sp.c:2170-2185) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_mem.h>
#include <fluent-bit/flb_sds.h>
#include <fluent-bit/flb_str.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_log.h>
#include <fluent-bit/flb_worker.h>
#include <fluent-bit/flb_slist.h>
#include <fluent-bit/flb_time.h>
#include <fluent-bit/stream_processor/flb_sp.h>
#include <fluent-bit/stream_processor/flb_sp_parser.h>
#include <fluent-bit/stream_processor/flb_sp_window.h>

/* ---- the logger: the worker context stays NULL and the print hooks do nothing */
FLB_TLS_DEFINE(struct flb_worker, flb_worker_ctx);
void flb_log_print(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; }
int flb_log_is_truncated(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; return 0; }
int flb_errno_print(int errnum, const char *file, int line) { (void) errnum; (void) file; (void) line; return 0; }
struct flb_worker *flb_worker_get(void) { return NULL; }
int flb_worker_log_level(struct flb_worker *worker) { (void) worker; return 0; }
int flb_log_cache_check_suppress(struct flb_log_cache *cache, char *msg_buf, size_t msg_size) { (void) cache; (void) msg_buf; (void) msg_size; return 0; }

/* ---- the engine's timer: a constant fd, nothing to arm */
int mk_event_timeout_create(struct mk_event_loop *loop, time_t sec, long nsec, void *data) { (void) loop; (void) sec; (void) nsec; (void) data; return 1000; }
int mk_event_timeout_destroy(struct mk_event_loop *loop, void *data) { (void) loop; (void) data; return 0; }
int mk_event_closesocket(int fd) { (void) fd; return 0; }

/* ---- CREATE STREAM: the in_stream_processor instance the result would be appended to is the engine's; the packaged bytes are
 * what this driver returns (task->stream stays NULL) */
int flb_sp_stream_create(const char *name, struct flb_sp_task *task, struct flb_sp *sp) { (void) name; (void) task; (void) sp; return 0; }
void flb_sp_stream_destroy(struct flb_sp_stream *stream, struct flb_sp *sp) { (void) stream; (void) sp; }

/* ---- the clock of package_results */
static struct flb_time g_now;
int __wrap_flb_time_get(struct flb_time *tm) { *tm = g_now; return 0; }
/* NOW() / UNIX_TIMESTAMP() read time(NULL) (flb_sp_func_time.c:54,75): the same given time */
time_t __wrap_time(time_t *t) { if (t) *t = g_now.tm.tv_sec; return g_now.tm.tv_sec; }

/* ---- parser/sql.l + sql.y by hand (see the header of this file) */
typedef void *yyscan_t;
typedef void *YY_BUFFER_STATE;
int flb_sp_lex_init(yyscan_t *s) { *s = NULL; return 0; }
int flb_sp_lex_destroy(yyscan_t s) { (void) s; return 0; }
YY_BUFFER_STATE flb_sp__scan_string(const char *str, yyscan_t s) { (void) s; return (YY_BUFFER_STATE) str; }
void flb_sp__delete_buffer(YY_BUFFER_STATE b, yyscan_t s) { (void) b; (void) s; }

enum { T_EOF = 0, T_IDENT, T_INT, T_FLOAT, T_STRING, T_BOOL, T_KW, T_CH, T_NEQ, T_LT, T_LTE, T_GT, T_GTE, T_BAD };
struct tok { int t; char *s; int i; float f; int ch; };
struct lex { const char *p; struct tok cur; };

static const char *KW[] = { "CREATE", "FLUSH", "STREAM", "SNAPSHOT", "WITH", "SELECT", "AS", "FROM", "WHERE", "AND", "OR", "NOT", "WINDOW",
    "LIMIT", "IS", "NULL", "SUM", "AVG", "COUNT", "MIN", "MAX", "TIMESERIES_FORECAST", "CONTAINS", "TIME", "TUMBLING", "HOPPING",
    "HOUR", "MINUTE", "SECOND", "NOW", "UNIX_TIMESTAMP", "RECORD_TAG", "RECORD_TIME", NULL };

static int ci_prefix(const char *p, const char *w)
{
    while (*w) { if (toupper((unsigned char) *p) != *w) return 0; p++; w++; }
    return 1;
}
static int is_ident_char(int c) { return isalnum(c) || c == '_' || c == '.'; }

/* flex picks the longest match, the earlier rule on a tie: a keyword only when the identifier rule does not match longer */
static void lex_next(struct lex *L)
{
    const char *p = L->p;
    struct tok *t = &L->cur;
    memset(t, 0, sizeof(*t));
    while (*p == ' ' || *p == '\t' || *p == '\n') p++;
    if (!*p) { t->t = T_EOF; L->p = p; return; }
    if (ci_prefix(p, "GROUP BY")) { t->t = T_KW; t->s = "GROUP BY"; L->p = p + 8; return; }
    if (ci_prefix(p, "ADVANCE BY")) { t->t = T_KW; t->s = "ADVANCE BY"; L->p = p + 10; return; }
    if (ci_prefix(p, "STREAM:")) { t->t = T_KW; t->s = "STREAM:"; L->p = p + 7; return; }
    if (ci_prefix(p, "TAG:")) { t->t = T_KW; t->s = "TAG:"; L->p = p + 4; return; }
    if (ci_prefix(p, "@RECORD")) { t->t = T_KW; t->s = "@RECORD"; L->p = p + 7; return; }
    if (*p == '_' || isalpha((unsigned char) *p)) {
        const char *q = p;
        size_t n;
        int k;
        while (is_ident_char((unsigned char) *q)) q++;
        n = q - p;
        for (k = 0; KW[k]; k++) {
            if (strlen(KW[k]) == n && ci_prefix(p, KW[k])) { t->t = T_KW; t->s = (char *) KW[k]; L->p = q; return; }
        }
        if (n == 4 && ci_prefix(p, "TRUE")) { t->t = T_BOOL; t->i = 1; L->p = q; return; }
        if (n == 5 && ci_prefix(p, "FALSE")) { t->t = T_BOOL; t->i = 0; L->p = q; return; }
        t->t = T_IDENT; t->s = flb_strndup(p, n); L->p = q;
        return;
    }
    if (isdigit((unsigned char) *p) || (*p == '-' && p[1] >= '1' && p[1] <= '9')) {
        const char *q = p;
        if (*q == '-') q++;
        if (*q == '0') q++;
        else while (isdigit((unsigned char) *q)) q++;
        if (*q == '.' && isdigit((unsigned char) q[1])) {
            q++;
            while (isdigit((unsigned char) *q)) q++;
            t->t = T_FLOAT; t->f = atof(p); L->p = q;
            return;
        }
        t->t = T_INT; t->i = atoi(p); L->p = q;
        return;
    }
    if (*p == '\'') {
        const char *q = p + 1;
        char *s;
        size_t j = 0;
        for (;;) {
            if (!*q) { t->t = T_BAD; L->p = q; return; }
            if (*q == '\'') { if (q[1] == '\'') { q += 2; continue; } break; }
            q++;
        }
        s = flb_malloc(q - p);
        for (const char *r = p + 1; r < q; r++) { s[j++] = *r; if (*r == '\'') r++; }
        s[j] = 0;
        t->t = T_STRING; t->s = s; L->p = q + 1;
        return;
    }
    if (p[0] == '!' && p[1] == '=') { t->t = T_NEQ; L->p = p + 2; return; }
    if (p[0] == '<' && p[1] == '>') { t->t = T_NEQ; L->p = p + 2; return; }
    if (p[0] == '<' && p[1] == '=') { t->t = T_LTE; L->p = p + 2; return; }
    if (p[0] == '>' && p[1] == '=') { t->t = T_GTE; L->p = p + 2; return; }
    if (p[0] == '<') { t->t = T_LT; L->p = p + 1; return; }
    if (p[0] == '>') { t->t = T_GT; L->p = p + 1; return; }
    if (strchr("*,=()[].;", *p)) { t->t = T_CH; t->ch = *p; L->p = p + 1; return; }
    t->t = T_BAD; L->p = p + 1;
}
static int is_kw(struct lex *L, const char *w) { return L->cur.t == T_KW && !strcmp(L->cur.s, w); }
static int is_ch(struct lex *L, int c) { return L->cur.t == T_CH && L->cur.ch == c; }
static int eat_kw(struct lex *L, const char *w) { if (is_kw(L, w)) { lex_next(L); return 1; } return 0; }
static int eat_ch(struct lex *L, int c) { if (is_ch(L, c)) { lex_next(L); return 1; } return 0; }

/* record_subkey: '[' STRING ']' ... -> cmd->tmp_subkeys */
static int p_subkeys(struct lex *L, struct flb_sp_cmd *cmd)
{
    while (is_ch(L, '[')) {
        lex_next(L);
        if (L->cur.t != T_STRING) return -1;
        flb_slist_add(cmd->tmp_subkeys, L->cur.s);
        flb_free(L->cur.s);
        lex_next(L);
        if (!eat_ch(L, ']')) return -1;
    }
    return 0;
}
static int p_alias(struct lex *L, struct flb_sp_cmd *cmd)
{
    if (eat_kw(L, "AS")) {
        if (L->cur.t != T_IDENT) return -1;
        flb_sp_cmd_alias_add(cmd, L->cur.s);
        lex_next(L);
    }
    return 0;
}
static int func_code(const char *kw)
{
    if (!strcmp(kw, "AVG")) return FLB_SP_AVG;
    if (!strcmp(kw, "SUM")) return FLB_SP_SUM;
    if (!strcmp(kw, "COUNT")) return FLB_SP_COUNT;
    if (!strcmp(kw, "MIN")) return FLB_SP_MIN;
    if (!strcmp(kw, "MAX")) return FLB_SP_MAX;
    if (!strcmp(kw, "TIMESERIES_FORECAST")) return FLB_SP_FORECAST;
    if (!strcmp(kw, "NOW")) return FLB_SP_NOW;
    if (!strcmp(kw, "UNIX_TIMESTAMP")) return FLB_SP_UNIX_TIMESTAMP;
    if (!strcmp(kw, "RECORD_TAG")) return FLB_SP_RECORD_TAG;
    if (!strcmp(kw, "RECORD_TIME")) return FLB_SP_RECORD_TIME;
    return -1;
}
static int p_record_key(struct lex *L, struct flb_sp_cmd *cmd)
{
    if (eat_ch(L, '*')) return flb_sp_cmd_key_add(cmd, -1, NULL);
    if (L->cur.t == T_IDENT) {
        char *name = L->cur.s;
        int ret;
        lex_next(L);
        if (p_subkeys(L, cmd) || p_alias(L, cmd)) return -1;
        ret = flb_sp_cmd_key_add(cmd, -1, name);
        flb_free(name);
        return ret;
    }
    if (L->cur.t == T_KW) {
        int code = func_code(L->cur.s);
        if (code < 0) return -1;
        lex_next(L);
        if (!eat_ch(L, '(')) return -1;
        if (code >= FLB_SP_NOW) {                       /* time_record_func '(' ')' key_alias */
            if (!eat_ch(L, ')') || p_alias(L, cmd)) return -1;
            return flb_sp_cmd_key_add(cmd, code, NULL);
        }
        if (code == FLB_SP_COUNT && eat_ch(L, '*')) {
            if (!eat_ch(L, ')') || p_alias(L, cmd)) return -1;
            return flb_sp_cmd_key_add(cmd, code, NULL);
        }
        if (L->cur.t != T_IDENT) return -1;
        {
            char *name = L->cur.s;
            int ret;
            lex_next(L);
            if (code == FLB_SP_FORECAST) {
                int secs;
                if (!eat_ch(L, ',') || L->cur.t != T_INT) return -1;
                secs = L->cur.i;
                lex_next(L);
                if (!eat_ch(L, ')') || p_alias(L, cmd)) return -1;
                ret = flb_sp_cmd_timeseries_forecast(cmd, code, name, secs);
            }
            else {
                if (p_subkeys(L, cmd) || !eat_ch(L, ')') || p_alias(L, cmd)) return -1;
                ret = flb_sp_cmd_key_add(cmd, code, name);
            }
            flb_free(name);
            return ret;
        }
    }
    return -1;
}
static struct flb_exp *p_key(struct lex *L, struct flb_sp_cmd *cmd)
{
    struct flb_exp *e;
    char *name = L->cur.s;
    lex_next(L);
    if (p_subkeys(L, cmd)) return NULL;
    e = flb_sp_cmd_condition_key(cmd, name);
    flb_free(name);
    return e;
}
static struct flb_exp *p_value(struct lex *L, struct flb_sp_cmd *cmd)
{
    struct flb_exp *e = NULL;
    if (L->cur.t == T_INT) e = flb_sp_cmd_condition_integer(cmd, L->cur.i);
    else if (L->cur.t == T_FLOAT) e = flb_sp_cmd_condition_float(cmd, L->cur.f);
    else if (L->cur.t == T_STRING) { e = flb_sp_cmd_condition_string(cmd, L->cur.s); flb_free(L->cur.s); }
    else if (L->cur.t == T_BOOL) e = flb_sp_cmd_condition_boolean(cmd, L->cur.i ? true : false);
    else return NULL;
    lex_next(L);
    return e;
}
static int is_value(struct lex *L) { return L->cur.t == T_INT || L->cur.t == T_FLOAT || L->cur.t == T_STRING || L->cur.t == T_BOOL; }

static struct flb_exp *p_condition(struct lex *L, struct flb_sp_cmd *cmd);

/* comparison | key | value | '(' condition ')' */
static struct flb_exp *p_primary(struct lex *L, struct flb_sp_cmd *cmd)
{
    struct flb_exp *left = NULL, *v;
    int plain_key = 0;
    if (eat_ch(L, '(')) {
        struct flb_exp *e = p_condition(L, cmd);
        if (!e || !eat_ch(L, ')')) return NULL;
        return flb_sp_cmd_operation(cmd, e, NULL, FLB_EXP_PAR);
    }
    if (is_value(L)) {
        v = p_value(L, cmd);
        return v ? flb_sp_cmd_operation(cmd, NULL, v, FLB_EXP_OR) : NULL;
    }
    if (is_kw(L, "@RECORD")) {
        lex_next(L);
        if (!eat_ch(L, '.')) return NULL;
        if (eat_kw(L, "CONTAINS")) {
            struct flb_exp *k;
            if (!eat_ch(L, '(') || L->cur.t != T_IDENT) return NULL;
            k = p_key(L, cmd);
            if (!k || !eat_ch(L, ')')) return NULL;
            left = flb_sp_record_function_add(cmd, "contains", k);
        }
        else if (eat_kw(L, "TIME")) {
            if (!eat_ch(L, '(') || !eat_ch(L, ')')) return NULL;
            left = flb_sp_record_function_add(cmd, "time", NULL);
        }
        else return NULL;
    }
    else if (L->cur.t == T_IDENT) {
        left = p_key(L, cmd);
        plain_key = 1;
    }
    if (!left) return NULL;
    if (plain_key && eat_kw(L, "IS")) {
        int neg = eat_kw(L, "NOT");
        struct flb_exp *c;
        if (!eat_kw(L, "NULL")) return NULL;
        c = flb_sp_cmd_comparison(cmd, left, flb_sp_cmd_condition_null(cmd), FLB_EXP_EQ);
        return neg ? flb_sp_cmd_operation(cmd, c, NULL, FLB_EXP_NOT) : c;
    }
    {
        int op = -1, neg = 0;
        if (is_ch(L, '=')) op = FLB_EXP_EQ;
        else if (L->cur.t == T_NEQ) { op = FLB_EXP_EQ; neg = 1; }
        else if (L->cur.t == T_LT) op = FLB_EXP_LT;
        else if (L->cur.t == T_LTE) op = FLB_EXP_LTE;
        else if (L->cur.t == T_GT) op = FLB_EXP_GT;
        else if (L->cur.t == T_GTE) op = FLB_EXP_GTE;
        if (op < 0) {
            /* a bare key is "condition: key" (an OR with nothing); a bare record function compares with true */
            if (plain_key) return flb_sp_cmd_operation(cmd, left, NULL, FLB_EXP_OR);
            return flb_sp_cmd_comparison(cmd, left, flb_sp_cmd_condition_boolean(cmd, true), FLB_EXP_EQ);
        }
        lex_next(L);
        v = p_value(L, cmd);
        if (!v) return NULL;
        left = flb_sp_cmd_comparison(cmd, left, v, op);
        return neg ? flb_sp_cmd_operation(cmd, left, NULL, FLB_EXP_NOT) : left;
    }
}
static struct flb_exp *p_condition(struct lex *L, struct flb_sp_cmd *cmd)
{
    struct flb_exp *left, *right;
    if (eat_kw(L, "NOT")) {
        left = p_condition(L, cmd);
        return left ? flb_sp_cmd_operation(cmd, left, NULL, FLB_EXP_NOT) : NULL;
    }
    left = p_primary(L, cmd);
    if (!left) return NULL;
    if (is_kw(L, "AND") || is_kw(L, "OR")) {
        int op = is_kw(L, "AND") ? FLB_EXP_AND : FLB_EXP_OR;
        lex_next(L);
        right = p_condition(L, cmd);
        return right ? flb_sp_cmd_operation(cmd, left, right, op) : NULL;
    }
    return left;
}
static int p_time_unit(struct lex *L)
{
    if (eat_kw(L, "SECOND")) return FLB_SP_TIME_SECOND;
    if (eat_kw(L, "MINUTE")) return FLB_SP_TIME_MINUTE;
    if (eat_kw(L, "HOUR")) return FLB_SP_TIME_HOUR;
    return -1;
}
static int p_select(struct lex *L, struct flb_sp_cmd *cmd)
{
    if (!eat_kw(L, "SELECT")) return -1;
    do { if (p_record_key(L, cmd)) return -1; } while (eat_ch(L, ','));
    if (!eat_kw(L, "FROM")) return -1;
    if (eat_kw(L, "STREAM:")) {
        if (L->cur.t != T_IDENT) return -1;
        flb_sp_cmd_source(cmd, FLB_SP_STREAM, L->cur.s);
        flb_free(L->cur.s);
        lex_next(L);
    }
    else if (eat_kw(L, "TAG:")) {
        if (L->cur.t != T_STRING) return -1;
        flb_sp_cmd_source(cmd, FLB_SP_TAG, L->cur.s);
        flb_free(L->cur.s);
        lex_next(L);
    }
    else return -1;
    if (eat_kw(L, "WINDOW")) {
        int size, unit, adv = 0, adv_unit = 0, type;
        if (eat_kw(L, "TUMBLING")) type = FLB_SP_WINDOW_TUMBLING;
        else if (eat_kw(L, "HOPPING")) type = FLB_SP_WINDOW_HOPPING;
        else return -1;
        if (!eat_ch(L, '(') || L->cur.t != T_INT) return -1;
        size = L->cur.i;
        lex_next(L);
        if ((unit = p_time_unit(L)) < 0) return -1;
        if (type == FLB_SP_WINDOW_HOPPING) {
            if (!eat_ch(L, ',') || !eat_kw(L, "ADVANCE BY") || L->cur.t != T_INT) return -1;
            adv = L->cur.i;
            lex_next(L);
            if ((adv_unit = p_time_unit(L)) < 0) return -1;
        }
        if (!eat_ch(L, ')')) return -1;
        flb_sp_cmd_window(cmd, type, size, unit, adv, adv_unit);
    }
    if (eat_kw(L, "WHERE")) {
        struct flb_exp *e = p_condition(L, cmd);
        if (!e) return -1;
        flb_sp_cmd_condition_add(cmd, e);
    }
    if (eat_kw(L, "GROUP BY")) {
        do {
            char *name;
            if (L->cur.t != T_IDENT) return -1;
            name = L->cur.s;
            lex_next(L);
            if (p_subkeys(L, cmd)) return -1;
            flb_sp_cmd_gb_key_add(cmd, name);
            flb_free(name);
        } while (eat_ch(L, ','));
    }
    if (eat_kw(L, "LIMIT")) {
        if (L->cur.t != T_INT) return -1;
        flb_sp_cmd_limit_add(cmd, L->cur.i);
        lex_next(L);
    }
    if (!eat_ch(L, ';')) return -1;
    cmd->type = FLB_SP_SELECT;
    return 0;
}
int flb_sp_parse(struct flb_sp_cmd *cmd, const char *query, void *scanner)
{
    struct lex L;
    (void) scanner;
    L.p = query;
    lex_next(&L);
    if (eat_kw(&L, "CREATE")) {
        char *name;
        if (!eat_kw(&L, "STREAM") || L.cur.t != T_IDENT) return 1;       /* snapshots: not on this path */
        name = L.cur.s;
        lex_next(&L);
        if (eat_kw(&L, "WITH")) {
            if (!eat_ch(&L, '(')) return 1;
            do {
                char *k;
                if (L.cur.t != T_IDENT) return 1;
                k = L.cur.s;
                lex_next(&L);
                if (!eat_ch(&L, '=') || L.cur.t != T_STRING) return 1;
                flb_sp_cmd_stream_prop_add(cmd, k, L.cur.s);
                flb_free(k); flb_free(L.cur.s);
                lex_next(&L);
            } while (eat_ch(&L, ','));
            if (!eat_ch(&L, ')')) return 1;
        }
        if (!eat_kw(&L, "AS") || p_select(&L, cmd)) return 1;
        flb_sp_cmd_stream_new(cmd, name);
        flb_free(name);
    }
    else if (p_select(&L, cmd)) return 1;
    return L.cur.t == T_EOF ? 0 : 1;
}

/* ---- the driver */
static void on_segv(int sig)
{
    void *bt[32];
    int n = backtrace(bt, 32);
    (void) sig;
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
static int rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }
static char *rd_str(void)
{
    uint32_t n;
    char *s;
    if (!rd(&n, 4)) return NULL;
    s = malloc(n + 1);
    if (n && !rd(s, n)) return NULL;
    s[n] = 0;
    return s;
}
static void wr_answer(int32_t ret, const void *out, uint64_t n)
{
    fwrite(&ret, 4, 1, stdout); fwrite(&n, 8, 1, stdout);
    if (n) fwrite(out, 1, n, stdout);
    fflush(stdout);
}

/* src/stream_processor/flb_sp.c (not in a header) */
int sp_process_data_aggr(const char *buf_data, size_t buf_size, const char *tag, int tag_len, struct flb_sp_task *task, struct flb_sp *sp,
                         int convert_str_to_num);
void package_results(const char *tag, int tag_len, char **out_buf, size_t *out_size, struct flb_sp_task *task);
int sp_process_hopping_slot(const char *tag, int tag_len, struct flb_sp_task *task);
int sp_process_data(const char *tag, int tag_len, const char *buf_data, size_t buf_size, char **out_buf, size_t *out_size, struct flb_sp_task *task,
                    struct flb_sp *sp);

int main(void)
{
    struct flb_config *config = flb_calloc(1, sizeof(struct flb_config));
    struct flb_sp *sp = flb_calloc(1, sizeof(struct flb_sp));
    struct flb_sp_task *task = NULL;
    uint32_t str_conv = 1;
    signal(SIGSEGV, on_segv);
    mk_list_init(&config->inputs);
    mk_list_init(&config->stream_processor_tasks);
    sp->config = config;
    mk_list_init(&sp->tasks);
    for (;;) {
        uint32_t op;
        if (!rd(&op, 4)) break;
        if (op == 1) {
            uint32_t sec, nsec;
            char *sql;
            if (!rd(&str_conv, 4) || !rd(&sec, 4) || !rd(&nsec, 4)) break;
            sql = rd_str();
            if (!sql) break;
            g_now.tm.tv_sec = sec; g_now.tm.tv_nsec = nsec;
            if (task) { flb_sp_task_destroy(task); task = NULL; }
            task = flb_sp_task_create(sp, "t", sql);
            if (task && task->stream) task->stream = NULL;
            /* 0: an aggregate task, 1: a task of flb_sp_do's other branch (sp_process_data), -1: refused */
            wr_answer(!task ? -1 : task->aggregate_keys == FLB_TRUE ? 0 : 1, NULL, 0);
            free(sql);
        }
        else if (op == 2) {
            uint64_t n;
            char *data, *out = NULL;
            size_t out_size = 0;
            int ret;
            if (!rd(&n, 8)) break;
            data = malloc(n + 1);
            if (n && !rd(data, n)) break;
            if (!task) { wr_answer(-1, NULL, 0); free(data); continue; }
            if (task->aggregate_keys != FLB_TRUE) {
                /* flb_sp.c:2059-2069, the other branch of flb_sp_do: the answer is sp_process_data's return value and buffer */
                ret = sp_process_data("t", 1, data, n, &out, &out_size, task, sp);
                wr_answer(ret, ret > 0 ? out : NULL, ret > 0 ? out_size : 0);
                if (ret > 0 && out) flb_free(out);
                free(data);
                continue;
            }
            /* tests/internal/stream_processor.c:92-150 (flb_sp_do_test), the aggregate branch */
            ret = sp_process_data_aggr(data, n, "t", 1, task, sp, (int) str_conv);
            if (ret != -1 && flb_sp_window_populate(task, data, n) != -1 && task->window.type == FLB_SP_WINDOW_DEFAULT) {
                package_results("t", 1, &out, &out_size, task);
                flb_sp_window_prune(task);
            }
            wr_answer(ret, out, out_size);
            if (out) flb_free(out);
            free(data);
        }
        else if (op == 3) {
            char *out = NULL;
            size_t out_size = 0;
            /* tests/internal/stream_processor.c:60-90 (flb_sp_fd_event_test), the window.fd branch */
            if (task && task->window.records > 0) package_results("t", 1, &out, &out_size, task);
            if (task) flb_sp_window_prune(task);
            wr_answer(0, out, out_size);
            if (out) flb_free(out);
        }
        else if (op == 4) {
            if (task) { flb_sp_task_destroy(task); task = NULL; }
            wr_answer(0, NULL, 0);
        }
        else if (op == 5) {
            wr_answer(task && task->window.type == FLB_SP_WINDOW_HOPPING ? sp_process_hopping_slot("t", 1, task) : -1, NULL, 0);
        }
        else break;
    }
    return 0;
}
