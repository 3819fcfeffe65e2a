/*
 * oracle/orx.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the regex semantics the reference hot path gets from Onigmo 6.2.0
 * (lib/onigmo) as configured by src/flb_regex.c:142-145: ONIG_ENCODING_UTF8,
 * ONIG_SYNTAX_RUBY, ONIG_OPTION_DEFAULT (+ i/m/x from the /pat/imx form).
 *
 * It is a plain backtracking matcher over an AST (leftmost-first alternation, greedy /
 * lazy / possessive repeats, "last iteration wins" captures, Onigmo's empty-loop exit),
 * i.e. the same search strategy as lib/onigmo/regexec.c:match_at (:1431) driven from every
 * start offset like onig_search (:3793) -- but written independently from the parser up.
 * Ruby-syntax facts restated here:
 *   - ^ and $ are LINE anchors (ONIG_OPTION_SINGLELINE off, lib/onigmo/regparse.c:39-76);
 *     \A \z \Z are string anchors; '.' excludes \n unless (?m)
 *   - \d \s \w \h and [[:posix:]] are ASCII-range here: ONIG_OPTION_ASCII_RANGE is merged in
 *     by onig_reg_init for Ruby syntax (lib/onigmo/regcomp.c:5842-5850); negations therefore
 *     contain every non-ASCII code point
 *   - plain (...) groups stop capturing as soon as one named group exists
 *     (ONIG_OPTION_CAPTURE_GROUP off; lib/onigmo/regparse.c:971-984)
 *   - names iterate in first-appearance order (lib/onigmo/regparse.c:582-597)
 * Pinned against the real engine (oracle/_ref/libonig_ref.so) by tests/test_oracle_regex.py
 * and against the reference KATs in tests/internal/regex.c, tests/internal/parser_regex.c.
 *
 * Not restated (compile error => callers treat the pattern as unsupported): back-references,
 * look-behind, \G, \K, \R, \X, absent operator, conditionals, subexpression calls, \p{..}
 * properties other than the POSIX set, non-ASCII case folding, class intersection (&&).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdint.h>
#include "orx.h"

enum { N_EMPTY, N_CHAR, N_ANY, N_CLASS, N_CAT, N_ALT, N_GROUP, N_REPEAT, N_ANCHOR, N_LOOK, N_ATOMIC };
enum { A_BOL, A_EOL, A_BOS, A_EOS, A_EOS_NL, A_WORDB, A_NWORDB };

typedef struct Range { uint32_t lo, hi; } Range;

typedef struct Node {
    int type;
    uint32_t ch;             /* N_CHAR */
    int multiline;           /* N_ANY: '.' also matches \n */
    Range *ranges;           /* N_CLASS (normalised, non-negated) */
    int nranges;
    struct Node **kids;      /* N_CAT / N_ALT */
    int nkids;
    struct Node *sub;        /* GROUP / REPEAT / LOOK / ATOMIC */
    int cap;                 /* N_GROUP: capture index or 0 */
    int min, max;            /* N_REPEAT: max = -1 => inf */
    int greedy, possessive;
    int anchor;              /* N_ANCHOR */
    int neg;                 /* N_LOOK negative */
} Node;

struct orx {
    Node *root;
    int ncap;                          /* number of capture groups (excluding 0) */
    int nnames;
    char *names[ORX_MAX_GROUPS];       /* in first-appearance order */
    int name_groups[ORX_MAX_GROUPS][8];
    int name_ngroups[ORX_MAX_GROUPS];
    Node **all; int nall, capall;      /* for freeing */
};

/* ------------------------------------------------------------------ parser */
typedef struct P {
    const unsigned char *s, *e, *p;
    struct orx *rx;
    int has_named;          /* pre-scan result */
    int ncap;
    char *err; int errlen;
    int failed;
} P;

#define OPT_I 1
#define OPT_M 2
#define OPT_X 4

static void perr(P *ps, const char *msg)
{
    if (!ps->failed) {
        ps->failed = 1;
        if (ps->err) snprintf(ps->err, ps->errlen, "%s at offset %d", msg, (int) (ps->p - ps->s));
    }
}

static Node *mk(P *ps, int type)
{
    struct orx *rx = ps->rx;
    Node *n = calloc(1, sizeof(Node));
    n->type = type;
    if (rx->nall == rx->capall) {
        rx->capall = rx->capall ? rx->capall * 2 : 64;
        rx->all = realloc(rx->all, sizeof(Node *) * rx->capall);
    }
    rx->all[rx->nall++] = n;
    return n;
}

static void add_kid(Node *n, Node *k)
{
    n->kids = realloc(n->kids, sizeof(Node *) * (n->nkids + 1));
    n->kids[n->nkids++] = k;
}

/* decode one UTF-8 code point of the PATTERN (patterns must be valid UTF-8) */
static uint32_t pat_char(P *ps)
{
    uint32_t c = *ps->p++;
    int n = 0;
    if (c < 0x80) return c;
    if (c >= 0xc2 && c <= 0xdf) { n = 1; c &= 0x1f; }
    else if (c >= 0xe0 && c <= 0xef) { n = 2; c &= 0x0f; }
    else if (c >= 0xf0 && c <= 0xf4) { n = 3; c &= 0x07; }
    else { perr(ps, "invalid UTF-8 in pattern"); return 0xfffd; }
    while (n-- > 0) {
        if (ps->p >= ps->e || (*ps->p & 0xc0) != 0x80) { perr(ps, "invalid UTF-8 in pattern"); return 0xfffd; }
        c = (c << 6) | (*ps->p++ & 0x3f);
    }
    return c;
}

/* ---- class building */
typedef struct RS { Range *r; int n, cap; } RS;

static void rs_add(RS *s, uint32_t lo, uint32_t hi)
{
    if (lo > hi) return;
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 8; s->r = realloc(s->r, sizeof(Range) * s->cap); }
    s->r[s->n].lo = lo; s->r[s->n].hi = hi; s->n++;
}

static int rcmp(const void *a, const void *b)
{
    const Range *x = a, *y = b;
    return x->lo < y->lo ? -1 : x->lo > y->lo ? 1 : 0;
}

static void rs_norm(RS *s)
{
    int i, j = 0;
    if (s->n == 0) return;
    qsort(s->r, s->n, sizeof(Range), rcmp);
    for (i = 1; i < s->n; i++) {
        if (s->r[i].lo <= s->r[j].hi + 1 && s->r[j].hi != 0xffffffffu) {
            if (s->r[i].hi > s->r[j].hi) s->r[j].hi = s->r[i].hi;
        }
        else s->r[++j] = s->r[i];
    }
    s->n = j + 1;
}

#define MAXCP 0x7fffffffu

static void rs_negate(RS *s)
{
    RS o = {0};
    uint32_t next = 0;
    int i;
    rs_norm(s);
    for (i = 0; i < s->n; i++) {
        if (s->r[i].lo > next) rs_add(&o, next, s->r[i].lo - 1);
        next = s->r[i].hi + 1;
    }
    if (next <= MAXCP) rs_add(&o, next, MAXCP);
    free(s->r);
    *s = o;
}

static void rs_union(RS *d, const RS *s) { int i; for (i = 0; i < s->n; i++) rs_add(d, s->r[i].lo, s->r[i].hi); }

/* ASCII case closure (Onigmo also folds a handful of non-ASCII code points; see file header) */
static void rs_icase(RS *s)
{
    int i, n = s->n;
    for (i = 0; i < n; i++) {
        uint32_t lo = s->r[i].lo, hi = s->r[i].hi, a, b;
        a = lo > 'a' ? lo : 'a'; b = hi < 'z' ? hi : 'z';
        if (a <= b) rs_add(s, a - 32, b - 32);
        a = lo > 'A' ? lo : 'A'; b = hi < 'Z' ? hi : 'Z';
        if (a <= b) rs_add(s, a + 32, b + 32);
    }
}

/* the ASCII-range ctype sets (ONIG_OPTION_ASCII_RANGE) */
static int add_ctype(RS *s, char t)
{
    switch (t) {
    case 'd': rs_add(s, '0', '9'); return 1;
    case 'w': rs_add(s, '0', '9'); rs_add(s, 'A', 'Z'); rs_add(s, 'a', 'z'); rs_add(s, '_', '_'); return 1;
    case 's': rs_add(s, 9, 13); rs_add(s, ' ', ' '); return 1;
    case 'h': rs_add(s, '0', '9'); rs_add(s, 'A', 'F'); rs_add(s, 'a', 'f'); return 1;
    }
    return 0;
}

static int add_posix(RS *s, const char *name, int len)
{
#define IS(x) (len == (int) strlen(x) && memcmp(name, x, len) == 0)
    if (IS("alpha")) { rs_add(s, 'A', 'Z'); rs_add(s, 'a', 'z'); }
    else if (IS("digit")) rs_add(s, '0', '9');
    else if (IS("alnum")) { rs_add(s, '0', '9'); rs_add(s, 'A', 'Z'); rs_add(s, 'a', 'z'); }
    else if (IS("upper")) rs_add(s, 'A', 'Z');
    else if (IS("lower")) rs_add(s, 'a', 'z');
    else if (IS("space")) { rs_add(s, 9, 13); rs_add(s, ' ', ' '); }
    else if (IS("blank")) { rs_add(s, 9, 9); rs_add(s, ' ', ' '); }
    else if (IS("cntrl")) { rs_add(s, 0, 31); rs_add(s, 127, 127); }
    else if (IS("punct")) { rs_add(s, 33, 47); rs_add(s, 58, 64); rs_add(s, 91, 96); rs_add(s, 123, 126); }
    else if (IS("graph")) rs_add(s, 33, 126);
    else if (IS("print")) rs_add(s, 32, 126);
    else if (IS("xdigit")) { rs_add(s, '0', '9'); rs_add(s, 'A', 'F'); rs_add(s, 'a', 'f'); }
    else if (IS("word")) { rs_add(s, '0', '9'); rs_add(s, 'A', 'Z'); rs_add(s, 'a', 'z'); rs_add(s, '_', '_'); }
    else if (IS("ascii")) rs_add(s, 0, 127);
    else return 0;
    return 1;
#undef IS
}

static int hexv(int c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

/* escape that denotes a single code point; returns 1 and sets *out, 0 if not such an escape */
static int esc_char(P *ps, uint32_t *out)
{
    int c = *ps->p;
    switch (c) {
    case 't': ps->p++; *out = 9; return 1;
    case 'n': ps->p++; *out = 10; return 1;
    case 'r': ps->p++; *out = 13; return 1;
    case 'f': ps->p++; *out = 12; return 1;
    case 'v': ps->p++; *out = 11; return 1;
    case 'a': ps->p++; *out = 7; return 1;
    case 'e': ps->p++; *out = 27; return 1;
    case 'x': {
        uint32_t v = 0; int n = 0;
        ps->p++;
        if (ps->p < ps->e && *ps->p == '{') {
            ps->p++;
            while (ps->p < ps->e && hexv(*ps->p) >= 0 && n < 8) { v = v * 16 + hexv(*ps->p++); n++; }
            if (ps->p >= ps->e || *ps->p != '}' || n == 0) { perr(ps, "bad \\x{}"); return 1; }
            ps->p++;
        }
        else {
            while (ps->p < ps->e && hexv(*ps->p) >= 0 && n < 2) { v = v * 16 + hexv(*ps->p++); n++; }
            if (n == 0) { perr(ps, "bad \\x"); return 1; }
            if (v >= 0x80) { perr(ps, "raw byte escape >= 0x80 unsupported"); return 1; }
        }
        *out = v; return 1;
    }
    case 'u': {
        uint32_t v = 0; int n = 0;
        ps->p++;
        while (ps->p < ps->e && hexv(*ps->p) >= 0 && n < 4) { v = v * 16 + hexv(*ps->p++); n++; }
        if (n != 4) { perr(ps, "bad \\u"); return 1; }
        *out = v; return 1;
    }
    case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': {
        /* \0, \0oo octal; \1-\9 are back-references outside classes (handled by caller) */
        uint32_t v = 0; int n = 0;
        if (c != '0') return 0;
        while (ps->p < ps->e && *ps->p >= '0' && *ps->p <= '7' && n < 3) { v = v * 8 + (*ps->p++ - '0'); n++; }
        if (v >= 0x80) { perr(ps, "raw byte escape >= 0x80 unsupported"); return 1; }
        *out = v; return 1;
    }
    case 'c': case 'C': case 'M':
        perr(ps, "control/meta escapes unsupported"); return 1;
    }
    return 0;
}

static void parse_class_body(P *ps, RS *out, int opts);

/* parses after '[' ; returns normalised (possibly negated) set */
static void parse_class(P *ps, RS *out, int opts)
{
    int neg = 0;
    RS s = {0};
    if (ps->p < ps->e && *ps->p == '^') { neg = 1; ps->p++; }
    parse_class_body(ps, &s, opts);
    if (opts & OPT_I) rs_icase(&s);
    rs_norm(&s);
    if (neg) rs_negate(&s);
    rs_union(out, &s);
    free(s.r);
}

static void parse_class_body(P *ps, RS *s, int opts)
{
    int first = 1;
    for (;;) {
        uint32_t lo, hi;
        int have = 0;
        if (ps->p >= ps->e) { perr(ps, "premature end of char-class"); return; }
        if (*ps->p == ']' && !first) { ps->p++; return; }
        if (*ps->p == ']' && first) {
            /* Onigmo: ']' right after '[' or '[^' is a literal (with a warning) if another ] follows */
            ps->p++; lo = ']'; have = 1;
        }
        first = 0;
        if (!have) {
            if (*ps->p == '[') {
                if (ps->p + 1 < ps->e && ps->p[1] == ':') {
                    const unsigned char *q = ps->p + 2;
                    int pneg = 0;
                    const unsigned char *nm;
                    if (q < ps->e && *q == '^') { pneg = 1; q++; }
                    nm = q;
                    while (q < ps->e && *q >= 'a' && *q <= 'z') q++;
                    if (q + 1 < ps->e && q[0] == ':' && q[1] == ']') {
                        RS t = {0};
                        if (!add_posix(&t, (const char *) nm, (int) (q - nm))) { perr(ps, "unknown POSIX bracket"); free(t.r); return; }
                        rs_norm(&t);
                        if (pneg) rs_negate(&t);
                        rs_union(s, &t);
                        free(t.r);
                        ps->p = q + 2;
                        continue;
                    }
                }
                /* nested class */
                ps->p++;
                {
                    RS t = {0};
                    parse_class(ps, &t, opts);
                    rs_union(s, &t);
                    free(t.r);
                }
                continue;
            }
            if (*ps->p == '&' && ps->p + 1 < ps->e && ps->p[1] == '&') { perr(ps, "class intersection unsupported"); return; }
            if (*ps->p == '\\') {
                ps->p++;
                if (ps->p >= ps->e) { perr(ps, "trailing backslash"); return; }
                {
                    int c = *ps->p;
                    if (c == 'd' || c == 'w' || c == 's' || c == 'h') { ps->p++; add_ctype(s, (char) c); continue; }
                    if (c == 'D' || c == 'W' || c == 'S' || c == 'H') {
                        RS t = {0};
                        ps->p++;
                        add_ctype(&t, (char) (c + 32));
                        rs_negate(&t);
                        rs_union(s, &t);
                        free(t.r);
                        continue;
                    }
                    if (c == 'p' || c == 'P' || c == 'R' || c == 'X') { perr(ps, "property escape unsupported"); return; }
                    if (c == 'b') { ps->p++; lo = 8; }
                    else if (esc_char(ps, &lo)) { if (ps->failed) return; }
                    else if (c >= '1' && c <= '7') {
                        uint32_t v = 0; int n = 0;
                        while (ps->p < ps->e && *ps->p >= '0' && *ps->p <= '7' && n < 3) { v = v * 8 + (*ps->p++ - '0'); n++; }
                        if (v >= 0x80) { perr(ps, "raw byte escape >= 0x80 unsupported"); return; }
                        lo = v;
                    }
                    else lo = pat_char(ps);
                }
            }
            else lo = pat_char(ps);
        }
        hi = lo;
        /* range? */
        if (ps->p + 1 < ps->e && ps->p[0] == '-' && ps->p[1] != ']') {
            const unsigned char *save = ps->p;
            ps->p++;
            if (*ps->p == '[') { ps->p = save; }          /* a-[..] : '-' literal next round */
            else if (*ps->p == '\\') {
                ps->p++;
                if (ps->p >= ps->e) { perr(ps, "trailing backslash"); return; }
                if (strchr("dwshDWSHpP", *ps->p)) { ps->p = save; }
                else if (*ps->p == 'b') { ps->p++; hi = 8; }
                else if (esc_char(ps, &hi)) { if (ps->failed) return; }
                else hi = pat_char(ps);
            }
            else hi = pat_char(ps);
            if (hi < lo) { perr(ps, "empty range in char class"); return; }
        }
        rs_add(s, lo, hi);
    }
}

static Node *mk_class_from(P *ps, RS *s)
{
    Node *n = mk(ps, N_CLASS);
    rs_norm(s);
    n->ranges = s->r; n->nranges = s->n;
    s->r = NULL; s->n = s->cap = 0;
    return n;
}

static Node *mk_char(P *ps, uint32_t c, int opts)
{
    if ((opts & OPT_I) && ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) {
        RS s = {0};
        rs_add(&s, c, c);
        rs_icase(&s);
        return mk_class_from(ps, &s);
    }
    if ((opts & OPT_I) && c >= 0x80) { perr(ps, "non-ASCII case folding unsupported"); }
    {
        Node *n = mk(ps, N_CHAR);
        n->ch = c;
        return n;
    }
}

static Node *parse_alt(P *ps, int *opts, int depth);

static void skip_x(P *ps, int opts)
{
    if (!(opts & OPT_X)) return;
    while (ps->p < ps->e) {
        int c = *ps->p;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') ps->p++;
        else if (c == '#') { while (ps->p < ps->e && *ps->p != '\n') ps->p++; }
        else break;
    }
}

static int parse_int(P *ps)
{
    long v = 0; int n = 0;
    while (ps->p < ps->e && *ps->p >= '0' && *ps->p <= '9') { v = v * 10 + (*ps->p++ - '0'); n++; if (v > 100000) v = 100001; }
    return n ? (int) v : -1;
}

static void register_name(P *ps, const unsigned char *nm, int len, int group)
{
    struct orx *rx = ps->rx;
    int i;
    for (i = 0; i < rx->nnames; i++) {
        if ((int) strlen(rx->names[i]) == len && memcmp(rx->names[i], nm, len) == 0) {
            if (rx->name_ngroups[i] < 8) rx->name_groups[i][rx->name_ngroups[i]++] = group;
            return;
        }
    }
    if (rx->nnames >= ORX_MAX_GROUPS) { perr(ps, "too many names"); return; }
    rx->names[rx->nnames] = malloc(len + 1);
    memcpy(rx->names[rx->nnames], nm, len);
    rx->names[rx->nnames][len] = 0;
    rx->name_groups[rx->nnames][0] = group;
    rx->name_ngroups[rx->nnames] = 1;
    rx->nnames++;
}

/* atom; returns NULL at ')' '|' or end */
static Node *parse_atom(P *ps, int *opts, int depth)
{
    int c;
    skip_x(ps, *opts);
    if (ps->p >= ps->e) return NULL;
    c = *ps->p;
    if (c == '|' || c == ')') return NULL;
    if (c == '(') {
        Node *g;
        ps->p++;
        if (depth > 200) { perr(ps, "nesting too deep"); return NULL; }
        if (ps->p < ps->e && *ps->p == '?') {
            ps->p++;
            if (ps->p >= ps->e) { perr(ps, "end pattern in group"); return NULL; }
            c = *ps->p;
            if (c == '#') {                      /* comment group */
                while (ps->p < ps->e && *ps->p != ')') ps->p++;
                if (ps->p >= ps->e) { perr(ps, "end pattern in group"); return NULL; }
                ps->p++;
                return mk(ps, N_EMPTY);
            }
            if (c == ':') {
                int o = *opts;
                ps->p++;
                g = mk(ps, N_GROUP);
                g->sub = parse_alt(ps, &o, depth + 1);
            }
            else if (c == '=' || c == '!') {
                int o = *opts;
                ps->p++;
                g = mk(ps, N_LOOK);
                g->neg = (c == '!');
                g->sub = parse_alt(ps, &o, depth + 1);
            }
            else if (c == '>') {
                int o = *opts;
                ps->p++;
                g = mk(ps, N_ATOMIC);
                g->sub = parse_alt(ps, &o, depth + 1);
            }
            else if (c == '<' || c == '\'') {
                const unsigned char *nm;
                int term = c == '<' ? '>' : '\'';
                int o = *opts;
                if (c == '<' && ps->p + 1 < ps->e && (ps->p[1] == '=' || ps->p[1] == '!')) { perr(ps, "look-behind unsupported"); return NULL; }
                ps->p++;
                nm = ps->p;
                while (ps->p < ps->e && *ps->p != term) {
                    int ch = *ps->p;
                    if (!((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_' || ch >= 0x80)) { perr(ps, "invalid group name"); return NULL; }
                    ps->p++;
                }
                if (ps->p >= ps->e || ps->p == nm) { perr(ps, "invalid group name"); return NULL; }
                if (*nm >= '0' && *nm <= '9') { perr(ps, "invalid group name"); return NULL; }
                g = mk(ps, N_GROUP);
                g->cap = ++ps->ncap;
                if (g->cap >= ORX_MAX_GROUPS) { perr(ps, "too many groups"); return NULL; }
                register_name(ps, nm, (int) (ps->p - nm), g->cap);
                ps->p++;
                g->sub = parse_alt(ps, &o, depth + 1);
            }
            else if (c == '~' || c == '(' || c == '&' || c == 'P') { perr(ps, "unsupported group construct"); return NULL; }
            else {
                /* option setting: (?imx-imx) or (?imx-imx:subexp) */
                int o = *opts, on = 1;
                for (;;) {
                    if (ps->p >= ps->e) { perr(ps, "end pattern in group"); return NULL; }
                    c = *ps->p;
                    if (c == 'i') { if (on) o |= OPT_I; else o &= ~OPT_I; }
                    else if (c == 'm') { if (on) o |= OPT_M; else o &= ~OPT_M; }
                    else if (c == 'x') { if (on) o |= OPT_X; else o &= ~OPT_X; }
                    else if (c == '-') on = 0;
                    else if (c == ')' || c == ':') break;
                    else { perr(ps, "undefined group option"); return NULL; }
                    ps->p++;
                }
                if (c == ')') {
                    /* isolated option: Onigmo makes THE REST of the enclosing group (alternation
                     * included) the body of the option node (lib/onigmo/regparse.c parse_exp,
                     * "option only" r == 2 path) */
                    ps->p++;
                    g = mk(ps, N_GROUP);
                    g->sub = parse_alt(ps, &o, depth + 1);
                    return g;
                }
                ps->p++;
                g = mk(ps, N_GROUP);
                g->sub = parse_alt(ps, &o, depth + 1);
            }
        }
        else {
            int o = *opts;
            g = mk(ps, N_GROUP);
            if (!ps->has_named) {
                g->cap = ++ps->ncap;
                if (g->cap >= ORX_MAX_GROUPS) { perr(ps, "too many groups"); return NULL; }
            }
            g->sub = parse_alt(ps, &o, depth + 1);
        }
        if (ps->failed) return NULL;
        if (ps->p >= ps->e || *ps->p != ')') { perr(ps, "end pattern with unmatched parenthesis"); return NULL; }
        ps->p++;
        return g;
    }
    if (c == '[') {
        RS s = {0};
        ps->p++;
        parse_class(ps, &s, *opts);
        if (ps->failed) { free(s.r); return NULL; }
        return mk_class_from(ps, &s);
    }
    if (c == '.') {
        Node *n = mk(ps, N_ANY);
        ps->p++;
        n->multiline = (*opts & OPT_M) ? 1 : 0;
        return n;
    }
    if (c == '^') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_BOL; return n; }
    if (c == '$') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_EOL; return n; }
    if (c == '*' || c == '+' || c == '?') { perr(ps, "target of repeat operator is not specified"); return NULL; }
    if (c == '\\') {
        uint32_t v;
        ps->p++;
        if (ps->p >= ps->e) { perr(ps, "end pattern at escape"); return NULL; }
        c = *ps->p;
        if (c == 'd' || c == 'w' || c == 's' || c == 'h' || c == 'D' || c == 'W' || c == 'S' || c == 'H') {
            RS s = {0};
            ps->p++;
            add_ctype(&s, (char) (c | 32));
            rs_norm(&s);
            if (!(c & 32)) rs_negate(&s);
            return mk_class_from(ps, &s);
        }
        if (c == 'A') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_BOS; return n; }
        if (c == 'z') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_EOS; return n; }
        if (c == 'Z') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_EOS_NL; return n; }
        if (c == 'b') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_WORDB; return n; }
        if (c == 'B') { Node *n = mk(ps, N_ANCHOR); ps->p++; n->anchor = A_NWORDB; return n; }
        if (c == 'G' || c == 'K' || c == 'R' || c == 'X' || c == 'k' || c == 'g' || c == 'p' || c == 'P') { perr(ps, "unsupported escape"); return NULL; }
        if (c >= '1' && c <= '9') { perr(ps, "back-reference unsupported"); return NULL; }
        if (esc_char(ps, &v)) { if (ps->failed) return NULL; return mk_char(ps, v, *opts); }
        v = pat_char(ps);
        return mk_char(ps, v, *opts);
    }
    return mk_char(ps, pat_char(ps), *opts);
}

/* quantifier suffixes */
static Node *parse_repeat(P *ps, int *opts, int depth)
{
    Node *a = parse_atom(ps, opts, depth);
    if (a == NULL || ps->failed) return a;
    for (;;) {
        int min, max, c, is_brace = 0;
        const unsigned char *save;
        skip_x(ps, *opts);
        if (ps->p >= ps->e) break;
        c = *ps->p;
        save = ps->p;
        if (c == '*') { min = 0; max = -1; ps->p++; }
        else if (c == '+') { min = 1; max = -1; ps->p++; }
        else if (c == '?') { min = 0; max = 1; ps->p++; }
        else if (c == '{') {
            int lo, hi;
            ps->p++;
            lo = parse_int(ps);
            if (ps->p < ps->e && *ps->p == ',') {
                ps->p++;
                hi = parse_int(ps);
                if (lo < 0 && hi < 0) { ps->p = save; break; }       /* "{,}" literal */
                if (lo < 0) lo = 0;
                /* hi < 0 => infinite */
            }
            else {
                if (lo < 0) { ps->p = save; break; }                  /* "{" literal */
                hi = lo;
            }
            if (ps->p >= ps->e || *ps->p != '}') { ps->p = save; break; }
            ps->p++;
            if (lo > 100000 || hi > 100000) { perr(ps, "too big number for repeat range"); return NULL; }
            if (hi >= 0 && lo > hi) {
                /* Ruby syntax: ONIG_SYN_OP_ESC... allows {n,m} with n>m => swapped & non-greedy?  Onigmo: error */
                perr(ps, "upper is smaller than lower in repeat range"); return NULL;
            }
            min = lo; max = hi; is_brace = 1;
        }
        else break;
        if (a->type == N_ANCHOR || a->type == N_LOOK) { perr(ps, "target of repeat operator is invalid"); return NULL; }
        {
            Node *r = mk(ps, N_REPEAT);
            r->sub = a; r->min = min; r->max = max; r->greedy = 1;
            if (ps->p < ps->e && *ps->p == '?') { ps->p++; r->greedy = 0; }
            else if (!is_brace && ps->p < ps->e && *ps->p == '+') { ps->p++; r->possessive = 1; }
            a = r;
        }
    }
    return a;
}

/* literal '{' fallback: parse_atom treats '{' as a plain char; handled because parse_repeat
 * resets ps->p to the '{' and breaks, after which the caller parses it as an atom. */

static Node *parse_cat(P *ps, int *opts, int depth)
{
    Node *cat = mk(ps, N_CAT);
    for (;;) {
        Node *r;
        const unsigned char *before = ps->p;
        r = parse_repeat(ps, opts, depth);
        if (ps->failed) return cat;
        if (r == NULL) break;
        /* an interval that fell back to a literal '{' leaves p at '{' */
        if (ps->p == before && r != NULL) { /* cannot happen */ break; }
        add_kid(cat, r);
    }
    return cat;
}

static Node *parse_alt(P *ps, int *opts, int depth)
{
    Node *first = parse_cat(ps, opts, depth);
    Node *alt;
    if (ps->failed) return first;
    if (ps->p >= ps->e || *ps->p != '|') return first;
    alt = mk(ps, N_ALT);
    add_kid(alt, first);
    while (ps->p < ps->e && *ps->p == '|') {
        ps->p++;
        add_kid(alt, parse_cat(ps, opts, depth));
        if (ps->failed) return alt;
    }
    return alt;
}

/* pre-scan: does the pattern contain a named group? (decides whether plain groups capture) */
static int scan_named(const unsigned char *s, const unsigned char *e)
{
    const unsigned char *p = s;
    int in_class = 0;
    while (p < e) {
        if (*p == '\\') { p += 2; continue; }
        if (in_class) {
            if (*p == '[') in_class++;
            else if (*p == ']') in_class--;
            p++;
            continue;
        }
        if (*p == '[') { in_class = 1; p++; if (p < e && *p == '^') p++; if (p < e && *p == ']') p++; continue; }
        if (*p == '(' && p + 2 < e && p[1] == '?' && ((p[2] == '<' && p + 3 < e && p[3] != '=' && p[3] != '!') || p[2] == '\'')) return 1;
        p++;
    }
    return 0;
}

orx_t *orx_compile(const char *pat, int len, unsigned options, char *err, int errlen)
{
    P ps;
    int opts = 0;
    struct orx *rx = calloc(1, sizeof(*rx));
    memset(&ps, 0, sizeof(ps));
    ps.s = ps.p = (const unsigned char *) pat;
    ps.e = ps.s + len;
    ps.rx = rx;
    ps.err = err; ps.errlen = errlen;
    if (err && errlen) err[0] = 0;
    if (options & ORX_OPT_IGNORECASE) opts |= OPT_I;
    if (options & ORX_OPT_MULTILINE) opts |= OPT_M;
    if (options & ORX_OPT_EXTEND) opts |= OPT_X;
    ps.has_named = scan_named(ps.s, ps.e);
    rx->root = parse_alt(&ps, &opts, 0);
    if (!ps.failed && ps.p < ps.e) perr(&ps, *ps.p == ')' ? "unmatched close parenthesis" : "trailing garbage");
    if (ps.failed) { orx_free(rx); return NULL; }
    rx->ncap = ps.ncap;
    return rx;
}

void orx_free(orx_t *rx)
{
    int i;
    if (!rx) return;
    for (i = 0; i < rx->nall; i++) { free(rx->all[i]->ranges); free(rx->all[i]->kids); free(rx->all[i]); }
    free(rx->all);
    for (i = 0; i < rx->nnames; i++) free(rx->names[i]);
    free(rx);
}

int orx_num_groups(const orx_t *rx) { return rx->ncap; }
int orx_num_names(const orx_t *rx) { return rx->nnames; }
const char *orx_name(const orx_t *rx, int i) { return rx->names[i]; }
int orx_name_ngroups(const orx_t *rx, int i) { return rx->name_ngroups[i]; }
int orx_name_group(const orx_t *rx, int i, int k) { return rx->name_groups[i][k]; }

/* ------------------------------------------------------------------ matcher */
typedef struct M {
    const unsigned char *s;
    int len;
    int beg[ORX_MAX_GROUPS], end[ORX_MAX_GROUPS];
    int match_end;
    long steps;
} M;

enum { K_NODE, K_REP, K_CAPEND, K_LOOKEND, K_CAT };

typedef struct Cont {
    int kind;
    Node *node;             /* K_NODE: node to match next; K_REP: the repeat node; K_CAT: the cat node */
    int idx;                /* K_CAT: next kid index */
    int count;              /* K_REP */
    int start;              /* K_REP: position at iteration start */
    int cap;                /* K_CAPEND */
    struct Cont *next;
} Cont;

/* decode the char at pos following Onigmo's length rules (valid UTF-8: one code point;
 * invalid lead/continuation: one byte whose code is the byte value) */
static int dec(const M *m, int pos, uint32_t *cp)
{
    const unsigned char *p = m->s + pos;
    int rem = m->len - pos;
    uint32_t c = p[0];
    if (c < 0x80) { *cp = c; return 1; }
    if (c >= 0xc2 && c <= 0xdf) {
        if (rem >= 2 && (p[1] & 0xc0) == 0x80) { *cp = ((c & 0x1f) << 6) | (p[1] & 0x3f); return 2; }
    }
    else if (c >= 0xe0 && c <= 0xef) {
        if (rem >= 3 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80) {
            int ok = 1;
            if (c == 0xe0 && p[1] < 0xa0) ok = 0;
            if (c == 0xed && p[1] > 0x9f) ok = 0;
            if (ok) { *cp = ((c & 0x0f) << 12) | ((p[1] & 0x3f) << 6) | (p[2] & 0x3f); return 3; }
        }
    }
    else if (c >= 0xf0 && c <= 0xf4) {
        if (rem >= 4 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80 && (p[3] & 0xc0) == 0x80) {
            int ok = 1;
            if (c == 0xf0 && p[1] < 0x90) ok = 0;
            if (c == 0xf4 && p[1] > 0x8f) ok = 0;
            if (ok) { *cp = ((c & 0x07) << 18) | ((p[1] & 0x3f) << 12) | ((p[2] & 0x3f) << 6) | (p[3] & 0x3f); return 4; }
        }
    }
    *cp = c;
    return 1;
}

static int in_class(const Node *n, uint32_t cp)
{
    int lo = 0, hi = n->nranges - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (cp < n->ranges[mid].lo) hi = mid - 1;
        else if (cp > n->ranges[mid].hi) lo = mid + 1;
        else return 1;
    }
    return 0;
}

static int is_word_byte(int c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }

/* single-char matchers: returns consumed length or 0 */
static int one(const M *m, const Node *n, int pos)
{
    uint32_t cp;
    int l;
    if (pos >= m->len) return 0;
    l = dec(m, pos, &cp);
    switch (n->type) {
    case N_CHAR: return cp == n->ch ? l : 0;
    case N_ANY: return (cp == '\n' && !n->multiline) ? 0 : l;
    case N_CLASS: return in_class(n, cp) ? l : 0;
    }
    return 0;
}

static int is_single(const Node *n) { return n->type == N_CHAR || n->type == N_ANY || n->type == N_CLASS; }

static int run(M *m, Cont *k, int pos);
static int mnode(M *m, Node *n, int pos, Cont *k);

static int anchor_ok(const M *m, int a, int pos)
{
    switch (a) {
    /* OP_BEGIN_LINE (lib/onigmo/regexec.c): after a newline only when not at end of string */
    case A_BOL: return pos == 0 || (m->s[pos - 1] == '\n' && pos != m->len);
    case A_EOL: return pos == m->len || m->s[pos] == '\n';
    case A_BOS: return pos == 0;
    case A_EOS: return pos == m->len;
    case A_EOS_NL: return pos == m->len || (pos == m->len - 1 && m->s[pos] == '\n');
    case A_WORDB: case A_NWORDB: {
        /* ASCII view of \b; non-ASCII word characters are a documented deviation */
        int l = pos > 0 && is_word_byte(m->s[pos - 1]);
        int r = pos < m->len && is_word_byte(m->s[pos]);
        return a == A_WORDB ? (l != r) : (l == r);
    }
    }
    return 0;
}

static int rep_step(M *m, Node *r, int count, int pos, Cont *k);

static int run(M *m, Cont *k, int pos)
{
    if (k == NULL) { m->match_end = pos; return 1; }
    switch (k->kind) {
    case K_NODE: return mnode(m, k->node, pos, k->next);
    case K_CAT: {
        Node *c = k->node;
        if (k->idx >= c->nkids) return run(m, k->next, pos);
        {
            Cont nk = *k;
            nk.idx = k->idx + 1;
            return mnode(m, c->kids[k->idx], pos, &nk);
        }
    }
    case K_REP:
        /* end of one iteration of a general repeat */
        if (pos == k->start && k->node->max < 0 && k->count >= k->node->min) {
            /* Onigmo null check (OP_NULL_CHECK_END): an empty iteration leaves the loop */
            return run(m, k->next, pos);
        }
        return rep_step(m, k->node, k->count + 1, pos, k->next);
    case K_CAPEND: {
        int old = m->end[k->cap];
        m->end[k->cap] = pos;
        if (run(m, k->next, pos)) return 1;
        m->end[k->cap] = old;
        return 0;
    }
    }
    return 0;
}

static int rep_step(M *m, Node *r, int count, int pos, Cont *k)
{
    Cont it;
    it.kind = K_REP; it.node = r; it.count = count; it.start = pos; it.next = k; it.idx = 0; it.cap = 0;
    if (count < r->min) return mnode(m, r->sub, pos, &it);
    if (r->max >= 0 && count >= r->max) return run(m, k, pos);
    if (r->greedy) {
        if (mnode(m, r->sub, pos, &it)) return 1;
        return run(m, k, pos);
    }
    if (run(m, k, pos)) return 1;
    return mnode(m, r->sub, pos, &it);
}

static int mnode(M *m, Node *n, int pos, Cont *k)
{
    int l;
    if (++m->steps > 50000000L) return 0;
    switch (n->type) {
    case N_EMPTY: return run(m, k, pos);
    case N_CHAR: case N_ANY: case N_CLASS:
        l = one(m, n, pos);
        if (!l) return 0;
        return run(m, k, pos + l);
    case N_ANCHOR:
        if (!anchor_ok(m, n->anchor, pos)) return 0;
        return run(m, k, pos);
    case N_CAT: {
        Cont c;
        c.kind = K_CAT; c.node = n; c.idx = 0; c.next = k; c.count = c.start = c.cap = 0;
        return run(m, &c, pos);
    }
    case N_ALT: {
        int i;
        for (i = 0; i < n->nkids; i++) if (mnode(m, n->kids[i], pos, k)) return 1;
        return 0;
    }
    case N_GROUP:
        if (n->cap) {
            Cont c;
            int ob = m->beg[n->cap];
            c.kind = K_CAPEND; c.cap = n->cap; c.next = k; c.node = NULL; c.idx = c.count = c.start = 0;
            m->beg[n->cap] = pos;
            if (mnode(m, n->sub, pos, &c)) return 1;
            m->beg[n->cap] = ob;
            return 0;
        }
        return mnode(m, n->sub, pos, k);
    case N_LOOK: {
        int sb[ORX_MAX_GROUPS], se[ORX_MAX_GROUPS], save_end = m->match_end, r;
        memcpy(sb, m->beg, sizeof(sb)); memcpy(se, m->end, sizeof(se));
        r = mnode(m, n->sub, pos, NULL);
        m->match_end = save_end;
        if (n->neg) {
            memcpy(m->beg, sb, sizeof(sb)); memcpy(m->end, se, sizeof(se));
            return r ? 0 : run(m, k, pos);
        }
        if (!r) return 0;
        if (run(m, k, pos)) return 1;
        memcpy(m->beg, sb, sizeof(sb)); memcpy(m->end, se, sizeof(se));
        return 0;
    }
    case N_ATOMIC: {
        int sb[ORX_MAX_GROUPS], se[ORX_MAX_GROUPS], save_end = m->match_end, e;
        memcpy(sb, m->beg, sizeof(sb)); memcpy(se, m->end, sizeof(se));
        if (!mnode(m, n->sub, pos, NULL)) return 0;
        e = m->match_end;
        m->match_end = save_end;
        if (run(m, k, e)) return 1;
        memcpy(m->beg, sb, sizeof(sb)); memcpy(m->end, se, sizeof(se));
        return 0;
    }
    case N_REPEAT:
        if (n->possessive) {
            /* a*+  ==  (?>a*) */
            int sb[ORX_MAX_GROUPS], se[ORX_MAX_GROUPS], save_end = m->match_end, e;
            Node tmp = *n;
            tmp.possessive = 0;
            memcpy(sb, m->beg, sizeof(sb)); memcpy(se, m->end, sizeof(se));
            if (!mnode(m, &tmp, pos, NULL)) return 0;
            e = m->match_end;
            m->match_end = save_end;
            if (run(m, k, e)) return 1;
            memcpy(m->beg, sb, sizeof(sb)); memcpy(m->end, se, sizeof(se));
            return 0;
        }
        if (is_single(n->sub)) {
            /* iterative fast path (keeps recursion depth independent of the run length) */
            int cnt = 0, p = pos, stackpos[64], *posv = stackpos, capv = 64, i, r = 0;
            if (n->greedy) {
                for (;;) {
                    if (cnt == capv) {
                        int *nv = malloc(sizeof(int) * capv * 2);
                        memcpy(nv, posv, sizeof(int) * capv);
                        if (posv != stackpos) free(posv);
                        posv = nv; capv *= 2;
                    }
                    posv[cnt] = p;
                    if (n->max >= 0 && cnt >= n->max) break;
                    l = one(m, n->sub, p);
                    if (!l) break;
                    p += l; cnt++;
                }
                for (i = cnt; i >= n->min; i--) {
                    if (run(m, k, posv[i])) { r = 1; break; }
                }
                if (posv != stackpos) free(posv);
                return r;
            }
            /* lazy */
            while (cnt < n->min) {
                l = one(m, n->sub, p);
                if (!l) return 0;
                p += l; cnt++;
            }
            for (;;) {
                if (run(m, k, p)) return 1;
                if (n->max >= 0 && cnt >= n->max) return 0;
                l = one(m, n->sub, p);
                if (!l) return 0;
                p += l; cnt++;
            }
        }
        return rep_step(m, n, 0, pos, k);
    }
    return 0;
}

int orx_search(const orx_t *rx, const char *s, int len, int *beg, int *end, int max)
{
    M m;
    int start, i;
    m.s = (const unsigned char *) s;
    m.len = len;
    m.steps = 0;
    for (start = 0; start <= len; ) {
        uint32_t cp;
        for (i = 0; i <= rx->ncap; i++) { m.beg[i] = -1; m.end[i] = -1; }
        m.match_end = -1;
        if (mnode(&m, rx->root, start, NULL)) {
            m.beg[0] = start; m.end[0] = m.match_end;
            for (i = 0; i <= rx->ncap && i < max; i++) {
                /* a group that was entered but never closed on the winning path reports unset */
                if (m.end[i] < 0 || m.beg[i] < 0) { beg[i] = -1; end[i] = -1; }
                else { beg[i] = m.beg[i]; end[i] = m.end[i]; }
            }
            return rx->ncap + 1;
        }
        if (start >= len) break;
        start += dec(&m, start, &cp);
    }
    return -1;
}

int orx_match(const orx_t *rx, const char *s, int len)
{
    int b[ORX_MAX_GROUPS], e[ORX_MAX_GROUPS];
    return orx_search(rx, s, len, b, e, ORX_MAX_GROUPS) >= 0 ? 1 : 0;
}
