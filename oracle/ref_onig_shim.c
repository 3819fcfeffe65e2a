/*
 * oracle/ref_onig_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin ctypes-friendly wrapper around the REAL reference regex engine
 * (Onigmo 6.2.0, vendored by the reference at lib/onigmo).  oracle/Makefile
 * compiles it together with the reference's own Onigmo sources, read in place
 * from /root/reference/lib/onigmo, into oracle/_ref/libonig_ref.so.  No
 * reference source is copied into this repository.
 *
 * The call pattern mirrors the reference wrapper src/flb_regex.c:
 *   onig_new(..., ONIG_ENCODING_UTF8, ONIG_SYNTAX_RUBY)   (src/flb_regex.c:142-145)
 *   onig_search(reg, str, end, str, end, region, NONE)   (src/flb_regex.c:199-204)
 *   onig_foreach_name() named-group iteration order      (src/flb_regex.c:306)
 */
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include "onigmo.h"

int ref_onig_new(const char *pat, int patlen, unsigned int options, void **out)
{
    OnigRegex reg;
    OnigErrorInfo einfo;
    int r = onig_new(&reg, (const OnigUChar *) pat, (const OnigUChar *) pat + patlen,
                     (OnigOptionType) options, ONIG_ENCODING_UTF8, ONIG_SYNTAX_RUBY,
                     &einfo);
    if (r != ONIG_NORMAL) {
        *out = NULL;
        return r;
    }
    *out = reg;
    return 0;
}

void ref_onig_free(void *reg)
{
    onig_free((OnigRegex) reg);
}

/* returns num_regs (>=1) on match and fills beg/end (up to max), -1 mismatch, < -1 error */
int ref_onig_search(void *reg, const char *s, int len, int *beg, int *end, int max)
{
    OnigRegion *region = onig_region_new();
    const OnigUChar *str = (const OnigUChar *) s;
    OnigPosition r = onig_search((OnigRegex) reg, str, str + len, str, str + len,
                                 region, ONIG_OPTION_NONE);
    int n, i;
    if (r < 0) {
        onig_region_free(region, 1);
        return (int) r;
    }
    n = region->num_regs;
    for (i = 0; i < n && i < max; i++) {
        beg[i] = (int) region->beg[i];
        end[i] = (int) region->end[i];
    }
    onig_region_free(region, 1);
    return n;
}

/* boolean search without a region (src/flb_regex.c:270-291) */
int ref_onig_match(void *reg, const char *s, int len)
{
    const OnigUChar *str = (const OnigUChar *) s;
    OnigPosition r = onig_search((OnigRegex) reg, str, str + len, str, str + len,
                                 NULL, ONIG_OPTION_NONE);
    if (r == ONIG_MISMATCH) return 0;
    if (r < 0) return (int) r;
    return 1;
}

struct name_acc { char *buf; int cap; int len; };

static int name_cb(const OnigUChar *name, const OnigUChar *name_end, int ngroups,
                   int *groups, OnigRegex reg, void *arg)
{
    struct name_acc *a = (struct name_acc *) arg;
    int i;
    int nlen = (int) (name_end - name);
    (void) reg;
    for (i = 0; i < ngroups; i++) {
        char num[16];
        int k = 0, g = groups[i], j;
        char tmp[16];
        if (a->len + nlen + 16 >= a->cap) return 1;
        memcpy(a->buf + a->len, name, nlen);
        a->len += nlen;
        a->buf[a->len++] = '=';
        if (g == 0) tmp[k++] = '0';
        while (g > 0) { tmp[k++] = (char) ('0' + g % 10); g /= 10; }
        for (j = 0; j < k; j++) num[j] = tmp[k - 1 - j];
        memcpy(a->buf + a->len, num, k);
        a->len += k;
        a->buf[a->len++] = '\n';
    }
    return 0;
}

/* writes "name=groupnum\n" lines in onig_foreach_name order; returns length */
int ref_onig_names(void *reg, char *buf, int cap)
{
    struct name_acc a;
    a.buf = buf; a.cap = cap; a.len = 0;
    onig_foreach_name((OnigRegex) reg, name_cb, &a);
    if (a.len < cap) a.buf[a.len] = '\0';
    return a.len;
}

/* timing loop for the cpu_baseline leg: onig_search with a region over n rows */
double ref_onig_bench(void *reg, const char *data, const long long *off, long long n,
                      long long *matched)
{
    struct timespec t0, t1;
    long long i, m = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < n; i++) {
        OnigRegion *region = onig_region_new();
        const OnigUChar *s = (const OnigUChar *) data + off[i];
        const OnigUChar *e = (const OnigUChar *) data + off[i + 1];
        if (onig_search((OnigRegex) reg, s, e, s, e, region, ONIG_OPTION_NONE) >= 0) m++;
        onig_region_free(region, 1);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *matched = m;
    return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/* membership probe for tools/gen_posix_ranges.py: out[c - lo] = 1 when the one-character text holding
 * code point c (UTF-8; surrogates skipped) is matched by reg */
void ref_onig_probe_codepoints(void *reg, unsigned lo, unsigned hi, unsigned char *out)
{
    unsigned c;
    for (c = lo; c <= hi; c++) {
        OnigUChar b[4];
        int n;
        out[c - lo] = 0;
        if (c >= 0xd800 && c <= 0xdfff) continue;
        if (c < 0x80) { b[0] = (OnigUChar) c; n = 1; }
        else if (c < 0x800) { b[0] = 0xc0 | (c >> 6); b[1] = 0x80 | (c & 0x3f); n = 2; }
        else if (c < 0x10000) { b[0] = 0xe0 | (c >> 12); b[1] = 0x80 | ((c >> 6) & 0x3f); b[2] = 0x80 | (c & 0x3f); n = 3; }
        else { b[0] = 0xf0 | (c >> 18); b[1] = 0x80 | ((c >> 12) & 0x3f); b[2] = 0x80 | ((c >> 6) & 0x3f); b[3] = 0x80 | (c & 0x3f); n = 4; }
        if (onig_search((OnigRegex) reg, b, b + n, b, b + n, NULL, ONIG_OPTION_NONE) >= 0) out[c - lo] = 1;
    }
}
