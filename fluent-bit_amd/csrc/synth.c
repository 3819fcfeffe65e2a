/*
 * synth.c -- seeded synthetic workload generator (harness utility, not on the filter path).
 *
 * Produces the inputs SURVEY.md section 8(d) specifies: apache "combined" access-log lines of
 * EXACTLY 256 bytes, wrapped as Fluent Bit V2 log events
 *     92 92 d7 00 <sec32be> <nsec32be> 80 81 a3 'log' da 01 00 <256 bytes>      (277 bytes)
 * (layout: src/flb_log_event_encoder.c:195-217 in the reference), plus the row-offset column
 * the kernels consume.  Field distributions: host IPv4 uniform; user 90 % "-"; time uniform
 * over one day with +0000 (tz_mixed=0) or mixed +-hhmm; method GET 80 % / POST 15 % / other;
 * status 200 70 %, 3xx 10 %, 404 10 %, 5xx 10 %; size log-uniform; referer/agent from 64-entry
 * pools; the request path is sized so that the line is exactly line_len bytes.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static inline uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static const char *const MON[12] = { "Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec" };
static const char *const OTHER_METHODS[4] = { "PUT", "DELETE", "HEAD", "OPTIONS" };
static const int TZS[8] = { 0, 100, 200, 530, 900, -500, -700, -800 };
static const char PATHCH[] = "abcdefghijklmnopqrstuvwxyz0123456789/-_.";

static int pool_str(char *dst, int kind, unsigned idx)
{
    /* 64-entry deterministic pools */
    if (kind == 0) {
        if (idx % 8 == 0) return sprintf(dst, "-");
        return sprintf(dst, "http://www.example%u.com/path/%u/index.html", idx % 17, idx);
    }
    switch (idx % 4) {
    case 0: return sprintf(dst, "Mozilla/5.0 (X11; Linux x86_64; rv:%u.0) Gecko/20100101 Firefox/%u.0", 60 + idx, 60 + idx);
    case 1: return sprintf(dst, "Mozilla/5.0 (Windows NT 10.0; Win64; x64) AppleWebKit/537.36 Chrome/%u.0.0.0", 90 + idx);
    case 2: return sprintf(dst, "curl/7.%u.0", idx);
    default: return sprintf(dst, "Mozilla/4.08 [en] (Win98; I ;Nav) build %u", idx);
    }
}

/* one line of exactly line_len bytes into dst; returns line_len, or -1 if it cannot fit */
static int gen_line(uint64_t *rng, char *dst, int line_len, int tz_mixed, uint32_t *epoch_out)
{
    char head[160], tail[512], ref[128], agent[128], method[16];
    uint64_t r = splitmix(rng);
    unsigned a = (r >> 0) & 255, b = (r >> 8) & 255, c = (r >> 16) & 255, d = (r >> 24) & 255;
    unsigned sod = (unsigned) ((r >> 32) % 86400u);
    uint64_t r2 = splitmix(rng);
    unsigned pm = (unsigned) (r2 % 100), ps = (unsigned) ((r2 >> 8) % 100), pu = (unsigned) ((r2 >> 16) % 100);
    int tz = tz_mixed ? TZS[(r2 >> 24) & 7] : 0;
    int status, hl, tl, pl, i;
    unsigned size;
    /* day: 2024-03-10 .. 2024-03-10 (one day), time uniform */
    int hh = sod / 3600, mm = (sod / 60) % 60, ss = sod % 60;
    const char *user = pu < 90 ? "-" : (pu < 95 ? "frank" : "alice");

    if (pm < 80) strcpy(method, "GET");
    else if (pm < 95) strcpy(method, "POST");
    else strcpy(method, OTHER_METHODS[(r2 >> 28) & 3]);
    if (ps < 70) status = 200;
    else if (ps < 80) status = 301 + (int) ((r2 >> 32) % 4);
    else if (ps < 90) status = 404;
    else status = 500 + (int) ((r2 >> 32) % 4);
    {
        unsigned bits = (unsigned) ((r2 >> 40) % 24);
        size = (unsigned) ((1u << bits) + ((r2 >> 45) & ((1u << bits) - 1)));
    }
    pool_str(ref, 0, (unsigned) ((r2 >> 50) & 63));
    pool_str(agent, 1, (unsigned) ((r2 >> 56) & 63));

    hl = sprintf(head, "%u.%u.%u.%u - %s [10/%s/2024:%02d:%02d:%02d %c%04d] \"%s /", a, b, c, d, user,
                 MON[2], hh, mm, ss, tz < 0 ? '-' : '+', tz < 0 ? -tz : tz, method);
    tl = sprintf(tail, " HTTP/1.1\" %d %u \"%s\" \"%s\"", status, size, ref, agent);
    pl = line_len - hl - tl;
    if (pl < 0) return -1;
    memcpy(dst, head, hl);
    {
        uint64_t pr = splitmix(rng);
        for (i = 0; i < pl; i++) {
            if ((i & 7) == 0) pr = splitmix(rng);
            dst[hl + i] = PATHCH[(pr >> ((i & 7) * 8)) % (sizeof(PATHCH) - 1)];
        }
    }
    memcpy(dst + hl + pl, tail, tl);
    if (epoch_out) {
        /* 2024-03-10T00:00:00Z = 1710028800 */
        int tzsec = (tz / 100) * 3600 + (tz % 100) * 60;
        *epoch_out = (uint32_t) (1710028800u + sod - tzsec);
    }
    return line_len;
}

/*
 * Fills `out` with n V2 records whose body is {"log": <line>} and `off` (n+1 entries) with the
 * record start offsets.  Record timestamps are 1700000000 + i/1000 s, nsec = (i%1000)*1000000.
 * Returns total bytes written, or 0 when `cap` is too small.
 */
uint64_t flbsynth_apache_records(uint64_t seed, uint64_t n, int line_len, int tz_mixed, char *out,
                                 uint64_t cap, uint64_t *off, uint32_t *epochs)
{
    uint64_t rng = seed, pos = 0, i;
    int hdr;
    for (i = 0; i < n; i++) {
        unsigned char *p;
        uint32_t sec = 1700000000u + (uint32_t) (i / 1000), nsec = (uint32_t) (i % 1000) * 1000000u;
        hdr = 4 + 8 + 1 + 1 + 4 + (line_len < 32 ? 1 : line_len < 256 ? 2 : line_len < 65536 ? 3 : 5);
        if (pos + hdr + line_len > cap) return 0;
        off[i] = pos;
        p = (unsigned char *) out + pos;
        *p++ = 0x92; *p++ = 0x92; *p++ = 0xd7; *p++ = 0x00;
        *p++ = sec >> 24; *p++ = sec >> 16; *p++ = sec >> 8; *p++ = sec;
        *p++ = nsec >> 24; *p++ = nsec >> 16; *p++ = nsec >> 8; *p++ = nsec;
        *p++ = 0x80;
        *p++ = 0x81; *p++ = 0xa3; *p++ = 'l'; *p++ = 'o'; *p++ = 'g';
        if (line_len < 32) *p++ = 0xa0 | line_len;
        else if (line_len < 256) { *p++ = 0xd9; *p++ = line_len; }
        else if (line_len < 65536) { *p++ = 0xda; *p++ = line_len >> 8; *p++ = line_len; }
        else { *p++ = 0xdb; *p++ = line_len >> 24; *p++ = line_len >> 16; *p++ = line_len >> 8; *p++ = line_len; }
        if (gen_line(&rng, (char *) p, line_len, tz_mixed, epochs ? &epochs[i] : NULL) < 0) return 0;
        pos += hdr + line_len;
    }
    off[n] = pos;
    return pos;
}
