/*
 * oracle/otime.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Restatement of the reference's BSD-derived strptime, src/flb_strptime.c:253-907, in the C
 * locale (the reference reads month/day names through nl_langinfo; fluent-bit never calls
 * setlocale, so they are the C-locale English names).  The reference keeps century / relyear /
 * fields in function statics (src/flb_strptime.c:261); this restatement keeps them in a
 * per-call state, which is observably identical for the single-threaded call pattern.
 * The tzname fallback of %Z (:630-650) is restated for a process whose zone is UTC.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <string.h>
#include <strings.h>
#include <stdlib.h>
#include <limits.h>
#include "otime.h"

#define TM_YEAR_BASE 1900
#define F_MON 1
#define F_MDAY 2
#define F_WDAY 4
#define F_YDAY 8
#define F_YEAR 16

static const char *const day_full[7] = { "Sunday", "Monday", "Tuesday", "Wednesday", "Thursday", "Friday", "Saturday" };
static const char *const day_ab[7] = { "Sun", "Mon", "Tue", "Wed", "Thu", "Fri", "Sat" };
static const char *const mon_full[12] = { "January", "February", "March", "April", "May", "June", "July",
                                          "August", "September", "October", "November", "December" };
static const char *const mon_ab[12] = { "Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec" };

/* src/flb_strptime.c:97-196 flb_known_timezones */
static const struct { const char *abbr; int off; int dst; } known_tz[] = {
    {"GMT", 0, 0}, {"UTC", 0, 0}, {"Z", 0, 0}, {"UT", 0, 0},
    {"EST", -5 * 3600, 0}, {"EDT", -4 * 3600, 1}, {"CST", -6 * 3600, 0}, {"CDT", -5 * 3600, 1},
    {"MST", -7 * 3600, 0}, {"MDT", -6 * 3600, 1}, {"PST", -8 * 3600, 0}, {"PDT", -7 * 3600, 1},
    {"AKST", -9 * 3600, 0}, {"AKDT", -8 * 3600, 1}, {"HST", -10 * 3600, 0}, {"HADT", -9 * 3600, 1},
    {"AST", -4 * 3600, 0}, {"ADT", -3 * 3600, 1}, {"NST", -12600, 0}, {"NDT", -9000, 1},
    {"WET", 0, 0}, {"WEST", 3600, 1}, {"CET", 3600, 0}, {"CEST", 7200, 1}, {"EET", 7200, 0},
    {"EEST", 10800, 1}, {"MSK", 10800, 0},
    {"ART", -10800, 0}, {"BRT", -10800, 0}, {"BRST", -7200, 1}, {"CLT", -14400, 0}, {"CLST", -10800, 1},
    {"AEST", 36000, 0}, {"AEDT", 39600, 1}, {"ACST", 34200, 0}, {"ACDT", 37800, 1}, {"AWST", 28800, 0},
    {"NZST", 43200, 0}, {"NZDT", 46800, 1},
    {"JST", 32400, 0}, {"KST", 32400, 0}, {"SGT", 28800, 0}, {"IST", 19800, 0}, {"GST", 14400, 0},
    {"ICT", 25200, 0}, {"WIB", 25200, 0}, {"WITA", 28800, 0}, {"WIT", 32400, 0}, {"MYT", 28800, 0},
    {"BDT", 21600, 0}, {"NPT", 20700, 0},
    {"WAT", 3600, 0}, {"CAT", 7200, 0}, {"EAT", 10800, 0}, {"SAST", 7200, 0},
    {"A", 3600, 0}, {"B", 7200, 0}, {"C", 10800, 0}, {"D", 14400, 0}, {"E", 18000, 0}, {"F", 21600, 0},
    {"G", 25200, 0}, {"H", 28800, 0}, {"I", 32400, 0}, {"K", 36000, 0}, {"L", 39600, 0}, {"M", 43200, 0},
    {"N", -3600, 0}, {"O", -7200, 0}, {"P", -10800, 0}, {"Q", -14400, 0}, {"R", -18000, 0},
    {"S", -21600, 0}, {"T", -25200, 0}, {"U", -28800, 0}, {"V", -32400, 0}, {"W", -36000, 0},
    {"X", -39600, 0}, {"Y", -43200, 0},
    {NULL, 0, 0}
};

struct st { int century, relyear, fields; };

static int conv_num(const unsigned char **buf, int *dest, int llim, int ulim)
{
    int result = 0, rulim = ulim;
    if (**buf < '0' || **buf > '9') return 0;
    do {
        result *= 10;
        result += *(*buf)++ - '0';
        rulim /= 10;
    } while ((result * 10 <= ulim) && rulim && **buf >= '0' && **buf <= '9');
    if (result < llim || result > ulim) return 0;
    *dest = result;
    return 1;
}

static int conv_num64(const unsigned char **buf, int64_t *dest, int64_t llim, int64_t ulim)
{
    int64_t result = 0, rulim = ulim;
    if (**buf < '0' || **buf > '9') return 0;
    do {
        if (result > 922337203685477580LL) return 0;
        result *= 10;
        if (result > 9223372036854775760LL) return 0;
        result += *(*buf)++ - '0';
        rulim /= 10;
        if (result >= 922337203685477580LL) return 0;
    } while ((result * 10 <= ulim) && rulim && **buf >= '0' && **buf <= '9');
    if (result < llim || result > ulim) return 0;
    *dest = result;
    return 1;
}

static int leaps_thru_end_of(int y)
{
    return (y >= 0) ? (y / 4 - y / 100 + y / 400) : -(leaps_thru_end_of(-(y + 1)) + 1);
}

#define isleap(y) (((y) % 4) == 0 && (((y) % 100) != 0 || ((y) % 400) == 0))

static const unsigned char *find_string(const unsigned char *bp, int *tgt, const char *const *n1, int c)
{
    int i;
    for (i = 0; i < c; i++) {
        size_t len = strlen(n1[i]);
        if (strncasecmp(n1[i], (const char *) bp, len) == 0) { *tgt = i; return bp + len; }
    }
    return NULL;
}

static const char *sp(const char *buf, const char *fmt, struct otm *tm, struct st *st, int initialize)
{
    unsigned char c;
    const unsigned char *bp, *ep;
    size_t len = 0;
    int i, offs, neg;
    static const int mon_lengths[2][12] = {
        { 31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 },
        { 31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 } };
    static const char *const nast[4] = { "EST", "CST", "MST", "PST" };
    static const char *const nadt[4] = { "EDT", "CDT", "MDT", "PDT" };

    if (initialize) {
        st->century = TM_YEAR_BASE; st->relyear = -1; st->fields = 0;
        tm->gmtoff = 0;
        tm->tm.tm_isdst = -1;
    }
    bp = (const unsigned char *) buf;
    while ((c = *fmt) != '\0') {
        if (isspace(c)) {
            while (isspace(*bp)) bp++;
            fmt++;
            continue;
        }
        if (*bp == '\0') return NULL;
        if ((c = *fmt++) != '%') goto literal;
again:
        switch (c = *fmt++) {
        case '%':
literal:
            if (c != *bp++) return NULL;
            break;
        case 'E': case 'O':
            goto again;
        case 'c':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%a %b %e %H:%M:%S %Y", tm, st, 0))) return NULL;
            break;
        case 'D':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%m/%d/%y", tm, st, 0))) return NULL;
            break;
        case 'F':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%Y-%m-%d", tm, st, 0))) return NULL;
            continue;
        case 'R':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%H:%M", tm, st, 0))) return NULL;
            break;
        case 'r':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%I:%M:%S %p", tm, st, 0))) return NULL;
            break;
        case 'T': case 'X':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%H:%M:%S", tm, st, 0))) return NULL;
            break;
        case 'x':
            if (!(bp = (const unsigned char *) sp((const char *) bp, "%m/%d/%y", tm, st, 0))) return NULL;
            break;
        case 'A': case 'a':
            for (i = 0; i < 7; i++) {
                len = strlen(day_full[i]);
                if (strncasecmp(day_full[i], (const char *) bp, len) == 0) break;
                len = strlen(day_ab[i]);
                if (strncasecmp(day_ab[i], (const char *) bp, len) == 0) break;
            }
            if (i == 7) return NULL;
            tm->tm.tm_wday = i;
            bp += len;
            st->fields |= F_WDAY;
            break;
        case 'B': case 'b': case 'h':
            for (i = 0; i < 12; i++) {
                len = strlen(mon_full[i]);
                if (strncasecmp(mon_full[i], (const char *) bp, len) == 0) break;
                len = strlen(mon_ab[i]);
                if (strncasecmp(mon_ab[i], (const char *) bp, len) == 0) break;
            }
            if (i == 12) return NULL;
            tm->tm.tm_mon = i;
            bp += len;
            st->fields |= F_MON;
            break;
        case 'C':
            if (!conv_num(&bp, &i, 0, 99)) return NULL;
            st->century = i * 100;
            break;
        case 'e':
            if (isspace(*bp)) bp++;
            /* FALLTHROUGH */
        case 'd':
            if (!conv_num(&bp, &tm->tm.tm_mday, 1, 31)) return NULL;
            st->fields |= F_MDAY;
            break;
        case 'k': case 'H':
            if (!conv_num(&bp, &tm->tm.tm_hour, 0, 23)) return NULL;
            break;
        case 'l': case 'I':
            if (!conv_num(&bp, &tm->tm.tm_hour, 1, 12)) return NULL;
            break;
        case 'j':
            if (!conv_num(&bp, &tm->tm.tm_yday, 1, 366)) return NULL;
            tm->tm.tm_yday--;
            st->fields |= F_YDAY;
            break;
        case 'M':
            if (!conv_num(&bp, &tm->tm.tm_min, 0, 59)) return NULL;
            break;
        case 'm':
            if (!conv_num(&bp, &tm->tm.tm_mon, 1, 12)) return NULL;
            tm->tm.tm_mon--;
            st->fields |= F_MON;
            break;
        case 'p':
            if (strncasecmp("AM", (const char *) bp, 2) == 0) {
                if (tm->tm.tm_hour > 12) return NULL;
                else if (tm->tm.tm_hour == 12) tm->tm.tm_hour = 0;
                bp += 2;
                break;
            }
            if (strncasecmp("PM", (const char *) bp, 2) == 0) {
                if (tm->tm.tm_hour > 12) return NULL;
                else if (tm->tm.tm_hour < 12) tm->tm.tm_hour += 12;
                bp += 2;
                break;
            }
            return NULL;
        case 'S':
            if (!conv_num(&bp, &tm->tm.tm_sec, 0, 60)) return NULL;
            break;
        case 's': {
            int64_t i64;
            time_t tt;
            if (!conv_num64(&bp, &i64, 0, INT64_MAX)) return NULL;
            tt = (time_t) i64;
            if (!gmtime_r(&tt, &tm->tm)) return NULL;
            tm->gmtoff = 0;
            tm->tm.tm_isdst = 0;
            st->fields = 0xffff;
            break;
        }
        case 'U': case 'W':
            if (!conv_num(&bp, &i, 0, 53)) return NULL;
            break;
        case 'w':
            if (!conv_num(&bp, &tm->tm.tm_wday, 0, 6)) return NULL;
            st->fields |= F_WDAY;
            break;
        case 'u':
            if (!conv_num(&bp, &i, 1, 7)) return NULL;
            tm->tm.tm_wday = i % 7;
            st->fields |= F_WDAY;
            continue;
        case 'g':
            if (!conv_num(&bp, &i, 0, 99)) return NULL;
            continue;
        case 'G':
            do bp++; while (isdigit(*bp));
            continue;
        case 'V':
            if (!conv_num(&bp, &i, 0, 53)) return NULL;
            continue;
        case 'Y':
            if (!conv_num(&bp, &i, 0, 9999)) return NULL;
            st->relyear = -1;
            tm->tm.tm_year = i - TM_YEAR_BASE;
            st->fields |= F_YEAR;
            break;
        case 'y':
            if (!conv_num(&bp, &st->relyear, 0, 99)) return NULL;
            break;
        case 'Z': {
            int k, found = 0;
            for (k = 0; known_tz[k].abbr; k++) {
                size_t al = strlen(known_tz[k].abbr);
                if (strncasecmp(known_tz[k].abbr, (const char *) bp, al) == 0) {
                    if (!isalnum((unsigned char) bp[al])) {
                        tm->tm.tm_isdst = known_tz[k].dst;
                        tm->gmtoff = known_tz[k].off;
                        bp += al;
                        found = 1;
                        break;
                    }
                }
            }
            if (!found) {
                if (strncmp((const char *) bp, "GMT", 3) == 0 || strncmp((const char *) bp, "UTC", 3) == 0) {
                    tm->tm.tm_isdst = 0; tm->gmtoff = 0; bp += 3;
                }
                else if (strncasecmp((const char *) bp, "UTC", 3) == 0) {
                    /* the last resort (:630-650): the names of the PROCESS's zone, tzname[], without case -- restated for a process
                     * whose zone is UTC (TZ unset, what a container has: tzname = {"UTC", "UTC"}, timezone = 0) */
                    tm->tm.tm_isdst = 0; tm->gmtoff = 0; bp += 3;
                }
                else return NULL;
            }
            continue;
        }
        case 'z':
            while (isspace(*bp)) bp++;
            neg = 0;
            switch (*bp++) {
            case 'G':
                if (*bp++ != 'M') return NULL;
                if (*bp++ != 'T') return NULL;
                tm->tm.tm_isdst = 0; tm->gmtoff = 0;
                continue;
            case 'U':
                if (*bp++ != 'T') return NULL;
                if (*bp == 'C') bp++;
                tm->tm.tm_isdst = 0; tm->gmtoff = 0;
                continue;
            case 'Z':
                tm->tm.tm_isdst = 0; tm->gmtoff = 0;
                continue;
            case '+': neg = 0; break;
            case '-': neg = 1; break;
            default:
                --bp;
                ep = find_string(bp, &i, nast, 4);
                if (ep != NULL) { tm->gmtoff = (-5 - i) * 3600; tm->tm.tm_isdst = 0; bp = ep; continue; }
                ep = find_string(bp, &i, nadt, 4);
                if (ep != NULL) { tm->tm.tm_isdst = 1; tm->gmtoff = (-4 - i) * 3600; bp = ep; continue; }
                return NULL;
            }
            if (!isdigit(bp[0]) || !isdigit(bp[1])) return NULL;
            offs = ((bp[0] - '0') * 10 + (bp[1] - '0')) * 3600;
            bp += 2;
            if (*bp == ':') bp++;
            if (isdigit(*bp)) {
                offs += (*bp++ - '0') * 10 * 60;
                if (!isdigit(*bp)) return NULL;
                offs += (*bp++ - '0') * 60;
            }
            if (neg) offs = -offs;
            tm->tm.tm_isdst = 0;
            tm->gmtoff = offs;
            continue;
        case 'n': case 't':
            while (isspace(*bp)) bp++;
            break;
        default:
            return NULL;
        }
    }

    if (st->relyear != -1) {
        if (st->century == TM_YEAR_BASE) {
            if (st->relyear <= 68) tm->tm.tm_year = st->relyear + 2000 - TM_YEAR_BASE;
            else tm->tm.tm_year = st->relyear + 1900 - TM_YEAR_BASE;
        }
        else tm->tm.tm_year = st->relyear + st->century - TM_YEAR_BASE;
        st->fields |= F_YEAR;
    }
    if (st->fields & F_YEAR) {
        const int year = (unsigned int) tm->tm.tm_year + (unsigned int) TM_YEAR_BASE;
        const int *mon_lens = mon_lengths[isleap(year)];
        if (!(st->fields & F_YDAY) && (st->fields & F_MON) && (st->fields & F_MDAY)) {
            tm->tm.tm_yday = tm->tm.tm_mday - 1;
            for (i = 0; i < tm->tm.tm_mon; i++) tm->tm.tm_yday += mon_lens[i];
            st->fields |= F_YDAY;
        }
        if (st->fields & F_YDAY) {
            int days = tm->tm.tm_yday;
            if (!(st->fields & F_WDAY)) {
                tm->tm.tm_wday = 4 + ((year - 1970) % 7) * (365 % 7) + leaps_thru_end_of(year - 1)
                                 - leaps_thru_end_of(1970 - 1) + tm->tm.tm_yday;
                tm->tm.tm_wday %= 7;
                if (tm->tm.tm_wday < 0) tm->tm.tm_wday += 7;
            }
            if (!(st->fields & F_MON)) {
                tm->tm.tm_mon = 0;
                while (tm->tm.tm_mon < 12 && days >= mon_lens[tm->tm.tm_mon]) days -= mon_lens[tm->tm.tm_mon++];
            }
            if (!(st->fields & F_MDAY)) tm->tm.tm_mday = days + 1;
        }
    }
    return (const char *) bp;
}

/* NOTE: the reference's statics (century/relyear/fields) persist across the two flb_strptime
 * calls that bracket %L (src/flb_parser.c:2016,2030) only until re-initialised: every
 * flb_strptime() entry passes initialize=1, so each call starts clean. */
const char *o_strptime(const char *buf, const char *fmt, struct otm *tm)
{
    struct st st;
    return sp(buf, fmt, tm, &st, 1);
}
