/*
 * oracle/otime.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of src/flb_strptime.c and the
 * time lookup of src/flb_parser.c:1876-2065 (+ include/fluent-bit/flb_parser.h:80-94).
 */
#ifndef ORACLE_OTIME_H
#define ORACLE_OTIME_H
#include <time.h>
#include <stdint.h>

struct otm {
    struct tm tm;
    long gmtoff;          /* flb_tm_gmtoff() */
};

/* flb_strptime(): returns pointer past the consumed input or NULL */
const char *o_strptime(const char *buf, const char *fmt, struct otm *tm);

#endif
