/*
 * oracle/orx.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of the Onigmo semantics used by
 * the reference hot path (see orx.c header).  Never linked into the product.
 */
#ifndef ORACLE_ORX_H
#define ORACLE_ORX_H

#define ORX_MAX_GROUPS 64

/* same bit values as lib/onigmo/onigmo.h ONIG_OPTION_IGNORECASE/EXTEND/MULTILINE */
#define ORX_OPT_IGNORECASE 1u
#define ORX_OPT_EXTEND     2u
#define ORX_OPT_MULTILINE  4u

typedef struct orx orx_t;

orx_t *orx_compile(const char *pat, int len, unsigned options, char *err, int errlen);
void orx_free(orx_t *rx);
/* returns number of registers (groups + 1) and fills beg/end, or -1 on mismatch */
int orx_search(const orx_t *rx, const char *s, int len, int *beg, int *end, int max);
int orx_match(const orx_t *rx, const char *s, int len);
int orx_num_groups(const orx_t *rx);
int orx_num_names(const orx_t *rx);
const char *orx_name(const orx_t *rx, int i);
int orx_name_ngroups(const orx_t *rx, int i);
int orx_name_group(const orx_t *rx, int i, int k);

#endif
