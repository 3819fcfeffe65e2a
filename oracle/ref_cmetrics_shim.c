/*
 * oracle/ref_cmetrics_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL cmetrics of the reference (lib/cmetrics/src + lib/cfl/src, compiled in place by
 * oracle/Makefile into _ref/libcmetrics_ref.so) the way filter_log_to_metrics does
 * (plugins/filter_log_to_metrics/log_to_metrics.c:806-843 create, :1107-1119 update): one counter, gauge or
 * histogram with label keys, updated per record with the label values, so that the arithmetic of the
 * oracle's restatement (oracle/oflb.c oflb_l2m_*) can be pinned on the real library: which series exist,
 * in which order, and their value / cumulative... bucket counts, count and sum bit for bit.
 */
#include <stdlib.h>
#include <string.h>
#include <cmetrics/cmetrics.h>
#include <cmetrics/cmt_counter.h>
#include <cmetrics/cmt_gauge.h>
#include <cmetrics/cmt_histogram.h>
#include <cmetrics/cmt_map.h>
#include <cmetrics/cmt_metric.h>

struct refcmt {
    int mode;                      /* 0 counter, 1 gauge, 2 histogram */
    struct cmt *cmt;
    struct cmt_counter *c;
    struct cmt_gauge *g;
    struct cmt_histogram *h;
    struct cmt_histogram_buckets *b;
    int nbuckets;
};

/* nbuckets < 0: the default buckets (cmt_histogram_buckets_default_create) */
void *refcmt_new(int mode, int nlabels, char **label_keys, int nbuckets, double *bounds)
{
    struct refcmt *r = calloc(1, sizeof(*r));
    r->mode = mode;
    r->cmt = cmt_create();
    if (mode == 0) r->c = cmt_counter_create(r->cmt, "log_metric", "counter", "m", "help", nlabels, label_keys);
    else if (mode == 1) r->g = cmt_gauge_create(r->cmt, "log_metric", "gauge", "m", "help", nlabels, label_keys);
    else {
        r->b = nbuckets < 0 ? cmt_histogram_buckets_default_create() : cmt_histogram_buckets_create_size(bounds, (size_t) nbuckets);
        r->nbuckets = (int) r->b->count;
        r->h = cmt_histogram_create(r->cmt, "log_metric", "histogram", "m", "help", r->b, nlabels, label_keys);
    }
    return r;
}

int refcmt_update(void *h, uint64_t ts, double val, int nlabels, char **label_vals)
{
    struct refcmt *r = h;
    if (r->mode == 0) return cmt_counter_inc(r->c, ts, nlabels, label_vals);
    if (r->mode == 1) return cmt_gauge_set(r->g, ts, val, nlabels, label_vals);
    return cmt_histogram_observe(r->h, ts, val, nlabels, label_vals);
}

/* n observations in record order, ONE label each: label_idx[i] picks its value from the table (bench.py's measured
 * ULP distance of the device's once-rounded histogram sums to the real cmetrics' sequential f64 sums) */
int refcmt_update_many(void *h, uint64_t n, const double *vals, const int32_t *label_idx, char **label_table)
{
    uint64_t i;
    int ret = 0;
    for (i = 0; i < n && ret == 0; i++) {
        char *lv[1] = { label_table[label_idx[i]] };
        ret = refcmt_update(h, (uint64_t) (i + 1), vals[i], 1, lv);
    }
    return ret;
}

static struct cmt_map *map_of(struct refcmt *r) { return r->mode == 0 ? r->c->map : r->mode == 1 ? r->g->map : r->h->map; }

int refcmt_nbuckets(void *h) { return ((struct refcmt *) h)->nbuckets; }
double refcmt_bound(void *h, int i) { return ((struct refcmt *) h)->b->upper_bounds[i]; }

/* series in list order (a map without label keys has the one static metric) */
int refcmt_nseries(void *h)
{
    struct cmt_map *m = map_of(h);
    if (m->label_count == 0) return m->metric_static_set ? 1 : 0;
    return cfl_list_size(&m->metrics);
}

static struct cmt_metric *nth(struct cmt_map *m, int idx)
{
    struct cfl_list *head;
    int i = 0;
    if (m->label_count == 0) return &m->metric;
    cfl_list_foreach(head, &m->metrics) {
        if (i++ == idx) return cfl_list_entry(head, struct cmt_metric, _head);
    }
    return NULL;
}

/* label value `li` of series `idx` (NULL when absent) */
const char *refcmt_label(void *h, int idx, int li)
{
    struct cmt_metric *mt = nth(map_of(h), idx);
    struct cfl_list *head;
    int i = 0;
    cfl_list_foreach(head, &mt->labels) {
        if (i++ == li) return cfl_list_entry(head, struct cmt_map_label, _head)->name;
    }
    return NULL;
}

double refcmt_value(void *h, int idx) { return cmt_metric_get_value(nth(map_of(h), idx)); }
uint64_t refcmt_bucket(void *h, int idx, int b) { return cmt_metric_hist_get_value(nth(map_of(h), idx), b); }
uint64_t refcmt_count(void *h, int idx) { return cmt_metric_hist_get_count_value(nth(map_of(h), idx)); }
double refcmt_sum(void *h, int idx) { return cmt_metric_hist_get_sum_value(nth(map_of(h), idx)); }

void refcmt_free(void *h)
{
    struct refcmt *r = h;
    cmt_destroy(r->cmt);
    free(r);
}
