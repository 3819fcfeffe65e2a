// l2m.cpp -- host side of filter_log_to_metrics (include/flb_gpu.h "filter_log_to_metrics").
//
//   flbgpu_filter_l2m_create  ~ cb_log_to_metrics_init: set_rules / set_labels / set_buckets
//                               (plugins/filter_log_to_metrics/log_to_metrics.c:216-312,355-497,540-595,655-968)
//   run_l2m_dev               ~ cb_log_to_metrics_filter (:970-1156), as kernel launches
//   flbgpu_l2m_snapshot       ~ what the cmt context holds after the callback (cmt_counter / cmt_gauge /
//                               cmt_histogram series, lib/cmetrics/src/cmt_map.c:377-452)
//   flbgpu_l2m_export / _finalize_row   the mergeable integer state behind the snapshot, for the
//                               all-reduce across GPUs
// The hidden emitter input, the flush timer and the cmetrics msgpack encoding are engine plumbing
// and stay with the engine (out of scope).
#include "host_int.hpp"
#include "numconv.hpp"
#include "l2m_lane.hpp"
#include "seqsum.hpp"

#include <algorithm>
#include <string>
#include <unordered_map>

using namespace flbgpu;

struct L2mMisc { unsigned long long first_bad; unsigned long long counts[3]; };

void l2m_state_destroy(L2mState *s) {
    if (!s) return;
    DevBuf *all[] = {&s->d_labels, &s->d_value_key, &s->d_bounds, &s->d_slot_hash, &s->d_slot_sid, &s->d_arena, &s->d_key_off,
                     &s->d_key_len, &s->d_series_hash, &s->d_rows, &s->d_ctr, &s->d_sid, &s->d_val, &s->d_tmp, &s->d_misc, &s->d_seq, &s->d_log_sid, &s->d_log_val, &s->d_nobad, &s->d_seqwork};
    for (auto *b : all) b->release();
    delete s;
}

// first parser entry of flb_ra_create(str) (src/flb_record_accessor.c:74-232): text before the
// first '$' -- or the whole string -- is a STRING part whose name is looked up as a top-level key
// (src/record_accessor/flb_ra_parser.c:224-249); "$TAG" / "$0" entries have no key.
// returns 1 ok, 0 ok-but-keyless (lookups fail at run time), -1 invalid
static int parse_first_part(const char *str, DevKey &k, std::string &why) {
    memset(&k, 0, sizeof(k));
    if (str[0] != '$') {
        const char *d = strchr(str, '$');
        size_t n = d ? (size_t) (d - str) : strlen(str);
        if (n == 0) { why = "empty accessor"; return -1; }
        if (n >= (size_t) MAX_KEY) { why = "key too long"; return -1; }
        memcpy(k.key, str, n);
        k.key_len = (int) n;
        return 1;
    }
    if (!str[1]) { why = "empty accessor"; return -1; }
    if ((str[1] >= '0' && str[1] <= '9') || !strncmp(str + 1, "TAG", 3)) { k.key_len = -1; return 0; }
    int quote = 0;
    size_t end;
    for (end = 1; str[end]; end++) {                       // :175-186
        if (str[end] == '\'') quote++;
        else if (str[end] == '.' && (quote & 1)) continue;
        else if (str[end] == '.' || str[end] == ' ' || str[end] == ',' || str[end] == '"') break;
    }
    std::string seg(str, end);
    if (!parse_ra(seg.c_str(), k, why)) return -1;
    return 1;
}

static bool zero_alloc(DevBuf &b, size_t bytes) {
    if (!b.ensure(bytes)) return false;
    HIPOK(hipMemset(b.p, 0, b.cap));
    return true;
}

// grows `b` to new_bytes keeping the first `keep` bytes, zero-filling the rest
static bool grow_keep(DevBuf &b, size_t new_bytes, size_t keep) {
    DevBuf nb;
    if (!zero_alloc(nb, new_bytes)) return false;
    if (keep && b.p) HIPOK(hipMemcpy(nb.p, b.p, keep, hipMemcpyDeviceToDevice));
    b.release();
    b = nb;
    return true;
}

L2mTable l2m_table_of(L2mState *s) {
    L2mTable t;
    L2mCtr *c = s->d_ctr.as<L2mCtr>();
    t.slot_hash = s->d_slot_hash.as<unsigned long long>();
    t.slot_sid = s->d_slot_sid.as<uint32_t>();
    t.cap_mask = s->cap - 1;
    t.max_series = s->max_series;
    t.n_series = &c->n_series;
    t.arena = s->d_arena.as<uint8_t>();
    t.arena_used = &c->arena_used;
    t.arena_cap = s->arena_cap;
    t.key_off = s->d_key_off.as<unsigned long long>();
    t.key_len = s->d_key_len.as<uint32_t>();
    t.series_hash = s->d_series_hash.as<unsigned long long>();
    t.overflow = &c->overflow;
    return t;
}

bool l2m_table_init(L2mState *s) {
    uint64_t cap = 1u << 16;
    if (const char *e = getenv("FLBGPU_L2M_INIT_CAP")) {
        uint64_t v = strtoull(e, nullptr, 10);
        cap = 16;
        while (cap < v) cap <<= 1;
    }
    s->cap = cap;
    s->max_series = (uint32_t) (cap / 2);
    s->arena_cap = std::max<uint64_t>(4096, cap * 32);
    if (const char *e = getenv("FLBGPU_L2M_INIT_ARENA")) s->arena_cap = std::max<uint64_t>(64, strtoull(e, nullptr, 10) & ~7ull);
    return zero_alloc(s->d_slot_hash, cap * 8) && zero_alloc(s->d_slot_sid, cap * 4) && zero_alloc(s->d_arena, s->arena_cap + 8) &&
           zero_alloc(s->d_key_off, (size_t) s->max_series * 8) && zero_alloc(s->d_key_len, (size_t) s->max_series * 4) &&
           zero_alloc(s->d_series_hash, (size_t) s->max_series * 8) &&
           zero_alloc(s->d_rows, (size_t) s->max_series * s->W * 8) && zero_alloc(s->d_ctr, sizeof(L2mCtr));
}

// doubles whatever ran out (series capacity and/or arena) and rebuilds the slot array
bool l2m_table_grow(L2mState *s, hipStream_t st) {
    L2mCtr c;
    HIPOK(hipMemcpy(&c, s->d_ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    const uint32_t ns = c.n_series;
    bool series_full = (uint64_t) ns * 2 >= s->max_series;
    bool arena_full = c.arena_used * 2 >= s->arena_cap;
    if (!series_full && !arena_full) series_full = arena_full = true;       // long probe run or one very long key
    if (series_full) {
        const uint32_t old_max = s->max_series;
        s->cap *= 2;
        s->max_series = (uint32_t) (s->cap / 2);
        s->d_slot_hash.release(); s->d_slot_sid.release();
        if (!zero_alloc(s->d_slot_hash, s->cap * 8) || !zero_alloc(s->d_slot_sid, s->cap * 4)) return false;
        if (!grow_keep(s->d_key_off, (size_t) s->max_series * 8, (size_t) old_max * 8) ||
            !grow_keep(s->d_key_len, (size_t) s->max_series * 4, (size_t) old_max * 4) ||
            !grow_keep(s->d_series_hash, (size_t) s->max_series * 8, (size_t) old_max * 8) ||
            !grow_keep(s->d_rows, (size_t) s->max_series * s->W * 8, (size_t) old_max * s->W * 8)) return false;
        L2mTable t = l2m_table_of(s);
        launch_l2m_rehash(t, ns, st);
        HIPOK(hipStreamSynchronize(st));
    }
    if (arena_full) {
        const uint64_t old = s->arena_cap;
        s->arena_cap *= 2;
        if (!grow_keep(s->d_arena, s->arena_cap + 8, old)) return false;
    }
    s->grows++;
    return true;
}

extern "C" flbgpu_filter *flbgpu_filter_l2m_create(const char *metric_mode, int nprops, const char *const *keys,
                                                   const char *const *values, int kubernetes_mode, const char *value_field,
                                                   int discard_logs) {
    static const char *k8s[5] = {"namespace_name", "pod_name", "container_name", "docker_id", "pod_id"};   // :43-50
    auto *f = new flbgpu_filter();
    f->kind = F_L2M;
    auto *s = new L2mState();
    f->l2m = s;
    s->discard_logs = discard_logs;
    auto fail = [&](const char *fmt, const std::string &a) { set_err(fmt, a.c_str()); delete f; return (flbgpu_filter *) nullptr; };
    // :731-757
    if (!metric_mode || !strcasecmp(metric_mode, "counter")) s->mode = L2M_COUNTER;
    else if (!strcasecmp(metric_mode, "gauge")) s->mode = L2M_GAUGE;
    else if (!strcasecmp(metric_mode, "histogram")) s->mode = L2M_HISTOGRAM;
    else return fail("log_to_metrics: invalid 'mode' value '%s'. Only 'counter', 'gauge' or 'histogram' types are allowed", metric_mode);
    std::string why;
    // set_rules :216-312 -- the field goes to flb_ra_create VERBATIM (no '$' is prepended)
    for (int i = 0; i < nprops; i++) {
        GrepRule r;
        memset(&r, 0, sizeof(r));
        if (!strcasecmp(keys[i], "regex")) r.type = GREP_REGEX;
        else if (!strcasecmp(keys[i], "exclude")) r.type = GREP_EXCLUDE;
        else continue;
        const char *v = values[i];
        while (*v == ' ') v++;
        const char *sp = strchr(v, ' ');
        if (!sp || sp == v || !sp[1]) return fail("log_to_metrics: invalid regex, expected field and regular expression%s", "");
        std::string field(v, sp - v);
        DevKey k;
        int st = parse_first_part(field.c_str(), k, why);
        if (st < 0) return fail("log_to_metrics: invalid record accessor? %s", "'" + field + "': " + why);
        // compile_rule parses an accessor itself: hand it a placeholder and patch the key in
        bool nonregular = false;
        rx::BtProgram *bt = nullptr;
        if (!compile_rule("$k", sp + 1, r, f->rule_blobs, why, &nonregular)) {
            // not a regular expression: a host rule (flbgpu.cpp "host rules") -- the whole rule list then runs as a hidden filter_grep
            // in front of the metric kernels (plugins/filter_log_to_metrics/log_to_metrics.c:314-353 grep_filter_data is grep's legacy loop)
            std::string e2;
            if (nonregular && !getenv("FLBGPU_NO_HOST_RULES")) {
                const char *ps, *pe;
                unsigned opts;
                rx::split_flb_pattern(sp + 1, &ps, &pe, &opts);
                bt = rx::bt_compile(ps, (size_t) (pe - ps), opts, e2);
            }
            if (!bt) return fail("log_to_metrics: %s", why + (e2.empty() ? "" : "; on the host: " + e2));
            memset(&r.dfa, 0, sizeof(r.dfa)); memset(&r.utf8, 0, sizeof(r.utf8));
            f->has_host_rules = true;
        }
        r.key = k;
        if ((int) f->rules.size() >= MAX_RULES) { if (bt) rx::bt_free(bt); return fail("log_to_metrics: too many rules%s", ""); }
        f->rules.push_back(r);
        f->host_rx.push_back(bt);
    }
    if (f->has_host_rules) {
        auto *g = new flbgpu_filter();
        g->kind = F_GREP;
        g->logical_op = OP_LEGACY;
        g->rules.swap(f->rules);
        g->host_rx.swap(f->host_rx);
        g->rule_blobs.swap(f->rule_blobs);
        g->has_host_rules = true;
        f->has_host_rules = false;
        f->l2m_gate = g;
        if (!filter_common_init(g) || !g->d_rules.ensure(g->rules.size() * sizeof(GrepRule)) ||
            hipMemcpy(g->d_rules.p, g->rules.data(), g->rules.size() * sizeof(GrepRule), hipMemcpyHostToDevice) != hipSuccess) {
            if (!*flbgpu_last_error()) set_err("log_to_metrics: device setup failed");
            delete f;
            return nullptr;
        }
    }
    // set_labels :355-497
    auto add_label = [&](const std::string &name, const char *accessor) {
        DevKey k;
        std::string w;
        if (parse_first_part(accessor, k, w) < 0) { memset(&k, 0, sizeof(k)); k.key_len = -1; }   // flb_warn + NULL accessor: label stays empty
        s->label_keys.push_back(name);
        s->labels.push_back(k);
    };
    if (kubernetes_mode)
        for (int i = 0; i < 5; i++) add_label(k8s[i], (std::string("$kubernetes['") + k8s[i] + "']").c_str());
    for (int i = 0; i < nprops; i++) {
        if (!strcasecmp(keys[i], "label_field")) add_label(values[i], values[i]);
        else if (!strcasecmp(keys[i], "add_label")) {
            const char *v = values[i];
            while (*v == ' ') v++;
            const char *sp = strchr(v, ' ');
            if (!sp || sp == v || !sp[1]) return fail("log_to_metrics: invalid label, expected name and key%s", "");
            add_label(std::string(v, sp - v), sp + 1);
        }
    }
    if ((int) s->labels.size() > L2M_MAX_LABELS) return fail("log_to_metrics: too many labels%s", "");
    // value_field :794-808
    memset(&s->value_key, 0, sizeof(s->value_key));
    if (s->mode != L2M_COUNTER) {
        if (!value_field || !*value_field) return fail("log_to_metrics: value_field is not set%s", "");
        if (parse_first_part(value_field, s->value_key, why) < 0) return fail("log_to_metrics: invalid record accessor key for value_field: %s", why);
    }
    // set_buckets :540-595 + defaults :811-822
    if (s->mode == L2M_HISTOGRAM) {
        for (int i = 0; i < nprops; i++) {
            if (strcasecmp(keys[i], "bucket")) continue;
            char *end;
            double d = strtod(values[i], &end);
            if (end == values[i]) return fail("log_to_metrics: Error during conversion of bucket '%s'", values[i]);
            s->bounds.push_back(d);
        }
        if (s->bounds.empty()) s->bounds = {0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0};
        else {
            // sort_doubles_ascending: a bubble sort on '>' (NaN bounds stay where the comparisons leave them)
            for (size_t i = 0; i + 1 < s->bounds.size(); i++)
                for (size_t j = 0; j + 1 < s->bounds.size() - i; j++)
                    if (s->bounds[j] > s->bounds[j + 1]) std::swap(s->bounds[j], s->bounds[j + 1]);
        }
        for (double b : s->bounds) if (b != b) return fail("log_to_metrics: NaN bucket bounds are not on the GPU path%s", "");
    }
    s->nb = (int) s->bounds.size();
    s->W = l2m_row_words(s->mode, s->nb);
    if (!filter_common_init(f)) { delete f; return nullptr; }
    bool ok = f->d_rules.ensure(std::max<size_t>(1, f->rules.size()) * sizeof(GrepRule)) &&
              s->d_labels.ensure(std::max<size_t>(1, s->labels.size()) * sizeof(DevKey)) && s->d_value_key.ensure(sizeof(DevKey)) &&
              s->d_bounds.ensure(std::max<size_t>(1, s->bounds.size()) * sizeof(double)) && l2m_table_init(s);
    if (ok && !f->rules.empty()) ok = hipMemcpy(f->d_rules.p, f->rules.data(), f->rules.size() * sizeof(GrepRule), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !s->labels.empty()) ok = hipMemcpy(s->d_labels.p, s->labels.data(), s->labels.size() * sizeof(DevKey), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = hipMemcpy(s->d_value_key.p, &s->value_key, sizeof(DevKey), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && s->nb) ok = hipMemcpy(s->d_bounds.p, s->bounds.data(), s->nb * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { if (!*flbgpu_last_error()) set_err("log_to_metrics: device setup failed"); delete f; return nullptr; }
    return f;
}

bool run_l2m_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, hipStream_t st, int *ret) {
    L2mState *s = f->l2m;
    const uint64_t n = in->n;
    *ret = s->discard_logs ? FLBGPU_FILTER_MODIFIED : FLBGPU_FILTER_NOTOUCH;
    f->last_in = n; f->last_out = s->discard_logs ? 0 : n;
    if (n == 0) return true;
    flbgpu_dev_chunk kept;
    if (f->l2m_gate) {
        // the rules in front (a hidden filter_grep with host rules): the metric kernels take what it keeps
        if (!l2m_gate_dev(f->l2m_gate, in, &kept, st)) return false;
        if (kept.bytes == 0) return true;                   // nothing passes the rules
        in = &kept;
    }
    if (!s->d_sid.ensure(n * 4) || !s->d_misc.ensure(sizeof(L2mMisc)) || !s->d_tmp.ensure(l2m_stale_tmp_elems(n) * 8)) return false;
    if (s->mode != L2M_COUNTER && !s->d_val.ensure(n * 8)) return false;
    L2mMisc hm;
    L2mCtr hc;
    const int cus = device_cus() > 0 ? device_cus() : 256;
    for (int attempt = 0;; attempt++) {
        if (attempt > 40) { set_err("log_to_metrics: series dictionary keeps overflowing"); return false; }
        memset(&hm, 0, sizeof(hm));
        hm.first_bad = ~0ull;
        HIPOK(hipMemcpyAsync(s->d_misc.p, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
        HIPOK(hipMemsetAsync(&s->d_ctr.as<L2mCtr>()->overflow, 0, sizeof(unsigned int), st));
        L2mArgs a;
        a.data = (const uint8_t *) in->data; a.row_off = in->row_off; a.n = n; a.bytes = in->bytes;
        a.rules = f->d_rules.as<GrepRule>(); a.nrules = (int) f->rules.size();
        a.labels = s->d_labels.as<DevKey>(); a.nlabels = (int) s->labels.size();
        a.value_key = s->d_value_key.as<DevKey>(); a.mode = s->mode;
        a.t = l2m_table_of(s);
        a.sid_col = s->d_sid.as<uint32_t>(); a.val_col = s->d_val.as<uint64_t>();
        a.first_bad = &s->d_misc.as<L2mMisc>()->first_bad; a.counts = s->d_misc.as<L2mMisc>()->counts;
        // the extraction with ONE lean walk per record on LDS pointers (l2mlane_kernels.inc) when the filter has no rules and its value
        // field and labels are top-level names; k_l2m_extract (a generic walk per name) otherwise
        static const bool lane_off = getenv("FLBGPU_L2M_LANE") && atoi(getenv("FLBGPU_L2M_LANE")) == 0;
        L2mLaneArgs la;
        bool lane_ok = !lane_off && a.nrules == 0 && ((uintptr_t) in->data & 15) == 0 && s->labels.size() <= 4;                 // (L2M_LV of l2m_dev.inc)
        if (lane_ok) {
            memset(&la, 0, sizeof(la));
            auto plain = [](const DevKey &k) { return k.key_len >= 1 && k.key_len <= 32 && k.nsub == 0; };
            for (size_t i = 0; lane_ok && i < s->labels.size(); i++) {
                if (!plain(s->labels[i])) { lane_ok = false; break; }
                la.slot_klen[i] = (uint8_t) s->labels[i].key_len;
                memcpy(la.slot_kw[i], s->labels[i].key, (size_t) s->labels[i].key_len);
            }
            la.nslots = (int) s->labels.size();
            la.value_slot = -1;
            if (lane_ok && s->mode != L2M_COUNTER) {
                if (!plain(s->value_key)) lane_ok = false;
                else {
                    la.value_slot = la.nslots++;
                    la.slot_klen[la.value_slot] = (uint8_t) s->value_key.key_len;
                    memcpy(la.slot_kw[la.value_slot], s->value_key.key, (size_t) s->value_key.key_len);
                }
            }
        }
        if (lane_ok) {
            const uint64_t avg = in->bytes / n + 1;
            uint64_t cap = (64 * avg * 5 / 4 + 512 + 15) & ~15ull, R = 64;
            if (cap > (uint64_t) l2m_lane_text_max()) { R = (uint64_t) (l2m_lane_text_max() - 16) * 92 / 100 / avg; cap = (uint64_t) l2m_lane_text_max(); }
            if (cap < 2048) cap = 2048;
            if (R < 1) R = 1;
            if (R > 64) R = 64;
            la.a = a; la.text_cap = (uint32_t) cap; la.rows_per_tile = (uint32_t) R;
            ProfScope ps(f, st, "k_l2m_lane");
            launch_l2m_lane(la, cus, st);
        }
        else { ProfScope ps(f, st, "k_l2m_extract"); launch_l2m_extract(a, cus, st); }
        HIPOK(hipMemcpyAsync(&hm, s->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&hc, s->d_ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (!hc.overflow && hm.counts[1] > 0) {
            { ProfScope ps(f, st, "k_l2m_generic"); launch_l2m_generic(a, st); }
            HIPOK(hipMemcpyAsync(&hm, s->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipMemcpyAsync(&hc, s->d_ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
        }
        if (!hc.overflow) break;
        if (!l2m_table_grow(s, st)) return false;          // nothing was aggregated yet: the pass simply runs again
    }
    s->last_obs = hm.counts[0]; s->last_deferred = hm.counts[1]; s->last_stale = hm.counts[2];
    if (hm.counts[2] > 0) {
        ProfScope ps(f, st, "k_l2m_stale");
        launch_l2m_stale(s->d_sid.as<uint32_t>(), s->d_val.as<uint64_t>(), n, &s->d_misc.as<L2mMisc>()->first_bad, s->d_tmp.as<uint64_t>(), st);
    }
    L2mAggArgs g;
    g.sid_col = s->d_sid.as<uint32_t>(); g.val_col = s->d_val.as<uint64_t>(); g.n = n;
    g.first_bad = &s->d_misc.as<L2mMisc>()->first_bad;
    g.rows = s->d_rows.as<unsigned long long>(); g.W = s->W; g.mode = s->mode; g.nb = s->nb;
    g.bounds = s->d_bounds.as<double>(); g.idx_base = s->idx_base; g.n_series = &s->d_ctr.as<L2mCtr>()->n_series;
    { ProfScope ps(f, st, "k_l2m_aggregate"); launch_l2m_aggregate(g, cus, st); }
    if (s->sum_order_ref == 2 && s->mode == L2M_HISTOGRAM) {
        // the interval's observations, in record order, for the flush-time chain (what a decode error cuts off is not among them)
        const uint64_t lim = hm.first_bad < n ? hm.first_bad : n;
        if (s->log_n + lim > s->log_cap) {
            uint64_t nc = s->log_cap ? s->log_cap : (1u << 20);
            while (nc < s->log_n + lim) nc *= 2;
            if (!grow_keep(s->d_log_sid, (size_t) nc * 4, (size_t) s->log_n * 4) || !grow_keep(s->d_log_val, (size_t) nc * 8, (size_t) s->log_n * 8)) return false;
            s->log_cap = nc;
        }
        if (lim) {
            HIPOK(hipMemcpyAsync(s->d_log_sid.as<uint32_t>() + s->log_n, g.sid_col, (size_t) lim * 4, hipMemcpyDeviceToDevice, st));
            HIPOK(hipMemcpyAsync(s->d_log_val.as<uint64_t>() + s->log_n, g.val_col, (size_t) lim * 8, hipMemcpyDeviceToDevice, st));
            s->log_n += lim;
        }
    }
    else if (s->sum_order_ref && s->mode == L2M_HISTOGRAM && hc.n_series) {
        // the reference's own sum next to the exact one (host_int.hpp L2mState::sum_order_ref)
        if (hc.n_series > s->seq_cap) {
            uint32_t nc = s->seq_cap ? s->seq_cap : 64;
            while (nc < hc.n_series) nc *= 2;
            if (!grow_keep(s->d_seq, (size_t) nc * sizeof(double), (size_t) s->seq_cap * sizeof(double))) return false;
            s->seq_cap = nc;
        }
        // (round 6: the observations sorted by series, then a lane / a wave per series -- kernels_seqsum.hip; round 4's kernel read the
        // whole call once per series: 345 ms per 10 M observations on ten series)
        const uint64_t lim = hm.first_bad < n ? hm.first_bad : n;
        const size_t wb = seqsum_work_bytes(lim, hc.n_series);
        if (!s->d_seqwork.ensure(wb)) return false;
        ProfScope ps(f, st, "k_l2m_seqsum");
        if (!launch_seqsum_sorted(g.sid_col, g.val_col, lim, s->d_seq.as<double>(), hc.n_series, s->d_seqwork.p, s->d_seqwork.cap, st)) { set_err("log_to_metrics: the reference-order sum failed to launch"); return false; }
    }
    HIPOK(hipStreamSynchronize(st));
    s->idx_base += n;
    return true;
}

// sum_order: 0 = the exact sum rounded once, 1 = also the reference's sequential sum (flbgpu_l2m_seq_sums), 2 = the sequential sum
// across ranks (the chain below: flbgpu_l2m_all_reduce / flbgpu_l2m_chain_*)
extern "C" int flbgpu_l2m_set_sum_order(flbgpu_filter *f, int reference) {
    if (!f || f->kind != F_L2M) return -1;
    f->l2m->sum_order_ref = reference == 2 ? 2 : reference ? 1 : 0;
    return 0;
}

// ------------------------------------------------------------------------------------------ the chain (sum_order 2)
// cmetrics adds every observation to a binary64 in the order the records come (lib/cmetrics/src/cmt_metric_histogram.c:124-137): no
// merge of per-rank sums gives those bits.  What gives them for ANY number of ranks is the fold itself, continued from rank to rank:
// the order is "the interval's records of rank 0, then of rank 1, ..." (the record index ranges flbgpu_l2m_set_index_base hands out),
// rank r starts its fold from the sums rank r - 1 ended on, the last rank's sums are the interval's.  A rank keeps its interval's
// observations (d_log_*: 12 bytes each) for that; the flush replays them with k_l2m_seqsum.
//   chain_begin  the sums the last flush ended on, for the union of the label tuples (0 for a new one)
//   seq_replay   this rank's turn: its local series start from `sums`, the interval's observations are added in order, `sums` gets the
//                result; the interval is closed
//   chain_end    every rank keeps the final sums
// flbgpu_l2m_all_reduce runs the three over RCCL (a broadcast per rank); the Python merge helper runs them over torch.distributed.
static bool l2m_local_keys(L2mState *s, std::vector<std::string> &keys) {
    L2mCtr c;
    if (hipMemcpy(&c, s->d_ctr.p, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) { set_err("log_to_metrics: device read failed"); return false; }
    const uint32_t ns = c.n_series;
    std::vector<unsigned long long> hoff(ns);
    std::vector<uint32_t> hlen(ns);
    std::vector<uint8_t> arena(c.arena_used);
    if (ns && (hipMemcpy(hoff.data(), s->d_key_off.p, (size_t) ns * 8, hipMemcpyDeviceToHost) != hipSuccess ||
               hipMemcpy(hlen.data(), s->d_key_len.p, (size_t) ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
               (!arena.empty() && hipMemcpy(arena.data(), s->d_arena.p, arena.size(), hipMemcpyDeviceToHost) != hipSuccess))) {
        set_err("log_to_metrics: device read failed");
        return false;
    }
    keys.resize(ns);
    for (uint32_t i = 0; i < ns; i++) keys[i].assign((const char *) arena.data() + hoff[i], hlen[i]);
    return true;
}
static std::vector<std::string> split_keys(uint64_t n, const uint64_t *key_off, const char *keys) {
    std::vector<std::string> out(n);
    for (uint64_t i = 0; i < n; i++) out[i].assign(keys + key_off[i], (size_t) (key_off[i + 1] - key_off[i]));
    return out;
}
static bool l2m_seq_replay(flbgpu_filter *f, const std::vector<std::string> &ukeys, std::vector<double> &G, hipStream_t st) {
    L2mState *s = f->l2m;
    std::vector<std::string> lkeys;
    if (!l2m_local_keys(s, lkeys)) return false;
    const uint32_t ns = (uint32_t) lkeys.size();
    if (ns == 0 || s->log_n == 0) { s->log_n = 0; return true; }
    std::unordered_map<std::string, uint32_t> index;
    for (size_t u = 0; u < ukeys.size(); u++) index.emplace(ukeys[u], (uint32_t) u);
    std::vector<double> hseq(ns, 0.0);
    std::vector<int64_t> uof(ns, -1);
    for (uint32_t i = 0; i < ns; i++) { auto it = index.find(lkeys[i]); if (it != index.end()) { uof[i] = it->second; hseq[i] = G[it->second]; } }
    if (ns > s->seq_cap) {
        uint32_t nc = s->seq_cap ? s->seq_cap : 64;
        while (nc < ns) nc *= 2;
        if (!s->d_seq.ensure((size_t) nc * sizeof(double))) return false;
        s->seq_cap = nc;
    }
    HIPOK(hipMemcpyAsync(s->d_seq.p, hseq.data(), (size_t) ns * sizeof(double), hipMemcpyHostToDevice, st));
    {
        const size_t wb = seqsum_work_bytes(s->log_n, ns);
        if (!s->d_seqwork.ensure(wb)) return false;
        ProfScope ps(f, st, "k_l2m_seqsum(chain)");
        if (!launch_seqsum_sorted(s->d_log_sid.as<uint32_t>(), s->d_log_val.as<uint64_t>(), s->log_n, s->d_seq.as<double>(), ns, s->d_seqwork.p, s->d_seqwork.cap, st)) {
            set_err("log_to_metrics: the reference-order sum failed to launch");
            return false;
        }
    }
    HIPOK(hipMemcpyAsync(hseq.data(), s->d_seq.p, (size_t) ns * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < ns; i++) if (uof[i] >= 0) G[(size_t) uof[i]] = hseq[i];
    s->log_n = 0;
    return true;
}
extern "C" int flbgpu_l2m_chain_begin(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, double *sums) {
    if (!f || f->kind != F_L2M || f->l2m->sum_order_ref != 2) { set_err("log_to_metrics: the chain needs sum_order 2"); return -1; }
    const std::vector<std::string> uk = split_keys(n_keys, key_off, keys);
    for (uint64_t u = 0; u < n_keys; u++) { auto it = f->l2m->chain_sums.find(uk[u]); sums[u] = it == f->l2m->chain_sums.end() ? 0.0 : it->second; }
    return 0;
}
extern "C" int flbgpu_l2m_seq_replay(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, double *sums) {
    if (!f || f->kind != F_L2M || f->l2m->sum_order_ref != 2 || f->l2m->mode != L2M_HISTOGRAM) { set_err("log_to_metrics: the chain needs a histogram with sum_order 2"); return -1; }
    const std::vector<std::string> uk = split_keys(n_keys, key_off, keys);
    std::vector<double> G(sums, sums + n_keys);
    if (!l2m_seq_replay(f, uk, G, f->stream)) return -1;
    for (uint64_t u = 0; u < n_keys; u++) sums[u] = G[u];
    return 0;
}
extern "C" int flbgpu_l2m_chain_end(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, const double *sums) {
    if (!f || f->kind != F_L2M || f->l2m->sum_order_ref != 2) { set_err("log_to_metrics: the chain needs sum_order 2"); return -1; }
    const std::vector<std::string> uk = split_keys(n_keys, key_off, keys);
    for (uint64_t u = 0; u < n_keys; u++) f->l2m->chain_sums[uk[u]] = sums[u];
    f->l2m->last_chain.assign(sums, sums + n_keys);
    return 0;
}
// the sums of the last flush (flbgpu_l2m_all_reduce with sum_order 2), in the order of its output
extern "C" int64_t flbgpu_l2m_chain_sums(flbgpu_filter *f, uint64_t max_series, double *sums) {
    if (!f || f->kind != F_L2M || f->l2m->sum_order_ref != 2) return -1;
    const std::vector<double> &c = f->l2m->last_chain;
    if (c.size() > max_series) return -(int64_t) c.size() - 2;
    for (size_t i = 0; i < c.size(); i++) sums[i] = c[i];
    return (int64_t) c.size();
}

// the sequential sums of the series flbgpu_l2m_export lists, in its order (first appearance); returns their number, -1 when the
// filter does not keep them
extern "C" int64_t flbgpu_l2m_seq_sums(flbgpu_filter *f, uint64_t max_series, double *sums) {
    if (!f || f->kind != F_L2M || f->l2m->sum_order_ref != 1 || f->l2m->mode != L2M_HISTOGRAM) return -1;
    L2mState *s = f->l2m;
    L2mCtr c;
    if (hipMemcpy(&c, s->d_ctr.p, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) { set_err("log_to_metrics: device read failed"); return -1; }
    const uint32_t ns = c.n_series;
    std::vector<uint64_t> hrows((size_t) ns * s->W);
    std::vector<double> hseq(ns, 0.0);
    if (ns) {
        if (hipMemcpy(hrows.data(), s->d_rows.p, hrows.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { set_err("log_to_metrics: device read failed"); return -1; }
        const uint32_t have = ns < s->seq_cap ? ns : s->seq_cap;
        if (have && hipMemcpy(hseq.data(), s->d_seq.p, (size_t) have * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { set_err("log_to_metrics: device read failed"); return -1; }
    }
    std::vector<uint32_t> live;
    for (uint32_t i = 0; i < ns; i++) if (hrows[(size_t) i * s->W + L2M_W_FIRST] != 0) live.push_back(i);
    std::sort(live.begin(), live.end(), [&](uint32_t a, uint32_t b) {
        return ~hrows[(size_t) a * s->W + L2M_W_FIRST] < ~hrows[(size_t) b * s->W + L2M_W_FIRST];
    });
    if (live.size() > max_series) return -(int64_t) live.size() - 2;
    for (size_t j = 0; j < live.size(); j++) sums[j] = hseq[live[j]];
    return (int64_t) live.size();
}

// ------------------------------------------------------------------------------------------ results
extern "C" int flbgpu_l2m_info(flbgpu_filter *f, int *mode, int *label_count, int *nbuckets, int *row_words) {
    if (!f || f->kind != F_L2M) return -1;
    L2mState *s = f->l2m;
    if (mode) *mode = s->mode;
    if (label_count) *label_count = (int) s->labels.size();
    if (nbuckets) *nbuckets = s->nb;
    if (row_words) *row_words = s->W;
    return 0;
}
extern "C" const char *flbgpu_l2m_label_key(flbgpu_filter *f, int i) {
    if (!f || f->kind != F_L2M || i < 0 || i >= (int) f->l2m->label_keys.size()) return nullptr;
    return f->l2m->label_keys[i].c_str();
}
extern "C" int flbgpu_l2m_bounds(flbgpu_filter *f, double *bounds) {
    if (!f || f->kind != F_L2M) return -1;
    for (int i = 0; i < f->l2m->nb; i++) bounds[i] = f->l2m->bounds[i];
    return f->l2m->nb;
}
extern "C" void flbgpu_l2m_set_index_base(flbgpu_filter *f, uint64_t base) { if (f && f->kind == F_L2M) f->l2m->idx_base = base; }
extern "C" void flbgpu_l2m_stats(flbgpu_filter *f, uint64_t *out5) {
    L2mState *s = f->l2m;
    out5[0] = s->last_obs; out5[1] = s->last_deferred; out5[2] = s->last_stale; out5[3] = s->grows; out5[4] = s->cap;
}

// mergeable state of the series that exist (first-appearance order): rows[n][row_words] and the
// label tuples as NUL-terminated strings back to back, key_off[n + 1] delimiting them.
// Returns the number of series, or -(needed count) - 1 ... see header.
extern "C" int64_t flbgpu_l2m_export(flbgpu_filter *f, uint64_t max_series, uint64_t *rows, uint64_t *key_off, char *keys,
                                     size_t keys_cap, size_t *keys_needed) {
    if (!f || f->kind != F_L2M) return -1;
    L2mState *s = f->l2m;
    L2mCtr c;
    if (hipMemcpy(&c, s->d_ctr.p, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) { set_err("log_to_metrics: device read failed"); return -1; }
    const uint32_t ns = c.n_series;
    std::vector<uint64_t> hrows((size_t) ns * s->W);
    std::vector<unsigned long long> hoff(ns);
    std::vector<uint32_t> hlen(ns);
    std::vector<uint8_t> arena(c.arena_used);
    bool ok = true;
    if (ns) {
        ok = hipMemcpy(hrows.data(), s->d_rows.p, hrows.size() * 8, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(hoff.data(), s->d_key_off.p, (size_t) ns * 8, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(hlen.data(), s->d_key_len.p, (size_t) ns * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             (arena.empty() || hipMemcpy(arena.data(), s->d_arena.p, arena.size(), hipMemcpyDeviceToHost) == hipSuccess);
    }
    if (!ok) { set_err("log_to_metrics: device read failed"); return -1; }
    // a series exists once a record touched it (dictionary entries created by rows behind a
    // decode error never were)
    std::vector<uint32_t> live;
    for (uint32_t i = 0; i < ns; i++) if (hrows[(size_t) i * s->W + L2M_W_FIRST] != 0) live.push_back(i);
    std::sort(live.begin(), live.end(), [&](uint32_t a, uint32_t b) {
        return ~hrows[(size_t) a * s->W + L2M_W_FIRST] < ~hrows[(size_t) b * s->W + L2M_W_FIRST];
    });
    size_t need = 0;
    for (uint32_t i : live) need += hlen[i];
    if (keys_needed) *keys_needed = need;
    if (live.size() > max_series || need > keys_cap) return -(int64_t) live.size() - 2;
    size_t ko = 0;
    for (size_t j = 0; j < live.size(); j++) {
        uint32_t i = live[j];
        memcpy(rows + j * s->W, &hrows[(size_t) i * s->W], (size_t) s->W * 8);
        key_off[j] = ko;
        memcpy(keys + ko, arena.data() + hoff[i], hlen[i]);
        ko += hlen[i];
    }
    key_off[live.size()] = ko;
    return (int64_t) live.size();
}

// the exact sum held as fixed-point digits -> the binary64 nearest to it (ties to even); NaN / infinities by their counts
uint64_t l2m_limbs_bits(const uint64_t *limbs, uint64_t n_nan, uint64_t n_pinf, uint64_t n_ninf) {
    uint64_t bits;
    if (n_nan || (n_pinf && n_ninf)) bits = nc::DBL_NAN_BITS;
    else if (n_pinf) bits = nc::DBL_INF_BITS;
    else if (n_ninf) bits = nc::DBL_INF_BITS | nc::DBL_SIGN;
    else {
        // carry-propagate the digits (they may be sums over GPUs), then sign-magnitude
        uint64_t d[L2M_NLIMB];
        long long carry = 0;
        for (int j = 0; j < L2M_NLIMB - 1; j++) {
            long long t = (long long) limbs[j] + carry;
            long long lo = t & 0xFFFFFFFFll;
            carry = (t - lo) >> 32;
            d[j] = (uint64_t) lo;
        }
        long long top = (long long) limbs[L2M_NLIMB - 1] + carry;
        bool neg = top < 0;
        uint64_t utop = (uint64_t) top;
        if (neg) {
            uint64_t c = 1;
            for (int j = 0; j < L2M_NLIMB - 1; j++) { uint64_t t = (~d[j] & 0xFFFFFFFFull) + c; d[j] = t & 0xFFFFFFFFull; c = t >> 32; }
            utop = ~utop + c;
        }
        d[L2M_NLIMB - 1] = utop;
        // highest set bit; digit j has weight 2^(32 j - 1074), the top digit is a full 64-bit word
        int hj = -1;
        for (int j = L2M_NLIMB - 1; j >= 0; j--) if (d[j]) { hj = j; break; }
        if (hj < 0) bits = 0;
        else {
            // gather the top 64 bits below (and including) the leading one + sticky
            const int lead = 63 - __builtin_clzll(d[hj]);                 // bit inside digit hj
            const int64_t top_pos = (int64_t) 32 * hj + lead;             // absolute bit index of the leading one
            uint64_t m = 0;
            bool sticky = false;
            for (int64_t k = 0; k < 64; k++) {
                int64_t pos = top_pos - k;
                uint64_t bit = 0;
                if (pos >= 0) {
                    int j = (int) (pos / 32);
                    if (j >= L2M_NLIMB - 1) bit = (d[L2M_NLIMB - 1] >> (pos - 32 * (L2M_NLIMB - 1))) & 1;
                    else bit = (d[j] >> (pos % 32)) & 1;
                }
                m = (m << 1) | bit;
            }
            const int64_t low_pos = top_pos - 63;                         // weight of m's bit 0
            for (int j = 0; j < L2M_NLIMB && !sticky; j++) {
                const int64_t lo = (int64_t) 32 * j;
                if (lo >= low_pos) break;
                const int64_t width = j < L2M_NLIMB - 1 ? 32 : 64;
                const int64_t below = std::min<int64_t>(low_pos - lo, width);     // bits of this digit under low_pos
                const uint64_t mask = below >= 64 ? ~0ull : ((1ull << below) - 1);
                if (d[j] & mask) sticky = true;
            }
            bits = nc::make_double_bits(m, low_pos - 1074, sticky) | (neg ? nc::DBL_SIGN : 0);
        }
    }
    return bits;
}

// one merged row -> the numbers cmetrics would hold.  Pure host arithmetic on integers.
extern "C" int flbgpu_l2m_finalize_row(int mode, int nbuckets, const uint64_t *row, double *value, uint64_t *buckets, uint64_t *count,
                                       double *sum) {
    *value = 0; *count = 0; *sum = 0;
    if (mode == L2M_COUNTER) {
        // cmt_counter_inc adds 1.0 to an f64 (lib/cmetrics/src/cmt_counter.c:100-116): exact up to 2^53, stuck there after
        uint64_t c = row[L2M_W_COUNT];
        if (c > (1ull << 53)) c = 1ull << 53;
        *value = (double) c;
        return 0;
    }
    if (mode == L2M_GAUGE) { memcpy(value, &row[L2M_W_LASTVAL], 8); return 0; }
    // histogram: cumulative buckets (lib/cmetrics/src/cmt_histogram.c:344-356)
    uint64_t run = 0;
    for (int b = 0; b <= nbuckets; b++) { run += row[L2M_W_BUCKET + b]; buckets[b] = run; }
    *count = run;
    const uint64_t bits = l2m_limbs_bits(row + L2M_W_LIMB, row[L2M_W_SPECIAL], row[L2M_W_SPECIAL + 1], row[L2M_W_SPECIAL + 2]);
    memcpy(sum, &bits, 8);
    return 0;
}

// device instantiation check of numconv.hpp (tests only)
extern "C" int flbgpu_nc_scan_double_dev(const char *strs, const uint32_t *off, uint32_t n, int mode, uint64_t *bits, int *consumed) {
    if (n == 0) return 0;
    void *d_s = nullptr, *d_o = nullptr, *d_b = nullptr, *d_c = nullptr;
    bool ok = hipMalloc(&d_s, off[n] + 16) == hipSuccess && hipMalloc(&d_o, (n + 1) * 4) == hipSuccess &&
              hipMalloc(&d_b, n * 8) == hipSuccess && hipMalloc(&d_c, n * 4) == hipSuccess;
    ok = ok && hipMemcpy(d_s, strs, off[n], hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_o, off, (n + 1) * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && l2m_test_numconv((const char *) d_s, (const uint32_t *) d_o, n, mode, (uint64_t *) d_b, (int *) d_c);
    ok = ok && hipMemcpy(bits, d_b, n * 8, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(consumed, d_c, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    (void) hipFree(d_s); (void) hipFree(d_o); (void) hipFree(d_b); (void) hipFree(d_c);
    if (!ok) set_err("numconv device self-test failed to run");
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------ RCCL
// The collective of SURVEY 8(e) behind the C ABI: a C engine process per GPU merges its log_to_metrics state with
// the other ranks' without leaving libflbgpu.so.  librccl is loaded on first use (the filters themselves never
// need it), so the library still loads on a box without RCCL.
#include <dlfcn.h>
namespace {
typedef int (*nccl_allreduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_allgather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_bcast_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_count_t)(void *, int *);
typedef int (*nccl_getid_t)(void *);
struct NcclId { char b[128]; };                      // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
typedef int (*nccl_initrank_t)(void **, int, NcclId, int);
typedef int (*nccl_destroy_t)(void *);
typedef const char *(*nccl_errstr_t)(int);
struct Rccl {
    void *h = nullptr;
    nccl_allreduce_t all_reduce = nullptr;
    nccl_allgather_t all_gather = nullptr;
    nccl_bcast_t bcast = nullptr;
    nccl_count_t comm_count = nullptr, comm_rank = nullptr;
    nccl_getid_t get_id = nullptr;
    nccl_initrank_t init_rank = nullptr;
    nccl_destroy_t destroy = nullptr;
    nccl_errstr_t errstr = nullptr;
};
Rccl g_rccl;
constexpr int NCCL_UINT64 = 5, NCCL_UINT8 = 1, NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2;     // rccl.h: ncclDataType_t / ncclRedOp_t

bool rccl_load() {
    if (g_rccl.h) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    // one RCCL runtime per process: a copy the host program already loaded (e.g. the one bundled with PyTorch) is reused
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (h) break; }
    if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { set_err("RCCL: librccl.so not found (%s)", dlerror()); return false; }
    Rccl r;
    r.h = h;
    r.all_reduce = (nccl_allreduce_t) dlsym(h, "ncclAllReduce"); r.all_gather = (nccl_allgather_t) dlsym(h, "ncclAllGather");
    r.bcast = (nccl_bcast_t) dlsym(h, "ncclBroadcast");
    r.comm_count = (nccl_count_t) dlsym(h, "ncclCommCount"); r.comm_rank = (nccl_count_t) dlsym(h, "ncclCommUserRank");
    r.get_id = (nccl_getid_t) dlsym(h, "ncclGetUniqueId"); r.init_rank = (nccl_initrank_t) dlsym(h, "ncclCommInitRank");
    r.destroy = (nccl_destroy_t) dlsym(h, "ncclCommDestroy"); r.errstr = (nccl_errstr_t) dlsym(h, "ncclGetErrorString");
    if (!r.all_reduce || !r.all_gather || !r.comm_count || !r.comm_rank || !r.get_id || !r.init_rank || !r.destroy) { set_err("RCCL: symbols missing in librccl"); return false; }
    g_rccl = r;
    return true;
}
#define NCCLOK(call)                                                                                           \
    do {                                                                                                       \
        int e_ = (call);                                                                                       \
        if (e_ != 0) { set_err("%s failed: %s", #call, g_rccl.errstr ? g_rccl.errstr(e_) : "?"); return -1; }  \
    } while (0)
}  // namespace

// variable-size all-gather over RCCL (sizes first, then the blobs padded to the largest): every rank ends with every rank's
// bytes, in rank order.  Staged through device buffers: RCCL moves HBM over xGMI.
bool rccl_all_gather_bytes(void *rccl_comm, hipStream_t st, const void *mine, size_t bytes, std::vector<std::vector<uint8_t>> &all) {
    if (!rccl_load()) return false;
    int world = 0;
    if (g_rccl.comm_count(rccl_comm, &world) != 0 || world <= 0) { set_err("rccl: ncclCommCount failed"); return false; }
    ScopedDevBuf d_a, d_b;                      // (released on every return path)
    uint64_t mysize = bytes;
    std::vector<uint64_t> sizes((size_t) world);
    bool ok = d_a.ensure(8) && d_b.ensure(8 * (size_t) world) && hipMemcpyAsync(d_a.p, &mysize, 8, hipMemcpyHostToDevice, st) == hipSuccess &&
              g_rccl.all_gather(d_a.p, d_b.p, 1, NCCL_UINT64, rccl_comm, st) == 0 &&
              hipMemcpyAsync(sizes.data(), d_b.p, 8 * (size_t) world, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    size_t mx = 8;
    if (ok) for (int r = 0; r < world; r++) mx = std::max<size_t>(mx, (size_t) sizes[r]);
    mx = (mx + 7) & ~(size_t) 7;
    std::vector<uint8_t> pad(mx, 0), flat(ok ? mx * (size_t) world : 0);
    if (bytes) memcpy(pad.data(), mine, bytes);
    ok = ok && d_a.ensure(mx) && d_b.ensure(mx * (size_t) world) && hipMemcpyAsync(d_a.p, pad.data(), mx, hipMemcpyHostToDevice, st) == hipSuccess &&
         g_rccl.all_gather(d_a.p, d_b.p, mx, NCCL_UINT8, rccl_comm, st) == 0 &&
         hipMemcpyAsync(flat.data(), d_b.p, mx * (size_t) world, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    d_a.release(); d_b.release();
    if (!ok) { if (!*flbgpu_last_error()) set_err("rccl: all-gather failed"); return false; }
    all.clear();
    for (int r = 0; r < world; r++) all.emplace_back(flat.begin() + (size_t) r * mx, flat.begin() + (size_t) r * mx + (size_t) sizes[r]);
    return true;
}

extern "C" int flbgpu_rccl_unique_id(void *id128) { if (!rccl_load()) return -1; NCCLOK(g_rccl.get_id(id128)); return 0; }
extern "C" int flbgpu_rccl_comm_init(void **comm, int nranks, const void *id128, int rank) {
    if (!rccl_load()) return -1;
    NcclId id;
    memcpy(id.b, id128, 128);
    NCCLOK(g_rccl.init_rank(comm, nranks, id, rank));
    return 0;
}
extern "C" int flbgpu_rccl_comm_destroy(void *comm) { if (!rccl_load()) return -1; NCCLOK(g_rccl.destroy(comm)); return 0; }

// One all-reduce of the partial aggregates per flush: the label dictionaries are made identical first
// (all-gather of the tuples), then the rows -- exact integer words -- merge with MAX (the two index words; the gauge
// value follows the winning index) and SUM (counts, bucket counts, fixed-point sum digits), on device buffers over
// RCCL.  Every rank receives the merged state in the format of flbgpu_l2m_export; the result does not depend on
// the rank count or on how the records were sharded.
extern "C" int64_t flbgpu_l2m_all_reduce(flbgpu_filter *f, void *rccl_comm, void *stream, uint64_t max_series, uint64_t *rows, uint64_t *key_off,
                                         char *keys, size_t keys_cap, size_t *keys_needed) {
    if (!f || f->kind != F_L2M) return -1;
    if (!rccl_load()) return -1;
    L2mState *s = f->l2m;
    hipStream_t st = stream ? (hipStream_t) stream : f->stream;
    const int W = s->W;
    int world = 0, rank = 0;
    NCCLOK(g_rccl.comm_count(rccl_comm, &world));
    NCCLOK(g_rccl.comm_rank(rccl_comm, &rank));
    // ---- this rank's live series
    std::vector<uint64_t> lrows, loff;
    std::vector<char> lkeys;
    size_t need = 0;
    uint64_t cap = 1024;
    int64_t ln;
    for (;;) {
        lrows.resize(cap * W); loff.resize(cap + 1); lkeys.resize(need + 16);
        ln = flbgpu_l2m_export(f, cap, lrows.data(), loff.data(), lkeys.data(), lkeys.size(), &need);
        if (ln >= 0) break;
        if (ln == -1) return -1;
        cap = (uint64_t) (-ln - 2) + 16;
    }
    // ---- all-gather of the label tuples: sizes, then the blobs padded to the largest
    ScopedDevBuf d_a, d_b;                      // (released on every return path)
    const size_t hdr = 2 * sizeof(uint64_t);
    uint64_t mine[2] = {(uint64_t) ln, (uint64_t) loff[ln]};
    std::vector<uint64_t> sizes((size_t) world * 2);
    if (!d_a.ensure(hdr) || !d_b.ensure(hdr * world)) return -1;
    if (hipMemcpyAsync(d_a.p, mine, hdr, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    NCCLOK(g_rccl.all_gather(d_a.p, d_b.p, 2, NCCL_UINT64, rccl_comm, st));
    if (hipMemcpyAsync(sizes.data(), d_b.p, hdr * world, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    size_t max_blob = 0;
    for (int r = 0; r < world; r++) { const size_t b = (size_t) (sizes[2 * r] + 1) * 8 + (size_t) sizes[2 * r + 1]; if (b > max_blob) max_blob = b; }
    max_blob = (max_blob + 7) & ~(size_t) 7;
    std::vector<uint8_t> blob(max_blob, 0), all(max_blob * world);
    memcpy(blob.data(), loff.data(), (size_t) (ln + 1) * 8);
    memcpy(blob.data() + (size_t) (ln + 1) * 8, lkeys.data(), (size_t) loff[ln]);
    if (!d_a.ensure(max_blob) || !d_b.ensure(max_blob * world)) return -1;
    if (hipMemcpyAsync(d_a.p, blob.data(), max_blob, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    NCCLOK(g_rccl.all_gather(d_a.p, d_b.p, max_blob, NCCL_UINT8, rccl_comm, st));
    if (hipMemcpyAsync(all.data(), d_b.p, max_blob * world, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    // ---- union of the tuples in (rank, local order): the same dictionary on every rank
    std::vector<std::string> ukeys;
    std::unordered_map<std::string, uint32_t> index;
    for (int r = 0; r < world; r++) {
        const uint8_t *b = all.data() + (size_t) r * max_blob;
        const uint64_t n = sizes[2 * r];
        const uint64_t *off = (const uint64_t *) b;
        const char *kb = (const char *) b + (n + 1) * 8;
        for (uint64_t i = 0; i < n; i++) {
            std::string k(kb + off[i], (size_t) (off[i + 1] - off[i]));
            if (index.emplace(k, (uint32_t) ukeys.size()).second) ukeys.push_back(k);
        }
    }
    const size_t n = ukeys.size();
    // ---- dense rows: [n][2] index words for MAX, [n][W - 2] for SUM (the gauge value travels with its index)
    std::vector<uint64_t> mx(n * 2, 0), sm(n * (size_t) (W - 2), 0);
    for (int64_t i = 0; i < ln; i++) {
        const std::string k(lkeys.data() + loff[i], (size_t) (loff[i + 1] - loff[i]));
        const size_t u = index[k];
        mx[2 * u] = lrows[(size_t) i * W]; mx[2 * u + 1] = lrows[(size_t) i * W + 1];
        memcpy(&sm[u * (size_t) (W - 2)], &lrows[(size_t) i * W + 2], (size_t) (W - 2) * 8);
    }
    std::vector<uint64_t> mine_idx(n);
    for (size_t u = 0; u < n; u++) mine_idx[u] = mx[2 * u + 1];
    if (n) {
        if (!d_a.ensure(mx.size() * 8)) return -1;
        if (hipMemcpyAsync(d_a.p, mx.data(), mx.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        NCCLOK(g_rccl.all_reduce(d_a.p, d_a.p, mx.size(), NCCL_UINT64, NCCL_MAX, rccl_comm, st));
        if (hipMemcpyAsync(mx.data(), d_a.p, mx.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        // ranks that do not own the winning index contribute 0 to the value word (word 2 = the first SUM word)
        for (size_t u = 0; u < n; u++) if (!(mine_idx[u] == mx[2 * u + 1] && mine_idx[u] != 0)) sm[u * (size_t) (W - 2)] = 0;
        if (!d_b.ensure(sm.size() * 8)) return -1;
        if (hipMemcpyAsync(d_b.p, sm.data(), sm.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        NCCLOK(g_rccl.all_reduce(d_b.p, d_b.p, sm.size(), NCCL_UINT64, NCCL_SUM, rccl_comm, st));
        if (hipMemcpyAsync(sm.data(), d_b.p, sm.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    }
    d_a.release(); d_b.release();
    // ---- snapshot order = first appearance (word 0 holds ~index of the creating record)
    std::vector<uint32_t> order(n);
    for (size_t u = 0; u < n; u++) order[u] = (uint32_t) u;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ~mx[2 * (size_t) a] < ~mx[2 * (size_t) b]; });
    if (s->sum_order_ref == 2 && s->mode == L2M_HISTOGRAM) {
        // the chain: rank after rank folds its interval's observations into the sums the rank in front ended on
        if (!g_rccl.bcast) { set_err("RCCL: ncclBroadcast missing in librccl"); return -1; }
        std::vector<double> G(n, 0.0);
        for (size_t u = 0; u < n; u++) { auto it = s->chain_sums.find(ukeys[u]); if (it != s->chain_sums.end()) G[u] = it->second; }
        ScopedDevBuf d_g;
        if (n && !d_g.ensure(n * sizeof(double))) return -1;
        for (int r = 0; r < world; r++) {
            if (r == rank && !l2m_seq_replay(f, ukeys, G, st)) return -1;
            if (n && world > 1) {
                if (hipMemcpyAsync(d_g.p, G.data(), n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
                NCCLOK(g_rccl.bcast(d_g.p, d_g.p, n, NCCL_FLOAT64, r, rccl_comm, st));
                if (hipMemcpyAsync(G.data(), d_g.p, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
            }
        }
        for (size_t u = 0; u < n; u++) s->chain_sums[ukeys[u]] = G[u];
        s->last_chain.resize(n);
        for (size_t j = 0; j < n; j++) s->last_chain[j] = G[order[j]];
    }
    size_t kneed = 0;
    for (auto &k : ukeys) kneed += k.size();
    if (keys_needed) *keys_needed = kneed;
    if (n > max_series || kneed > keys_cap) return -(int64_t) n - 2;
    size_t ko = 0;
    for (size_t j = 0; j < n; j++) {
        const uint32_t u = order[j];
        rows[j * W] = mx[2 * (size_t) u]; rows[j * W + 1] = mx[2 * (size_t) u + 1];
        memcpy(&rows[j * W + 2], &sm[(size_t) u * (W - 2)], (size_t) (W - 2) * 8);
        key_off[j] = ko;
        memcpy(keys + ko, ukeys[u].data(), ukeys[u].size());
        ko += ukeys[u].size();
    }
    key_off[n] = ko;
    return (int64_t) n;
}
