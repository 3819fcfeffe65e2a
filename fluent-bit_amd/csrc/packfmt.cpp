// packfmt.cpp -- host side of the msgpack -> JSON output formatter (flb_pack_msgpack_to_json_format,
// /root/reference src/flb_pack.c:1320-1600; kernels in kernels_fmt.hip / fmt_dev.inc).
//
// Two passes over the record rows -- size, scan, emit -- like the filters.  What the decoder makes sequential is
// handled between them on the host: the walk ends at the first row it refuses (src/flb_log_event_decoder.c:342-384),
// group openers govern the rows that follow them (:416-485) and 1000 consecutive skipped rows end the walk (:389-394);
// the last two need a pass of their own (k_fmt_groups), run only for chunks that hold markers.
#include "host_int.hpp"

using namespace flbgpu;

namespace {
struct FmtWords { unsigned long long first_bad, first_fail, skip_limit, counts[4]; };
}

extern "C" flbgpu_filter *flbgpu_jsonfmt_create(int json_format, int date_format, const char *date_key, int date_key_len,
                                                int escape_unicode, int convert_nan_to_null) {
    if (json_format < 1 || json_format > 3) { set_err("json_format %d: json (1), stream (2) or lines (3)", json_format); return nullptr; }
    if (date_key && date_key_len >= 0 && (date_format < 0 || date_format > 4)) { set_err("date_format %d is not a FLB_PACK_JSON_DATE_* value", date_format); return nullptr; }
    flbgpu_filter *f = new flbgpu_filter();
    f->kind = F_JSONFMT;
    if (!filter_common_init(f)) { delete f; return nullptr; }
    JsonFmtCfg &c = f->jcfg;
    c.json_format = json_format; c.date_format = date_format; c.escape_unicode = escape_unicode ? 1 : 0; c.nan_to_null = convert_nan_to_null ? 1 : 0;
    c.has_date = (date_key && date_key_len >= 0) ? 1 : 0;
    c.date_key_len = c.has_date ? (uint32_t) date_key_len : 0;
    c.date_key_is_internal = c.has_date && date_key_len == 12 && memcmp(date_key, "__internal__", 12) == 0;
    c.date_key = nullptr;
    if (c.has_date) {
        if (!f->d_datekey.ensure(c.date_key_len + 16) ||
            (c.date_key_len && hipMemcpy(f->d_datekey.p, date_key, c.date_key_len, hipMemcpyHostToDevice) != hipSuccess)) {
            set_err("device copy of the date key failed");
            delete f;
            return nullptr;
        }
        c.date_key = f->d_datekey.as<uint8_t>();
    }
    return f;
}

// out->data: the JSON text in HBM (out->bytes bytes; valid until the next call on this formatter), out->row_off /
// out->n: the offset column of the rows (row r's text is [row_off[r], row_off[r+1]); in json format the closing ']' is
// the last byte, outside every row).  Returns 0, or -1 where the reference returns NULL (nothing to print, or a record
// whose temporary map does not unpack) and on errors (flbgpu_last_error() then is not empty).
extern "C" int flbgpu_jsonfmt_run_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in_raw, flbgpu_dev_chunk *out) {
    if (!f || f->kind != F_JSONFMT) { set_err("not a JSON formatter"); return -1; }
    set_err("");
    memset(out, 0, sizeof(*out));
    hipStream_t st = f->stream;
    flbgpu_dev_chunk in;
    bool garbage = false;
    if (!resolve_raw_chunk(f, in_raw, &in, &garbage)) return -1;
    const int cus = device_cus() > 0 ? device_cus() : 256;
    const bool json = f->jcfg.json_format == 1;
    uint64_t n = in.n;
    auto fail = [&](const char *what) { set_err("%s", what); return -1; };
    uint64_t total = 0;
    if (n > 0) {
        if (!f->d_misc.ensure(sizeof(FmtWords)) || !f->hp_misc.ensure(sizeof(FmtWords) + sizeof(uint64_t))) return -1;
        FmtWords *dm = f->d_misc.as<FmtWords>();
        FmtWords &hm = *f->hp_misc.as<FmtWords>();
        uint64_t &htotal = *(uint64_t *) (f->hp_misc.as<uint8_t>() + sizeof(FmtWords));
        if (!f->d_len.ensure(n * sizeof(uint32_t)) || !f->d_off.ensure((n + 1) * sizeof(uint64_t)) || !f->d_status.ensure(((n + 63) / 64) * sizeof(uint64_t)) ||
            !f->d_scan_tmp.ensure(scan_tmp_elems(n) * sizeof(uint64_t))) return -1;
        JsonFmtArgs a;
        a.data = (const uint8_t *) in.data; a.row_off = in.row_off; a.n = n; a.cfg = f->jcfg; a.len = f->d_len.as<uint32_t>();
        a.bytes = in.bytes; a.slow = f->d_status.as<uint64_t>();
        a.g_row = nullptr; a.first_bad = &dm->first_bad; a.first_fail = &dm->first_fail; a.counts = dm->counts;
        a.out_off = nullptr; a.out = nullptr;
        auto size_pass = [&]() -> bool {
            memset(&hm, 0, sizeof(hm));
            hm.first_bad = hm.first_fail = hm.skip_limit = ~0ull;
            if (hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st) != hipSuccess) return false;
            { ProfScope ps(f, st, "k_fmt_size"); launch_fmt_size(a, cus, st); }
            if (hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
            return hipStreamSynchronize(st) == hipSuccess;
        };
        if (!size_pass()) return fail("size pass failed");
        if (hm.counts[1] > 0 || hm.counts[2] >= 1000) {
            // markers in the chunk: the rows' group state first, then the sizes again with the group attributes
            const unsigned long long first_bad = hm.first_bad;
            if (!f->d_grow.ensure(n * sizeof(uint32_t))) return -1;
            JsonGroupArgs g;
            g.data = a.data; g.row_off = a.row_off; g.n = first_bad < n ? first_bad : n; g.g_row = f->d_grow.as<uint32_t>(); g.skip_limit = &dm->skip_limit;
            if (hipMemsetAsync(f->d_grow.p, 0, n * sizeof(uint32_t), st) != hipSuccess) return fail("memset failed");
            { ProfScope ps(f, st, "k_fmt_groups"); launch_fmt_groups(g, st); }
            unsigned long long skip_limit = ~0ull;
            if (hipMemcpyAsync(&skip_limit, &dm->skip_limit, sizeof(skip_limit), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return fail("group pass failed");
            if (skip_limit < n) n = skip_limit;
            a.n = n;
            a.g_row = f->d_grow.as<uint32_t>();
            if (n > 0 && !size_pass()) return fail("size pass failed");
            if (n == 0) { memset(&hm, 0, sizeof(hm)); hm.first_bad = hm.first_fail = ~0ull; }
        }
        // the decoder's walk ends at the first row it refuses
        if (hm.first_bad < n) n = hm.first_bad;
        if (hm.first_fail < n) return -1;                       // the reference returns NULL (flb_msgpack_raw_to_json_sds fails)
        a.n = n;
        if (n > 0) {
            { ProfScope ps(f, st, "k_scan"); launch_scan(f->d_len.as<uint32_t>(), n, f->d_scan_tmp.as<uint64_t>(), f->d_off.as<uint64_t>(), st); }
            htotal = 0;
            if (hipMemcpyAsync(&htotal, f->d_off.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return fail("scan failed");
            total = htotal;
        }
        if (!f->d_out.ensure(total + 16)) return -1;
        if (total > 0) {
            a.out_off = f->d_off.as<uint64_t>(); a.out = f->d_out.as<uint8_t>();
            { ProfScope ps(f, st, "k_fmt_emit"); launch_fmt_emit(a, cus, st); }
        }
    }
    else if (!f->d_out.ensure(16)) return -1;
    if (json) {
        // "[" rows "]": an empty walk still prints "[]"
        const char tail[2] = {'[', ']'};
        if (total == 0) { if (hipMemcpyAsync(f->d_out.p, tail, 2, hipMemcpyHostToDevice, st) != hipSuccess) return fail("copy failed"); total = 2; }
        else { if (hipMemcpyAsync(f->d_out.as<uint8_t>() + total, tail + 1, 1, hipMemcpyHostToDevice, st) != hipSuccess) return fail("copy failed"); total += 1; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail("emit pass failed");
    if (n > 0) {
        unsigned long long mism = 0;
        if (hipMemcpy(&mism, &f->d_misc.as<FmtWords>()->counts[3], sizeof(mism), hipMemcpyDeviceToHost) != hipSuccess) return fail("device read failed");
        if (mism) { set_err("emit pass disagrees with the size pass on %llu rows", mism); return -1; }
    }
    if (total == 0) return -1;                                  // lines / stream with nothing to print: NULL (:1594-1597)
    out->data = f->d_out.p; out->row_off = n > 0 ? f->d_off.as<uint64_t>() : nullptr; out->n = n; out->bytes = total;
    return 0;
}

// flb_pack_msgpack_to_json_format on a host chunk: *out is malloc()'d (NUL terminated like an flb_sds_t; release with free())
extern "C" int flbgpu_jsonfmt_run(flbgpu_filter *f, const void *data, size_t bytes, char **out_buf, size_t *out_size) {
    if (!f || f->kind != F_JSONFMT) { set_err("not a JSON formatter"); return -1; }
    *out_buf = nullptr; *out_size = 0;
    flbgpu_dev_chunk in, out;
    memset(&in, 0, sizeof(in));
    if (bytes > 0) {
        size_t consumed = 0;
        const uint64_t *row_off = nullptr;
        int64_t n = staged_upload(f, (const uint8_t *) data, bytes, &consumed, &row_off);
        if (n < 0) return -1;
        in.data = f->h_in_data.p; in.row_off = row_off; in.n = (uint64_t) n; in.bytes = consumed;
        if (n == 0) { in.row_off = nullptr; in.bytes = 0; }
    }
    if (flbgpu_jsonfmt_run_dev(f, &in, &out) != 0) return -1;
    char *hb = (char *) malloc(out.bytes + 1);
    if (!hb) { set_err("out of memory"); return -1; }
    if (!staged_download(f, hb, out.data, out.bytes)) { free(hb); set_err("device to host copy failed"); return -1; }
    hb[out.bytes] = 0;
    *out_buf = hb; *out_size = out.bytes;
    return 0;
}

// one-shot form with the reference's argument list (date_key_len < 0: no date key)
extern "C" int flbgpu_pack_msgpack_to_json_format(const char *data, uint64_t bytes, int json_format, int date_format, const char *date_key,
                                                  int date_key_len, int escape_unicode, int convert_nan_to_null, char **out_buf, size_t *out_size) {
    *out_buf = nullptr; *out_size = 0;
    if (json_format < 1 || json_format > 3) return -1;             // any other format prints nothing (:1584-1597)
    flbgpu_filter *f = flbgpu_jsonfmt_create(json_format, date_format, date_key, date_key_len, escape_unicode, convert_nan_to_null);
    if (!f) return -1;
    const int r = flbgpu_jsonfmt_run(f, data, (size_t) bytes, out_buf, out_size);
    flbgpu_filter_destroy(f);
    return r;
}
