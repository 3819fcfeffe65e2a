// json.cpp -- host side of the JSON -> msgpack path (include/flb_gpu.h "flb_pack_json").
//
//   flbgpu_pack_json / flbgpu_pack_json_recs  ~ flb_pack_json / flb_pack_json_recs (src/flb_pack.c:670-688)
//   flbgpu_json_run_dev                       the batched sibling: every text row is one such call;
//                                             with `events` the rows that are one JSON object become
//                                             V2 log events, ready for the filter chain
// Kernels: json_kernels.inc (size pass, scan, emit pass; generic twins for deep / hard rows).
#include "host_int.hpp"
#include "jtile.hpp"

using namespace flbgpu;

struct JsonMisc { unsigned long long counts[8]; };     // dev.hpp JsonArgs::counts

struct flbgpu_json {
    hipStream_t stream = nullptr;
    DevBuf d_len, d_rec, d_cons, d_rt, d_st, d_off, d_tmp, d_out, d_misc, d_cnt, h_text, h_off;
    DevBuf d_spec, d_off2, d_keep, d_tiles;      // the tile pass: its output, the second leg's offsets and copy lengths, the look-back words
    uint64_t n = 0;
    uint64_t stats[3] = {0, 0, 0};
    uint64_t tile_stats[4] = {0, 0, 0, 0};       // rows the tile pass wrote / left to the row-per-lane kernels, its launches, tokens
    uint64_t tile_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dbg_prof = 0, dbg_lb_off = 0;            // flbgpu_json_tile_debug (timing experiments)
    DevBuf d_prof;
    ~flbgpu_json() {
        DevBuf *all[] = {&d_len, &d_rec, &d_cons, &d_rt, &d_st, &d_off, &d_tmp, &d_out, &d_misc, &d_cnt, &h_text, &h_off, &d_spec, &d_off2, &d_keep, &d_tiles, &d_prof};
        for (auto *b : all) b->release();
        if (stream) (void) hipStreamDestroy(stream);
    }
};

extern "C" flbgpu_json *flbgpu_json_create(void) {
    auto *j = new flbgpu_json();
    if (hipStreamCreate(&j->stream) != hipSuccess) { set_err("hipStreamCreate failed: no HIP device? (libflbgpu has no CPU path)"); delete j; return nullptr; }
    return j;
}
extern "C" void flbgpu_json_destroy(flbgpu_json *j) { delete j; }

static bool json_run(flbgpu_json *j, const flbgpu_dev_chunk *in, int events, uint32_t ts_sec, uint32_t ts_nsec, flbgpu_dev_chunk *out) {
    const uint64_t n = in->n;
    hipStream_t st = j->stream;
    j->n = n;
    memset(out, 0, sizeof(*out));
    j->stats[0] = j->stats[1] = j->stats[2] = 0;
    if (n == 0) return true;
    if (!j->d_len.ensure(n * 4) || !j->d_rec.ensure(n * 4) || !j->d_cons.ensure(n * 4) || !j->d_rt.ensure(n) || !j->d_st.ensure(n) ||
        !j->d_off.ensure((n + 1) * 8) || !j->d_tmp.ensure(scan_tmp_elems(n) * 8) || !j->d_misc.ensure(sizeof(JsonMisc)) ||
        !j->d_cnt.ensure(n * 8 * 4)) return false;
    HIPOK(hipMemsetAsync(j->d_misc.p, 0, sizeof(JsonMisc), st));
    JsonArgs a;
    memset(&a, 0, sizeof(a));
    a.text = (const uint8_t *) in->data; a.row_off = in->row_off; a.n = n;
    a.out_len = j->d_len.as<uint32_t>(); a.records = j->d_rec.as<uint32_t>(); a.consumed = j->d_cons.as<uint32_t>();
    a.root_type = j->d_rt.as<uint8_t>(); a.status = j->d_st.as<uint8_t>(); a.cnt = j->d_cnt.as<uint32_t>();
    a.events = events; a.ts_sec = ts_sec; a.ts_nsec = ts_nsec;
    a.counts = j->d_misc.as<JsonMisc>()->counts;
    const int cus = device_cus() > 0 ? device_cus() : 256;
    JsonMisc hm;
    // ---- first leg: the tile pass (jtile_kernels.inc) -- a wave per tile of rows, the text read once, the output placed by a look-back
    // over the tiles.  It takes the rows that are one JSON object (and blank rows) and leaves the rest marked JS_TDEFER.
    static const bool tile_off = getenv("FLBGPU_JSON_TILE") && atoi(getenv("FLBGPU_JSON_TILE")) == 0;
    const bool tile = !tile_off && n >= 2 && ((uintptr_t) in->data & 15) == 0;
    j->tile_stats[0] = j->tile_stats[1] = j->tile_stats[2] = j->tile_stats[3] = 0;
    if (tile) {
        const uint64_t avg = in->bytes / n + 1;
        uint64_t R = (uint64_t) (json_lane_text_bytes() - 16) * 92 / 100 / avg;
        if (R < 1) R = 1;
        if (R > 64) R = 64;
        JtArgs t;
        memset(&t, 0, sizeof(t));
        t.ntiles = (n + R - 1) / R;
        t.rows_per_tile = (uint32_t) R;
        const uint64_t nunits = json_lane_units(t.ntiles);
        if (!j->d_tiles.ensure(nunits * 8 + 8)) return false;
        if (j->dbg_prof) { if (!j->d_prof.ensure(64)) return false; HIPOK(hipMemsetAsync(j->d_prof.p, 0, 64, st)); t.prof = j->d_prof.as<unsigned long long>(); }
        t.lb_off = j->dbg_lb_off;
        uint64_t cap = in->bytes + (events ? 13 * n : 0) + 64;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (!j->d_spec.ensure(cap + 16)) return false;
            HIPOK(hipMemsetAsync(j->d_tiles.p, 0, nunits * 8, st));
            if (attempt) HIPOK(hipMemsetAsync(j->d_misc.p, 0, sizeof(JsonMisc), st));
            { unsigned long long *pr = t.prof; const int lb = t.lb_off; const uint64_t nt = t.ntiles; const uint32_t rp = t.rows_per_tile;
              memset(&t, 0, sizeof(t)); t.prof = pr; t.lb_off = lb; t.ntiles = nt; t.rows_per_tile = rp; }
            t.j = a;
            t.j.out = j->d_spec.as<uint8_t>();
            t.off_out = j->d_off.as<uint64_t>();
            t.tile_state = j->d_tiles.as<unsigned long long>();
            t.ticket = &j->d_misc.as<JsonMisc>()->counts[7];
            t.out_cap = cap;
            launch_json_lane(t, cus, st);
            HIPOK(hipMemcpyAsync(&hm, j->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
            j->tile_stats[2]++;
            if (j->dbg_prof) HIPOK(hipMemcpy(j->tile_prof, j->d_prof.p, 64, hipMemcpyDeviceToHost));
            if (!hm.counts[5]) break;
            if (hm.counts[5] >> 32) { set_err("JSON tile pass: a tile waited for the tiles in front of it and gave up"); return false; }
            if (attempt) { set_err("JSON tile pass: no room for its own output size"); return false; }
            cap = hm.counts[4] + 64;                  // the pass is deterministic: the size it reports is the size it needs
        }
        j->tile_stats[0] = n - hm.counts[3]; j->tile_stats[1] = hm.counts[3]; j->tile_stats[3] = hm.counts[6];
        if (hm.counts[3] == 0) {
            j->stats[0] = 0; j->stats[1] = hm.counts[1]; j->stats[2] = 0;
            out->data = j->d_spec.p; out->row_off = j->d_off.as<uint64_t>(); out->n = n; out->bytes = hm.counts[4];
            return true;
        }
        // ---- second leg: the rows the pass left go through the row-per-lane kernels; what it wrote is copied to its final place
        if (!j->d_keep.ensure(n * 4) || !j->d_off2.ensure((n + 1) * 8)) return false;
        HIPOK(hipMemcpyAsync(j->d_keep.p, j->d_len.p, n * 4, hipMemcpyDeviceToDevice, st));
        a.tile_mode = 1;
    }
    launch_json_size(a, cus, st);
    HIPOK(hipMemcpyAsync(&hm, j->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    const bool generic = hm.counts[0] > 0;
    if (generic) launch_json_generic(a, false, st);
    uint64_t *final_off = a.tile_mode ? j->d_off2.as<uint64_t>() : j->d_off.as<uint64_t>();
    launch_scan(a.out_len, n, j->d_tmp.as<uint64_t>(), final_off, st);
    uint64_t total = 0;
    HIPOK(hipMemcpyAsync(&total, final_off + n, 8, hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(&hm, j->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    if (!j->d_out.ensure(total + 16)) return false;
    a.out_off = final_off; a.out = j->d_out.as<uint8_t>();
    if (a.tile_mode) {
        GatherArgs g;
        memset(&g, 0, sizeof(g));
        g.data = j->d_spec.as<uint8_t>(); g.row_off = j->d_off.as<uint64_t>(); g.n = n; g.keep_len = j->d_keep.as<uint32_t>();
        g.out_off = final_off; g.out = j->d_out.as<uint8_t>();
        launch_gather(g, st);
    }
    launch_json_emit(a, cus, st);
    if (generic) launch_json_generic(a, true, st);
    HIPOK(hipStreamSynchronize(st));
    j->stats[0] = hm.counts[0]; j->stats[1] = hm.counts[1]; j->stats[2] = hm.counts[2];
    out->data = j->d_out.p; out->row_off = final_off; out->n = n; out->bytes = total;
    return true;
}

extern "C" int flbgpu_json_run_dev(flbgpu_json *j, const flbgpu_dev_chunk *text_rows, int events, uint32_t ts_sec, uint32_t ts_nsec,
                                   flbgpu_dev_chunk *out) {
    return json_run(j, text_rows, events, ts_sec, ts_nsec, out) ? 0 : -1;
}

extern "C" int flbgpu_json_row_info(flbgpu_json *j, uint64_t first, uint64_t count, uint32_t *records, uint32_t *consumed,
                                    uint8_t *root_type, uint8_t *status) {
    if (first + count > j->n) { set_err("row range out of bounds"); return -1; }
    if (count == 0) return 0;
    bool ok = true;
    if (records) ok = ok && hipMemcpy(records, j->d_rec.as<uint32_t>() + first, count * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (consumed) ok = ok && hipMemcpy(consumed, j->d_cons.as<uint32_t>() + first, count * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (root_type) ok = ok && hipMemcpy(root_type, j->d_rt.as<uint8_t>() + first, count, hipMemcpyDeviceToHost) == hipSuccess;
    if (status) {
        ok = ok && hipMemcpy(status, j->d_st.as<uint8_t>() + first, count, hipMemcpyDeviceToHost) == hipSuccess;
        for (uint64_t i = 0; ok && i < count; i++) status[i] &= 0x3f;     // bits 6 and 7 only route rows between kernels
    }
    if (!ok) set_err("device read failed");
    return ok ? 0 : -1;
}

extern "C" void flbgpu_json_stats(flbgpu_json *j, uint64_t *out3) { out3[0] = j->stats[0]; out3[1] = j->stats[1]; out3[2] = j->stats[2]; }
extern "C" void flbgpu_json_tile_stats(flbgpu_json *j, uint64_t *out4) { for (int i = 0; i < 4; i++) out4[i] = j->tile_stats[i]; }
extern "C" void flbgpu_json_tile_debug(flbgpu_json *j, int prof, int no_lookback, uint64_t *phases8) {
    j->dbg_prof = prof; j->dbg_lb_off = no_lookback;
    if (phases8) for (int i = 0; i < 8; i++) phases8[i] = j->tile_prof[i];
}

extern "C" int flbgpu_pack_json_recs(const char *js, size_t len, char **buffer, size_t *size, int *root_type, int *out_records,
                                     size_t *consumed) {
    static thread_local flbgpu_json *tl = nullptr;
    if (!js || !buffer || !size) return -1;
    if (len > 0xFFFFFFFFull) { set_err("flbgpu_pack_json: a text over 4 GiB must be split into rows"); return -1; }
    if (!tl) tl = flbgpu_json_create();
    if (!tl) return -1;
    flbgpu_json *j = tl;
    uint64_t off[2] = {0, len};
    if (!j->h_text.ensure(len + 16) || !j->h_off.ensure(sizeof(off))) return -1;
    if ((len && hipMemcpy(j->h_text.p, js, len, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(j->h_off.p, off, sizeof(off), hipMemcpyHostToDevice) != hipSuccess) { set_err("host to device copy failed"); return -1; }
    flbgpu_dev_chunk in, out;
    in.data = j->h_text.p; in.row_off = j->h_off.as<uint64_t>(); in.n = 1; in.bytes = len;
    if (!json_run(j, &in, 0, 0, 0, &out)) return -1;
    uint32_t rec = 0, cons = 0;
    uint8_t rt = 0, st = 0;
    if (flbgpu_json_row_info(j, 0, 1, &rec, &cons, &rt, &st) != 0) return -1;
    if (st != 0) return -1;                              // nothing parsed (flb_pack.c:441-452)
    char *hb = nullptr;
    if (out.bytes) {
        hb = (char *) malloc(out.bytes);
        if (!hb || hipMemcpy(hb, out.data, out.bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(hb); set_err("device to host copy failed"); return -1; }
    }
    *buffer = hb;                                        // NULL with size 0 for blank input, like msgpack_sbuffer
    *size = out.bytes;
    if (root_type && rec) *root_type = rt;
    if (out_records) *out_records = (int) rec;
    if (consumed) *consumed = cons;
    return 0;
}

extern "C" int flbgpu_pack_json(const char *js, size_t len, char **buffer, size_t *size, int *root_type, size_t *consumed) {
    int records = 0;
    return flbgpu_pack_json_recs(js, len, buffer, size, root_type, &records, consumed);
}

// rows of an NDJSON buffer: every line including its '\n' (the last one may lack it)
extern "C" int64_t flbgpu_split_lines_host(const void *data, size_t bytes, uint64_t *row_off, size_t cap) {
    const char *d = (const char *) data;
    size_t pos = 0;
    int64_t n = 0;
    while (pos < bytes) {
        if ((size_t) n + 1 >= cap) return -1;
        row_off[n++] = pos;
        const char *nl = (const char *) memchr(d + pos, '\n', bytes - pos);
        pos = nl ? (size_t) (nl - d) + 1 : bytes;
    }
    if (cap > 0) row_off[n] = pos;
    return n;
}
