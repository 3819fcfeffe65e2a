// tail.cpp -- in_tail's line packing behind the C ABI: the buffer of a tailed file cut into lines, every line one log event
// (plugins/in_tail/tail_file.c:689-1040 process_content, the plain path, + :552-604 flb_tail_file_pack_line).  Kernels:
// tail_kernels.inc.  What comes out is a device chunk (bytes + one row per newline, skipped lines as empty rows) that the
// filters take as it is (flbgpu_filter_chain_run_dev), or a malloc()'d host buffer.
#include "host_int.hpp"

using namespace flbgpu;

struct flbgpu_tail {
    TailArgs a;                       // configuration part filled at create
    hipStream_t stream = nullptr;
    DevBuf d_masks, d_tcnt, d_toff, d_scan_tmp, d_nl, d_len, d_off, d_out, d_misc, d_in;
    ~flbgpu_tail() {
        d_masks.release(); d_tcnt.release(); d_toff.release(); d_scan_tmp.release(); d_nl.release(); d_len.release(); d_off.release();
        d_out.release(); d_misc.release(); d_in.release();
        if (stream) (void) hipStreamDestroy(stream);
    }
};

static bool put_str(std::vector<uint8_t> &b, const char *s) {
    const size_t n = strlen(s);
    if (n > 255) return false;
    if (n < 32) b.push_back((uint8_t) (0xa0 | n));
    else { b.push_back(0xd9); b.push_back((uint8_t) n); }
    b.insert(b.end(), s, s + n);
    return true;
}

// Key (default "log"), Path_Key / the file's name, Offset_Key / the file's stream offset, Skip_Empty_Lines (plugins/in_tail/tail.c
// config map); the timestamp every record of a call carries (the reference stamps flb_time_get() per record)
extern "C" flbgpu_tail *flbgpu_tail_create(const char *key, const char *path_key, const char *path, const char *offset_key, int skip_empty_lines) {
    if (flbgpu_device_cus() <= 0) { set_err("flbgpu_init has not run: libflbgpu has no CPU path"); return nullptr; }
    auto *t = new flbgpu_tail();
    memset(&t->a, 0, sizeof(t->a));
    std::vector<uint8_t> pa, pb, pc;
    bool ok = put_str(pc, key && key[0] ? key : "log");
    if (path_key && path_key[0]) ok = ok && put_str(pa, path_key) && put_str(pa, path ? path : "");
    if (offset_key && offset_key[0]) ok = ok && put_str(pb, offset_key);
    if (!ok || pa.size() + pb.size() + pc.size() > sizeof(t->a.pre)) { set_err("in_tail: key / path_key / path / offset_key too long for the GPU path"); delete t; return nullptr; }
    t->a.la = (uint32_t) pa.size(); t->a.lb = (uint32_t) pb.size(); t->a.lc = (uint32_t) pc.size();
    memcpy(t->a.pre, pa.data(), pa.size());
    memcpy(t->a.pre + pa.size(), pb.data(), pb.size());
    memcpy(t->a.pre + pa.size() + pb.size(), pc.data(), pc.size());
    t->a.skip_empty_lines = skip_empty_lines ? 1 : 0;
    if (hipStreamCreate(&t->stream) != hipSuccess) { set_err("hipStreamCreate failed"); delete t; return nullptr; }
    return t;
}

extern "C" void flbgpu_tail_destroy(flbgpu_tail *t) { delete t; }

struct TailWords { unsigned long long lead, lines; };

// text in HBM -> records in HBM.  *processed: bytes consumed (the file's buffer keeps what follows); *lines: records produced.
extern "C" int flbgpu_tail_run_dev(flbgpu_tail *t, const void *d_text, uint64_t bytes, uint64_t stream_offset, uint32_t ts_sec, uint32_t ts_nsec,
                                   flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *lines) {
    auto fail = [](const char *w) -> int { set_err("in_tail: %s", w); return -1; };
    memset(out, 0, sizeof(*out));
    *processed = 0; *lines = 0;
    if (!t) return fail("no context");
    if (bytes == 0) return 0;
    hipStream_t st = t->stream;
    const uint8_t *text = (const uint8_t *) d_text;
    const size_t ntiles = tl_tiles(bytes);
    if (!t->d_masks.ensure((bytes + 63) / 64 * sizeof(uint64_t)) || !t->d_tcnt.ensure(ntiles * sizeof(uint32_t)) ||
        !t->d_toff.ensure((ntiles + 1) * sizeof(uint64_t)) || !t->d_scan_tmp.ensure(scan_tmp_elems(ntiles) * sizeof(uint64_t)) ||
        !t->d_misc.ensure(sizeof(TailWords))) return -1;
    TailWords *dw = t->d_misc.as<TailWords>();
    if (hipMemsetAsync(dw, 0, sizeof(TailWords), st) != hipSuccess) return fail("memset failed");
    launch_tl_lead(text, bytes, &dw->lead, st);
    launch_tl_count(text, bytes, t->d_masks.as<uint64_t>(), t->d_tcnt.as<uint32_t>(), st);
    launch_scan(t->d_tcnt.as<uint32_t>(), ntiles, t->d_scan_tmp.as<uint64_t>(), t->d_toff.as<uint64_t>(), st);
    uint64_t nl = 0;
    TailWords hw;
    if (hipMemcpyAsync(&nl, t->d_toff.as<uint64_t>() + ntiles, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&hw, dw, sizeof(hw), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("mask pass failed");
    if (nl == 0) { *processed = hw.lead; return 0; }        // no complete line yet: only the leading NULs are consumed
    if (!t->d_nl.ensure(nl * sizeof(uint64_t)) || !t->d_len.ensure(nl * sizeof(uint32_t)) || !t->d_off.ensure((nl + 1) * sizeof(uint64_t)) ||
        !t->d_scan_tmp.ensure(scan_tmp_elems(nl > ntiles ? nl : ntiles) * sizeof(uint64_t))) return -1;
    launch_tl_fill(t->d_masks.as<uint64_t>(), bytes, t->d_toff.as<uint64_t>(), t->d_nl.as<uint64_t>(), st);
    TailArgs a = t->a;
    a.text = text; a.bytes = bytes; a.lead = &dw->lead; a.nl_pos = t->d_nl.as<uint64_t>(); a.nl = nl; a.out_len = t->d_len.as<uint32_t>();
    a.out_off = t->d_off.as<uint64_t>(); a.out = nullptr; a.lines = &dw->lines; a.stream_offset = stream_offset; a.ts_sec = ts_sec; a.ts_nsec = ts_nsec;
    launch_tl_size(a, st);
    launch_scan(a.out_len, nl, t->d_scan_tmp.as<uint64_t>(), t->d_off.as<uint64_t>(), st);
    uint64_t total = 0, last_nl = 0;
    if (hipMemcpyAsync(&total, t->d_off.as<uint64_t>() + nl, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&last_nl, t->d_nl.as<uint64_t>() + (nl - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&hw, dw, sizeof(hw), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("size pass failed");
    *processed = last_nl + 1;
    *lines = hw.lines;
    if (!t->d_out.ensure(total + 16)) return -1;
    if (total > 0) {
        a.out = t->d_out.as<uint8_t>();
        launch_tl_emit(a, flbgpu_device_cus(), st);
        if (hipStreamSynchronize(st) != hipSuccess) return fail("emit pass failed");
    }
    out->data = t->d_out.p; out->row_off = t->d_off.as<uint64_t>(); out->n = nl; out->bytes = total;
    return 0;
}

// the same on a host buffer; *out_buf is malloc()'d (nullptr when no line was complete)
extern "C" int flbgpu_tail_run(flbgpu_tail *t, const void *text, size_t bytes, uint64_t stream_offset, uint32_t ts_sec, uint32_t ts_nsec,
                               void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *lines) {
    *out_buf = nullptr; *out_size = 0;
    if (!t) { set_err("in_tail: no context"); return -1; }
    if (bytes && (!t->d_in.ensure(bytes + 16) || hipMemcpyAsync(t->d_in.p, text, bytes, hipMemcpyHostToDevice, t->stream) != hipSuccess)) { set_err("in_tail: upload failed"); return -1; }
    flbgpu_dev_chunk out;
    if (flbgpu_tail_run_dev(t, t->d_in.p, bytes, stream_offset, ts_sec, ts_nsec, &out, processed, lines) != 0) return -1;
    if (out.bytes == 0) return 0;
    void *hb = malloc(out.bytes);
    if (!hb) { set_err("out of memory"); return -1; }
    if (hipMemcpy(hb, out.data, out.bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(hb); set_err("in_tail: download failed"); return -1; }
    *out_buf = hb; *out_size = out.bytes;
    return 0;
}
