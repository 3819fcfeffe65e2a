// flbgpu.cpp -- host side of libflbgpu.so: the C ABI declared in include/flb_gpu.h.
//
// Mirrors the reference's host-side structure for the filter hot path:
//   flbgpu_parser        ~ struct flb_parser            (include/fluent-bit/flb_parser.h:41-70)
//   flbgpu_filter(grep)  ~ struct grep_ctx + rules      (plugins/filter_grep/grep.h)
//   flbgpu_filter(parser)~ struct filter_parser_ctx     (plugins/filter_parser/filter_parser.h)
//   flbgpu_filter_run    ~ cb_filter                    (include/fluent-bit/flb_filter.h:57-81)
// Configuration-time work (regex compile, rule parsing, time-format analysis) happens here on the
// host exactly once; per-record work happens only in kernels.hip.  There is no CPU data path.
#include <condition_variable>
#include <functional>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include "host_int.hpp"
#include "grep_lane.hpp"
#include "perm.hpp"

using namespace flbgpu;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int g_cus = 0;

int flbgpu::device_cus() { return g_cus; }

void flbgpu::set_err(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    if (getenv("FLBGPU_DEBUG")) fprintf(stderr, "[flbgpu] %s\n", buf);
}

extern "C" const char *flbgpu_last_error(void) { return g_err.c_str(); }

extern "C" int flbgpu_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_err("no HIP device available (%s): libflbgpu has no CPU path", e != hipSuccess ? hipGetErrorString(e) : "count=0");
        return -1;
    }
    if (device < 0 || device >= n) { set_err("device %d out of range (0..%d)", device, n - 1); return -1; }
    if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice(%d) failed", device); return -1; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) g_cus = prop.multiProcessorCount;
    if (!upload_time_tables()) { set_err("uploading the strptime tables failed"); return -1; }
    return 0;
}

extern "C" int flbgpu_device_cus(void) { return g_cus; }

// the "now" of year-less Time_Formats: time(NULL) unless a test pinned it
static int64_t g_time_now = 0;
extern "C" void flbgpu_set_time_now(int64_t now) { g_time_now = now; }

// ------------------------------------------------------------------------------------------ device buffers
template <class T> static size_t put(std::vector<uint8_t> &blob, const std::vector<T> &v) {
    size_t off = (blob.size() + 15) & ~(size_t) 15;
    blob.resize(off + v.size() * sizeof(T) + 16);
    if (!v.empty()) memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

bool flbgpu::upload_cap(const rx::TableSet &t, TableBlob &blob, DevCap &out) {
    std::vector<uint8_t> b;
    std::vector<uint8_t> cls(t.cls, t.cls + 512);
    // hot block first (staged into LDS as one piece): rdelta | ft | ft2 | cls | col
    // device encoding of the forward tables: a plain entry carries the dword index of the next
    // ROW (row << wsh, 19 bits: rx.cpp caps the table at 2 MiB) so that a step's address is one add
    // and one shift-add; the two capture actions move up to bits 19 / 25.  Special entries keep
    // the layout of rx.hpp (the rare path decodes them).
    auto dev_entry = [&](uint32_t e) -> uint32_t {
        if (e & FT_SPECIAL) return e;
        return ((e & 0xFFF) << t.wsh) | (((e >> 12) & 63) << 19) | (((e >> 18) & 63) << 25);
    };
    std::vector<uint32_t> ftd(t.ft.size()), ft2d(t.ft2.size());
    for (size_t i = 0; i < t.ft.size(); i++) ftd[i] = dev_entry(t.ft[i]);
    for (size_t i = 0; i < t.ft2.size(); i++) ft2d[i] = dev_entry(t.ft2[i]);
    size_t o_rd = t.wide ? put(b, t.rdelta32_p) : put(b, t.rdelta_p), o_ft = put(b, ftd), o_f2 = put(b, ft2d), o_cls = put(b, cls), o_col = put(b, t.col);
    size_t hot_end = (b.size() + 15) & ~(size_t) 15;
    b.resize(hot_end);
    size_t o_ri = put(b, t.r_info), o_vm = put(b, t.vmask);
    size_t o_lo = put(b, t.list_off), o_le = put(b, t.list_ent), o_to = put(b, t.tag_off), o_td = put(b, t.tag_data);
    size_t o_xl = put(b, std::vector<uint8_t>(t.xl, t.xl + 512));
    int nwr = 0;
    const unsigned int (*wr)[2] = rx::unicode_word_ranges(&nwr);
    std::vector<uint32_t> wrv;
    if (t.word_variants) for (int i = 0; i < nwr; i++) { wrv.push_back(wr[i][0]); wrv.push_back(wr[i][1]); }
    size_t o_wr = put(b, wrv);
    const size_t o_cc = put(b, std::vector<uint64_t>{0});          // corner_count (dev.hpp DevCap)
    HIPOK(hipMalloc(&blob.dev, b.size()));
    HIPOK(hipMemcpy(blob.dev, b.data(), b.size(), hipMemcpyHostToDevice));
    const uint8_t *d = (const uint8_t *) blob.dev;
    memset(&out, 0, sizeof(out));
    out.rdelta = (const uint16_t *) (d + o_rd); out.ft = (const uint32_t *) (d + o_ft); out.ft2 = (const uint32_t *) (d + o_f2);
    out.cls = d + o_cls; out.col = d + o_col; out.xl = d + o_xl;
    out.r_info = d + o_ri; out.vmask = (const uint32_t *) (d + o_vm); out.list_off = (const uint32_t *) (d + o_lo);
    out.list_ent = (const uint32_t *) (d + o_le); out.tag_off = (const uint32_t *) (d + o_to); out.tag_data = d + o_td;
    out.ncls = t.ncls; out.nR = t.nR; out.r_init = t.r_init; out.VW = t.VW; out.nX = t.nX; out.NK = t.NK; out.NKp = t.NKp;
    out.kind_edge = t.kind_edge; out.ascii_only = t.ascii_only ? 1 : 0; out.cls_shift = t.cls_shift; out.fc_shift = t.fc_shift;
    out.wsh = t.wsh; out.col_eot = t.col_eot; out.wide = t.wide ? 1 : 0;
    out.word_variants = t.word_variants ? 1 : 0; out.wr = (const uint32_t *) (d + o_wr); out.nwr = t.word_variants ? nwr : 0;
    out.hot_base = d; out.hot_bytes = (uint32_t) hot_end;
    out.off_rdelta = (uint32_t) o_rd; out.off_ft = (uint32_t) o_ft; out.off_ft2 = (uint32_t) o_f2;
    out.off_cls = (uint32_t) o_cls; out.off_col = (uint32_t) o_col;
    out.stub = t.stub ? 1 : 0;
    out.corner_flags = 0; out.corner_count = (unsigned long long *) (d + o_cc);
    return true;
}

// what stands behind the ascii set of a pattern: its utf8 table set, or the NFA engine's tables (rx.hpp NfaSet -> dev.hpp DevNfa)
bool flbgpu::upload_utf8(const rx::Program &prog, TableBlob &blob, DevCap &out) {
    if (!prog.utf8_nfa) {
        if (!upload_cap(prog.utf8, blob, out)) return false;
        out.corner_flags = (int) prog.corner_flags;
        return true;
    }
    const rx::NfaSet &t = prog.nfa;
    std::vector<uint8_t> b;
    const size_t o_cb = put(b, t.cls_byte), o_ml = put(b, t.mb_lo), o_mc = put(b, t.mb_cls), o_am = put(b, t.amask), o_ck = put(b, t.ckind);
    const size_t o_pr = put(b, t.pred), o_ms = put(b, t.mstart), o_lo = put(b, t.list_off), o_le = put(b, t.list_ent);
    const size_t o_to = put(b, t.tag_off), o_td = put(b, t.tag_data);
    const size_t o_cc = put(b, std::vector<uint64_t>{0});
    HIPOK(hipMalloc(&blob.dev, b.size()));
    HIPOK(hipMemcpy(blob.dev, b.data(), b.size(), hipMemcpyHostToDevice));
    const uint8_t *d = (const uint8_t *) blob.dev;
    memset(&out, 0, sizeof(out));
    out.nfa_on = 1;
    out.corner_flags = (int) prog.corner_flags; out.corner_count = (unsigned long long *) (d + o_cc);
    DevNfa &n = out.nfa;
    n.cls_byte = (const uint16_t *) (d + o_cb); n.mb_lo = (const uint32_t *) (d + o_ml); n.mb_cls = (const uint16_t *) (d + o_mc);
    n.amask = (const uint32_t *) (d + o_am); n.ckind = d + o_ck; n.pred = (const uint32_t *) (d + o_pr); n.mstart = d + o_ms;
    n.list_off = (const uint32_t *) (d + o_lo); n.list_ent = (const uint32_t *) (d + o_le); n.tag_off = (const uint32_t *) (d + o_to);
    n.tag_data = d + o_td;
    n.P = t.P; n.VW = t.VW; n.NK = t.NK; n.kind_edge = t.kind_edge; n.ncls = t.ncls; n.nmb = (int) t.mb_lo.size();
    return true;
}

bool flbgpu::upload_fx(const rx::TableSet &t, int ncap, TableBlob &blob, DevFx &out, bool pair) {
    std::vector<uint8_t> b;
    if (!build_fx(t, ncap, b, out, pair) || !out.ok) { out.ok = 0; return true; }
    HIPOK(hipMalloc(&blob.dev, b.size()));
    HIPOK(hipMemcpy(blob.dev, b.data(), b.size(), hipMemcpyHostToDevice));
    out.base = (const uint8_t *) blob.dev;
    return true;
}

bool flbgpu::upload_dfa(const rx::TableSet &t, TableBlob &blob, DevDfa &out) {
    std::vector<uint8_t> b;
    std::vector<uint8_t> cls(t.cls, t.cls + 256);
    size_t o_cls = put(b, cls), o_dd = put(b, t.ddelta), o_df = put(b, t.d_final);
    HIPOK(hipMalloc(&blob.dev, b.size()));
    HIPOK(hipMemcpy(blob.dev, b.data(), b.size(), hipMemcpyHostToDevice));
    const uint8_t *d = (const uint8_t *) blob.dev;
    memset(&out, 0, sizeof(out));
    out.cls = d + o_cls; out.ddelta = (const uint16_t *) (d + o_dd); out.d_final = d + o_df;
    out.ncls = t.ncls; out.nD = t.nD; out.d_init = t.d_init;
    return true;
}

// The fixed-layout plan once more, COMPILED for the single-pass kernel (tile_kernels.inc time_fast_compiled): instead of an
// interpreter loop over the ops -- a scalar load, a switch and a register select per op and per byte: a sixth of the kernel's
// iteration on the apache format -- the literals of the whole text are ONE masked compare per dword, the digit positions ONE SWAR
// test per dword, and every field sits at an offset the kernel reads from this block.  32 dwords behind the fx3 tables in the LDS:
//   [0..7] literal mask  [8..15] literal bytes  [16..23] 0x80 at every digit position
//   [24] offsets mday | hour << 8 | min << 16 | sec << 24        (0xFF: no such field)
//   [25] offsets mon2 | year << 8 | mon3 << 16 | tz << 24
//   [26] upper limits, [27] lower limits of [24]'s fields       [28] mon2 upper | lower << 8 | spaces << 16 | length << 24
//   [29] offsets of up to four whitespace positions              [31] 1: usable
static void compile_time_plan(const TimePlan &pl, uint32_t out[32]) {
    memset(out, 0, 32 * sizeof(uint32_t));
    if (!pl.ok || pl.len > 32) return;
    uint8_t *lm = (uint8_t *) out, *lv = (uint8_t *) (out + 8), *dm = (uint8_t *) (out + 16);
    uint8_t off[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};     // mday hour min sec mon2 year mon3 tz
    uint8_t hi[5] = {0, 0, 0, 0, 0}, lo[5] = {0, 0, 0, 0, 0}, sp[4] = {0, 0, 0, 0};
    int nsp = 0;
    auto take = [&](int f, uint8_t o) -> bool { if (off[f] != 0xFF) return false; off[f] = o; return true; };
    for (int k = 0; k < pl.nops; k++) {
        const TimeOp &op = pl.ops[k];
        switch (op.kind) {
        case TP_LIT: lm[op.off] = 0xFF; lv[op.off] = op.a; break;
        case TP_SPACE: if (nsp >= 4) return; sp[nsp++] = op.off; break;
        case TP_NUM2: {
            const int f = op.a == TPF_MDAY ? 0 : op.a == TPF_HOUR ? 1 : op.a == TPF_MIN ? 2 : op.a == TPF_SEC ? 3 : 4;
            if (!take(f, op.off)) return;
            hi[f] = op.b; lo[f] = pl.lo[k];
            dm[op.off] = dm[op.off + 1] = 0x80;
            break;
        }
        case TP_YEAR4: if (!take(5, op.off)) return; for (int q = 0; q < 4; q++) dm[op.off + q] = 0x80; break;
        case TP_MON3: if (!take(6, op.off)) return; break;
        case TP_TZ5: if (!take(7, op.off)) return; for (int q = 1; q < 5; q++) dm[op.off + q] = 0x80; break;
        default: return;
        }
    }
    if (off[4] != 0xFF && off[6] != 0xFF) return;                          // two months
    out[24] = off[0] | (off[1] << 8) | (off[2] << 16) | ((uint32_t) off[3] << 24);
    out[25] = off[4] | (off[5] << 8) | (off[6] << 16) | ((uint32_t) off[7] << 24);
    out[26] = hi[0] | (hi[1] << 8) | (hi[2] << 16) | ((uint32_t) hi[3] << 24);
    out[27] = lo[0] | (lo[1] << 8) | (lo[2] << 16) | ((uint32_t) lo[3] << 24);
    out[28] = hi[4] | (lo[4] << 8) | ((uint32_t) nsp << 16) | ((uint32_t) pl.len << 24);
    out[29] = sp[0] | (sp[1] << 8) | (sp[2] << 16) | ((uint32_t) sp[3] << 24);
    out[31] = 1;
}

// ------------------------------------------------------------------------------------------ parser
// what tzif_parse_data keeps of a zone's file (tzif.hpp)
struct TzTable { std::vector<int64_t> trans; std::vector<uint8_t> ttype; std::vector<int32_t> gmtoff; int default_type = 0; };

struct flbgpu_parser {
    std::string name;
    rx::Program prog;
    TableBlob blob_ascii, blob_utf8, blob_fx, blob_fx2, blob_fx2b;
    DevFx fx2b;                       // the four-port pair tables behind the three-port ones (dev.fx2): a filter falls back to them (flbgpu_filter::fx5_off)
    DevParser dev;                 // host copy (device pointers inside)
    flbgpu_filter *self_filter = nullptr;   // lazily created for flbgpu_parser_do
    DevDecoders decs;              // Decode_Field / Decode_Field_As (flbgpu_parser_add_decoder), uploaded when a filter takes the parser
    void *d_decs = nullptr;
    rx::BtProgram *bt = nullptr;   // a HOST parser: the Regex is not a regular expression, the backtracking matcher answers on the host (rxbt.inc)
    TzTable zone;                  // Time_Zone (flbgpu_parser_set_time_zone), uploaded when a filter takes the parser
    void *d_zone = nullptr;
    flbgpu_parser() { memset(&decs, 0, sizeof(decs)); }
    ~flbgpu_parser() { if (d_decs) (void) hipFree(d_decs); if (d_zone) (void) hipFree(d_zone); if (bt) rx::bt_free(bt); }
};

// ---- Time_Zone / Time_System_Timezone (src/flb_parser.c:805-1049 flb_parser_create_with_time_zone)
// The zone's TZif file read the way tzif_load / tzif_parse_data do (src/flb_parser.c:452-537, 359-450): $TZDIR or
// /usr/share/zoneinfo, the 64-bit block of a version 2+ file, else the 32-bit one; the footer string is not read.
static uint32_t tz_be32(const unsigned char *b) { return ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | b[3]; }
static bool tz_block(const unsigned char *buf, size_t size, int time_size, TzTable &z) {
    if (size < 44) return false;
    const uint32_t timecnt = tz_be32(buf + 32), typecnt = tz_be32(buf + 36);
    if (typecnt == 0 || timecnt > 0x7FFFFFFFu || typecnt > 0x7FFFFFFFu) return false;
    size_t off = 44;
    if (off + (size_t) timecnt * time_size + timecnt + (size_t) typecnt * 6 > size) return false;
    z.trans.resize(timecnt); z.ttype.resize(timecnt); z.gmtoff.resize(typecnt);
    for (uint32_t i = 0; i < timecnt; i++, off += time_size) {
        if (time_size == 8) z.trans[i] = (int64_t) (((uint64_t) tz_be32(buf + off) << 32) | tz_be32(buf + off + 4));
        else z.trans[i] = (int32_t) tz_be32(buf + off);
    }
    for (uint32_t i = 0; i < timecnt; i++) { z.ttype[i] = buf[off + i]; if (z.ttype[i] >= typecnt) return false; }
    off += timecnt;
    z.default_type = 0;
    bool have_std = false;
    for (uint32_t i = 0; i < typecnt; i++, off += 6) {
        z.gmtoff[i] = (int32_t) tz_be32(buf + off);
        if (!have_std && buf[off + 4] == 0) { z.default_type = (int) i; have_std = true; }
    }
    // the lookups index the types with a byte and walk all of them per record
    return typecnt <= 256;
}
static bool tz_load(const char *iana_zone, TzTable &z, std::string &why) {
    const char *tzdir = getenv("TZDIR");
    if (!tzdir || !tzdir[0]) tzdir = "/usr/share/zoneinfo";
    const std::string path = std::string(tzdir) + "/" + iana_zone;
    if (path.size() >= 4096) { why = "path too long"; return false; }
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { why = "invalid or unavailable time_zone (no " + path + ")"; return false; }
    std::vector<unsigned char> buf;
    unsigned char chunk[4096];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0) { buf.insert(buf.end(), chunk, chunk + got); if (buf.size() > (1u << 24)) break; }
    fclose(fp);
    why = "could not load time_zone";
    if (buf.size() < 44 || memcmp(buf.data(), "TZif", 4) != 0) return false;
    const unsigned char version = buf[4];
    if (version == '2' || version == '3' || version == '4') {
        const uint32_t isutcnt = tz_be32(&buf[20]), isstdcnt = tz_be32(&buf[24]), leapcnt = tz_be32(&buf[28]), timecnt = tz_be32(&buf[32]),
                       typecnt = tz_be32(&buf[36]), charcnt = tz_be32(&buf[40]);
        const size_t block = (size_t) timecnt * 4 + timecnt + (size_t) typecnt * 6 + charcnt + (size_t) leapcnt * 8 + isstdcnt + isutcnt;
        if (44 + block + 44 > buf.size() || memcmp(&buf[44 + block], "TZif", 4) != 0) return false;
        return tz_block(&buf[44 + block], buf.size() - 44 - block, 8, z);
    }
    return tz_block(buf.data(), buf.size(), 4, z);
}

// Time_Zone <IANA name> on a parser with a Time_Format (src/flb_parser.c:988-1022): refused together with Time_Offset or
// Time_System_Timezone, refused when the zone's file is not there.  The records of such a parser take the interpreter
// for the time text (no compiled plan): the seconds come from tz::tm2time over the table (tzif.hpp).
extern "C" int flbgpu_parser_set_time_zone(flbgpu_parser *p, const char *iana_zone) {
    if (!p) { set_err("parser time_zone: missing argument"); return -1; }
    if (!iana_zone || !iana_zone[0]) return 0;
    if (!p->dev.has_time) { set_err("parser '%s': time_zone requires time_format", p->name.c_str()); return -1; }
    if (p->dev.zone_mode == 2) { set_err("parser '%s': time_zone cannot be combined with time_system_timezone", p->name.c_str()); return -1; }
    if (p->dev.time_offset_given) { set_err("parser '%s': time_zone cannot be combined with time_offset", p->name.c_str()); return -1; }
    // validate_time_zone (:606-638): the name has to be one of the reference's built-in zone index before its file is looked for
    static const char *const known[] = {
#include "tz_names.inc"
    };
    size_t lo = 0, hi = sizeof(known) / sizeof(known[0]);
    bool listed = false;
    while (lo < hi && !listed) {
        const size_t mid = (lo + hi) / 2;
        const int c = strcmp(iana_zone, known[mid]);
        if (c == 0) listed = true;
        else if (c < 0) hi = mid;
        else lo = mid + 1;
    }
    if (!listed) { set_err("parser '%s': invalid or unavailable time_zone '%s' (not a name of the zone index)", p->name.c_str(), iana_zone); return -1; }
    TzTable z;
    std::string why;
    if (!tz_load(iana_zone, z, why)) { set_err("parser '%s': %s '%s'", p->name.c_str(), why.c_str(), iana_zone); return -1; }
    p->zone = z;
    p->dev.zone_mode = 1;
    p->dev.plan.ok = 0;
    if (p->d_zone) { (void) hipFree(p->d_zone); p->d_zone = nullptr; }
    if (p->self_filter) { delete p->self_filter; p->self_filter = nullptr; }
    return 0;
}

// Time_System_Timezone On (src/flb_parser.c:986, include/fluent-bit/flb_parser.h:80-94: mktime() of the fields with
// tm_isdst = -1, whatever zone the text itself named).  Taken when the process's zone is UTC without rules -- mktime is
// timegm then; any other process zone is refused (mktime's choice inside gaps and overlaps is the C library's).
extern "C" int flbgpu_parser_set_system_timezone(flbgpu_parser *p, int on) {
    if (!p) { set_err("parser time_system_timezone: missing argument"); return -1; }
    if (!on) return 0;
    if (p->dev.zone_mode == 1) { set_err("parser '%s': time_zone cannot be combined with time_system_timezone", p->name.c_str()); return -1; }
    tzset();
    if (timezone != 0 || daylight != 0) { set_err("parser '%s': Time_System_Timezone with a process zone that is not UTC is not supported", p->name.c_str()); return -1; }
    p->dev.zone_mode = 2;
    p->dev.time_offset = 0;                                 // (src/flb_parser.c:1027: the fixed offset is not applied)
    p->dev.plan.ok = 0;
    if (p->self_filter) { delete p->self_filter; p->self_filter = nullptr; }
    return 0;
}

// CPU test hook: tzif_tm2time (src/flb_parser.c:560-590) of `local_epoch` = timegm() of the parsed fields, over the
// zone's file -- the same tz::tm2time text the kernels run.  Returns 0, -1 when the zone does not load.
extern "C" int flbgpu_tz_tm2time(const char *iana_zone, int64_t local_epoch, int64_t *out) {
    TzTable z;
    std::string why;
    if (!iana_zone || !out || !tz_load(iana_zone, z, why)) { set_err("time_zone: %s '%s'", why.c_str(), iana_zone ? iana_zone : ""); return -1; }
    *out = tz::tm2time(z.trans.data(), z.ttype.data(), z.gmtoff.data(), (int) z.trans.size(), (int) z.gmtoff.size(), z.default_type, local_epoch);
    return 0;
}

// One rule of a parser's decoder list: "Decode_Field[_As] <backend> <key> [try_next|do_next]" (conf/parsers.conf, parsed by
// src/flb_parser_decoder.c:593-776 flb_parser_decoder_list_create): rules of one key are kept together in configuration order,
// a key with a Decode_Field rule appends the decoded object's pairs to the record (add_extra_keys :701-703).
extern "C" int flbgpu_parser_add_decoder(flbgpu_parser *p, int as, const char *backend, const char *key, const char *action) {
    if (!p || !backend || !key) { set_err("parser decoder: missing argument"); return -1; }
    int bk;
    if (!strcasecmp(backend, "json")) bk = 0;
    else if (!strcasecmp(backend, "escaped")) bk = 1;
    else if (!strcasecmp(backend, "escaped_utf8")) bk = 2;
    else if (!strcasecmp(backend, "mysql_quoted")) bk = 3;
    else { set_err("parser '%s': field decoder '%s' not found", p->name.c_str(), backend); return -1; }
    int act = DEC_A_NONE;
    if (action && *action) {
        if (!strcasecmp(action, "try_next")) act = DEC_A_TRY_NEXT;
        else if (!strcasecmp(action, "do_next")) act = DEC_A_DO_NEXT;
        else { set_err("parser '%s': unknown decoder action '%s'", p->name.c_str(), action); return -1; }
    }
    const size_t kl = strlen(key);
    if (kl == 0 || kl > sizeof(p->decs.d[0].key)) { set_err("parser '%s': decoder key too long", p->name.c_str()); return -1; }
    DevDecoder *d = nullptr;
    for (uint32_t i = 0; i < p->decs.n; i++) if (p->decs.d[i].key_len == kl && !memcmp(p->decs.d[i].key, key, kl)) d = &p->decs.d[i];
    if (!d) {
        if (p->decs.n >= MAX_DEC_KEYS) { set_err("parser '%s': more than %d keys with decoders", p->name.c_str(), MAX_DEC_KEYS); return -1; }
        d = &p->decs.d[p->decs.n++];
        memcpy(d->key, key, kl);
        d->key_len = (uint32_t) kl;
    }
    if (d->nrules >= MAX_DEC_RULES) { set_err("parser '%s': more than %d decoder rules for one key", p->name.c_str(), MAX_DEC_RULES); return -1; }
    DevDecRule &r = d->rules[d->nrules++];
    r.type = as ? DEC_T_AS : DEC_T_DEFAULT; r.backend = (uint8_t) bk; r.action = (uint8_t) act; r.pad = 0;
    if (!as) d->add_extra_keys = 1;
    if (p->d_decs) { (void) hipFree(p->d_decs); p->d_decs = nullptr; }       // (uploaded again by the next filter that takes the parser)
    if (p->self_filter) { delete p->self_filter; p->self_filter = nullptr; }
    return 0;
}

// src/flb_parser.c:1806-1870 flb_parser_tzone_offset
static int tzone_offset(const char *str, int len, int *tmdiff) {
    const char *p = str;
    *tmdiff = 0;
    if (*p == 'Z') return 0;
    if (*p != '+' && *p != '-') return -1;
    if (len < 4) return -1;
    int neg = (*p++ == '-');
    const char *end = str + len;
    long hour = ((p[0] - '0') * 10) + (p[1] - '0'), min;
    if (end - p == 5 && p[2] == ':') min = ((p[3] - '0') * 10) + (p[4] - '0');
    else min = ((p[2] - '0') * 10) + (p[3] - '0');
    if (hour < 0 || hour > 59 || min < 0 || min > 59) return -1;
    *tmdiff = (int) (hour * 3600 + min * 60);
    if (neg) *tmdiff = -*tmdiff;
    return 0;
}

// expands the composite strptime directives (src/flb_strptime.c:300-355) and checks that only
// directives implemented by the device interpreter remain
static bool expand_time_fmt(const char *fmt, std::string &out, std::string &why) {
    for (const char *p = fmt; *p; p++) {
        if (*p != '%') { out += *p; continue; }
        p++;
        while (*p == 'E' || *p == 'O') p++;
        switch (*p) {
        // "%\x01" marks where the reference's recursive call for the composite returns (kdev.inc DK_FINAL)
        case 'T': case 'X': out += "%H:%M:%S%\x01"; break;
        case 'D': case 'x': out += "%m/%d/%y%\x01"; break;
        case 'F': out += "%Y-%m-%d%\x01"; break;
        case 'R': out += "%H:%M%\x01"; break;
        case 'r': out += "%I:%M:%S %p%\x01"; break;
        case 'c': out += "%a %b %e %H:%M:%S %Y%\x01"; break;
        case '\0': why = "dangling % in time format"; return false;
        default:
            if (!strchr("%AaBbhCedkHlIjMmpSsUWVwugGYyzZnt", *p)) { why = std::string("unsupported time directive %") + *p; return false; }
            out += '%'; out += *p;
        }
    }
    return true;
}

// Fixed-layout plan of an expanded Time_Format (dev.hpp TimePlan).  Conservative: every directive must
// have one width (%d %m %H %M %S two digits, %Y four, %b/%B/%h a three-letter abbreviation, %z
// sign + hhmm), whitespace in the format must be single and followed by something that is not
// whitespace, and year, month and day must all be there.  The limits are src/flb_strptime.c's.
static void build_time_plan(const std::string &fmt, bool has_frac, TimePlan &pl) {
    memset(&pl, 0, sizeof(pl));
    if (has_frac || fmt.empty()) return;
    int off = 0, n = 0;
    bool year = false, mon = false, mday = false, prev_space = false;
    auto push = [&](int kind, int width, int a, int b, int lo) -> bool {
        if (n >= 32 || off + width > 32) return false;
        pl.ops[n].kind = (uint8_t) kind; pl.ops[n].off = (uint8_t) off; pl.ops[n].a = (uint8_t) a; pl.ops[n].b = (uint8_t) b;
        pl.lo[n] = (uint8_t) lo;
        n++; off += width;
        return true;
    };
    for (size_t i = 0; i < fmt.size(); i++) {
        const unsigned char c = (unsigned char) fmt[i];
        if (c == ' ' || (c >= 9 && c <= 13)) {
            if (prev_space || i + 1 == fmt.size()) return;
            if (!push(TP_SPACE, 1, 0, 0, 0)) return;
            prev_space = true;
            continue;
        }
        prev_space = false;
        if (c != '%') {
            if (c == 0 || !push(TP_LIT, 1, c, 0, 0)) return;
            continue;
        }
        if (++i >= fmt.size()) return;
        if (fmt[i] == '\x01') continue;           // end of a composite directive: nothing to read
        bool ok;
        switch (fmt[i]) {
        case 'd': ok = push(TP_NUM2, 2, TPF_MDAY, 31, 1); mday = true; break;
        case 'H': ok = push(TP_NUM2, 2, TPF_HOUR, 23, 0); break;
        case 'M': ok = push(TP_NUM2, 2, TPF_MIN, 59, 0); break;
        case 'S': ok = push(TP_NUM2, 2, TPF_SEC, 60, 0); break;
        case 'm': ok = push(TP_NUM2, 2, TPF_MON1, 12, 1); mon = true; break;
        case 'Y': ok = push(TP_YEAR4, 4, 0, 0, 0); year = true; break;
        case 'b': case 'B': case 'h': ok = push(TP_MON3, 3, 0, 0, 0); mon = true; break;
        case 'z': ok = push(TP_TZ5, 5, 0, 0, 0); break;
        default: return;
        }
        if (!ok) return;
    }
    if (!year || !mon || !mday) return;
    pl.len = off;
    pl.nops = n;
    pl.ok = 1;
}

// Format regex (regex != NULL) or Format json (is_json)
static flbgpu_parser *parser_create_impl(bool is_json, const char *name, const char *regex, int skip_empty,
                                         const char *time_fmt, const char *time_key, const char *time_offset,
                                         int time_keep, int time_strict, const char *types) {
    if (!is_json && !regex) { set_err("parser '%s': missing regex", name ? name : ""); return nullptr; }
    if (is_json) regex = "";
    // (a %Z DIRECTIVE: "%%Z" is a literal percent sign and a Z -- ADVICE r5)
    auto has_zone_directive = [](const char *f) { for (; *f; f++) if (*f == '%') { f++; if (*f == 'Z') return true; if (!*f) break; } return false; };
    // %Z's last resort for a zone text that is in neither of flb_strptime's tables is the PROCESS's zone -- tzname[] and -timezone
    // (src/flb_strptime.c:611-650).  Rounds 4 and 5 restated that for a process without a zone and refused %Z in any other; round 6
    // reads the two names and the offset here, once, like the reference's tzset() does, and the device compares with THEM.
    char zn[2][16] = {{0}, {0}};
    int zn_len[2] = {0, 0}, zn_off = 0, zn_set = 0;
    if (time_fmt && has_zone_directive(time_fmt)) {
        tzset();
        for (int i = 0; i < 2; i++) {
            const char *nm = tzname[i] ? tzname[i] : "";
            const size_t L = strlen(nm);
            if (L > 15) { set_err("parser '%s': Time_Format with %%Z in a process whose zone name '%s' is longer than 15 bytes is not supported", name ? name : "", nm); return nullptr; }
            memcpy(zn[i], nm, L); zn_len[i] = (int) L;
        }
        zn_off = (int) -timezone; zn_set = 1;
    }
    auto *p = new flbgpu_parser();
    p->name = name ? name : "";
    DevParser &d = p->dev;
    memset(&d, 0, sizeof(d));
    d.tz_names = zn_set; d.tzn_gmtoff = zn_off; d.tzn_len[0] = zn_len[0]; d.tzn_len[1] = zn_len[1];
    memcpy(d.tzn, zn, sizeof(zn));
    if (!is_json) {
        const char *s, *e;
        unsigned opts;
        rx::split_flb_pattern(regex, &s, &e, &opts);
        std::string err;
        if (!rx::compile(s, (size_t) (e - s), opts, true, p->prog, err)) {
            // not a regular expression (look-around, back-references, ...): a HOST parser -- the device does everything but the
            // capture search, which the backtracking matcher runs on the located values (host_parser_rx below)
            std::string e2;
            if (p->prog.nonregular && !getenv("FLBGPU_NO_HOST_RULES")) p->bt = rx::bt_compile(s, (size_t) (e - s), opts, e2);
            if (p->bt && rx::bt_ngroups(p->bt) > 31) { e2 = "more than 31 capture groups"; rx::bt_free(p->bt); p->bt = nullptr; }
            if (!p->bt) {
                set_err("parser '%s': cannot compile regex for the GPU path: %s%s%s", p->name.c_str(), err.c_str(), e2.empty() ? "" : "; on the host: ", e2.c_str());
                delete p;
                return nullptr;
            }
            p->prog = rx::Program();
            p->prog.ngroups = rx::bt_ngroups(p->bt);
            p->prog.names = rx::bt_names(p->bt);
            p->prog.name_groups = rx::bt_name_groups(p->bt);
            p->prog.slot2cap.assign(2 * (size_t) (p->prog.ngroups + 1), 0xFF);
            int fi = 0;
            for (size_t i = 0; i < p->prog.names.size(); i++)
                for (int g : p->prog.name_groups[i]) {
                    if (fi < 120) { p->prog.slot2cap[2 * g] = (uint8_t) (2 * fi); p->prog.slot2cap[2 * g + 1] = (uint8_t) (2 * fi + 1); }
                    fi++;
                }
        }
        else if (!upload_cap(p->prog.ascii, p->blob_ascii, d.ascii) || !upload_utf8(p->prog, p->blob_utf8, d.utf8)) { delete p; return nullptr; }
    }
    d.is_json = is_json ? 1 : 0;
    d.ngroups = p->prog.ngroups;
    d.nregs_minus1 = p->prog.ngroups;
    d.skip_empty = skip_empty;
    d.time_keep = time_keep;
    d.time_strict = time_strict;
    memset(d.slot2cap, 0xFF, sizeof(d.slot2cap));
    for (size_t i = 0; i < p->prog.slot2cap.size() && i < sizeof(d.slot2cap); i++) d.slot2cap[i] = p->prog.slot2cap[i];
    // time format analysis: src/flb_parser.c:906-1040
    if (time_fmt && time_fmt[0]) {
        std::string tf = time_fmt;
        bool with_year = tf.find("%Y") != std::string::npos || tf.find("%y") != std::string::npos || tf.find("%s") != std::string::npos;
        d.yearless = with_year ? 0 : 1;       // "%Y " + fmt over "<current year> " + text, see DevParser::yearless
        d.has_time = 1;
        d.time_with_tz = (tf.find("%z") != std::string::npos || tf.find("%Z") != std::string::npos ||
                          tf.find("%SZ") != std::string::npos || tf.find("%S.%LZ") != std::string::npos) ? 1 : 0;
        std::string f1 = tf, f2;
        size_t lpos = tf.find("%L");
        if (lpos != std::string::npos) { f1 = tf.substr(0, lpos); f2 = tf.substr(lpos + 2); d.has_frac = 1; }
        std::string x1, x2, why;
        if (!expand_time_fmt(f1.c_str(), x1, why) || !expand_time_fmt(f2.c_str(), x2, why)) {
            set_err("parser '%s': %s", p->name.c_str(), why.c_str());
            delete p;
            return nullptr;
        }
        if (x1.size() >= MAX_TIMEFMT || x2.size() >= MAX_TIMEFMT) { set_err("parser '%s': Time_Format too long", p->name.c_str()); delete p; return nullptr; }
        strcpy(d.fmt1, x1.c_str());
        strcpy(d.fmt2, x2.c_str());
        build_time_plan(x1, d.has_frac != 0, d.plan);
        if (time_offset && time_offset[0]) {
            int diff = 0;
            if (tzone_offset(time_offset, (int) strlen(time_offset), &diff) == -1) { set_err("parser '%s': invalid Time_Offset", p->name.c_str()); delete p; return nullptr; }
            d.time_offset = diff;
            d.time_offset_given = 1;
        }
    }
    // Types: src/flb_parser.c:1130-1182
    std::vector<std::pair<std::string, int>> tys;
    if (types && types[0]) {
        const char *q = types;
        while (*q) {
            while (*q == ' ') q++;
            if (!*q) break;
            const char *sp = strchr(q, ' ');
            if (!sp) sp = q + strlen(q);
            const char *colon = (const char *) memchr(q, ':', sp - q);
            if (colon) {
                std::string key(q, colon - q), ty(colon + 1, sp - colon - 1);
                int t = TY_STRING;
                if (!strcasecmp(ty.c_str(), "integer")) t = TY_INT;
                else if (!strcasecmp(ty.c_str(), "bool")) t = TY_BOOL;
                else if (!strcasecmp(ty.c_str(), "float")) t = TY_FLOAT;
                else if (!strcasecmp(ty.c_str(), "hex")) t = TY_HEX;
                tys.emplace_back(key, t);
            }
            q = *sp ? sp + 1 : sp;
        }
    }
    // named fields in onig_foreach_name order
    const char *tkey = (time_key && time_key[0]) ? time_key : "time";
    if (strlen(tkey) >= sizeof(d.tkey)) { set_err("parser '%s': Time_Key too long", p->name.c_str()); delete p; return nullptr; }
    d.tkey_len = (int) strlen(tkey);
    memcpy(d.tkey, tkey, (size_t) d.tkey_len);
    size_t noff = 0;
    for (size_t i = 0; i < p->prog.names.size(); i++) {
        for (int g : p->prog.name_groups[i]) {
            if (d.nfields >= MAX_NAMES || noff + p->prog.names[i].size() > sizeof(d.names)) { set_err("parser '%s': too many named groups", p->name.c_str()); delete p; return nullptr; }
            int f = d.nfields++;
            d.field_group[f] = g;
            d.field_name_off[f] = (int) noff;
            d.field_name_len[f] = (int) p->prog.names[i].size();
            memcpy(d.names + noff, p->prog.names[i].data(), p->prog.names[i].size());
            noff += p->prog.names[i].size();
            d.field_is_time[f] = (d.has_time && p->prog.names[i] == tkey) ? 1 : 0;
            d.field_type[f] = TY_NONE;
            for (auto &ty : tys) if (ty.first == p->prog.names[i]) { d.field_type[f] = ty.second; break; }   // first match wins
        }
    }
    {
        // packed "str header + name" per field: the emit kernel stores them a dword at a time
        size_t w = 0;
        memset(d.keywords, 0, sizeof(d.keywords));
        for (int f = 0; f < d.nfields; f++) {
            uint8_t tmp[8 + 1024];
            size_t nb = 0;
            const uint32_t nl = (uint32_t) d.field_name_len[f];
            if (nl < 32) tmp[nb++] = (uint8_t) (0xa0 | nl);
            else if (nl < 256) { tmp[nb++] = 0xd9; tmp[nb++] = (uint8_t) nl; }
            else { tmp[nb++] = 0xda; tmp[nb++] = (uint8_t) (nl >> 8); tmp[nb++] = (uint8_t) nl; }
            memcpy(tmp + nb, d.names + d.field_name_off[f], nl);
            nb += nl;
            d.kw_off[f] = (int) w;
            d.kw_bytes[f] = (int) nb;
            if (w + (nb + 3) / 4 > sizeof(d.keywords) / 4) { set_err("parser '%s': field names too long", p->name.c_str()); delete p; return nullptr; }
            memcpy((uint8_t *) (d.keywords + w), tmp, nb);
            w += (nb + 3) / 4;
        }
    }
    d.fwd_first = (regex[0] == '^' || (regex[0] == '\\' && regex[1] == 'A') || (regex[0] == '/' && (regex[1] == '^' || (regex[1] == '\\' && regex[2] == 'A')))) ? 1 : 0;
    if (getenv("FLBGPU_NO_FWD_FIRST")) d.fwd_first = 0;
    d.time_field = -1;
    d.plain_types = 1;
    {
        int ntime = 0;
        for (int f = 0; f < d.nfields; f++) {
            if (d.field_is_time[f]) { ntime++; d.time_field = f; }
            if (d.field_type[f] != TY_NONE && d.field_type[f] != TY_STRING) d.plain_types = 0;
        }
        if (ntime != 1) d.time_field = -1;
    }
    // compact forward tables of the single-pass tile kernel (start-anchored patterns: the forward walk needs no reverse pass)
    if (!is_json && !p->bt && d.fwd_first && d.nregs_minus1 > 0 && d.nfields > 0 && !getenv("FLBGPU_NO_TILE") && !p->prog.ascii_stub) {
        if (!upload_fx(p->prog.ascii, 2 * d.nfields, p->blob_fx, d.fx)) { delete p; return nullptr; }
        // the same tables without special entries (fx.cpp build_fx3: 8-byte cells, two capture writes per step): what k_parser_reg<.., FX3> walks
        if (d.fx.ok) {
            std::vector<uint8_t> b3;
            // the two-position form (fx4) when it fits the LDS beside the capture columns, else one position per read (fx3)
            // (round 5) ... with THREE capture-write ports per cell (fx5) when no cell of the pattern needs two writes at one position,
            // else with four (fx4).  FLBGPU_FX=3 / 4 selects the older forms.
            const char *fxe = getenv("FLBGPU_FX");
            const bool want4 = !(fxe && fxe[0] == '3');
            if (want4 && !(fxe && fxe[0] == '4')) {
                if (!build_fx3(p->prog.ascii, 2 * d.nfields, b3, d.fx2, 2)) { delete p; return nullptr; }
                if (!d.fx2.ok || b3.size() > 48 * 1024) { b3.clear(); memset(&d.fx2, 0, sizeof(d.fx2)); }
            }
            if (want4 && !d.fx2.ok && !build_fx3(p->prog.ascii, 2 * d.nfields, b3, d.fx2, 1)) { delete p; return nullptr; }
            if (!want4 || !d.fx2.ok || b3.size() > 48 * 1024) {
                b3.clear();
                if (!build_fx3(p->prog.ascii, 2 * d.nfields, b3, d.fx2, 0)) { delete p; return nullptr; }
            }
            if (d.fx2.ok) {
                // the compiled time plan rides behind the tables (the last 128 bytes of what the kernel stages into the LDS)
                uint32_t ctp[32];
                compile_time_plan(d.plan, ctp);
                const size_t at = b3.size();
                b3.resize(at + sizeof(ctp));
                memcpy(b3.data() + at, ctp, sizeof(ctp));
                d.fx2.bytes = (uint32_t) b3.size();
                if (hipMalloc(&p->blob_fx2.dev, b3.size()) != hipSuccess || hipMemcpy(p->blob_fx2.dev, b3.data(), b3.size(), hipMemcpyHostToDevice) != hipSuccess) {
                    set_err("parser '%s': upload failed", p->name.c_str()); delete p; return nullptr;
                }
                d.fx2.base = (const uint8_t *) p->blob_fx2.dev;
                memset(&p->fx2b, 0, sizeof(p->fx2b));
                if (d.fx2.pair_bias == 2) {
                    // fx5 sends a record with two capture writes at one even position (an EMPTY field, `""`) to the generic kernel.  Data full
                    // of such records is better served by the four-port cells: both forms are uploaded, a filter switches when it sees
                    // the fast walk hand on more than 1 row in 64 (parser_size_pass, note_fx5)
                    std::vector<uint8_t> b4;
                    if (build_fx3(p->prog.ascii, 2 * d.nfields, b4, p->fx2b, 1) && p->fx2b.ok && b4.size() <= 48 * 1024) {
                        const size_t at4 = b4.size();
                        b4.resize(at4 + sizeof(ctp));
                        memcpy(b4.data() + at4, ctp, sizeof(ctp));
                        p->fx2b.bytes = (uint32_t) b4.size();
                        if (hipMalloc(&p->blob_fx2b.dev, b4.size()) != hipSuccess || hipMemcpy(p->blob_fx2b.dev, b4.data(), b4.size(), hipMemcpyHostToDevice) != hipSuccess) {
                            set_err("parser '%s': upload failed", p->name.c_str()); delete p; return nullptr;
                        }
                        p->fx2b.base = (const uint8_t *) p->blob_fx2b.dev;
                    }
                    else p->fx2b.ok = 0;
                }
            }
        }
    }
    return p;
}

extern "C" flbgpu_parser *flbgpu_parser_create(const char *name, const char *regex, int skip_empty,
                                               const char *time_fmt, const char *time_key, const char *time_offset,
                                               int time_keep, int time_strict, const char *types) {
    return parser_create_impl(false, name, regex, skip_empty, time_fmt, time_key, time_offset, time_keep, time_strict, types);
}

extern "C" flbgpu_parser *flbgpu_parser_create_json(const char *name, const char *time_fmt, const char *time_key,
                                                    const char *time_offset, int time_keep, int time_strict) {
    // Types are ignored by the json format (tests/internal/parser_json.c: test_types_is_not_supported);
    // Decode_Field has no place in this ABI
    return parser_create_impl(true, name, nullptr, 1, time_fmt, time_key, time_offset, time_keep, time_strict, nullptr);
}

// Format logfmt / ltsv (src/flb_parser_logfmt.c, src/flb_parser_ltsv.c): same plumbing as Format json
// (no regex; the value is walked pair by pair on the device), other scanner (pkv_dev.inc)
extern "C" flbgpu_parser *flbgpu_parser_create_kv(const char *name, const char *format, const char *time_fmt, const char *time_key,
                                                  const char *time_offset, int time_keep, int time_strict, int logfmt_no_bare_keys,
                                                  const char *types) {
    int kv = 0;
    if (format && !strcasecmp(format, "logfmt")) kv = 1;
    else if (format && !strcasecmp(format, "ltsv")) kv = 2;
    if (!kv) { set_err("parser '%s': format '%s' is not logfmt or ltsv", name ? name : "", format ? format : ""); return nullptr; }
    flbgpu_parser *p = parser_create_impl(true, name, nullptr, 1, time_fmt, time_key, time_offset, time_keep, time_strict, nullptr);
    if (!p) return nullptr;
    // Types: "key:type key:type" (src/flb_parser.c:1130-1182); the pair writer looks the cast up by key name
    if (types && types[0]) {
        const char *q = types;
        size_t noff = 0;
        while (*q) {
            while (*q == ' ') q++;
            if (!*q) break;
            const char *sp = strchr(q, ' ');
            if (!sp) sp = q + strlen(q);
            const char *colon = (const char *) memchr(q, ':', sp - q);
            if (colon) {
                const size_t kl = (size_t) (colon - q);
                std::string ty(colon + 1, sp - colon - 1);
                int t = TY_STRING;
                if (!strcasecmp(ty.c_str(), "integer")) t = TY_INT;
                else if (!strcasecmp(ty.c_str(), "bool")) t = TY_BOOL;
                else if (!strcasecmp(ty.c_str(), "float")) t = TY_FLOAT;
                else if (!strcasecmp(ty.c_str(), "hex")) t = TY_HEX;
                if (p->dev.nkvtypes >= MAX_NAMES || noff + kl > sizeof(p->dev.names)) { set_err("parser '%s': too many Types", name ? name : ""); flbgpu_parser_destroy(p); return nullptr; }
                const int i = p->dev.nkvtypes++;
                p->dev.kvtype_off[i] = (int) noff; p->dev.kvtype_len[i] = (int) kl; p->dev.kvtype_kind[i] = t;
                memcpy(p->dev.names + noff, q, kl);
                noff += kl;
            }
            q = *sp ? sp + 1 : sp;
        }
    }
    p->dev.kv_format = kv;
    p->dev.no_bare_keys = (kv == 1 && logfmt_no_bare_keys) ? 1 : 0;
    return p;
}

// ------------------------------------------------------------------------------------------ keys
// grammar: src/record_accessor/ra.l:54-67, ra.y:60-99
bool flbgpu::parse_ra(const char *pat, DevKey &k, std::string &why) {
    memset(&k, 0, sizeof(k));
    k.is_ra = 1;
    const char *p = pat;
    if (*p != '$') { why = "record accessor must start with $"; return false; }
    p++;
    if (!((*p >= 'A' && *p <= 'Z') || (*p >= 'a' && *p <= 'z') || *p == '_')) { why = "unsupported record accessor (only $key['sub'][n] forms are on the GPU path)"; return false; }
    const char *q = p;
    while ((*q >= 'A' && *q <= 'Z') || (*q >= 'a' && *q <= 'z') || (*q >= '0' && *q <= '9') || *q == '_' || *q == '.' || *q == '-' || *q == '/') q++;
    if (q - p >= MAX_KEY) { why = "key too long"; return false; }
    if ((q - p) == 3 && !strncmp(p, "TAG", 3)) { why = "$TAG accessors are not on the GPU path"; return false; }
    memcpy(k.key, p, q - p);
    k.key_len = (int) (q - p);
    p = q;
    size_t so = 0;
    while (*p == '[') {
        if (k.nsub >= MAX_SUBKEYS) { why = "too many subkeys"; return false; }
        p++;
        int s = k.nsub;
        if (*p == '\'') {
            p++;
            k.sub_off[s] = (int) so;
            for (;;) {
                if (!*p) { why = "unterminated subkey string"; return false; }
                if (*p == '\'') {
                    if (p[1] == '\'') { if (so >= sizeof(k.sub_str)) { why = "subkeys too long"; return false; } k.sub_str[so++] = '\''; p += 2; continue; }
                    p++;
                    break;
                }
                if (so >= sizeof(k.sub_str)) { why = "subkeys too long"; return false; }
                k.sub_str[so++] = *p++;
            }
            k.sub_len[s] = (int) so - k.sub_off[s];
        }
        else if (*p >= '0' && *p <= '9') {
            k.sub_is_index[s] = 1;
            k.sub_index[s] = atoi(p);
            while (*p >= '0' && *p <= '9') p++;
        }
        else { why = "bad subkey"; return false; }
        if (*p != ']') { why = "bad subkey"; return false; }
        p++;
        k.nsub++;
    }
    if (*p) { why = "trailing characters in record accessor"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------ filters
bool filter_common_init(flbgpu_filter *f) {
    HIPOK(hipStreamCreate(&f->stream));
    HIPOK(hipEventCreate(&f->ev0));
    HIPOK(hipEventCreate(&f->ev1));
    return true;
}

void prof_resolve(flbgpu_filter *f) {
    for (auto &p : f->pending) {
        float ms = 0;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            bool hit = false;
            for (auto &k : f->kp) if (!strcmp(k.name, p.name)) { k.ms += ms; k.launches++; hit = true; break; }
            if (!hit) { KernelProf k; k.name = p.name; k.ms = ms; k.launches = 1; f->kp.push_back(k); }
        }
        (void) hipEventDestroy(p.e0);
        (void) hipEventDestroy(p.e1);
    }
    f->pending.clear();
}

extern "C" void flbgpu_filter_profile(flbgpu_filter *f, int enable) { prof_resolve(f); f->prof = enable != 0; f->kp.clear(); }

extern "C" int flbgpu_filter_profile_read(flbgpu_filter *f, int max, const char **names, double *ms, uint64_t *launches) {
    prof_resolve(f);
    int n = 0;
    for (auto &k : f->kp) {
        if (n >= max) break;
        names[n] = k.name; ms[n] = k.ms; launches[n] = k.launches; n++;
    }
    return n;
}

extern "C" uint64_t flbgpu_filter_regex_corners(flbgpu_filter *f) {
    if (!f) return 0;
    uint64_t total = 0;
    auto add = [&](const DevCap &u) {
        unsigned long long v = 0;
        if (u.corner_flags && u.corner_count && hipMemcpy(&v, u.corner_count, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) total += v;
    };
    for (auto *p : f->parsers) if (!p->dev.is_json) add(p->dev.utf8);
    for (const GrepRule &r : f->rules) add(r.utf8);
    return total;
}

extern "C" void flbgpu_filter_last_counts(flbgpu_filter *f, uint64_t *in_records, uint64_t *out_records) {
    if (in_records) *in_records = f->last_in;
    if (out_records) *out_records = f->last_out;
}

extern "C" flbgpu_filter *flbgpu_filter_parser_create(const char *key_name, int reserve_data, int preserve_key,
                                                      int nparsers, flbgpu_parser **parsers) {
    if (!key_name) { set_err("filter_parser: missing 'key_name'"); return nullptr; }
    if (nparsers <= 0) { set_err("filter_parser: Invalid 'parser'"); return nullptr; }
    auto *f = new flbgpu_filter();
    f->kind = F_PARSER;
    memset(&f->pcfg, 0, sizeof(f->pcfg));
    f->pcfg.reserve_data = reserve_data;
    f->pcfg.preserve_key = preserve_key;
    f->pcfg.nparsers = nparsers;
    if (key_name[0] == '$') {
        std::string why;
        if (!parse_ra(key_name, f->pcfg.key, why)) { set_err("filter_parser: invalid record accessor pattern '%s': %s", key_name, why.c_str()); delete f; return nullptr; }
    }
    else {
        size_t n = strlen(key_name);
        if (n >= MAX_KEY) { set_err("filter_parser: key_name too long"); delete f; return nullptr; }
        memcpy(f->pcfg.key.key, key_name, n);
        f->pcfg.key.key_len = (int) n;
    }
    std::vector<DevParser> dp;
    // a HOST parser (a Regex that is not a regular expression, csrc/rxbt.inc): as the only entry of a list it runs in the place of
    // k_parser_rx (host_parser_rx); in a list of several parsers its answers for the chunk's values are computed before the list's
    // kernel runs, which reads them where it would walk a device parser's tables (host_list_rx, k_parser_generic)
    {
        int nhost = 0;
        for (int i = 0; i < nparsers; i++) if (parsers[i] && parsers[i]->bt) nhost++;
        if (nparsers > 1 && nhost > MAX_HOST_PARSERS) {
            set_err("filter_parser: %d of the %d parsers have a Regex that is not a regular expression (look-around, back-reference, atomic group ...): "
                    "a list takes up to %d of them", nhost, nparsers, MAX_HOST_PARSERS);
            delete f;
            return nullptr;
        }
        f->host_list = nparsers > 1 && nhost > 0;
    }
    int host_slot = 0;
    for (int i = 0; i < nparsers; i++) {
        f->parsers.push_back(parsers[i]);
        parsers[i]->dev.tz_trans = nullptr; parsers[i]->dev.tz_gmtoff = nullptr; parsers[i]->dev.tz_ttype = nullptr;
        if (parsers[i]->dev.zone_mode == 1) {
            // one block: transitions (8 bytes each), offsets (4), transition types (1)
            const TzTable &z = parsers[i]->zone;
            const size_t nt = z.trans.size(), ny = z.gmtoff.size(), bytes = nt * 8 + ny * 4 + nt + 16;
            if (!parsers[i]->d_zone) {
                std::vector<uint8_t> blk(bytes, 0);
                if (nt) memcpy(blk.data(), z.trans.data(), nt * 8);
                memcpy(blk.data() + nt * 8, z.gmtoff.data(), ny * 4);
                if (nt) memcpy(blk.data() + nt * 8 + ny * 4, z.ttype.data(), nt);
                if (hipMalloc(&parsers[i]->d_zone, bytes) != hipSuccess ||
                    hipMemcpy(parsers[i]->d_zone, blk.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
                    set_err("device copy of the time zone table failed"); delete f; return nullptr;
                }
            }
            const uint8_t *b = (const uint8_t *) parsers[i]->d_zone;
            parsers[i]->dev.tz_trans = (const int64_t *) b; parsers[i]->dev.tz_gmtoff = (const int32_t *) (b + nt * 8); parsers[i]->dev.tz_ttype = b + nt * 8 + ny * 4;
            parsers[i]->dev.tz_timecnt = (int) nt; parsers[i]->dev.tz_typecnt = (int) ny; parsers[i]->dev.tz_default = z.default_type;
        }
        parsers[i]->dev.decs = nullptr;
        if (parsers[i]->decs.n > 0) {
            if (!parsers[i]->d_decs &&
                (hipMalloc(&parsers[i]->d_decs, sizeof(DevDecoders)) != hipSuccess ||
                 hipMemcpy(parsers[i]->d_decs, &parsers[i]->decs, sizeof(DevDecoders), hipMemcpyHostToDevice) != hipSuccess)) {
                set_err("filter_parser: uploading the decoders failed"); delete f; return nullptr;
            }
            parsers[i]->dev.decs = (const DevDecoders *) parsers[i]->d_decs;
            f->has_decoders = true;
        }
        dp.push_back(parsers[i]->dev);
        dp.back().host_only = parsers[i]->bt ? 1 : 0;
        dp.back().host_slot = parsers[i]->bt ? (host_slot++ & (MAX_HOST_PARSERS - 1)) : 0;
        if ((uint32_t) parsers[i]->dev.nfields * 2 > f->caps_stride) f->caps_stride = (uint32_t) parsers[i]->dev.nfields * 2;
    }
    if (f->caps_stride == 0) f->caps_stride = 2;
    for (int i = 0; i < nparsers; i++)
        if (parsers[i]->dev.is_json && f->caps_stride < 8) f->caps_stride = 8;   // the span columns also hold JSON container counts
    f->caps_stride = (f->caps_stride + 3) & ~3u;             // 16-byte rows
    if (!filter_common_init(f) || !f->d_parsers.ensure(dp.size() * sizeof(DevParser))) { delete f; return nullptr; }
    if (hipMemcpy(f->d_parsers.p, dp.data(), dp.size() * sizeof(DevParser), hipMemcpyHostToDevice) != hipSuccess) { set_err("upload failed"); delete f; return nullptr; }
    return f;
}

bool flbgpu::compile_rule(const std::string &ra_field, const char *pattern, GrepRule &r, std::vector<TableBlob *> &blobs, std::string &why, bool *nonregular) {
    std::string w2;
    if (!parse_ra(ra_field.c_str(), r.key, w2)) { why = "invalid record accessor? '" + ra_field + "': " + w2; return false; }
    const char *ps, *pe;
    unsigned opts;
    rx::split_flb_pattern(pattern, &ps, &pe, &opts);
    rx::Program prog;
    std::string err;
    if (!rx::compile(ps, (size_t) (pe - ps), opts, false, prog, err)) {
        why = std::string("could not compile regex pattern '") + pattern + "' for the GPU path: " + err;
        if (nonregular) *nonregular = prog.nonregular;
        return false;
    }
    auto *b1 = new TableBlob(), *b2 = new TableBlob();
    blobs.push_back(b1);
    blobs.push_back(b2);
    if (!upload_dfa(prog.ascii, *b1, r.dfa) || !upload_utf8(prog, *b2, r.utf8)) { why = flbgpu_last_error(); return false; }
    return true;
}

extern "C" flbgpu_filter *flbgpu_filter_grep_create(int nrules, const char *const *kinds, const char *const *values,
                                                    const char *logical_op) {
    auto *f = new flbgpu_filter();
    f->kind = F_GREP;
    f->logical_op = OP_LEGACY;
    if (logical_op) {
        size_t len = strlen(logical_op);
        if (len == 3 && !strncasecmp("AND", logical_op, 3)) f->logical_op = OP_AND;
        else if (len == 2 && !strncasecmp("OR", logical_op, 2)) f->logical_op = OP_OR;
    }
    int first_rule = 0;
    std::vector<std::string> rule_field, rule_pat;
    for (int i = 0; i < nrules; i++) {
        GrepRule r;
        memset(&r, 0, sizeof(r));
        if (!strcasecmp(kinds[i], "regex")) r.type = GREP_REGEX;
        else if (!strcasecmp(kinds[i], "exclude")) r.type = GREP_EXCLUDE;
        else continue;
        if (f->logical_op != OP_LEGACY && first_rule != 0 && first_rule != r.type) {
            set_err("filter_grep: Both 'regex' and 'exclude' are set.");
            delete f;
            return nullptr;
        }
        first_rule = r.type;
        // flb_utils_split(val, ' ', 1): src/flb_utils.c:386-462
        const char *v = values[i];
        while (*v == ' ') v++;
        const char *sp = strchr(v, ' ');
        if (!sp || sp == v || !sp[1]) { set_err("filter_grep: invalid regex, expected field and regular expression"); delete f; return nullptr; }
        std::string field(v, sp - v);
        if (field[0] != '$') field = "$" + field;
        std::string why;
        bool nonregular = false;
        rx::BtProgram *bt = nullptr;
        if (!compile_rule(field, sp + 1, r, f->rule_blobs, why, &nonregular)) {
            // not a regular expression: a HOST rule -- the device finds the value, the backtracking matcher answers (grep_host_pass)
            std::string e2;
            if (nonregular && !getenv("FLBGPU_NO_HOST_RULES")) {
                const char *ps, *pe;
                unsigned opts;
                rx::split_flb_pattern(sp + 1, &ps, &pe, &opts);
                bt = rx::bt_compile(ps, (size_t) (pe - ps), opts, e2);
            }
            if (!bt) { set_err("filter_grep: %s%s%s", why.c_str(), e2.empty() ? "" : "; on the host: ", e2.c_str()); delete f; return nullptr; }
            memset(&r.dfa, 0, sizeof(r.dfa)); memset(&r.utf8, 0, sizeof(r.utf8));
            f->has_host_rules = true;
        }
        if ((int) f->rules.size() >= MAX_RULES) { set_err("filter_grep: more than %d rules", MAX_RULES); if (bt) rx::bt_free(bt); delete f; return nullptr; }
        f->rules.push_back(r);
        f->host_rx.push_back(bt);
        rule_field.push_back(field);
        rule_pat.push_back(sp + 1);
    }
    // Logical_Op OR (plugins/filter_grep/grep.c:250-284): the rules all have one type and the record's fate is "does ANY rule
    // match" -- so the rules that test the SAME field are one search for the alternation of their patterns, one automaton
    // pass over the value instead of one per rule (BASELINE configs[2]: 16 + 16 rules on five fields).  Each pattern keeps
    // its own /../imx options as an inline group; a group whose automaton would exceed the table budget is split in halves.
    if (f->logical_op == OP_OR && f->rules.size() >= 2 && !f->has_host_rules && !getenv("FLBGPU_NO_MERGE")) {
        std::vector<std::string> keys;
        std::vector<std::vector<int>> members;
        for (size_t i = 0; i < f->rules.size(); i++) {
            size_t k = 0;
            while (k < keys.size() && keys[k] != rule_field[i]) k++;
            if (k == keys.size()) { keys.push_back(rule_field[i]); members.emplace_back(); }
            members[k].push_back((int) i);
        }
        auto inline_group = [&](const std::string &pat) -> std::string {
            const char *ps, *pe;
            unsigned opts;
            rx::split_flb_pattern(pat.c_str(), &ps, &pe, &opts);
            std::string fl;
            if (opts & rx::OPT_IGNORECASE) fl += 'i';
            if (opts & rx::OPT_MULTILINE) fl += 'm';
            if (opts & rx::OPT_EXTEND) fl += 'x';
            std::string g = "(?" + fl + ":" + std::string(ps, pe);
            if (opts & rx::OPT_EXTEND) g += "\n";           // (a trailing # comment must not swallow the closing parenthesis)
            return g + ")";
        };
        std::vector<GrepRule> out;
        const int type = f->rules[0].type;
        std::function<void(const std::string &, const std::vector<int> &)> merge = [&](const std::string &field, const std::vector<int> &idx) {
            if (idx.size() == 1) { out.push_back(f->rules[(size_t) idx[0]]); return; }
            std::string pat;
            for (size_t q = 0; q < idx.size(); q++) pat += (q ? "|" : "") + inline_group(rule_pat[(size_t) idx[q]]);
            GrepRule r;
            memset(&r, 0, sizeof(r));
            r.type = type;
            std::string why;
            if (compile_rule(field, pat.c_str(), r, f->rule_blobs, why)) { out.push_back(r); return; }
            const std::vector<int> a(idx.begin(), idx.begin() + (long) idx.size() / 2), b(idx.begin() + (long) idx.size() / 2, idx.end());
            merge(field, a);
            merge(field, b);
        };
        for (size_t k = 0; k < keys.size(); k++) merge(keys[k], members[k]);
        f->rules.swap(out);
        f->host_rx.assign(f->rules.size(), nullptr);
    }
    if (!filter_common_init(f) || !f->d_rules.ensure(std::max<size_t>(1, f->rules.size()) * sizeof(GrepRule))) { delete f; return nullptr; }
    if (!f->rules.empty() &&
        hipMemcpy(f->d_rules.p, f->rules.data(), f->rules.size() * sizeof(GrepRule), hipMemcpyHostToDevice) != hipSuccess) {
        set_err("upload failed");
        delete f;
        return nullptr;
    }
    return f;
}

extern "C" void flbgpu_filter_destroy(flbgpu_filter *f) { delete f; }

// which builds a filter_parser instance runs (host_int.hpp Probe: no choice is for good): out[0] what the last call ran (bit 0 the single
// pass, 1 the three-port pair tables, 2 the time lookup in k_pg_emit, 3 the plain emit build, 4 the rows walked in the order of their lengths), out[1..4] whether the single pass / the
// three-port tables / the emit-side lookup / the plain build are set aside right now, out[5] tries of a build that was set aside,
// out[6] tries that brought it back, out[7] device-level calls so far
extern "C" int flbgpu_filter_paths(flbgpu_filter *f, uint64_t *out8) {
    if (!f || !out8) return -1;
    out8[0] = f->last_path; out8[1] = f->tile.off; out8[2] = f->fx5.off; out8[3] = f->defer.off; out8[4] = f->plain.off;
    out8[5] = f->tile.probes + f->fx5.probes + f->defer.probes + f->plain.probes;
    out8[6] = f->tile.returns + f->fx5.returns + f->defer.returns + f->plain.returns;
    out8[7] = f->calls;
    return 0;
}

// host rules of a filter (see "host rules" below): out[0] = how many of its rules / parsers run on the host's backtracking matcher,
// out[1] = values it has searched so far, out[2] = of these, searches that ended on the backtrack budget (answered "no match", as the
// reference answers its own limit), out[3] = records a host parser did not take (duplicate Key_Name entries, values >= 64 KB)
extern "C" int flbgpu_filter_host_rules(flbgpu_filter *f, uint64_t *out4) {
    if (!f || !out4) return -1;
    uint64_t k = 0;
    for (auto *b : f->host_rx) if (b) k++;
    for (auto *p : f->parsers) if (p->bt) k++;
    out4[0] = k; out4[1] = f->host_values; out4[2] = f->host_budget_over; out4[3] = f->host_unhandled;
    if (f->l2m_gate) { for (auto *b : f->l2m_gate->host_rx) if (b) out4[0]++; out4[1] += f->l2m_gate->host_values; out4[2] += f->l2m_gate->host_budget_over; }
    return 0;
}

extern "C" void flbgpu_parser_destroy(flbgpu_parser *p) {
    if (!p) return;
    if (p->self_filter) delete p->self_filter;
    delete p;
}

// ------------------------------------------------------------------------------------------ run (device level)
struct MiscWords { unsigned long long first_bad; unsigned long long max_row; unsigned long long counts[18]; unsigned int ov_count; unsigned int kept_count; };      // counts[16], [17]: ParserMatchArgs::len_stat
static const unsigned int OV_CAP = 1u << 16;      // (record, index) pairs of FParserCfg's side list

// Pass 1 of filter_parser on a device chunk: every record decoded, located, matched and sized (locate / rx /
// finish, the generic kernel for the records outside the fast shape).  On return out_len[r] (f->d_len) holds the
// size of record r's output (0: nothing is emitted for it), *n_valid the rows in front of the first decoder
// error, hm the counters.  The emit pass (or the fused pair's decide + emit) follows.
// pair mode (filter_grep follows and is evaluated inline): grep's rules for k_parser_rx, keep_len for k_parser_finish
struct PairCtx { PgInline pg; uint32_t *keep_len; const flbgpu_filter *fg; uint32_t *desc; uint32_t dstride; };

// ------------------------------------------------------------------------------------------ host rules
// A Regex / Parser entry that is not a regular expression (look-around, atomic groups, possessive repeats, back-references, \Z \G \K)
// would abort start-up in the reference's place (src/flb_filter.c:691-697) if the filter refused it.  It does not: the DEVICE still
// decodes every record, resolves the keys, evaluates the other rules, times, sizes and writes the records; only the search of that
// one pattern runs on the host -- the product's own backtracking matcher (rxbt.inc; nothing under oracle/) over the values the device
// located -- and its answers go back as one bit per row (filter_grep) or as the capture columns k_parser_rx would have written
// (filter_parser).  Two passes over the chunk and a round trip through host memory: a correct slow path, counted
// (flbgpu_filter_host_rules), never the timed one.
//
// the caller's copy of the chunk a host-level call uploaded (the values are read from it instead of a copy back)
struct HostView { const void *dev = nullptr; const uint8_t *host = nullptr; size_t bytes = 0; };
static thread_local HostView g_hostview;

template <class F> static void parallel_rows(uint64_t n, F fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (n < 4096 || nt < 2) { fn((uint64_t) 0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = ((n + nt - 1) / nt + 63) & ~63ull;           // (whole words of a bit column per thread)
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t a = (uint64_t) t * per, b = a + per < n ? a + per : n;
        if (a < b) th.emplace_back([=]() { fn(a, b); });
    }
    for (auto &t : th) t.join();
}

// the chunk's bytes and row offsets on the host
static bool host_chunk(const flbgpu_dev_chunk *in, hipStream_t st, std::vector<uint8_t> &copy, const uint8_t **data, std::vector<uint64_t> &off) {
    off.resize((size_t) in->n + 1);
    if (hipMemcpyAsync(off.data(), in->row_off, ((size_t) in->n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (g_hostview.dev == in->data && g_hostview.host && g_hostview.bytes >= in->bytes) *data = g_hostview.host;
    else {
        copy.resize((size_t) in->bytes + 16);
        if (hipMemcpyAsync(copy.data(), in->data, (size_t) in->bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        *data = copy.data();
    }
    return hipStreamSynchronize(st) == hipSuccess;
}

// A call on a small chunk (what the engine appends at a time) is bound by the waits between its launches, not by its kernels: the
// stages then launch AHEAD of the counters they normally wait for -- the fix-up pass, the rule decisions, the scan and the writer with
// room for the usual output -- and wait once at the end; kernels guard themselves on the device (fix-up: the per-wave lists,
// writers: ParserEmitArgs::out_cap).  Whatever the counters then show that the launched kernels did not cover (rows for the generic
// kernel or the strptime interpreter, a bad record, records for the exact writer, an output over the room) runs the stage again the
// usual way.  Set around chain_dev by the entry points; off for everything else.
struct SpecCall {
    bool on = false;
    bool last = false;              // the stage being run ends the chain (chain_dev)
    uint8_t *sink = nullptr;        // host-level call: page-locked slab the last stage's output is written into by the device
    uint64_t sink_cap = 0;
    bool sunk = false;              // ... and it is there
};
static thread_local SpecCall g_spec;
struct SpecOff {                    // one stage run the usual way
    bool was;
    SpecOff() : was(g_spec.on) { g_spec.on = false; }
    ~SpecOff() { g_spec.on = was; }
};
static const uint64_t SPEC_MAX_RECORDS = 262144, SPEC_MAX_BYTES = 8u << 20;
static inline bool spec_wanted(uint64_t n, uint64_t bytes) { return n <= SPEC_MAX_RECORDS && bytes <= SPEC_MAX_BYTES && !getenv("FLBGPU_NO_SPEC"); }
// what parser_size_pass reads from the counters between its launches, for a pass launched ahead: false = run it the usual way
// fx5 (three write ports) hands a record with an empty field at an even position to the generic kernel: when the fast walk does not
// settle more than 1 row in 64 of a chunk, this filter takes the four-port tables for the next calls (both are on the device) and tries
// the three-port ones again later (host_int.hpp Probe)
static void note_fx5(flbgpu_filter *f, const MiscWords &hm, uint64_t n) {
    if (!f->last_fx5 || n < 64 || !f->parsers[0]->fx2b.ok) return;
    if (hm.counts[9] * 64 > n) f->fx5.bad(); else f->fx5.good();
}
// values the forward walk from boundary 0 does not settle take the reverse pass with tables in global memory: when that is the rule for
// this pattern / this data, the phase kernels (tables in LDS) are the better choice for the next calls.  NOT after a launch that walked
// the three-port tables while the four-port ones stand by (round 5): what it handed on is those tables' doing -- every even-length line
// without the pattern's optional tail ends in a cell with two writes at one position: 46 % of bench.py's mixed shapes --, note_fx5 answers
// it with the other tables.
static void note_unsettled(flbgpu_filter *f, const MiscWords &hm, uint64_t n) {
    if (n < 64 || !(f->last_path & 1u)) return;
    if (f->last_fx5 && f->parsers[0]->fx2b.ok && hm.counts[9] * 64 > n) return;
    if (hm.counts[9] * 8 > n || hm.counts[10] * 4 > n) f->tile.bad(); else f->tile.good();
}
// The register kernel's walk is position-synchronous: a wave steps as far as its longest record.  counts[16] / counts[17] = what the walk
// steps through in chunk order / what the rows hold (dev.hpp ParserMatchArgs::len_stat, perm.hpp): when the lines of this data differ
// that much in length, the next calls walk the rows in the order of their lengths (kernels_perm.hip); back when they no longer do.
static void note_lengths(flbgpu_filter *f, const MiscWords &hm) {
    if (hm.counts[17] == 0) return;
    const double ratio = (double) hm.counts[16] / (double) hm.counts[17];
    if (ratio > 1.35) f->sort_rows = true;
    else if (ratio < 1.15) f->sort_rows = false;
}
static bool ahead_counters_ok(flbgpu_filter *f, const MiscWords &hm, uint64_t n) {
    note_lengths(f, hm);
    note_fx5(f, hm, n);
    note_unsettled(f, hm, n);
    return hm.counts[8] == 0 && hm.counts[2] == 0 && hm.first_bad >= n;
}

// k_parser_rx's part for a host parser: the capture search of parser 0 on the values k_parser_locate found, by the backtracking
// matcher; writes what that kernel writes -- the span columns, RF_RXOK, the time text column
static bool host_parser_rx(flbgpu_filter *f, const flbgpu_dev_chunk *in, const ParserMatchArgs &ma, hipStream_t st) {
    const uint64_t n = ma.n;
    const flbgpu_parser *hp = f->parsers[0];
    const DevParser &d = hp->dev;
    const int ncap = 2 * d.nfields;
    std::vector<uint32_t> info(3 * (size_t) n);
    HIPOK(hipMemcpyAsync(info.data(), ma.info, info.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    std::vector<uint8_t> copy;
    std::vector<uint64_t> off;
    const uint8_t *hd = nullptr;
    if (!host_chunk(in, st, copy, &hd, off)) { set_err("filter_parser: copying the chunk back for the host parser failed"); return false; }
    std::vector<uint32_t> caps((size_t) ncap * n, 0u), tbuf((size_t) TBUF_WORDS * n, 0u);
    std::atomic<uint64_t> over{0}, vals{0}, unhandled{0};
    const uint64_t total_bytes = in->bytes;
    const int ng = rx::bt_ngroups(hp->bt);
    parallel_rows(n, [&](uint64_t a, uint64_t b) {
        std::vector<int> beg((size_t) ng + 1), end((size_t) ng + 1);
        uint64_t ov = 0, nv = 0, un = 0;
        for (uint64_t r = a; r < b; r++) {
            uint32_t fl = info[r];
            if (fl & RF_GENERIC) { info[r] = fl & ~(uint32_t) RF_GENERIC; un++; continue; }     // (duplicate Key_Name entries, a value of 64 KB or more: not taken)
            if (!(fl & RF_CAND)) continue;
            const uint32_t vo = info[(size_t) n + r], vlen = info[2 * (size_t) n + r];
            if (off[r] + vo + (uint64_t) vlen > total_bytes) continue;
            const uint8_t *val = hd + off[r] + vo;
            const int res = rx::bt_search(hp->bt, val, (int) vlen, beg.data(), end.data());
            nv++;
            if (res == -4) ov++;
            if (res <= 0) continue;
            for (int q = 0; q < d.nfields; q++) {
                const int g = d.field_group[q];
                caps[(size_t) (2 * q) * n + r] = beg[(size_t) g] >= 0 ? (uint32_t) beg[(size_t) g] : CAP_UNSET;
                caps[(size_t) (2 * q + 1) * n + r] = end[(size_t) g] >= 0 ? (uint32_t) end[(size_t) g] : CAP_UNSET;
            }
            info[r] = fl | RF_RXOK;
            if (d.time_field >= 0) {
                const uint32_t tb = caps[(size_t) (2 * d.time_field) * n + r], te = caps[(size_t) (2 * d.time_field + 1) * n + r];
                if (tb != CAP_UNSET && te != CAP_UNSET && te >= tb && te - tb <= 4 * TBUF_WORDS) {
                    uint8_t tx[4 * TBUF_WORDS];
                    memset(tx, 0, sizeof(tx));
                    memcpy(tx, val + tb, te - tb);
                    for (int k = 0; k < TBUF_WORDS; k++) { uint32_t w; memcpy(&w, tx + 4 * k, 4); tbuf[(size_t) k * n + r] = w; }
                }
            }
        }
        over += ov; vals += nv; unhandled += un;
    });
    f->host_budget_over += over.load(); f->host_values += vals.load(); f->host_unhandled += unhandled.load();
    HIPOK(hipMemcpyAsync(ma.caps, caps.data(), caps.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIPOK(hipMemcpyAsync(ma.info, info.data(), (size_t) n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (ma.tbuf) HIPOK(hipMemcpyAsync(ma.tbuf, tbuf.data(), tbuf.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIPOK(hipStreamSynchronize(st));            // (the vectors above are the source of the copies)
    return true;
}

// Host parsers inside a list of several parsers (plugins/filter_parser/filter_parser.c:286-323 tries the list in order on every value):
// each one's capture search on the values k_parser_locate found, by the backtracking matcher, BEFORE the list's kernel runs -- per host
// parser [1 + 2 * nfields][n] words (dev.hpp ParserMatchArgs::host_res).  Every candidate row goes to that kernel (RF_GENERIC).
static bool host_list_rx(flbgpu_filter *f, const flbgpu_dev_chunk *in, ParserMatchArgs &ma, hipStream_t st) {
    const uint64_t n = ma.n;
    std::vector<uint32_t> info(3 * (size_t) n);
    HIPOK(hipMemcpyAsync(info.data(), ma.info, info.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    std::vector<uint8_t> copy;
    std::vector<uint64_t> off;
    const uint8_t *hd = nullptr;
    if (!host_chunk(in, st, copy, &hd, off)) { set_err("filter_parser: copying the chunk back for the host parsers failed"); return false; }
    const uint64_t total_bytes = in->bytes;
    int slot = 0;
    std::atomic<uint64_t> over{0}, vals{0}, unhandled{0};
    for (size_t q = 0; q < f->parsers.size(); q++) {
        const flbgpu_parser *hp = f->parsers[q];
        if (!hp->bt) continue;
        const DevParser &d = hp->dev;
        const int ncap = 2 * d.nfields, ng = rx::bt_ngroups(hp->bt);
        std::vector<uint32_t> res((size_t) (1 + ncap) * n, 0u);
        const bool first = slot == 0;
        parallel_rows(n, [&](uint64_t a, uint64_t b) {
            std::vector<int> beg((size_t) ng + 1), end((size_t) ng + 1);
            uint64_t ov = 0, nv = 0, un = 0;
            for (uint64_t r = a; r < b; r++) {
                const uint32_t fl = info[r];
                if (fl & RF_GENERIC) { if (first) un++; continue; }      // (duplicate Key_Name entries: the value was not located as one -- the matcher's answer is "no")
                if (!(fl & RF_CAND)) continue;
                const uint32_t vo = info[(size_t) n + r], vlen = info[2 * (size_t) n + r];
                if (off[r] + vo + (uint64_t) vlen > total_bytes || vlen > 0x7FFFFFF0u) continue;
                const int rr = rx::bt_search(hp->bt, hd + off[r] + vo, (int) vlen, beg.data(), end.data());
                nv++;
                if (rr == -4) ov++;
                if (rr <= 0) continue;
                res[r] = vo + 1;
                for (int k = 0; k < d.nfields; k++) {
                    const int g = d.field_group[k];
                    res[(size_t) (1 + 2 * k) * n + r] = beg[(size_t) g] >= 0 ? (uint32_t) beg[(size_t) g] : CAP_UNSET;
                    res[(size_t) (2 + 2 * k) * n + r] = end[(size_t) g] >= 0 ? (uint32_t) end[(size_t) g] : CAP_UNSET;
                }
            }
            over += ov; vals += nv; unhandled += un;
        });
        if (!f->d_hres[slot].ensure(res.size() * sizeof(uint32_t))) return false;
        HIPOK(hipMemcpy(f->d_hres[slot].p, res.data(), res.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        ma.host_res[slot] = f->d_hres[slot].as<uint32_t>();
        slot++;
    }
    f->host_budget_over += over.load(); f->host_values += vals.load(); f->host_unhandled += unhandled.load();
    for (uint64_t r = 0; r < n; r++) if (info[r] & RF_CAND) info[r] |= RF_GENERIC;
    HIPOK(hipMemcpy(ma.info, info.data(), (size_t) n * sizeof(uint32_t), hipMemcpyHostToDevice));
    return true;
}

static bool parser_size_pass(flbgpu_filter *f, const flbgpu_dev_chunk *in, hipStream_t st, MiscWords **dm_out, MiscWords **hm_out, uint64_t *n_valid,
                             PairCtx *pair = nullptr, bool *ahead = nullptr) {
    uint64_t n = in->n;
    if (!f->d_misc.ensure(sizeof(MiscWords))) return false;
    MiscWords *dm = f->d_misc.as<MiscWords>();
    // host mirror of the counters + the output size, page-locked so that the small copies are real
    // asynchronous DMA transfers
    if (!f->hp_misc.ensure(sizeof(MiscWords) + 4 * sizeof(uint64_t))) return false;
    MiscWords &hm = *f->hp_misc.as<MiscWords>();
    *dm_out = dm; *hm_out = &hm;
    const uint64_t *row_off = in->row_off;
    const uint8_t *data = (const uint8_t *) in->data;
    // scratch sizing: one state id per byte boundary of the longest record
    memset(&hm, 0, sizeof(hm));
    hm.first_bad = ~0ull;
    // (a call launched ahead: the counter block, the tail copy and the pair's keep_len column are set by ONE kernel -- k_call_prep, in
    // the register kernel's branch below; the other branches issue the commands here)
    bool prep_pending = g_spec.on;
    auto prep_now = [&]() -> bool {
        if (!prep_pending) return true;
        prep_pending = false;
        HIPOK(hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
        if (pair) HIPOK(hipMemsetAsync(pair->keep_len, 0xFF, n * sizeof(uint32_t), st));
        return true;
    };
    if (!prep_pending) HIPOK(hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
    // scratch of the fast path: one reverse-DFA state checkpoint per CHK_STEP bytes, per lane, for
    // values up to 4 KiB (longer ones go to the generic kernel, whose scratch is sized from the
    // longest row -- measured only when that kernel is needed)
    uint32_t chk_len = 4096 / CHK_STEP + 3;
    int cus = g_cus > 0 ? g_cus : 256;
    // one 1024-thread workgroup (16 waves) per CU shares one LDS copy of parser 0's hot ASCII
    // tables (160 KiB of LDS per CU); bigger tables are read through L2 instead
    int rx_threads = MATCH_BLOCK;
    if (getenv("FLBGPU_RX_THREADS")) { rx_threads = atoi(getenv("FLBGPU_RX_THREADS")); if (rx_threads < 64 || rx_threads > MATCH_BLOCK || (rx_threads & 63)) rx_threads = MATCH_BLOCK; }
    uint32_t tab_bytes = f->parsers[0]->dev.ascii.hot_bytes;
    uint32_t caps_bytes = (uint32_t) rx_threads * (f->caps_stride + 1) * (uint32_t) sizeof(uint16_t);   // + dummy column
    if (getenv("FLBGPU_NO_LDS")) tab_bytes = 0;
    const uint32_t lds_cap = 160 * 1024;
    if (tab_bytes + caps_bytes > lds_cap) {
        // tables win the LDS when only one of the two fits
        if (tab_bytes <= lds_cap) caps_bytes = 0;
        else { tab_bytes = 0; if (caps_bytes > lds_cap) caps_bytes = 0; }
    }
    uint32_t lds_bytes = tab_bytes;                          // staged table bytes (0: tables stay in global memory)
    // single-pass tile kernel (tile_kernels.inc) instead of locate / rx / finish: parser 0 start-anchored with compact
    // tables; a workgroup's waves share one copy of the tables, every wave owns a record tile + its capture columns
    const char *tmode0 = getenv("FLBGPU_TILE_MODE");
    // (pair cells -- two positions per table read -- are built and tested but measured SLOWER, DESIGN 4.0: off unless FLBGPU_PAIR2=1)
    // (fx2 = the tables without special entries, k_parser_reg<.., FX3>; FLBGPU_FX3=0: the tables with look-ahead / pair entries)
    const char *fx3env = getenv("FLBGPU_FX3");
    const bool use_fx2 = f->parsers[0]->dev.fx2.ok && !(tmode0 && !strcmp(tmode0, "tile")) && !(fx3env && fx3env[0] == '0');
    const bool has_fx5 = use_fx2 && f->parsers[0]->dev.fx2.pair_bias == 2 && f->parsers[0]->fx2b.ok;
    const bool fx5_fallback = has_fx5 && !f->fx5.use(f->calls);
    if (has_fx5 && (fx5_fallback ? 1 : 0) != f->fx_on_device) {
        // this filter's device copy of parser 0 gets the tables this call walks in the slot the kernels read
        HIPOK(hipMemcpyAsync((uint8_t *) f->d_parsers.as<DevParser>() + offsetof(DevParser, fx2), fx5_fallback ? &f->parsers[0]->fx2b : &f->parsers[0]->dev.fx2, sizeof(DevFx),
                             hipMemcpyHostToDevice, st));
        f->fx_on_device = fx5_fallback ? 1 : 0;
    }
    const DevFx &fx = use_fx2 ? (fx5_fallback ? f->parsers[0]->fx2b : f->parsers[0]->dev.fx2) : f->parsers[0]->dev.fx;
    bool use_tile = fx.ok && !f->parsers[0]->dev.is_json && !getenv("FLBGPU_NO_TILE") && !f->has_decoders;
    if (use_tile) use_tile = f->tile.use(f->calls);
    for (int q = 0; q < f->parsers[0]->dev.nfields; q++) if (f->parsers[0]->dev.field_name_len[q] > 250) use_tile = false;   // (TileCfg::name_cost is a byte)
    uint32_t tile_wave_bytes = 0, tile_pg_room = 0;
    // two builds of the single pass: value bytes in registers (k_parser_reg, 16 waves per CU: the default) or the
    // records in an LDS tile (k_parser_tile, FLBGPU_TILE_MODE=tile)
    const char *tmode = getenv("FLBGPU_TILE_MODE");
    const bool tile_in_lds = tmode && !strcmp(tmode, "tile");
    if (use_tile) {
        tile_wave_bytes = tile_in_lds ? (uint32_t) TILE_BYTES + fx.nslots * 128u : fx.nslots * 128u + 64u * (4 * TBUF_WORDS + 4);
        // (the two-position tables: a block of 68 bytes of capture slots per lane, tile_kernels.inc CAP_STEP)
        if (!tile_in_lds && use_fx2 && fx.pair_bias && tile_wave_bytes < 64u * 68u) tile_wave_bytes = 64u * 68u;
        if (pair) for (int i = 0; i < pair->pg.nrules; i++) if (tile_pg_room + ((pair->pg.rule_lds_bytes[i] + 15) & ~15u) <= 8192) tile_pg_room += (pair->pg.rule_lds_bytes[i] + 15) & ~15u;
        int waves = (int) ((lds_cap - 64 - fx.bytes - tile_pg_room) / tile_wave_bytes);
        // the register kernel: 12 waves per CU with the compact tables (the build without spills, tile_kernels.inc), 16 on request
        // (FLBGPU_TILE_WAVES=16) and for the older table form
        const int wcap = tile_in_lds ? 8 : 16;
        int wmax = tile_in_lds ? 8 : (use_fx2 ? 12 : 16);
        if (waves > wcap) waves = wcap;
        if (getenv("FLBGPU_TILE_WAVES")) { int w = atoi(getenv("FLBGPU_TILE_WAVES")); if (w >= 1 && w <= wcap) wmax = w; }
        if (waves > wmax) waves = wmax;
        if (waves < 2) use_tile = false;
        else { rx_threads = waves * 64; caps_bytes = 1; }
    }
    int grid = cus;
    // (several workgroups per CU: the register kernel's waves beyond the 16 of one workgroup -- FLBGPU_TILE_GRID_MULT, with FLBGPU_TILE_WAVES)
    if (use_tile && getenv("FLBGPU_TILE_GRID_MULT")) { const int m = atoi(getenv("FLBGPU_TILE_GRID_MULT")); if (m >= 1 && m <= 4) grid = cus * m; }
    uint64_t need_blocks = (n + rx_threads - 1) / rx_threads;
    if ((uint64_t) grid > need_blocks) grid = (int) need_blocks;
    if (!f->d_rid.ensure((size_t) grid * (rx_threads / 64) * 64 * chk_len * sizeof(uint16_t))) return false;
    if (!f->d_info.ensure(n * REC_NCOLS * sizeof(uint32_t)) || !f->d_caps.ensure(n * f->caps_stride * sizeof(uint32_t)) ||
        !f->d_null.ensure(n * sizeof(uint64_t)) || !f->d_len.ensure(n * sizeof(uint32_t)) ||
        !f->d_off.ensure((n + 1) * sizeof(uint64_t)) || !f->d_scan_tmp.ensure(scan_tmp_elems(n) * sizeof(uint64_t)) ||
        !f->d_status.ensure(n * TBUF_WORDS * sizeof(uint32_t)))
        return false;
    {
        // year-less Time_Formats: today's date (UTC, src/flb_parser.c:1982) into the device copies of the parsers
        bool any = false;
        for (auto *pp : f->parsers) any = any || pp->dev.yearless;
        if (any) {
            time_t now = g_time_now > 0 ? (time_t) g_time_now : time(NULL);
            struct tm tmy;
            gmtime_r(&now, &tmy);
            for (size_t q = 0; q < f->parsers.size(); q++) {
                DevParser &d = f->parsers[q]->dev;
                if (!d.yearless) continue;
                d.now_year = tmy.tm_year + 1900; d.now_mon = tmy.tm_mon; d.now_mday = tmy.tm_mday;
                HIPOK(hipMemcpyAsync((uint8_t *) (f->d_parsers.as<DevParser>() + q) + offsetof(DevParser, now_year), &d.now_year, 3 * sizeof(int),
                                     hipMemcpyHostToDevice, st));
            }
        }
    }
    if (!f->d_ov.ensure((size_t) OV_CAP * 2 * sizeof(unsigned long long))) return false;
    f->pcfg.ov_pairs = f->d_ov.as<unsigned long long>(); f->pcfg.ov_count = &dm->ov_count; f->pcfg.ov_cap = OV_CAP;
    ParserMatchArgs ma;
    ma.tbuf = f->d_status.as<uint32_t>();                    // (the status buffer is grep's; a parser filter uses it for the time text)
    ma.data = data; ma.row_off = row_off; ma.n = n; ma.cfg = f->pcfg; ma.parsers = f->d_parsers.as<DevParser>();
    ma.info = f->d_info.as<uint32_t>(); ma.caps = f->d_caps.as<uint32_t>(); ma.caps_stride = f->caps_stride;
    ma.null_mask = f->d_null.as<uint64_t>(); ma.out_len = f->d_len.as<uint32_t>(); ma.chk = f->d_rid.as<uint16_t>();
    ma.chk_len = chk_len; ma.chk_nfa_off = chk_len; ma.lds_bytes = lds_bytes; ma.caps_lds_off = tab_bytes; ma.caps_in_lds = caps_bytes ? 1 : 0;
    ma.lds_total = tab_bytes + caps_bytes; ma.debug_skip = getenv("FLBGPU_DEBUG_SKIP") ? (uint32_t) atoi(getenv("FLBGPU_DEBUG_SKIP")) : 0; ma.first_bad = &dm->first_bad; ma.counts = dm->counts;
    ma.bytes = in->bytes;
    ma.pg = nullptr; ma.pg_lds_off = 0; ma.pg_keep_len = nullptr; ma.self = nullptr; ma.desc = nullptr; ma.dstride = 0; ma.fix_first = 0; ma.fix_list = nullptr; ma.fix_count = &dm->counts[12]; ma.tail_buf = nullptr; ma.tail_start = 0; ma.stage_lds_off = 0; ma.stage_bytes = 0; ma.stage_nbuf = 0; ma.trace = nullptr; ma.trace_iters = 0;
    for (auto &hr : ma.host_res) hr = nullptr;
    ma.perm = nullptr; ma.len_stat = nullptr;
    ma.tile_lds_off = 0; ma.tile_wave_bytes = tile_wave_bytes; ma.use_fx2 = use_fx2 ? (fx.pair_bias == 2 ? 5 : fx.pair_bias ? 4 : 3) : 0;
    f->last_fx5 = use_tile && !tile_in_lds && ma.use_fx2 == 5;
    f->last_path = (use_tile ? 1u : 0u) | (f->last_fx5 ? 2u : 0u);
    if (use_tile) { ma.lds_bytes = 0; ma.caps_lds_off = 0; ma.lds_total = fx.bytes; }
    if (pair) {
        // the rules' match-only DFA blocks behind the tables and the span columns in k_parser_rx's LDS, while they fit
        uint32_t at = (ma.lds_total + 15) & ~15u, used = 0;
        ma.pg_lds_off = at;
        const uint32_t room = use_tile ? at + tile_pg_room : lds_cap;
        for (int i = 0; i < pair->pg.nrules; i++) {
            const uint32_t blob = pair->pg.rule_lds_bytes[i];
            pair->pg.rule_lds_off[i] = 0xFFFFFFFFu;
            if (at + used + ((blob + 15) & ~15u) <= room) { pair->pg.rule_lds_off[i] = used; used += (blob + 15) & ~15u; }
        }
        ma.lds_total = at + used;
        if (!f->d_pg.ensure(sizeof(PgInline))) return false;
        HIPOK(hipMemcpyAsync(f->d_pg.p, &pair->pg, sizeof(PgInline), hipMemcpyHostToDevice, st));
        ma.pg = f->d_pg.as<PgInline>();
        ma.pg_keep_len = pair->keep_len;
        ma.desc = pair->desc; ma.dstride = pair->dstride;
    }
    {
        // the parser's and the rules' per-record configuration for the single-pass kernels, by value (dev.hpp TileCfg)
        TileCfg &tc = ma.tc;
        memset(&tc, 0, sizeof(tc));
        const DevParser &d0 = f->parsers[0]->dev;
        tc.nfields = d0.nfields; tc.skip_empty = d0.skip_empty; tc.time_field = d0.time_field; tc.time_keep = d0.time_keep;
        tc.nregs_minus1 = d0.nregs_minus1; tc.time_with_tz = d0.time_with_tz; tc.time_offset = d0.time_offset; tc.plain_types = d0.plain_types;
        tc.plan = d0.plan;
        for (int q = 0; q < d0.nfields && q < MAX_NAMES; q++) {
            if (d0.field_is_time[q]) tc.is_time_mask |= 1u << q;
            const uint32_t nl = (uint32_t) d0.field_name_len[q], cost = (nl < 32 ? 1 : nl < 256 ? 2 : nl < 65536 ? 3 : 5) + nl;
            tc.name_cost[q] = (uint8_t) cost;
        }
        if (pair) {
            tc.pg_on = 1; tc.pg_nrules = pair->pg.nrules; tc.pg_logical_op = pair->pg.logical_op;
            tc.pg_static_drop = pair->pg.static_drop; tc.pg_time_fields = pair->pg.time_fields;
            tc.pg_fast = pair->pg.nrules <= TILE_RULES ? 1 : 0;
            // the time lookup left to k_pg_emit (dev.hpp TileCfg::defer_time; FLBGPU_DEFER_TIME=0: in the single pass, as before round 5)
            if (pair->desc && d0.time_field >= 0 && d0.plan.ok && !d0.time_keep && d0.plan.len <= 4 * TBUF_WORDS && f->defer.use(f->calls) &&
                !(getenv("FLBGPU_DEFER_TIME") && getenv("FLBGPU_DEFER_TIME")[0] == '0')) {
                int nyear = 0;
                for (int k = 0; k < d0.plan.nops; k++) if (d0.plan.ops[k].kind == TP_YEAR4) { nyear++; tc.year_off = d0.plan.ops[k].off; }
                uint32_t ntime = 0;
                for (int q = 0; q < d0.nfields; q++) ntime += d0.field_is_time[q] ? 1u : 0u;
                uint32_t ctp[32];
                compile_time_plan(d0.plan, ctp);                       // (k_pg_emit reads the text through the compiled form)
                tc.defer_time = nyear == 1 && ntime == 1 && ctp[31] ? 1 : 0;
                if (tc.defer_time) f->last_path |= 4u;
            }
            for (int i = 0; i < pair->pg.nrules; i++) {
                tc.pg_named |= pair->pg.rule_fmask[i];
                if (pair->pg.rule_lds_off[i] == 0xFFFFFFFFu) tc.pg_fast = 0;
                if (i < TILE_RULES) {
                    const DevDfa &df = pair->fg->rules[(size_t) i].dfa;
                    TileRule &tr = tc.rule[i];
                    tr.type = (uint32_t) pair->fg->rules[(size_t) i].type; tr.fmask = pair->pg.rule_fmask[i]; tr.lds_off = pair->pg.rule_lds_off[i];
                    tr.ncls = (uint32_t) df.ncls; tr.d_init = (uint32_t) df.d_init;
                    tr.o_dd = (uint32_t) ((const uint8_t *) df.ddelta - df.cls); tr.o_df = (uint32_t) (df.d_final - df.cls);
                }
            }
        }
    }
    // records outside the fast path (UTF-8 input, several parsers / candidate keys, ...): one general kernel
    auto run_generic = [&]() -> bool {
        launch_max_row_len(row_off, n, &dm->max_row, st);
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        uint32_t gchk_len = (uint32_t) (hm.max_row / CHK_STEP) + 3;
        // a parser whose non-ASCII side is the NFA engine keeps position SETS, one per NFA_CHK byte boundaries (nfa_dev.inc
        // nfa_chk_words 32-bit words = twice as many 16-bit slots), in its own part of a lane's scratch: behind the table walkers' slots
        const uint32_t nfa_off = gchk_len;
        for (auto *pp : f->parsers)
            if (pp->dev.utf8.nfa_on) {
                const uint32_t need = nfa_off + 2u * (uint32_t) (hm.max_row / rx::NFA_CHK + 2) * (uint32_t) (pp->dev.utf8.nfa.VW + 1) + 2;
                if (need > gchk_len) gchk_len = need;
            }
        int ggrid = cus * 8;
        while (ggrid > 1 && (size_t) ggrid * 4 * 64 * gchk_len * sizeof(uint16_t) > ((size_t) 2 << 30)) ggrid /= 2;
        if (!f->d_rid2.ensure((size_t) ggrid * 4 * 64 * gchk_len * sizeof(uint16_t))) return false;
        ParserMatchArgs mg = ma;
        mg.chk = f->d_rid2.as<uint16_t>();
        mg.chk_len = gchk_len;
        mg.chk_nfa_off = nfa_off;
        { ProfScope ps(f, st, "k_parser_generic"); launch_parser_generic(mg, ggrid, st); }
        return true;
    };
    if ((f->parsers[0]->dev.is_json || !use_tile || tile_in_lds) && !prep_now()) return false;
    if (f->host_list) {
        // host parsers in a list of several: locate on the device, every host parser's capture search on the host, then the list's own
        // kernel over every candidate -- it reads the answers where it would walk a device parser's tables
        if (!prep_now()) return false;
        ParserMatchArgs ml = ma;
        ml.caps_in_lds = 1; ml.chk_len = 0x7FFFFFFFu;
        { ProfScope ps(f, st, "k_parser_locate"); launch_parser_locate(ml, cus, st); }
        if (!host_list_rx(f, in, ma, st)) return false;
        if (!run_generic()) return false;
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
    }
    else if (f->parsers[0]->bt) {
        // a HOST parser (the Regex is not a regular expression): locate on the device, the capture search on the host, finish on the
        // device.  Every single candidate is located as one (no length limit of a device walker applies).
        ParserMatchArgs ml = ma;
        ml.caps_in_lds = 1; ml.chk_len = 0x7FFFFFFFu;
        { ProfScope ps(f, st, "k_parser_locate"); launch_parser_locate(ml, cus, st); }
        if (!host_parser_rx(f, in, ma, st)) return false;
        { ProfScope ps(f, st, "k_parser_finish"); launch_parser_finish(ma, cus, st); }
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
    }
    else if (f->parsers[0]->dev.is_json) {
        // Format json: one size kernel replaces locate / rx / finish
        { ProfScope ps(f, st, "k_pjson_size"); launch_pjson_size(ma, cus, st); }
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (hm.counts[2] > 0) {
            if (f->pcfg.nparsers > 1) { if (!run_generic()) return false; }      // the other parsers of the list get their turn
            else { ProfScope ps(f, st, "k_pjson_size_generic"); launch_pjson_size_generic(ma, st); }
        }
    }
    else if (use_tile) {
        ma.tile_lds_off = (ma.lds_total + 15) & ~15u;
        ma.lds_total = ma.tile_lds_off + (uint32_t) (rx_threads / 64) * tile_wave_bytes + 64;
        if (!tile_in_lds) {
            // staged ingest (tile_kernels.inc): what is left of the LDS becomes the workgroup's ring of staging buffers
            uint32_t skb = 20, want = 0;                                  // (off unless asked for: the per-lane burst is faster)
            if (getenv("FLBGPU_STAGE_KB")) { int v = atoi(getenv("FLBGPU_STAGE_KB")); if (v >= 1 && v <= (int) STAGE_MAXK) skb = (uint32_t) v; }
            if (getenv("FLBGPU_STAGE_NBUF")) { int v = atoi(getenv("FLBGPU_STAGE_NBUF")); if (v >= 0 && v <= 8) want = (uint32_t) v; }
            const uint32_t at = (ma.lds_total + 15) & ~15u, per = skb * 1024u + STAGE_SLACK;
            uint32_t nb = at + 16 < lds_cap ? (lds_cap - at - 16) / per : 0;
            if (nb > want) nb = want;
            if (((uintptr_t) data & 15) != 0) nb = 0;                 // (the coalesced loads are aligned 16-byte loads)
            if (nb) { ma.stage_lds_off = at; ma.stage_bytes = skb * 1024u; ma.stage_nbuf = nb; ma.lds_total = at + 16 + nb * per; }
        }
        if (!tile_in_lds) {
            // a zero-padded copy of the chunk's last bytes: the call-free kernel loads 16 bytes at a time without bounds tests
            const size_t T = 4096, tb = in->bytes < T ? (size_t) in->bytes : T;
            if (!f->d_tail.ensure(T + 512)) return false;
            if (prep_pending) {
                prep_pending = false;
                launch_call_prep(dm, (uint32_t) sizeof(hm), f->d_tail.as<uint8_t>(), (uint32_t) (T + 512), data, in->bytes, (uint32_t) tb,
                                 pair ? pair->keep_len : nullptr, n, st);
            }
            else {
                HIPOK(hipMemsetAsync(f->d_tail.p, 0, T + 512, st));
                if (tb) HIPOK(hipMemcpyAsync(f->d_tail.p, data + (in->bytes - tb), tb, hipMemcpyDeviceToDevice, st));
            }
            ma.tail_buf = f->d_tail.as<uint8_t>(); ma.tail_start = in->bytes - tb;
        }
        if (!f->d_args.ensure(sizeof(ParserMatchArgs)) || !f->hp_args.ensure(sizeof(ParserMatchArgs))) return false;
        // the list of the rows the fast pass hands to the fix-up launch (dev.hpp ParserMatchArgs::fix_list)
        if (!tile_in_lds && n < 0xFFFFFFFFull) {
            // (a region per wave of the launch: what a wave sees at most, rounded up to whole iterations; behind them one count per wave)
            const uint64_t nw = (uint64_t) grid * (uint64_t) (rx_threads / 64);
            const uint64_t stride = ((n + nw * 64 - 1) / (nw * 64)) * 64;
            if (!f->d_fix.ensure((nw * stride + nw + nw + 1) * sizeof(uint32_t))) return false;     // (lists, counts, the batches in front of every wave)
            ma.fix_list = f->d_fix.as<uint32_t>();
            ma.fix_count = (unsigned long long *) (f->d_fix.as<uint32_t>() + nw * stride);
        }
        // the rows in the order of their lengths (dev.hpp ParserMatchArgs::perm; note_lengths above decides from the last call's counters;
        // FLBGPU_SORT_ROWS=0 / 1: never / always, for measurements); a call launched ahead of its sizes keeps the chunk's order
        if (!tile_in_lds && ma.fix_list && n >= 4096) {
            static const int sort_env = getenv("FLBGPU_SORT_ROWS") ? atoi(getenv("FLBGPU_SORT_ROWS")) : -1;
            ma.len_stat = &dm->counts[16];
            if ((sort_env >= 0 ? sort_env != 0 : f->sort_rows) && !g_spec.on) {
                const size_t wb = row_perm_work_bytes(n);
                if (f->d_perm.ensure(n * sizeof(uint32_t)) && f->d_permwork.ensure(wb)) {
                    ProfScope ps(f, st, "k_row_perm");
                    if (!launch_row_perm(row_off, n, f->d_perm.as<uint32_t>(), f->d_permwork.p, wb, &dm->counts[16], st)) { set_err("filter_parser: ordering the rows by length failed"); return false; }
                    ma.perm = f->d_perm.as<uint32_t>();
                    ma.stage_nbuf = 0;
                    f->last_path |= 16u;
                }
            }
        }
        ma.self = f->d_args.as<ParserMatchArgs>();
        memcpy(f->hp_args.p, &ma, sizeof(ma));                   // (page-locked: the copy below is a real asynchronous transfer)
        HIPOK(hipMemcpyAsync(f->d_args.p, f->hp_args.p, sizeof(ma), hipMemcpyHostToDevice, st));
        // diagnostics: FLBGPU_TRACE=<iterations> FLBGPU_TRACE_FILE=<path> -- phase time stamps of every wave's first iterations
        void *d_trace = nullptr;
        size_t trace_bytes = 0;
        if (!tile_in_lds && getenv("FLBGPU_TRACE") && getenv("FLBGPU_TRACE_FILE")) {
            const int iters = atoi(getenv("FLBGPU_TRACE"));
            if (iters > 0 && iters <= 4096) {
                trace_bytes = (size_t) grid * (rx_threads / 64) * (size_t) iters * 8 * sizeof(unsigned long long);
                if (hipMalloc(&d_trace, trace_bytes) == hipSuccess && hipMemsetAsync(d_trace, 0, trace_bytes, st) == hipSuccess) {
                    ma.trace = (unsigned long long *) d_trace; ma.trace_iters = (uint32_t) iters | (getenv("FLBGPU_TRACE_FIXUP") ? 0x80000000u : 0u);
                }
            }
        }
        if (tile_in_lds) { ProfScope ps(f, st, "k_parser_tile"); launch_parser_tile(ma, grid, rx_threads, st); }
        else { ProfScope ps(f, st, "k_parser_reg"); launch_parser_reg(ma, grid, rx_threads, false, st); }
        if (ahead && g_spec.on && !tile_in_lds && ma.fix_list && !d_trace) {
            // launched ahead (SpecCall): the fix-up pass in the main pass's shape, every wave takes what it listed (usually nothing);
            // *hm_out is not valid before the caller's wait, and ahead_counters_ok is the caller's to ask
            { ProfScope ps(f, st, "k_parser_reg_fixup"); launch_parser_reg(ma, grid, rx_threads, true, st); }
            *ahead = true;
            *n_valid = n;
            return true;
        }
        auto trace_out = [&]() {
            std::vector<unsigned long long> ht(trace_bytes / sizeof(unsigned long long));
            if (hipMemcpyAsync(ht.data(), d_trace, trace_bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
                if (FILE *tf = fopen(getenv("FLBGPU_TRACE_FILE"), "wb")) {
                    const uint32_t hdr[4] = {(uint32_t) grid, (uint32_t) (rx_threads / 64), ma.trace_iters & 0x7FFFFFFFu, 8};
                    fwrite(hdr, sizeof(hdr), 1, tf); fwrite(ht.data(), 1, trace_bytes, tf); fclose(tf);
                }
            }
            (void) hipFree(d_trace);
            d_trace = nullptr;
            ma.trace = nullptr; ma.trace_iters = 0;
        };
        if (d_trace && !(ma.trace_iters >> 31)) trace_out();
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (f->last_fx5 && n >= 64 && f->parsers[0]->fx2b.ok && hm.counts[9] * 64 > n && !d_trace) {
            // The three-port tables hand more than one row in 64 of THIS chunk on (a try of the set-aside tables on data that has not
            // changed, or the first chunk of such data): the kernels behind the walk would now pay for every such row (6.1 ms of
            // k_parser_generic on bench.py's mixed shapes, where the pass itself takes 0.9) -- the call again from the top instead, with the
            // four-port tables (note_fx5's verdict, taken here; the caller sees `again`)
            f->fx5.bad();
            f->again = true;
            return false;
        }
        if (!tile_in_lds && hm.counts[10] > 0) {
            // rows the call-free kernel only flagged (another layout, the last records of the chunk): the same pass with the
            // general locate, guarded loads and the reverse-pass fallback, for those rows
            ma.fix_first = hm.counts[11] <= n ? n - hm.counts[11] : 0;
            uint64_t fix_blocks = ((n - (ma.fix_first & ~63ull)) + (uint64_t) rx_threads - 1) / (uint64_t) rx_threads;
            if (fix_blocks > (uint64_t) grid) fix_blocks = (uint64_t) grid;
            if (ma.fix_list) fix_blocks = (uint64_t) grid;                       // the same shape: wave w takes what wave w listed
            { ProfScope ps(f, st, "k_parser_reg_fixup"); launch_parser_reg(ma, (int) fix_blocks, rx_threads, true, st); }
            if (d_trace) trace_out();
            HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
        }
        if (d_trace) trace_out();
        note_lengths(f, hm);
        note_fx5(f, hm, n);
        note_unsettled(f, hm, n);
        if (hm.counts[8] > 0) {
            // rows whose time text needs the strptime interpreter, sizes that depend on the record's bytes, ...
            { ProfScope ps(f, st, "k_parser_finish"); launch_parser_finish(ma, cus, st); }
            HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
        }
        if (hm.counts[2] > 0 && !run_generic()) return false;
    }
    else {
        { ProfScope ps(f, st, "k_parser_locate"); launch_parser_locate(ma, cus, st); }
        { ProfScope ps(f, st, "k_parser_rx"); launch_parser_rx(ma, grid, rx_threads, st); }
        { ProfScope ps(f, st, "k_parser_finish"); launch_parser_finish(ma, cus, st); }
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (hm.counts[2] > 0 && !run_generic()) return false;
    }
    if (hm.first_bad < n) n = hm.first_bad;                 // the decoder loop ends at the first bad record
    *n_valid = n;
    return true;
}

// the record writer's per-field configuration for the kernel arguments (dev.hpp EmitCfg)
static void fill_emit_cfg(const flbgpu_filter *f, EmitCfg &ec) {
    memset(&ec, 0, sizeof(ec));
    const DevParser &d = f->parsers[0]->dev;
    if (d.is_json || d.kv_format || !d.plain_types || f->pcfg.reserve_data || f->pcfg.preserve_key || d.nfields > MAX_NAMES) return;
    for (int q = 0; q < d.nfields; q++) {
        const int end = d.kw_off[q] + (d.kw_bytes[q] + 3) / 4;
        if (d.kw_off[q] > 255 || d.kw_bytes[q] > 255 || end > (int) (sizeof(ec.keywords) / 4)) return;
        ec.kw_off[q] = (uint8_t) d.kw_off[q]; ec.kw_bytes[q] = (uint8_t) d.kw_bytes[q];
        memcpy(ec.keywords + d.kw_off[q], d.keywords + d.kw_off[q], (size_t) (end - d.kw_off[q]) * 4);
    }
    ec.nfields = d.nfields; ec.nregs_minus1 = d.nregs_minus1; ec.ok = 1;
}

static bool run_parser_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, hipStream_t st, int *ret) {
    uint64_t n = in->n;
    *ret = FLBGPU_FILTER_NOTOUCH;
    f->last_in = 0; f->last_out = 0;
    if (n == 0) return true;
    MiscWords *dm = nullptr, *hmp = nullptr;
    bool ahead = false;
    if (!parser_size_pass(f, in, st, &dm, &hmp, &n, nullptr, &ahead)) {
        if (f->again) { f->again = false; return run_parser_dev(f, in, out, st, ret); }
        return false;
    }
    MiscWords &hm = *hmp;
    uint64_t &total = *(uint64_t *) (f->hp_misc.as<uint8_t>() + sizeof(MiscWords));
    const uint64_t *row_off = in->row_off;
    const uint8_t *data = (const uint8_t *) in->data;
    const int cus = g_cus > 0 ? g_cus : 256;
    if (n == 0) return true;
    // rows whose winning parser has Decode_Field rules: sized again with the decoders applied (dec_dev.inc)
    DecArgs dca;
    int dec_blocks = 0;
    memset((void *) &dca, 0, sizeof(dca));
    if (f->has_decoders) {
        launch_max_row_len(row_off, n, &dm->max_row, st);
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        // a region holds a packed map or a decoded text: 3 x the longest record + room for the field names is never reached
        const uint64_t cap = (uint64_t) hm.max_row * 3 + 4096;
        uint64_t lanes = (n + 63) / 64 * 64;
        const uint64_t budget = (uint64_t) 1 << 31;
        while (lanes > 64 && lanes * DEC_REGIONS * cap > budget) lanes = (lanes / 2 + 63) / 64 * 64;
        if (lanes > 64 * 4096) lanes = 64 * 4096;
        if (cap > 0xFFFFFFF0ull || !f->d_dec.ensure(lanes * DEC_REGIONS * cap)) { set_err("filter_parser: no memory for the decoders' scratch"); return false; }
        dec_blocks = (int) (lanes / 64);
        dca.e.data = data; dca.e.row_off = row_off; dca.e.n = n; dca.e.cfg = f->pcfg; dca.e.parsers = f->d_parsers.as<DevParser>();
        dca.e.n_cols = in->n; dca.e.info = f->d_info.as<uint32_t>(); dca.e.caps = f->d_caps.as<uint32_t>(); dca.e.caps_stride = f->caps_stride;
        dca.e.null_mask = f->d_null.as<uint64_t>(); dca.e.out_len = f->d_len.as<uint32_t>(); dca.e.out_off = nullptr; dca.e.out = nullptr; dca.e.bytes = in->bytes;
        dca.info_w = f->d_info.as<uint32_t>(); dca.out_len_w = f->d_len.as<uint32_t>(); dca.scratch = f->d_dec.as<uint8_t>(); dca.cap = (uint32_t) cap;
        dca.mode = 0; dca.err = &dm->counts[10]; dca.ndec = &dm->counts[11];
        HIPOK(hipMemsetAsync(&dm->counts[10], 0, 2 * sizeof(unsigned long long), st));
        { ProfScope ps(f, st, "k_parser_dec_size"); launch_parser_dec(dca, dec_blocks, st); }
    }
    // write offsets + the number of emitted records (what flb_mp_count_log_records would report) in one pass
    { ProfScope ps(f, st, "k_scan"); launch_scan(f->d_len.as<uint32_t>(), n, f->d_scan_tmp.as<uint64_t>(), f->d_off.as<uint64_t>(), st, &dm->counts[1]); }
    total = 0;
    if (ahead) {
        // launched ahead (SpecCall): the writer with room for the usual output, counters + size + (host-level call) the output itself
        // written to page-locked memory by the device, ONE wait
        if (!f->d_out.ensure(in->bytes * 2 + n * 64 + 65536 + 16)) return false;
        ParserEmitArgs ea;
        ea.data = data; ea.row_off = row_off; ea.n = n; ea.cfg = f->pcfg; ea.parsers = f->d_parsers.as<DevParser>();
        ea.n_cols = in->n; ea.info = f->d_info.as<uint32_t>(); ea.caps = f->d_caps.as<uint32_t>(); ea.caps_stride = f->caps_stride;
        ea.null_mask = f->d_null.as<uint64_t>(); ea.out_len = f->d_len.as<uint32_t>(); ea.out_off = f->d_off.as<uint64_t>();
        ea.out = f->d_out.as<uint8_t>(); ea.bytes = in->bytes; ea.out_cap = f->d_out.cap - 16;
        fill_emit_cfg(f, ea.ec);
        { ProfScope ps(f, st, "k_parser_emit"); launch_parser_emit(ea, cus, st); }
        uint8_t *sink = g_spec.last ? g_spec.sink : nullptr;
        launch_finish_to_host(ea.out, ea.out_cap, ea.out_off + n, sink, g_spec.sink_cap, dm, &hm, (uint32_t) sizeof(hm), &total, st);
        HIPOK(hipStreamSynchronize(st));
        if (!ahead_counters_ok(f, hm, n) || hm.counts[3] > 0 || hm.ov_count > OV_CAP || total > ea.out_cap) {
            SpecOff usual;
            return run_parser_dev(f, in, out, st, ret);
        }
        f->last_in = hm.counts[0];
        if (total == 0) return true;
        out->data = f->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = total;
        f->last_out = hm.counts[1] - (hm.counts[14] < hm.counts[1] ? hm.counts[14] : hm.counts[1]);   // (see below)
        if (sink && total <= g_spec.sink_cap) g_spec.sunk = true;
        *ret = FLBGPU_FILTER_MODIFIED;
        return true;
    }
    HIPOK(hipMemcpyAsync(&total, f->d_off.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    f->last_in = hm.counts[0];
    if (hm.ov_count > OV_CAP) {
        set_err("filter_parser: more than %u parsed Key_Name entries at body map index 64 and up in one chunk", OV_CAP);
        return false;
    }
    if (total == 0) return true;                            // encoder produced nothing: NOTOUCH (+ error log)
    if (!f->d_out.ensure(total + 16)) return false;
    ParserEmitArgs ea;
    ea.data = data; ea.row_off = row_off; ea.n = n; ea.cfg = f->pcfg; ea.parsers = f->d_parsers.as<DevParser>();
    ea.n_cols = in->n; ea.info = f->d_info.as<uint32_t>(); ea.caps = f->d_caps.as<uint32_t>(); ea.caps_stride = f->caps_stride;
    ea.null_mask = f->d_null.as<uint64_t>(); ea.out_len = f->d_len.as<uint32_t>(); ea.out_off = f->d_off.as<uint64_t>();
    ea.out = f->d_out.as<uint8_t>(); ea.bytes = in->bytes; ea.out_cap = 0;
    fill_emit_cfg(f, ea.ec);
    { ProfScope ps(f, st, "k_parser_emit"); launch_parser_emit(ea, cus, st); }
    if (hm.counts[3] > 0) { ProfScope ps(f, st, "k_parser_emit_exact"); launch_parser_emit_exact(ea, st); }
    if (f->has_decoders && hm.counts[11] > 0) {
        dca.e.out_off = ea.out_off; dca.e.out = ea.out; dca.mode = 1;
        { ProfScope ps(f, st, "k_parser_dec_emit"); launch_parser_dec(dca, dec_blocks, st); }
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
    }
    HIPOK(hipStreamSynchronize(st));
    if (f->has_decoders && hm.counts[10] > 0) { set_err("filter_parser: a decoded field outgrew the decoders' scratch (%llu records)", hm.counts[10]); return false; }
    out->data = f->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = total;
    // rows with length 0 (dropped/skipped) remain as empty rows; a record whose PARSED time reads as a group marker to the decoder
    // (ff ff ff ff / fe: -1 s, -2 s, 2106-02-07 06:28:14 / :15) is written but not counted: flb_filter_do re-counts the filter's output
    // with the log event decoder (src/flb_filter.c:272, src/flb_mp.c:49-71), which hides it
    f->last_out = hm.counts[1] - (hm.counts[14] < hm.counts[1] ? hm.counts[14] : hm.counts[1]);
    *ret = FLBGPU_FILTER_MODIFIED;
    return true;
}

// a filter_grep's tables and names into the one-pass kernel's arguments (glane_kernels.inc): the tables behind `used` in LDS, every rule's
// top-level name in one of the slots the kernel's single walk looks for (two filters that run as one pass share them)
static bool lane_rules(const flbgpu_filter *f, GrepArgs &g, GrepLaneArgs &la, uint32_t &used) {
    const uint32_t room = grep_lane_table_room(), start = used;
    if (f->rules.empty() || f->rules.size() > (size_t) MAX_RULES) return false;
    for (size_t i = 0; i < f->rules.size(); i++) {
        const DevDfa &df = f->rules[i].dfa;
        const uint32_t blob = grep_lane_table_bytes((uint32_t) df.nD, (uint32_t) df.ncls);
        if ((uint64_t) df.nD * (uint64_t) (df.ncls + 1) >= 0xFFFEull) return false;      // (a cell holds a row's offset in 16 bits)
        g.rule_lds_off[i] = used; g.rule_lds_bytes[i] = blob;
        used += blob;
        if (used > room) return false;
    }
    g.rules_lds_total = used - start;
    for (size_t i = 0; i < f->rules.size(); i++) {
        const DevKey &k = f->rules[i].key;
        int slot = -1;
        if (k.key_len < 1 || k.key_len > 32) return false;
        uint32_t kw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(kw, k.key, (size_t) k.key_len);                                            // (little endian host, zero padded)
        for (int s = 0; s < la.nslots; s++)
            if (la.slot_klen[s] == (uint8_t) k.key_len && memcmp(la.slot_kw[s], kw, sizeof(kw)) == 0) { slot = s; break; }
        if (slot < 0) {
            if (la.nslots >= GREP_SLOTS) return false;
            slot = la.nslots++;
            la.slot_klen[slot] = (uint8_t) k.key_len;
            memcpy(la.slot_kw[slot], kw, sizeof(kw));
        }
        g.rule_slot[i] = (uint8_t) slot;
    }
    return true;
}

// the rows a wave of the one-pass kernel takes and the LDS they need.  A large chunk is asked (one small launch over the row offsets: the
// longest run of 64 rows), a small one gets the mean and half as much again (rows an earlier filter emptied come in runs); a record that
// does not fit all the same is decided and copied from the chunk itself
static bool lane_geometry(const flbgpu_dev_chunk *in, MiscWords *dm, hipStream_t st, GrepLaneArgs &la) {
    const uint64_t n = in->n, avg = in->bytes / n + 1;
    uint64_t R = 64, cap = (64 * avg * 3 / 2 + 512 + 15) & ~15ull;
    if (n >= 65536) {
        HIPOK(hipMemsetAsync(&dm->counts[13], 0, 8, st));
        launch_tile_max(in->row_off, n, 64, &dm->counts[13], st);
        unsigned long long mx = 0;
        HIPOK(hipMemcpyAsync(&mx, &dm->counts[13], 8, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        cap = (mx + 32 + 15) & ~15ull;
    }
    if (cap > (uint64_t) grep_lane_text_max()) {
        R = (uint64_t) (grep_lane_text_max() - 16) * 92 / 100 / avg;      // records longer than a quarter kilobyte: fewer of them to a wave
        cap = (uint64_t) grep_lane_text_max();
    }
    if (cap < 2048) cap = 2048;
    la.text_cap = (uint32_t) cap;
    if (R < 1) R = 1;
    if (R > 64) R = 64;
    la.ntiles = (n + R - 1) / R;
    la.rows_per_tile = (uint32_t) R;
    return true;
}

static bool grep_lane_off() {
    static const bool off = getenv("FLBGPU_GREP_LANE") && atoi(getenv("FLBGPU_GREP_LANE")) == 0;
    return off;
}

static bool run_grep_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, hipStream_t st, int *ret,
                         bool trailing_garbage) {
    uint64_t n = in->n;
    *ret = FLBGPU_FILTER_NOTOUCH;
    f->last_in = 0; f->last_out = 0;
    if (n == 0) return true;
    if (!f->d_misc.ensure(sizeof(MiscWords))) return false;
    MiscWords *dm = f->d_misc.as<MiscWords>();
    // host mirror of the counters + the output size, page-locked so that the small copies are real
    // asynchronous DMA transfers
    if (!f->hp_misc.ensure(sizeof(MiscWords) + 4 * sizeof(uint64_t))) return false;
    MiscWords &hm = *f->hp_misc.as<MiscWords>();
    uint64_t &total = *(uint64_t *) (f->hp_misc.as<uint8_t>() + sizeof(MiscWords));
    memset(&hm, 0, sizeof(hm));
    hm.first_bad = ~0ull;
    HIPOK(hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
    if (!f->d_len.ensure(n * sizeof(uint32_t)) || !f->d_status.ensure(n * sizeof(uint32_t)) ||
        !f->d_off.ensure((n + 1) * sizeof(uint64_t)) || !f->d_scan_tmp.ensure(scan_tmp_elems(n) * sizeof(uint64_t)))
        return false;
    GrepArgs ga;
    ga.data = (const uint8_t *) in->data; ga.row_off = in->row_off; ga.n = n; ga.bytes = in->bytes; ga.rules = f->d_rules.as<GrepRule>();
    ga.nrules = (int) f->rules.size(); ga.logical_op = f->logical_op; ga.keep_len = f->d_len.as<uint32_t>();
    ga.status = f->d_status.as<uint32_t>(); ga.first_bad = &dm->first_bad; ga.counts = dm->counts;
    {
        // rule DFA blobs into what the two resident workgroups leave of the LDS (8 KB each: 160 KB - 2 x 4 x 18 KB of tiles)
        uint32_t used = 0;
        const uint32_t room = getenv("FLBGPU_GREP_NO_LDS_RULES") ? 0 : 8192 - 64;
        for (size_t i = 0; i < f->rules.size() && i < (size_t) MAX_RULES; i++) {
            const DevDfa &df = f->rules[i].dfa;
            const uint32_t blob = (uint32_t) ((df.d_final + df.nD) - df.cls);
            ga.rule_lds_off[i] = 0xFFFFFFFFu; ga.rule_lds_bytes[i] = blob;
            if (used + ((blob + 15) & ~15u) <= room) { ga.rule_lds_off[i] = used; used += (blob + 15) & ~15u; }
        }
        ga.rules_lds_total = used;
    }
    {
        // the rules' distinct top-level keys: one map walk per record finds them all (kdev.inc grep_walk)
        ga.nslots = 0;
        memset(ga.slot_rule, 0, sizeof(ga.slot_rule)); memset(ga.rule_slot, 0, sizeof(ga.rule_slot));
        // (a single rule keeps its own lookup: one walk either way, and the slot compare costs 19 % there -- 1.79 against 2.13 ms per 10 M)
        bool fits = !getenv("FLBGPU_GREP_NO_HITS") && f->rules.size() >= 2 && f->rules.size() <= (size_t) MAX_RULES;
        for (size_t i = 0; fits && i < f->rules.size(); i++) {
            const DevKey &k = f->rules[i].key;
            int slot = -1;
            for (int s = 0; s < ga.nslots; s++) {
                const DevKey &o = f->rules[ga.slot_rule[s]].key;
                if (o.key_len == k.key_len && memcmp(o.key, k.key, (size_t) k.key_len) == 0) { slot = s; break; }
            }
            if (slot < 0) {
                if (ga.nslots >= GREP_SLOTS) { fits = false; break; }
                slot = ga.nslots++;
                ga.slot_rule[slot] = (uint8_t) i;
            }
            ga.rule_slot[i] = (uint8_t) slot;
        }
        if (!fits) ga.nslots = 0;
    }
    ga.host_collect = 0;
    if (f->has_host_rules) {
        // host rules, pass 1: where their values sit; the host's matcher answers; the answers go back as one bit per row
        const size_t words = (size_t) ((n + 31) / 32);
        size_t nh = 0;
        for (auto *b : f->host_rx) if (b) nh++;
        if (!f->d_hspans.ensure(nh * n * 2 * sizeof(uint32_t)) || !f->d_hbits.ensure(nh * words * sizeof(uint32_t))) return false;
        HIPOK(hipMemsetAsync(f->d_hspans.p, 0xFF, nh * n * 2 * sizeof(uint32_t), st));
        std::vector<GrepRule> rr = f->rules;
        {
            size_t k = 0;
            for (size_t i = 0; i < rr.size(); i++) if (f->host_rx[i]) { rr[i].host_mode = 1; rr[i].host_io = f->d_hspans.as<uint32_t>() + k * n * 2; k++; }
        }
        HIPOK(hipStreamSynchronize(st));
        HIPOK(hipMemcpy(f->d_rules.p, rr.data(), rr.size() * sizeof(GrepRule), hipMemcpyHostToDevice));
        ga.host_collect = 1;
        { ProfScope ps(f, st, "k_grep_match(values for the host rules)"); launch_grep_match(ga, g_cus > 0 ? g_cus : 256, st); }
        ga.host_collect = 0;
        std::vector<uint32_t> spans(nh * n * 2);
        HIPOK(hipMemcpyAsync(spans.data(), f->d_hspans.p, spans.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        std::vector<uint8_t> copy;
        std::vector<uint64_t> off;
        const uint8_t *hd = nullptr;
        if (!host_chunk(in, st, copy, &hd, off)) { set_err("filter_grep: copying the chunk back for the host rules failed"); return false; }
        std::vector<uint32_t> bits(nh * words, 0u);
        std::atomic<uint64_t> over{0}, vals{0};
        size_t k = 0;
        for (size_t i = 0; i < rr.size(); i++) {
            if (!f->host_rx[i]) continue;
            const uint32_t *sp = spans.data() + k * n * 2;
            uint32_t *bw = bits.data() + k * words;
            const rx::BtProgram *bt = f->host_rx[i];
            const uint64_t total_bytes = in->bytes;
            parallel_rows(n, [&, sp, bw, bt, total_bytes](uint64_t a, uint64_t b) {
                uint64_t ov = 0, nv = 0;
                for (uint64_t r = a; r < b; r++) {
                    const uint32_t len = sp[2 * r + 1], o = sp[2 * r];
                    if (len == 0xFFFFFFFFu || off[r] + o + (uint64_t) len > total_bytes || len > 0x7FFFFFF0u) continue;
                    const int res = rx::bt_search(bt, hd + off[r] + o, (int) len, nullptr, nullptr);
                    nv++;
                    if (res == -4) ov++;
                    if (res > 0) bw[r >> 5] |= 1u << (r & 31);
                }
                over += ov; vals += nv;
            });
            rr[i].host_mode = 2;
            rr[i].host_io = f->d_hbits.as<uint32_t>() + k * words;
            k++;
        }
        f->host_budget_over += over.load(); f->host_values += vals.load();
        HIPOK(hipMemcpy(f->d_hbits.p, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPOK(hipMemcpy(f->d_rules.p, rr.data(), rr.size() * sizeof(GrepRule), hipMemcpyHostToDevice));
        // (what pass 1 counted is dropped)
        memset(&hm, 0, sizeof(hm));
        hm.first_bad = ~0ull;
        HIPOK(hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
    }
    const bool ahead = g_spec.on;
    // ---- ONE pass (glane_kernels.inc): a lane decides its record, the kept records are placed by a look-back over the workgroups and
    // leave LDS at once -- no scan, no second read of the chunk.  Taken whenever every rule runs on the device with its tables in LDS
    // (a call launched ahead of its sizes, SpecCall, keeps the three launches: its writer must not wait for anybody).
    if (!grep_lane_off() && !ahead && !f->has_host_rules && ((uintptr_t) in->data & 15) == 0 && !f->rules.empty() && f->rules.size() <= (size_t) MAX_RULES) {
        GrepLaneArgs la;
        memset(&la, 0, sizeof(la));
        la.g = ga;
        uint32_t used = 0;
        // all the tables in LDS, every rule's top-level key in a slot (the filter's own, in the order of its rules: what the functions
        // of k_grep_match, which decide the odd records, go by as well)
        const bool fits = lane_rules(f, la.g, la, used);
        if (fits) {
            la.g.nslots = la.nslots;
            for (size_t i = 0; i < f->rules.size(); i++) la.g.slot_rule[la.g.rule_slot[i]] = (uint8_t) i;
            if (!lane_geometry(in, dm, st, la)) return false;
            const uint64_t nunits = grep_lane_units(la.ntiles);
            if (!f->d_units.ensure(nunits * 8 + 8) || !f->d_out.ensure(in->bytes + 32)) return false;
            HIPOK(hipMemsetAsync(f->d_units.p, 0, nunits * 8, st));
            la.off_out = f->d_off.as<uint64_t>(); la.out = f->d_out.as<uint8_t>(); la.out_cap = in->bytes;
            la.unit_state = f->d_units.as<unsigned long long>();
            la.ticket = &dm->counts[12]; la.words = &dm->counts[10];
            static const bool lane_prof = getenv("FLBGPU_GREP_PROF") != nullptr;      // (measurement only: the kernel stamps its phases)
            ScopedDevBuf d_prof;
            if (lane_prof) { if (!d_prof.ensure(64)) return false; HIPOK(hipMemsetAsync(d_prof.p, 0, 64, st)); la.prof = d_prof.as<unsigned long long>(); }
            { ProfScope ps(f, st, "k_grep_lane"); launch_grep_lane(la, g_cus > 0 ? g_cus : 256, st); }
            HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
            if (lane_prof) {
                unsigned long long ph[8];
                HIPOK(hipMemcpy(ph, d_prof.p, 64, hipMemcpyDeviceToHost));
                unsigned long long tot = 0;
                for (int i = 0; i < 8; i++) tot += ph[i];
                fprintf(stderr, "k_grep_lane phases (%% of %.0f cycles a record): staging %.1f, walk %.1f, values %.1f, automata %.1f, verdict %.1f, look-back %.1f, copy %.1f; text_cap %u rows %u\n",
                        (double) tot / (double) n, 100.0 * ph[0] / tot, 100.0 * ph[1] / tot, 100.0 * ph[2] / tot, 100.0 * ph[3] / tot, 100.0 * (ph[4] > ph[1] + ph[2] + ph[3] ? ph[4] - ph[1] - ph[2] - ph[3] : 0) / tot,
                        100.0 * ph[5] / tot, 100.0 * ph[6] / tot, la.text_cap, la.rows_per_tile);
            }
            if (hm.counts[11]) { set_err("filter_grep: the one-pass kernel %s", (hm.counts[11] >> 32) ? "gave up waiting for the workgroups in front" : "found no room for its output"); return false; }
            f->last_in = hm.counts[0];
            f->last_out = hm.counts[0];
            if (hm.first_bad != ~0ull || trailing_garbage) return true;           // (as below: a decoder error anywhere and the filter answers NOTOUCH)
            if (hm.counts[0] == hm.counts[1]) return true;
            f->last_out = hm.counts[1];
            out->data = f->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = hm.counts[10];
            *ret = FLBGPU_FILTER_MODIFIED;
            return true;
        }
    }
    { ProfScope ps(f, st, "k_grep_match"); launch_grep_match(ga, g_cus > 0 ? g_cus : 256, st); }
    { ProfScope ps(f, st, "k_scan"); launch_scan(f->d_len.as<uint32_t>(), n, f->d_scan_tmp.as<uint64_t>(), f->d_off.as<uint64_t>(), st); }
    total = 0;
    uint8_t *sink = ahead && g_spec.last ? g_spec.sink : nullptr;
    if (ahead) {
        // launched ahead (SpecCall): the kept records are never more than the chunk -- the gather needs no size; one wait
        if (!f->d_out.ensure(in->bytes + 32)) return false;
        GatherArgs ta;
        ta.data = (const uint8_t *) in->data; ta.row_off = in->row_off; ta.n = n; ta.keep_len = f->d_len.as<uint32_t>();
        ta.out_off = f->d_off.as<uint64_t>(); ta.out = f->d_out.as<uint8_t>(); ta.out_cap = f->d_out.cap - 16;
        { ProfScope ps(f, st, "k_gather"); launch_gather(ta, st); }
        launch_finish_to_host(ta.out, ta.out_cap, ta.out_off + n, sink, g_spec.sink_cap, dm, &hm, (uint32_t) sizeof(hm), &total, st);
    }
    else {
        HIPOK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&total, f->d_off.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    HIPOK(hipStreamSynchronize(st));
    f->last_in = hm.counts[0];
    f->last_out = hm.counts[0];
    // Any decoder error (bad event shape, truncated or malformed msgpack) ends the reference's
    // loop with an error code: whatever was kept, the filter answers NOTOUCH
    // (plugins/filter_grep/grep.c:356-385).  "we keep everything" compares RECORD COUNTS, and
    // group markers are not records (old_size == new_size, :362-371).
    if (hm.first_bad != ~0ull || trailing_garbage) return true;
    if (hm.counts[0] == hm.counts[1]) return true;
    f->last_out = hm.counts[1];
    if (ahead) {
        if (sink && total <= g_spec.sink_cap) g_spec.sunk = true;
    }
    else {
        if (!f->d_out.ensure(total + 16)) return false;
        GatherArgs ta;
        ta.data = (const uint8_t *) in->data; ta.row_off = in->row_off; ta.n = n; ta.keep_len = f->d_len.as<uint32_t>();
        ta.out_off = f->d_off.as<uint64_t>(); ta.out = f->d_out.as<uint8_t>(); ta.out_cap = 0;
        { ProfScope ps(f, st, "k_gather"); launch_gather(ta, st); }
        HIPOK(hipStreamSynchronize(st));
    }
    out->data = f->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = total;
    *ret = FLBGPU_FILTER_MODIFIED;
    return true;
}

// Two filter_grep instances that follow each other in a chain, as ONE pass of the one-pass kernel.  flb_filter_do hands the second
// instance what the first one keeps (src/flb_filter.c:247-269) and grep rewrites no record, so a record leaves the pair iff both keep
// it: one read of the chunk, one walk of a record for the names of both, one output.  What each instance would have reported on its
// own (records in and out, bytes out, MODIFIED or NOTOUCH: plugins/filter_grep/grep.c:356-385) comes from the pass's counters.
static bool grep_pair_fusable(const flbgpu_filter *a, const flbgpu_filter *b, const flbgpu_dev_chunk *in) {
    static const bool off = getenv("FLBGPU_GREP_PAIR") && atoi(getenv("FLBGPU_GREP_PAIR")) == 0;
    if (off || grep_lane_off() || g_spec.on) return false;
    if (a->kind != F_GREP || b->kind != F_GREP || a->has_host_rules || b->has_host_rules) return false;
    if (a->rules.empty() || b->rules.empty() || a == b) return false;
    return in->n > 0 && in->row_off && ((uintptr_t) in->data & 15) == 0;
}

// 1: done (s2 and *out filled; *out = *in when neither instance answers MODIFIED), 0: not for this pair (the instances run one by one)
static int run_grep_pair(flbgpu_filter *f, flbgpu_filter *f2, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, hipStream_t st,
                         bool trailing_garbage, flbgpu_chain_stat *s2) {
    const uint64_t n = in->n;
    GrepLaneArgs la;
    memset(&la, 0, sizeof(la));
    uint32_t used = 0;
    if (!lane_rules(f, la.g, la, used)) return 0;
    const int own = la.nslots;
    if (!lane_rules(f2, la.g2, la, used)) return 0;
    if (!f->d_misc.ensure(sizeof(MiscWords)) || !f->hp_misc.ensure(sizeof(MiscWords) + 4 * sizeof(uint64_t))) return 0;
    MiscWords *dm = f->d_misc.as<MiscWords>();
    MiscWords &hm = *f->hp_misc.as<MiscWords>();
    memset(&hm, 0, sizeof(hm));
    hm.first_bad = ~0ull;
    if (hipMemcpyAsync(dm, &hm, sizeof(hm), hipMemcpyHostToDevice, st) != hipSuccess) return 0;
    if (!f->d_len.ensure(n * sizeof(uint32_t)) || !f->d_status.ensure(n * sizeof(uint32_t)) || !f->d_off.ensure((n + 1) * sizeof(uint64_t))) return 0;
    GrepArgs &ga = la.g;
    ga.data = (const uint8_t *) in->data; ga.row_off = in->row_off; ga.n = n; ga.bytes = in->bytes; ga.rules = f->d_rules.as<GrepRule>();
    ga.nrules = (int) f->rules.size(); ga.logical_op = f->logical_op; ga.keep_len = f->d_len.as<uint32_t>();
    ga.status = f->d_status.as<uint32_t>(); ga.first_bad = &dm->first_bad; ga.counts = dm->counts;
    // (the odd records go through the functions of k_grep_match: the first instance's names are the first slots, its own walk finds
    // them; the second instance's rules look their values up one by one there)
    ga.nslots = own;
    for (size_t i = 0; i < f->rules.size(); i++) ga.slot_rule[ga.rule_slot[i]] = (uint8_t) i;
    GrepArgs &gb = la.g2;
    gb.data = ga.data; gb.row_off = ga.row_off; gb.n = n; gb.bytes = ga.bytes; gb.rules = f2->d_rules.as<GrepRule>();
    gb.nrules = (int) f2->rules.size(); gb.logical_op = f2->logical_op; gb.keep_len = ga.keep_len; gb.status = ga.status;
    gb.first_bad = ga.first_bad; gb.counts = ga.counts; gb.nslots = 0;
    la.two = 1;
    if (!lane_geometry(in, dm, st, la)) return 0;
    const uint64_t nunits = grep_lane_units(la.ntiles);
    if (!f->d_units.ensure(nunits * 8 + 8) || !f2->d_out.ensure(in->bytes + 32)) return 0;
    if (hipMemsetAsync(f->d_units.p, 0, nunits * 8, st) != hipSuccess) return 0;
    la.off_out = f->d_off.as<uint64_t>(); la.out = f2->d_out.as<uint8_t>(); la.out_cap = in->bytes;
    la.unit_state = f->d_units.as<unsigned long long>();
    la.ticket = &dm->counts[12]; la.words = &dm->counts[10];
    { ProfScope ps(f, st, "k_grep_lane(two instances)"); launch_grep_lane(la, g_cus > 0 ? g_cus : 256, st); }
    if (hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
    if (hm.counts[11]) return 0;
    f->calls++; f2->calls++;
    const uint64_t dec = hm.counts[0], kept1 = hm.counts[1], kept2 = hm.counts[2], bytes1 = hm.counts[3], bytes2 = hm.counts[10];
    memset(s2, 0, 2 * sizeof(*s2));
    *out = *in;
    f->last_in = f->last_out = dec; f2->last_in = f2->last_out = dec;
    s2[0].ret = s2[1].ret = FLBGPU_FILTER_NOTOUCH;
    s2[0].in_records = s2[0].out_records = s2[1].in_records = s2[1].out_records = dec;
    s2[0].out_bytes = s2[1].out_bytes = in->bytes;
    if (hm.first_bad != ~0ull || trailing_garbage) return 1;             // a decoder error: both instances meet it, both answer NOTOUCH
    const bool mod1 = kept1 != dec;
    if (mod1) {
        f->last_out = kept1;
        s2[0].ret = FLBGPU_FILTER_MODIFIED; s2[0].out_records = kept1; s2[0].out_bytes = bytes1;
        if (bytes1 == 0) {                                               // nothing left: flb_filter_do stops in front of the second instance
            memset(&s2[1], 0, sizeof(s2[1]));
            f2->last_in = f2->last_out = 0;
            memset(out, 0, sizeof(*out));
            out->data = f2->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = 0;
            return 1;
        }
        f2->last_in = f2->last_out = kept1;
        s2[1].in_records = s2[1].out_records = kept1; s2[1].out_bytes = bytes1;
    }
    const bool mod2 = kept2 != (mod1 ? kept1 : dec);
    if (mod2) {
        f2->last_out = kept2;
        s2[1].ret = FLBGPU_FILTER_MODIFIED; s2[1].out_records = kept2; s2[1].out_bytes = bytes2;
    }
    if (mod1 || mod2) { out->data = f2->d_out.p; out->row_off = f->d_off.as<uint64_t>(); out->n = n; out->bytes = bytes2; }
    return 1;
}

// the rule gate of a filter_log_to_metrics whose rules run on the host (host_int.hpp flbgpu_filter::l2m_gate)
bool l2m_gate_dev(flbgpu_filter *gate, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *kept, hipStream_t st) {
    const bool last0 = g_spec.last;
    g_spec.last = false;                                   // (its output is no stage's output: nothing goes to the caller's slab)
    int ret = FLBGPU_FILTER_NOTOUCH;
    memset(kept, 0, sizeof(*kept));
    const bool ok = run_grep_dev(gate, in, kept, st, &ret, false);
    g_spec.last = last0;
    if (!ok) return false;
    if (gate->hp_misc.p && gate->hp_misc.as<MiscWords>()->first_bad != ~0ull) {
        // (grep answers NOTOUCH on a decoder error; the reference's log_to_metrics counts the records in front of it -- with the rules
        // on the host that prefix is not rebuilt: the chunk is not counted, loudly)
        set_err("log_to_metrics: a record of the chunk does not decode and the filter's rules run on the host: the chunk is not counted");
        return false;
    }
    if (ret != FLBGPU_FILTER_MODIFIED) *kept = *in;         // every record passes
    return true;
}

// ------------------------------------------------------------------------------------------ fused pair
// flb_filter_do over [filter_parser, filter_grep] without the parsed chunk (fused_kernels.inc).  Only grep's output
// and the two filters' record / byte counts are observable, so filter_parser's emit pass, grep's decode of the
// parsed chunk and the gather are replaced by: rules evaluated on the capture spans (k_pg_decide), a scan, and an
// emit of the kept records only (k_pg_emit).
// Configuration covered: one Format regex parser without Types casts, plain Key_Name, Reserve_Data / Preserve_Key
// off; any rule set.
static bool pair_fusable(const flbgpu_filter *fp, const flbgpu_filter *fg) {
    if (getenv("FLBGPU_NO_FUSE")) return false;
    if (fp->kind != F_PARSER || fg->kind != F_GREP) return false;
    if (fp->has_decoders) return false;
    if (fp->pcfg.nparsers != 1 || fp->pcfg.key.is_ra || fp->pcfg.reserve_data || fp->pcfg.preserve_key) return false;
    const DevParser &d = fp->parsers[0]->dev;
    if (d.is_json || !d.plain_types || d.nfields > 32 || d.nfields == 0 || d.nregs_minus1 <= 0) return false;
    if (fg->rules.empty()) return false;
    // a rule (or the parser) whose non-ASCII side is the NFA engine: the fused kernels do not carry that walker (it would sit, as a
    // call, inside the hot single-pass kernel): the unfused kernels take the pair
    if (d.utf8.nfa_on || d.ascii.stub) return false;
    for (const GrepRule &r : fg->rules) if (r.utf8.nfa_on) return false;
    if (fp->parsers[0]->bt || fp->host_list || fg->has_host_rules) return false;  // (host rules: the unfused kernels)
    return true;
}

// 1: ran fused (*out / stats set), 0: this chunk needs the unfused kernels, -1: failure
static std::atomic<uint64_t> g_fused_failures{0};          // fused passes that failed on the device (flbgpu_diag_fused_failures)
extern "C" uint64_t flbgpu_diag_fused_failures(void) { return g_fused_failures.load(); }

namespace flbgpu { void launch_pg_emit_plain(const PgEmitArgs &a, int nfields, int cus, hipStream_t st); }   // (fused_kernels.inc: k_pg_emit<true>)

static int run_pair_fused(flbgpu_filter *fp, flbgpu_filter *fg, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, flbgpu_chain_stat *stats2) {
    hipStream_t st = fp->stream;
    uint64_t n = in->n;
    if (n == 0) return 0;
    MiscWords *dm = nullptr, *hmp = nullptr;
    const DevParser &d0 = fp->parsers[0]->dev;
    if (!fp->d_keep.ensure(n * sizeof(uint32_t))) return -1;
    // grep's rules for the inline evaluation (k_parser_rx on the spans in LDS) and for k_pg_decide
    static thread_local PairCtx pc;                    // (page of host memory the async upload reads from)
    memset(&pc.pg, 0, sizeof(pc.pg));
    pc.keep_len = fp->d_keep.as<uint32_t>();
    pc.fg = fg;
    // row descriptors of the kept records (dev.hpp ParserMatchArgs::desc): 7 dwords + the spans as u16 pairs, whole 64-byte sectors
    pc.dstride = (uint32_t) ((7 + d0.nfields + 15) / 16 * 16);
    pc.desc = nullptr;
    if (!getenv("FLBGPU_NO_DESC") && fp->d_desc.ensure(n * (size_t) pc.dstride * sizeof(uint32_t))) pc.desc = fp->d_desc.as<uint32_t>();
    pc.pg.rules = fg->d_rules.as<GrepRule>(); pc.pg.nrules = (int) fg->rules.size(); pc.pg.logical_op = fg->logical_op;
    for (int f = 0; f < d0.nfields; f++) {
        if (!d0.field_is_time[f]) continue;
        if (d0.time_keep) pc.pg.time_fields |= 1u << f; else pc.pg.static_drop |= 1u << f;
    }
    for (size_t i = 0; i < fg->rules.size(); i++) {
        // the parser's named fields the rule's key names (flb_ra_key.c:118: the last one present decides)
        uint32_t m = 0;
        const DevKey &k = fg->rules[i].key;
        for (int f = 0; f < d0.nfields; f++)
            if (d0.field_name_len[f] == k.key_len && !memcmp(d0.names + d0.field_name_off[f], k.key, (size_t) k.key_len)) m |= 1u << f;
        pc.pg.rule_fmask[i] = m;
        // the rule's match-only DFA (cls | ddelta | d_final, one blob: upload_dfa)
        const DevDfa &df = fg->rules[i].dfa;
        pc.pg.rule_lds_bytes[i] = (uint32_t) ((df.d_final + df.nD) - df.cls);
    }
    // filter_parser's pass 1: every record sized (out_len), spans and record columns in HBM; keep_len of the records
    // whose rules could be settled on the spans
    if (!g_spec.on && hipMemsetAsync(pc.keep_len, 0xFF, n * sizeof(uint32_t), st) != hipSuccess) return -1;      // (ahead: parser_size_pass, with the counters)
    bool ahead = false;
    if (!parser_size_pass(fp, in, st, &dm, &hmp, &n, &pc, &ahead)) {
        if (fp->again) { fp->again = false; return run_pair_fused(fp, fg, in, out, stats2); }
        return -1;
    }
    MiscWords &hm = *hmp;
    uint64_t &total = *(uint64_t *) (fp->hp_misc.as<uint8_t>() + sizeof(MiscWords));
    // (counts[14]: a parsed time filter_grep's decoder would read as a group marker -- the unfused kernels restate that, the pair does not)
    if (!ahead && (n == 0 || hm.counts[3] > 0 || hm.ov_count > 0 || hm.counts[14] > 0)) return 0;           // (records for k_parser_emit_exact: unfused)
    const int cus = g_cus > 0 ? g_cus : 256;
    PgDecideArgs da;
    memset(&da, 0, sizeof(da));
    da.data = (const uint8_t *) in->data; da.row_off = in->row_off; da.n = n; da.n_cols = in->n;
    da.info = fp->d_info.as<uint32_t>(); da.caps = fp->d_caps.as<uint32_t>(); da.out_len = fp->d_len.as<uint32_t>();
    da.rules = pc.pg.rules; da.nrules = pc.pg.nrules; da.logical_op = pc.pg.logical_op;
    for (int i = 0; i < pc.pg.nrules; i++) {
        da.rule_fmask[i] = pc.pg.rule_fmask[i];
        da.rule_lds_off[i] = 0xFFFFFFFFu;
        if (da.lds_total + pc.pg.rule_lds_bytes[i] <= 48 * 1024) { da.rule_lds_off[i] = da.lds_total; da.rule_lds_bytes[i] = pc.pg.rule_lds_bytes[i]; da.lds_total += (pc.pg.rule_lds_bytes[i] + 15) & ~15u; }
    }
    da.keep_len = pc.keep_len; da.counts = dm->counts;
    // rows the inline evaluation left open (unparsed records, records of the generic kernel, rules on a kept time field)
    if (ahead || hm.counts[7] > 0 || hm.counts[2] > 0) { ProfScope ps(fp, st, "k_pg_decide"); launch_pg_decide(da, cus, st); }
    { ProfScope ps(fp, st, "k_scan"); launch_scan(da.keep_len, n, fp->d_scan_tmp.as<uint64_t>(), fp->d_off.as<uint64_t>(), st); }
    total = 0;
    auto emit_args = [&](PgEmitArgs &ea) {
        memset(&ea, 0, sizeof(ea));
        ea.data = da.data; ea.row_off = da.row_off; ea.cfg = fp->pcfg; ea.parsers = fp->d_parsers.as<DevParser>(); ea.n_cols = in->n;
        ea.info = da.info; ea.caps = da.caps; ea.null_mask = fp->d_null.as<uint64_t>();
        ea.keep_len = da.keep_len; ea.n = n; ea.out_off = fp->d_off.as<uint64_t>(); ea.out = fg->d_out.as<uint8_t>();
        ea.desc = pc.desc; ea.dstride = pc.dstride; ea.bytes = in->bytes;
        ea.counts = dm->counts;
        fill_emit_cfg(fp, ea.ec);
    };
    // RF_TIMEPEND rows k_pg_emit could not settle (a time text the fixed-layout plan refuses, a time the encoder refuses): the call
    // is repeated with the lookup inside the single pass, where the strptime interpreter stands behind the plan -- and stays there
    auto defer_failed = [&](unsigned long long c13) -> bool {
        if (!(fp->last_path & 4u)) return false;                // (the lookup was inside the pass)
        if (c13 == 0) { fp->defer.good(); return false; }
        fp->defer.bad();
        return true;
    };
    // the plain build of k_pg_emit: a kept record it leaves alone (counts[15]) is written by the general build behind it; when that is more
    // than 1 kept record in 16 the general build alone runs the next calls
    auto plain_wanted = [&](const PgEmitArgs &ea) -> bool {
        const bool can = ea.ec.ok && pc.desc != nullptr && !getenv("FLBGPU_EMIT_GENERAL");
        const bool use = can && fp->plain.use(fp->calls);
        if (use) fp->last_path |= 8u;
        return use;
    };
    auto plain_result = [&](unsigned long long left, unsigned long long kept) { if (left * 16 > kept && kept >= 64) fp->plain.bad(); else fp->plain.good(); };
    if (ahead) {
        // launched ahead (SpecCall): the writer with room for the usual output, counters + size + (host-level call) the output itself
        // written to page-locked memory by the device, ONE wait
        if (!fg->d_out.ensure(in->bytes * 2 + n * 64 + 65536 + 16)) return -1;
        PgEmitArgs ea;
        emit_args(ea);
        ea.out_cap = fg->d_out.cap - 16;
        // (the plain build when the parser's fields are plain strings and the descriptors exist; a record it does not take -- counts[15] --
        // sends the call round again the usual way, where the general build follows)
        const bool plain_emit = plain_wanted(ea);
        { ProfScope ps(fp, st, "k_pg_emit"); if (plain_emit) launch_pg_emit_plain(ea, d0.nfields, cus, st); else launch_pg_emit(ea, d0.nfields, cus, st); }
        uint8_t *sink = g_spec.last ? g_spec.sink : nullptr;
        launch_finish_to_host(ea.out, ea.out_cap, ea.out_off + n, sink, g_spec.sink_cap, dm, &hm, (uint32_t) sizeof(hm), &total, st);
        if (hipStreamSynchronize(st) != hipSuccess) return -1;
        if (plain_emit) plain_result(hm.counts[15], hm.counts[5]);
        if (defer_failed(hm.counts[13]) || (plain_emit && hm.counts[15] > 0) || !ahead_counters_ok(fp, hm, n) || hm.counts[3] > 0 || hm.counts[14] > 0 || hm.ov_count > 0 || total > ea.out_cap) {
            SpecOff usual;
            return run_pair_fused(fp, fg, in, out, stats2);
        }
        if (hm.counts[6] == 0 || hm.counts[5] == hm.counts[6]) return 0;
        fp->last_in = hm.counts[0]; fp->last_out = hm.counts[6];
        fg->last_in = hm.counts[6]; fg->last_out = hm.counts[5];
        if (stats2) {
            stats2[0].ret = FLBGPU_FILTER_MODIFIED; stats2[0].in_records = hm.counts[0]; stats2[0].out_records = hm.counts[6]; stats2[0].out_bytes = hm.counts[4];
            stats2[1].ret = FLBGPU_FILTER_MODIFIED; stats2[1].in_records = hm.counts[6]; stats2[1].out_records = hm.counts[5]; stats2[1].out_bytes = total;
        }
        memset(out, 0, sizeof(*out));
        if (total == 0) return 1;
        out->data = fg->d_out.p; out->row_off = fp->d_off.as<uint64_t>(); out->n = n; out->bytes = total;
        if (sink && total <= g_spec.sink_cap) g_spec.sunk = true;
        return 1;
    }
    if (hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&total, fp->d_off.as<uint64_t>() + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return -1;
    // a parser that emits nothing (NOTOUCH: grep would see the ORIGINAL chunk) or a grep that keeps everything
    // (NOTOUCH: the chain's output is the parser's): the unfused kernels
    if (hm.counts[6] == 0 || hm.counts[5] == hm.counts[6] || hm.counts[14] > 0) return 0;
    fp->last_in = hm.counts[0]; fp->last_out = hm.counts[6];
    fg->last_in = hm.counts[6]; fg->last_out = hm.counts[5];
    if (stats2) {
        stats2[0].ret = FLBGPU_FILTER_MODIFIED; stats2[0].in_records = hm.counts[0]; stats2[0].out_records = hm.counts[6]; stats2[0].out_bytes = hm.counts[4];
        stats2[1].ret = FLBGPU_FILTER_MODIFIED; stats2[1].in_records = hm.counts[6]; stats2[1].out_records = hm.counts[5]; stats2[1].out_bytes = total;
    }
    memset(out, 0, sizeof(*out));
    if (total == 0) return 1;                               // every record dropped: MODIFIED with an empty output
    if (!fg->d_out.ensure(total + 16)) return -1;
    PgEmitArgs ea;
    emit_args(ea);
    const bool plain_emit = plain_wanted(ea);
    { ProfScope ps(fp, st, "k_pg_emit"); if (plain_emit) launch_pg_emit_plain(ea, d0.nfields, cus, st); else launch_pg_emit(ea, d0.nfields, cus, st); }
    unsigned long long *c13 = (unsigned long long *) (fp->hp_misc.as<uint8_t>() + sizeof(MiscWords) + sizeof(uint64_t));     // counts[13 .. 15]
    c13[0] = c13[1] = c13[2] = 0;
    if (hipMemcpyAsync(c13, &dm->counts[13], 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (defer_failed(c13[0])) return run_pair_fused(fp, fg, in, out, stats2);
    if (plain_emit) plain_result(c13[2], hm.counts[5]);
    if (plain_emit && c13[2] > 0) {
        // records the plain build left alone (no descriptor, another parser's, larger than its staging area): the general build over the chunk
        { ProfScope ps(fp, st, "k_pg_emit_general"); launch_pg_emit(ea, d0.nfields, cus, st); }
        if (hipStreamSynchronize(st) != hipSuccess) return -1;
    }
    out->data = fg->d_out.p; out->row_off = fp->d_off.as<uint64_t>(); out->n = n; out->bytes = total;
    return 1;
}

// one cb_filter call on a device-resident chunk; `garbage` = undecodable bytes follow the rows
static int run_any_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, hipStream_t st, bool garbage) {
    int ret = FLBGPU_FILTER_NOTOUCH;
    bool ok;
    f->calls++;
    if (f->kind == F_L2M) {
        ok = run_l2m_dev(f, in, st, &ret);
        if (ok && ret == FLBGPU_FILTER_MODIFIED && out) memset(out, 0, sizeof(*out));   // discard_logs: every record dropped
    }
    else ok = f->kind == F_PARSER ? run_parser_dev(f, in, out, st, &ret) : run_grep_dev(f, in, out, st, &ret, garbage);
    if (!ok) return FLBGPU_FILTER_NOTOUCH;      // errors degrade to NOTOUCH (SURVEY 8b "Errors")
    return ret;
}

// Whether the bytes behind the last whole record end where msgpack-c's executor would have consumed
// everything it was given: it takes complete fields (a type byte, a length field, a payload) and stops in
// front of the first one that is cut short, so a chunk that ends exactly on a field boundary leaves the
// decoder's offset at the end of the buffer -- which flb_log_event_decoder_get_last_result and filter_grep
// read as a clean end (lib/msgpack-c/include/msgpack/unpack_template.h:242-247,439-447;
// src/flb_log_event_decoder.c:334-342; plugins/filter_grep/grep.c:357-360).  The reserved byte 0xc1 and a
// 33rd open container (MSGPACK_EMBED_STACK_SIZE) are errors wherever they stand.
static bool tail_is_clean(const uint8_t *d, size_t len, size_t start) {
    uint64_t count[32];
    int top = 0;
    size_t p = start;
    for (;;) {
        if (p >= len) return true;
        const uint8_t c = d[p];
        const size_t q = p + 1;
        size_t e, k = 0, lb = 0;
        uint64_t n = 0;
        bool container = false;
        if (c <= 0x7f || c >= 0xe0 || c == 0xc0 || c == 0xc2 || c == 0xc3) e = q;
        else if (c == 0xc1) return false;
        else if (c >= 0xa0 && c <= 0xbf) { k = c & 0x1f; if (len - q < k) return q == len; e = q + k; }
        else if (c >= 0x90 && c <= 0x9f) { container = true; n = c & 0x0f; e = q; }
        else if (c >= 0x80 && c <= 0x8f) { container = true; n = 2u * (c & 0x0f); e = q; }
        else {
            switch (c) {
            case 0xcc: case 0xd0: k = 1; break;
            case 0xcd: case 0xd1: k = 2; break;
            case 0xce: case 0xd2: case 0xca: k = 4; break;
            case 0xcf: case 0xd3: case 0xcb: k = 8; break;
            case 0xd4: k = 2; break;
            case 0xd5: k = 3; break;
            case 0xd6: k = 5; break;
            case 0xd7: k = 9; break;
            case 0xd8: k = 17; break;
            case 0xc4: case 0xc7: case 0xd9: lb = 1; break;
            case 0xc5: case 0xc8: case 0xda: case 0xdc: case 0xde: lb = 2; break;
            default: lb = 4; break;                          // c6 c9 db dd df
            }
            if (lb == 0) { if (len - q < k) return q == len; e = q + k; }
            else {
                if (len - q < lb) return q == len;
                for (size_t i = 0; i < lb; i++) n = (n << 8) | d[q + i];
                const size_t q2 = q + lb;
                if (c == 0xdc || c == 0xdd) { container = true; e = q2; }
                else if (c == 0xde || c == 0xdf) { container = true; n *= 2; e = q2; }
                else {
                    k = (size_t) n + ((c == 0xc7 || c == 0xc8 || c == 0xc9) ? 1 : 0);       // ext: type byte + data
                    if (len - q2 < k) return q2 == len;
                    e = q2 + k;
                }
            }
        }
        if (container) {
            if (top >= 32) return false;
            if (n > 0) { count[top++] = n; p = e; continue; }
        }
        p = e;
        for (;;) {
            if (top == 0) return false;                      // a whole object after all: the caller's walk said otherwise
            if (--count[top - 1] > 0) break;
            top--;
        }
    }
}

extern "C" int flbgpu_tail_clean_host(const void *data, size_t bytes, size_t consumed) {
    return consumed >= bytes || tail_is_clean((const uint8_t *) data, bytes, consumed) ? 1 : 0;
}

// A device chunk handed over without its offset column (row_off == NULL): the filter's indexer finds
// the records first (flbgpu_index_dev).  Returns false on failure; *garbage = undecodable bytes follow.
bool flbgpu::resolve_raw_chunk(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *resolved, bool *garbage) {
    *resolved = *in;
    *garbage = false;
    if (in->row_off != nullptr || in->bytes == 0) return true;
    if (!f->indexer) f->indexer = flbgpu_indexer_create();
    if (!f->indexer) return false;
    size_t consumed = 0;
    if (flbgpu_index_dev(f->indexer, in->data, (size_t) in->bytes, resolved, &consumed) < 0) return false;
    *garbage = false;
    if (consumed != in->bytes) {
        // the unfinished object behind the last record (at most one record long) decides: cut inside a
        // field = a decoder error, cut on a field boundary = a clean end
        const size_t tl = (size_t) in->bytes - consumed;
        std::vector<uint8_t> tail(tl);
        if (hipMemcpy(tail.data(), (const uint8_t *) in->data + consumed, tl, hipMemcpyDeviceToHost) != hipSuccess) return false;
        *garbage = !tail_is_clean(tail.data(), tl, 0);
    }
    return true;
}

// nothing decodes: every callback's loop ends at once; log_to_metrics still answers
// MODIFIED/empty when discard_logs is set (log_to_metrics.c:1141-1145)
static int empty_chunk_result(flbgpu_filter *const *filters, int nfilters, bool garbage, flbgpu_dev_chunk *out) {
    flbgpu_filter *f = filters[0];
    if (nfilters == 1 && f->kind == F_L2M) {
        flbgpu_dev_chunk in0;
        memset(&in0, 0, sizeof(in0));
        if (run_any_dev(f, &in0, out, f->stream, garbage) == FLBGPU_FILTER_MODIFIED) return FLBGPU_FILTER_MODIFIED;
    }
    return FLBGPU_FILTER_NOTOUCH;
}

extern "C" int flbgpu_filter_run_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, void *stream) {
    flbgpu_dev_chunk r;
    bool garbage = false;
    if (!resolve_raw_chunk(f, in, &r, &garbage)) return FLBGPU_FILTER_NOTOUCH;
    if (in->row_off == nullptr && in->bytes > 0 && r.n == 0) {
        flbgpu_filter *one[1] = {f};
        return empty_chunk_result(one, 1, garbage, out);
    }
    return run_any_dev(f, &r, out, stream ? (hipStream_t) stream : f->stream, garbage);
}

// flb_filter_do (src/flb_filter.c:121-325) over device-resident chunks: every MODIFIED output
// becomes the next filter's input, an empty MODIFIED output ends the chain (:247-269), a NOTOUCH
// filter leaves the working chunk alone.
static int chain_dev(flbgpu_filter *const *filters, int n, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, bool garbage,
                     flbgpu_chain_stat *stats) {
    flbgpu_dev_chunk cur = *in;
    bool modified = false;
    for (int i = 0; i < n; i++) {
        flbgpu_dev_chunk o;
        memset(&o, 0, sizeof(o));
        if (i + 1 < n && pair_fusable(filters[i], filters[i + 1])) {
            // the pair in one pass; the parser's stage always answers MODIFIED when this path completes
            flbgpu_chain_stat s2[2];
            memset(s2, 0, sizeof(s2));
            g_spec.last = i + 2 == n;
            filters[i]->calls++;
            const int fr = run_pair_fused(filters[i], filters[i + 1], &cur, &o, s2);
            if (fr == 1) {
                if (stats) { stats[i] = s2[0]; stats[i + 1] = s2[1]; }
                modified = true;
                cur = o;
                i++;
                if (o.bytes == 0) {
                    if (stats) for (int j = i + 1; j < n; j++) { memset(&stats[j], 0, sizeof(stats[j])); }
                    break;
                }
                continue;
            }
            if (fr < 0) {
                // a device / allocation failure inside the single pass (0 = "not fusable for this chunk"): the reference's
                // filter would log and hand the data on untouched (SURVEY 8b "Errors"); the error text stays in flbgpu_last_error
                g_fused_failures.fetch_add(1, std::memory_order_relaxed);
                *out = *in;
                return FLBGPU_FILTER_NOTOUCH;
            }
        }
        if (i + 1 < n && grep_pair_fusable(filters[i], filters[i + 1], &cur)) {
            flbgpu_chain_stat s2[2];
            if (run_grep_pair(filters[i], filters[i + 1], &cur, &o, filters[i]->stream, garbage && !modified, s2) == 1) {
                if (stats) { stats[i] = s2[0]; stats[i + 1] = s2[1]; }
                const bool mod = s2[0].ret == FLBGPU_FILTER_MODIFIED || s2[1].ret == FLBGPU_FILTER_MODIFIED;
                i++;
                if (!mod) continue;
                modified = true;
                cur = o;
                if (o.bytes == 0) {
                    if (stats) for (int j = i + 1; j < n; j++) { memset(&stats[j], 0, sizeof(stats[j])); }
                    break;
                }
                continue;
            }
        }
        g_spec.last = i + 1 == n;
        int ret = run_any_dev(filters[i], &cur, &o, filters[i]->stream, garbage && !modified);
        if (stats) {
            stats[i].ret = ret;
            stats[i].in_records = filters[i]->last_in;
            stats[i].out_records = ret == FLBGPU_FILTER_MODIFIED ? filters[i]->last_out : filters[i]->last_in;
            stats[i].out_bytes = ret == FLBGPU_FILTER_MODIFIED ? o.bytes : cur.bytes;
        }
        if (ret != FLBGPU_FILTER_MODIFIED) continue;
        modified = true;
        cur = o;
        if (o.bytes == 0) {                     // all records removed, no data to continue processing
            if (stats) for (int j = i + 1; j < n; j++) { memset(&stats[j], 0, sizeof(stats[j])); }
            break;
        }
    }
    *out = cur;
    return modified ? FLBGPU_FILTER_MODIFIED : FLBGPU_FILTER_NOTOUCH;
}

extern "C" int flbgpu_filter_chain_run_dev(flbgpu_filter *const *filters, int nfilters, const flbgpu_dev_chunk *in,
                                           flbgpu_dev_chunk *out, flbgpu_chain_stat *stats) {
    if (stats) memset(stats, 0, sizeof(*stats) * (size_t) (nfilters > 0 ? nfilters : 0));
    if (nfilters <= 0) return FLBGPU_FILTER_NOTOUCH;
    flbgpu_dev_chunk r;
    bool garbage = false;
    if (!resolve_raw_chunk(filters[0], in, &r, &garbage)) return FLBGPU_FILTER_NOTOUCH;
    if (in->row_off == nullptr && in->bytes > 0 && r.n == 0) return empty_chunk_result(filters, nfilters, garbage, out);
    g_spec = SpecCall();
    g_spec.on = spec_wanted(r.n, r.bytes);
    const int ret = chain_dev(filters, nfilters, &r, out, garbage, stats);
    g_spec = SpecCall();
    return ret;
}

// ------------------------------------------------------------------------------------------ host indexer
static inline bool h_skip(const uint8_t *d, size_t len, size_t *pos) {
    size_t p = *pos;
    uint64_t remaining = 1;
    // the usual event head [[ext8(type 0) ...], meta, body]: four objects in one compare
    if (len - p >= 13 && d[p] == 0x92 && d[p + 1] == 0x92 && d[p + 2] == 0xd7) {
        p += 12;
        remaining = 2;
        // ... and what an input plugin appends for a line of text -- no metadata, {one short key: a str} -- without the loop
        if (len - p >= 8 && d[p] == 0x80 && d[p + 1] == 0x81 && (d[p + 2] & 0xe0) == 0xa0) {
            const size_t v = p + 3 + (d[p + 2] & 31);
            if (len - p >= 3 + (size_t) (d[p + 2] & 31) + 5) {
                const uint8_t c = d[v];
                size_t e = 0;
                if ((c & 0xe0) == 0xa0) e = v + 1 + (c & 31);
                else if (c == 0xd9) e = v + 2 + d[v + 1];
                else if (c == 0xda) e = v + 3 + (((size_t) d[v + 1] << 8) | d[v + 2]);
                else if (c == 0xdb) e = v + 5 + (((size_t) d[v + 1] << 24) | ((size_t) d[v + 2] << 16) | ((size_t) d[v + 3] << 8) | d[v + 4]);
                if (e && e <= len) { *pos = e; return true; }
            }
        }
    }
    while (remaining > 0) {
        if (p >= len) return false;
        uint8_t c = d[p++];
        size_t need = 0, payload = 0;
        uint64_t kids = 0;
        if (c <= 0x7f || c >= 0xe0) { }
        else if (c >= 0xa0 && c <= 0xbf) payload = c & 31;
        else if (c >= 0x90 && c <= 0x9f) kids = c & 15;
        else if (c >= 0x80 && c <= 0x8f) kids = 2ull * (c & 15);
        else {
            int kind = 0;   // 1 payload-length, 2 array count, 3 map count, 4 ext
            switch (c) {
            case 0xc0: case 0xc2: case 0xc3: break;
            case 0xc1: return false;
            case 0xc4: need = 1; kind = 1; break;
            case 0xc5: need = 2; kind = 1; break;
            case 0xc6: need = 4; kind = 1; break;
            case 0xc7: need = 1; kind = 4; break;
            case 0xc8: need = 2; kind = 4; break;
            case 0xc9: need = 4; kind = 4; break;
            case 0xca: payload = 4; break;
            case 0xcb: payload = 8; break;
            case 0xcc: case 0xd0: payload = 1; break;
            case 0xcd: case 0xd1: payload = 2; break;
            case 0xce: case 0xd2: payload = 4; break;
            case 0xcf: case 0xd3: payload = 8; break;
            case 0xd4: payload = 2; break;
            case 0xd5: payload = 3; break;
            case 0xd6: payload = 5; break;
            case 0xd7: payload = 9; break;
            case 0xd8: payload = 17; break;
            case 0xd9: need = 1; kind = 1; break;
            case 0xda: need = 2; kind = 1; break;
            case 0xdb: need = 4; kind = 1; break;
            case 0xdc: need = 2; kind = 2; break;
            case 0xdd: need = 4; kind = 2; break;
            case 0xde: need = 2; kind = 3; break;
            case 0xdf: need = 4; kind = 3; break;
            }
            if (need) {
                if (len - p < need) return false;
                uint64_t v = 0;
                for (size_t i = 0; i < need; i++) v = (v << 8) | d[p + i];
                p += need;
                if (kind == 1) payload = (size_t) v;
                else if (kind == 2) kids = v;
                else if (kind == 3) kids = 2 * v;
                else payload = (size_t) v + 1;
            }
        }
        if (len - p < payload) return false;
        p += payload;
        remaining--;
        remaining += kids;
        if (kids > len - p) return false;          // every element needs at least one byte
    }
    *pos = p;
    return true;
}

extern "C" int64_t flbgpu_index_host(const void *data, size_t bytes, uint64_t *row_off, size_t cap, size_t *consumed) {
    const uint8_t *d = (const uint8_t *) data;
    size_t pos = 0;
    int64_t n = 0;
    while (pos < bytes) {
        size_t q = pos;
        __builtin_prefetch(d + pos + 2048); __builtin_prefetch(d + pos + 2112);
        if (!h_skip(d, bytes, &q)) break;
        if ((size_t) n + 1 >= cap) break;
        row_off[n++] = pos;
        pos = q;
    }
    if (cap > 0) row_off[n] = pos;
    if (consumed) *consumed = pos;
    return n;
}

// ------------------------------------------------------------------------------------------ run (host level)
static const size_t STAGE_SLAB = 8u << 20;

// Where the wall time of the last host-level call of this thread went, in microseconds (flbgpu_host_phases; bench.py host_level):
// 0 record boundaries on the host, 1 caller's buffer -> pinned slabs, 2 waiting for the upload, 3 the device chain (launches, kernels
// and the waits for the sizes between them), 4 output -> pinned slabs -> caller's buffer (with the wait for the copy), 5 malloc of the
// output, 6 the whole call.
enum { HP_INDEX, HP_COPY_IN, HP_UPLOAD_WAIT, HP_CHAIN, HP_DOWNLOAD, HP_MALLOC, HP_TOTAL, HP_N };
static thread_local double g_phase_us[HP_N];
static inline double wall_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct PhaseScope {
    int k; double t0;
    explicit PhaseScope(int k_) : k(k_), t0(wall_us()) {}
    ~PhaseScope() { g_phase_us[k] += wall_us() - t0; }
};
extern "C" int flbgpu_host_phases(double *out, int cap) {
    for (int i = 0; i < HP_N && i < cap; i++) out[i] = g_phase_us[i];
    return HP_N;
}

static bool stage_init(flbgpu_filter *f) {
    for (int i = 0; i < 2; i++) {
        if (!f->ev_stage[i] && hipEventCreateWithFlags(&f->ev_stage[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    return true;
}

// chunks of this size and more are indexed on the device (flbgpu_index_dev); smaller ones are
// cheaper to walk on the host than to launch the indexer's kernels for
static const size_t DEV_INDEX_MIN = 8u << 20;

// record boundaries of d[from..] appended to f->hp_off (pinned) until `until` is reached
static bool host_index_range(flbgpu_filter *f, const uint8_t *d, size_t bytes, size_t until, size_t *pos, int64_t *n, size_t *cap, bool *stop) {
    uint64_t *off = f->hp_off.as<uint64_t>();
    while (!*stop && *pos < until) {
        size_t q = *pos;
        __builtin_prefetch(d + *pos + 2048); __builtin_prefetch(d + *pos + 2112);
        if (!h_skip(d, bytes, &q)) { *stop = true; break; }
        if ((size_t) *n + 2 > *cap) {
            *cap *= 2;
            if (!f->hp_off.ensure(*cap * sizeof(uint64_t), (size_t) *n * sizeof(uint64_t))) return false;
            off = f->hp_off.as<uint64_t>();
        }
        off[(*n)++] = *pos;
        *pos = q;
    }
    return true;
}

// Copies `data` into f->h_in_data through the pinned slabs and finds the record boundaries -- on the
// host while the slabs are in flight, or on the device once the bytes are there.  Returns the
// record count (-1 on a HIP failure), the bytes covered by whole records and the device offsets.
// pageable <-> pinned copies of the host-level calls: one core moves ~13 GB/s, a PCIe 5 x16 link ~55 GB/s, so slabs of a
// megabyte and more are split over a few helper threads (created once, parked on a condition variable in between)
namespace {
struct CopyPool {
    static constexpr int HELPERS = 3;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::thread th[HELPERS];
    uint8_t *dst = nullptr; const uint8_t *src = nullptr; size_t part = 0, total = 0;
    uint64_t gen = 0; int pending = 0; bool quit = false, started = false;
    void worker(int i) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            uint8_t *d = dst; const uint8_t *s = src; const size_t p = part, t = total;
            lk.unlock();
            const size_t lo = (size_t) (i + 1) * p, hi = lo + p < t ? lo + p : t;
            if (lo < t) memcpy(d + lo, s + lo, hi - lo);
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
    std::mutex user;               // ONE job slot: a second caller (another filter's thread) copies by itself instead of queueing
    void copy(void *d, const void *s, size_t n) {
        if (n < (1u << 20) || getenv("FLBGPU_COPY_THREADS_OFF")) { memcpy(d, s, n); return; }
        std::unique_lock<std::mutex> own(user, std::try_to_lock);
        if (!own.owns_lock()) { memcpy(d, s, n); return; }
        std::unique_lock<std::mutex> lk(mu);
        if (!started) { for (int i = 0; i < HELPERS; i++) th[i] = std::thread(&CopyPool::worker, this, i); started = true; }
        dst = (uint8_t *) d; src = (const uint8_t *) s; total = n; part = (((n + HELPERS) / (HELPERS + 1)) + 63) & ~(size_t) 63;      // (rounded UP: the four shares cover n)
        pending = HELPERS; gen++;
        lk.unlock();
        cv_go.notify_all();
        memcpy(d, s, part < n ? part : n);                     // the caller's share
        lk.lock();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        if (started) for (auto &t : th) if (t.joinable()) t.join();
    }
};
CopyPool g_copy;
}  // namespace

// (diagnostics: the slab copy of the host-level calls, for the CPU-only unit test)
extern "C" void flbgpu_diag_copy(void *dst, const void *src, size_t n) { g_copy.copy(dst, src, n); }

int64_t flbgpu::staged_upload(flbgpu_filter *f, const uint8_t *d, size_t bytes, size_t *consumed, const uint64_t **row_off, bool no_wait) {
    hipStream_t st = f->stream;
    if (!stage_init(f) || !f->h_in_data.ensure(bytes + 16)) return -1;
    bool dev_index = bytes >= DEV_INDEX_MIN && !getenv("FLBGPU_HOST_INDEX");
    size_t cap = bytes / 96 + 1024;
    if (!f->hp_off.ensure(cap * sizeof(uint64_t))) return -1;
    const size_t slab = bytes < STAGE_SLAB ? bytes : STAGE_SLAB;
    if (!f->hp_stage[0].ensure(slab) || (bytes > slab && !f->hp_stage[1].ensure(slab))) return -1;
    size_t pos = 0, sent = 0;
    int64_t n = 0;
    bool stop = false;
    int k = 0;
    if (no_wait && !dev_index && bytes <= slab) {
        // a small chunk (SpecCall): the copy is on its way while the host finds the record boundaries, and nothing waits here -- the
        // stage that follows is on the same stream (the bytes behind the last whole record travel too: nobody reads them)
        if (hipStreamSynchronize(st) != hipSuccess) return -1;                      // (idle unless an earlier call ended early)
        { PhaseScope ph(HP_COPY_IN); g_copy.copy(f->hp_stage[0].p, d, bytes); }
        if (hipMemcpyAsync(f->h_in_data.p, f->hp_stage[0].p, bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(f->ev_stage[0], st) != hipSuccess) return -1;
        { PhaseScope ph(HP_INDEX); if (!host_index_range(f, d, bytes, bytes, &pos, &n, &cap, &stop)) return -1; }
        uint64_t *off = f->hp_off.as<uint64_t>();
        off[n] = pos;
        *consumed = pos;
        *row_off = f->h_in_off.as<uint64_t>();
        if (n == 0) return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
        if (!f->h_in_off.ensure((size_t) (n + 1) * sizeof(uint64_t))) return -1;
        *row_off = f->h_in_off.as<uint64_t>();
        if (hipMemcpyAsync(f->h_in_off.p, off, (size_t) (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        return n;
    }
    while (sent < bytes) {
        const size_t end = sent + slab < bytes ? sent + slab : bytes;
        if (!dev_index) { PhaseScope ph(HP_INDEX); if (!host_index_range(f, d, bytes, end, &pos, &n, &cap, &stop)) return -1; }
        // bytes past the last whole record are never read by the kernels; they are not uploaded
        const size_t upto = stop ? (pos < end ? pos : end) : end;
        if (upto > sent) {
            if (hipEventSynchronize(f->ev_stage[k]) != hipSuccess) return -1;
            { PhaseScope ph(HP_COPY_IN); g_copy.copy(f->hp_stage[k].p, d + sent, upto - sent); }
            if (hipMemcpyAsync((uint8_t *) f->h_in_data.p + sent, f->hp_stage[k].p, upto - sent, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(f->ev_stage[k], st) != hipSuccess) return -1;
            k ^= 1;
        }
        if (stop) break;
        sent = end;
    }
    if (dev_index) {
        if (hipStreamSynchronize(st) != hipSuccess) { set_err("host to device copy failed"); return -1; }
        if (!f->indexer) f->indexer = flbgpu_indexer_create();
        flbgpu_dev_chunk ch;
        int64_t nd = f->indexer ? flbgpu_index_dev(f->indexer, f->h_in_data.p, bytes, &ch, consumed) : -1;
        if (nd >= 0) { *row_off = ch.row_off; return nd; }
        // a chunk the device indexer gives up on (see flbgpu_index_dev): sequential walk
        if (!host_index_range(f, d, bytes, bytes, &pos, &n, &cap, &stop)) return -1;
    }
    uint64_t *off = f->hp_off.as<uint64_t>();
    off[n] = pos;
    *consumed = pos;
    *row_off = f->h_in_off.as<uint64_t>();
    if (n == 0) return 0;
    if (!f->h_in_off.ensure((size_t) (n + 1) * sizeof(uint64_t))) return -1;
    *row_off = f->h_in_off.as<uint64_t>();
    if (hipMemcpyAsync(f->h_in_off.p, off, (size_t) (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    // the staging slabs and the offsets are reused by the next call
    PhaseScope ph(HP_UPLOAD_WAIT);
    if (hipStreamSynchronize(st) != hipSuccess) { set_err("host to device copy failed"); return -1; }
    return n;
}

// device -> caller's (pageable) buffer through the two pinned slabs
bool flbgpu::staged_download(flbgpu_filter *f, void *dst, const void *src, size_t bytes) {
    hipStream_t st = f->stream;
    if (!stage_init(f)) return false;
    const size_t slab = bytes < STAGE_SLAB ? bytes : STAGE_SLAB;
    if (!f->hp_stage[0].ensure(slab) || (bytes > slab && !f->hp_stage[1].ensure(slab))) return false;
    size_t issued = 0, done = 0, len[2] = {0, 0};
    int ki = 0, kd = 0;
    // one slab in flight while the previous one is copied out
    while (done < bytes) {
        while (issued < bytes && issued - done < 2 * slab && (issued == done || ki != kd)) {
            len[ki] = bytes - issued < slab ? bytes - issued : slab;
            if (hipMemcpyAsync(f->hp_stage[ki].p, (const uint8_t *) src + issued, len[ki], hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipEventRecord(f->ev_stage[ki], st) != hipSuccess) return false;
            issued += len[ki];
            ki ^= 1;
        }
        if (hipEventSynchronize(f->ev_stage[kd]) != hipSuccess) return false;
        g_copy.copy((uint8_t *) dst + done, f->hp_stage[kd].p, len[kd]);
        done += len[kd];
        kd ^= 1;
    }
    return true;
}

extern "C" int flbgpu_filter_chain_run(flbgpu_filter *const *filters, int nfilters, const void *data, size_t bytes,
                                       void **out_buf, size_t *out_size, flbgpu_chain_stat *stats) {
    if (stats) memset(stats, 0, sizeof(*stats) * (size_t) (nfilters > 0 ? nfilters : 0));
    if (bytes == 0 || nfilters <= 0) return FLBGPU_FILTER_NOTOUCH;
    flbgpu_filter *f = filters[0];
    // Record boundaries are found on the host (msgpack is sequential) slab by slab; each slab goes
    // through a pinned staging buffer to the device while the next one is being indexed.
    size_t consumed = 0;
    const uint64_t *row_off = nullptr;
    memset(g_phase_us, 0, sizeof(g_phase_us));
    PhaseScope ph_all(HP_TOTAL);
    // a small chunk: everything launched ahead, one wait (SpecCall); the output comes back through the second slab, written by the device
    bool small = spec_wanted(0, bytes) && f->hp_stage[1].ensure(STAGE_SLAB);
    int64_t n = staged_upload(f, (const uint8_t *) data, bytes, &consumed, &row_off, small);
    if (n < 0) return FLBGPU_FILTER_NOTOUCH;
    small = small && spec_wanted((uint64_t) n, bytes);
    struct ViewScope {
        ViewScope(const void *dev, const void *host, size_t bytes) { g_hostview.dev = dev; g_hostview.host = (const uint8_t *) host; g_hostview.bytes = bytes; }
        ~ViewScope() { g_hostview = HostView(); }
    } view_scope(f->h_in_data.p, data, bytes);                 // (host rules read the values from the caller's copy)
    struct SpecScope {
        SpecScope(bool on, flbgpu_filter *f0) { g_spec = SpecCall(); g_spec.on = on; if (on) { g_spec.sink = (uint8_t *) f0->hp_stage[1].p; g_spec.sink_cap = STAGE_SLAB; } }
        ~SpecScope() { g_spec = SpecCall(); }
    } spec_scope(small, f);
    bool garbage = consumed != bytes && !tail_is_clean((const uint8_t *) data, bytes, consumed);
    if (n == 0) {
        flbgpu_dev_chunk o0;
        memset(&o0, 0, sizeof(o0));
        if (empty_chunk_result(filters, nfilters, garbage, &o0) == FLBGPU_FILTER_MODIFIED) { *out_buf = NULL; *out_size = 0; return FLBGPU_FILTER_MODIFIED; }
        return FLBGPU_FILTER_NOTOUCH;
    }
    flbgpu_dev_chunk in, out;
    in.data = f->h_in_data.p; in.row_off = row_off; in.n = (uint64_t) n; in.bytes = consumed;
    memset(&out, 0, sizeof(out));
    {
        PhaseScope ph(HP_CHAIN);
        if (chain_dev(filters, nfilters, &in, &out, garbage, stats) != FLBGPU_FILTER_MODIFIED) return FLBGPU_FILTER_NOTOUCH;
    }
    if (out.bytes == 0) { *out_buf = NULL; *out_size = 0; return FLBGPU_FILTER_MODIFIED; }
    void *hb;
    { PhaseScope ph(HP_MALLOC); hb = malloc(out.bytes); }
    if (!hb) return FLBGPU_FILTER_NOTOUCH;
    PhaseScope ph_dl(HP_DOWNLOAD);
    if (g_spec.sunk && out.bytes <= g_spec.sink_cap) g_copy.copy(hb, g_spec.sink, out.bytes);
    else if (!staged_download(f, hb, out.data, out.bytes)) {
        free(hb);
        set_err("device to host copy failed");
        return FLBGPU_FILTER_NOTOUCH;
    }
    *out_buf = hb;
    *out_size = out.bytes;
    return FLBGPU_FILTER_MODIFIED;
}

extern "C" int flbgpu_filter_run(flbgpu_filter *f, const void *data, size_t bytes, void **out_buf, size_t *out_size) {
    flbgpu_filter *one[1] = {f};
    return flbgpu_filter_chain_run(one, 1, data, bytes, out_buf, out_size, nullptr);
}

// flb_parser_do on a batch of one: wraps the value as {"k": value} and runs filter_parser with
// Key_Name k, then strips the event header from the output.
extern "C" int flbgpu_parser_do(flbgpu_parser *p, const char *buf, size_t length, void **out_buf, size_t *out_size,
                                int64_t *out_sec, int64_t *out_nsec) {
    if (!p->self_filter) {
        flbgpu_parser *arr[1] = {p};
        p->self_filter = flbgpu_filter_parser_create("k", 0, 0, 1, arr);
        if (!p->self_filter) return -1;
    }
    std::vector<uint8_t> rec;
    const uint8_t hdr[] = {0x92, 0x92, 0xd7, 0x00, 0, 0, 0, 0, 0, 0, 0, 0, 0x80, 0x81, 0xa1, 'k', 0xdb};
    rec.insert(rec.end(), hdr, hdr + sizeof(hdr));
    for (int i = 3; i >= 0; i--) rec.push_back((uint8_t) (length >> (8 * i)));
    rec.insert(rec.end(), (const uint8_t *) buf, (const uint8_t *) buf + length);
    void *ob = nullptr;
    size_t os = 0;
    if (flbgpu_filter_run(p->self_filter, rec.data(), rec.size(), &ob, &os) != FLBGPU_FILTER_MODIFIED || os < 13) { free(ob); return -1; }
    const uint8_t *o = (const uint8_t *) ob;
    size_t body = 13;       // 92 92 d7 00 <8> 80
    // whether a parser accepted the value: the record's RF_PARSED flag (column 0 of the filter's record columns;
    // an unparsed record comes back as the canonical re-pack of the wrapper, which a pattern like
    // ^(?<k>.*)$ would reproduce byte for byte)
    uint32_t flags = 0;
    if (hipMemcpy(&flags, p->self_filter->d_info.p, sizeof(flags), hipMemcpyDeviceToHost) != hipSuccess) { set_err("device read failed"); free(ob); return -1; }
    if (!(flags & RF_PARSED)) { free(ob); return -1; }
    uint32_t sec = ((uint32_t) o[4] << 24) | (o[5] << 16) | (o[6] << 8) | o[7];
    uint32_t nsec = ((uint32_t) o[8] << 24) | (o[9] << 16) | (o[10] << 8) | o[11];
    *out_sec = sec; *out_nsec = nsec;
    size_t ms = os - body;
    void *m = malloc(ms ? ms : 1);
    if (!m) { free(ob); set_err("out of memory"); return -1; }
    memcpy(m, o + body, ms);
    free(ob);
    *out_buf = m; *out_size = ms;
    // return value of flb_parser_do: the end of the last named group that took part in the match, in
    // name-iteration order (cb_results / last_pos, src/flb_regex.c:52-54, src/flb_parser_regex.c:44-112).
    // The capture spans of the record are still in the filter's columns ([span][n] with n == 1).
    const int nf = p->dev.nfields;
    std::vector<uint32_t> caps((size_t) 2 * nf + 2, CAP_UNSET);
    if (nf > 0 && hipMemcpy(caps.data(), p->self_filter->d_caps.p, (size_t) 2 * nf * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
        set_err("device read failed");
        return (int) length;
    }
    int last_pos = -1;
    for (int f = 0; f < nf; f++)
        if (caps[2 * f] != CAP_UNSET && caps[2 * f + 1] != CAP_UNSET) last_pos = (int) caps[2 * f + 1];
    return last_pos >= 0 ? last_pos : (int) length;
}

// ------------------------------------------------------------------------------------------ device helpers
extern "C" void *flbgpu_dev_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { set_err("hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" void flbgpu_dev_free(void *p) { if (p) (void) hipFree(p); }
extern "C" int flbgpu_memcpy_h2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
extern "C" int flbgpu_memcpy_d2h(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
extern "C" int flbgpu_sync(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }
