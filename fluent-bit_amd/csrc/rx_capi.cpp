// rx_capi.cpp -- C entry points over the regex table compiler (diagnostics / self-test part of
// the C-ABI, declared in include/flb_gpu.h).  The simulate_* calls execute the compiled TABLES on
// the host so the CPU-only unit tests can check them against the golden vectors; the filters
// never call them.
#include <cstring>
#include <string>
#include "rx.hpp"
#include "host_int.hpp"

extern "C" {

void *flbgpu_rx_compile(const char *pattern, int len, unsigned options, int want_captures, char *err, int errlen)
{
    auto *p = new rx::Program();
    std::string e;
    if (!rx::compile(pattern, (size_t) len, options, want_captures != 0, *p, e)) {
        if (err && errlen > 0) { strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
        delete p;
        return nullptr;
    }
    if (err && errlen > 0) err[0] = 0;
    return p;
}

void flbgpu_rx_free(void *h) { delete (rx::Program *) h; }

/* the backtracking matcher for look-around, atomic groups, possessive repeats, back-references, \Z \G \K (rxbt.inc): what the filters
 * run on the host for a rule / parser the GPU engines cannot take.  flbgpu_rxbt_search: groups + 1 on a match (beg / end may be NULL),
 * -1 no match, -4 the backtrack budget was spent. */
void *flbgpu_rxbt_compile(const char *pattern, int len, unsigned options, char *err, int errlen)
{
    std::string e;
    rx::BtProgram *p = rx::bt_compile(pattern, (size_t) len, options, e);
    if (err && errlen > 0) { strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
    return p;
}
void flbgpu_rxbt_free(void *h) { rx::bt_free((rx::BtProgram *) h); }
int flbgpu_rxbt_search(void *h, const char *s, int len, int *beg, int *end)
{
    return rx::bt_search((const rx::BtProgram *) h, (const uint8_t *) s, len, beg, end);
}
/* 1: flbgpu_rx_compile refuses the pattern because of such a construct (and flbgpu_rxbt_compile is the one to ask) */
int flbgpu_rx_is_nonregular(const char *pattern, int len, unsigned options)
{
    rx::Program p;
    std::string e;
    if (rx::compile(pattern, (size_t) len, options, true, p, e)) return 0;
    return p.nonregular ? 1 : 0;
}

int flbgpu_rx_simulate_capture(void *h, const char *s, int len, int *beg, int *end)
{
    return rx::simulate_capture(*(rx::Program *) h, (const uint8_t *) s, len, beg, end);
}

int flbgpu_rx_simulate_match(void *h, const char *s, int len)
{
    return rx::simulate_match(*(rx::Program *) h, (const uint8_t *) s, len);
}

/* info[0..11] = ascii{ncls, nD, nR, nX, NK, list entries}, utf8{ncls, nR, nX, NK, list entries}, ngroups */
void flbgpu_rx_info(void *h, int *info)
{
    auto *p = (rx::Program *) h;
    info[0] = p->ascii.ncls; info[1] = p->ascii.nD; info[2] = p->ascii.nR; info[3] = p->ascii.nX;
    info[4] = p->ascii.NK; info[5] = (int) p->ascii.list_ent.size();
    info[6] = p->utf8.ncls; info[7] = p->utf8.nR; info[8] = p->utf8.nX; info[9] = p->utf8.NK;
    info[10] = (int) p->utf8.list_ent.size(); info[11] = p->ngroups;
}

void flbgpu_rx_debug_stats(long *out3) { rx::debug_stats(out3); }

/* which engines stand behind the handle: bit 0 the NFA engine answers for values with a byte >= 0x80, bit 1 for every value (the
 * ascii set is a stub); info[0] = positions, [1] = words per set, [2] = character classes, [3] = context kinds, [4] = list entries,
 * [5] = code point intervals.  why (may be NULL): what the table compiler said when it gave up. */
int flbgpu_rx_engine(void *h, int *info6, char *why, int whylen)
{
    auto *p = (rx::Program *) h;
    if (info6) {
        info6[0] = p->nfa.P; info6[1] = p->nfa.VW; info6[2] = p->nfa.ncls; info6[3] = p->nfa.NK;
        info6[4] = (int) p->nfa.list_ent.size(); info6[5] = (int) p->nfa.mb_lo.size();
    }
    if (why && whylen > 0) { strncpy(why, p->why_nfa.c_str(), whylen - 1); why[whylen - 1] = 0; }
    return (p->utf8_nfa ? 1 : 0) | (p->ascii_stub ? 2 : 0);
}

/* 1: the text can meet one of the two corners where the reference's own answer depends on its search optimizer (rx::corner): the
 * filters count such values (flbgpu_filter_regex_corners); info (may be NULL): the pattern's corner flags */
int flbgpu_rx_corner(void *h, const char *s, int len, int *flags)
{
    auto *p = (rx::Program *) h;
    if (flags) *flags = (int) p->corner_flags;
    return rx::corner(p->corner_flags, (const uint8_t *) s, len) ? 1 : 0;
}

/* test aid: a random text drawn from the pattern (rx::sample); returns its length (cut to cap), -1 when the pattern does not parse */
int flbgpu_rx_sample(const char *pattern, int len, unsigned options, unsigned long long seed, char *out, int cap)
{
    std::string o, e;
    if (!rx::sample(pattern, (size_t) len, options, seed, o, e)) return -1;
    const int n = (int) o.size() < cap ? (int) o.size() : cap;
    memcpy(out, o.data(), (size_t) n);
    return n;
}

/* The compact tables of the single-pass tile kernel (fx.cpp) executed on the host with the kernel's rules.
 * >= 0: groups, beg/end of the NAMED groups filled (-1 elsewhere), beg[0] = 0, end[0] = end of the match;
 * -1: the forward walk from boundary 0 does not settle this text (the kernel falls back to the classic walk);
 * -2: a byte >= 0x80 (UTF-8 tables); -4: the pattern has no compact tables. */
static int simulate_fx_tables(void *h, const char *s, int len, int *beg, int *end, bool pair, bool use_tail = true);
int flbgpu_rx_simulate_fx(void *h, const char *s, int len, int *beg, int *end) { return simulate_fx_tables(h, s, len, beg, end, false); }
/* the same with every position walked (the pattern's tail, DevFx::tail_min, not used): what the skipping walk must equal */
int flbgpu_rx_simulate_fx_walk_all(void *h, const char *s, int len, int *beg, int *end) { return simulate_fx_tables(h, s, len, beg, end, false, false); }
/* the same over the tables with a cell per pair of byte classes (k_parser_reg<PAIR2>: two positions per table read); -4 also when they do not fit */
int flbgpu_rx_simulate_fx2(void *h, const char *s, int len, int *beg, int *end) { return simulate_fx_tables(h, s, len, beg, end, true); }
/* the tables without special entries (fx.cpp build_fx3: 8-byte cells, two capture writes per step -- k_parser_reg<.., FX3>); info2
 * (may be NULL): rows, bytes of the tables */
int flbgpu_rx_simulate_fx3(void *h, const char *s, int len, int *beg, int *end, int *info2)
{
    auto *p = (rx::Program *) h;
    int ncap = 0;
    for (uint8_t c : p->slot2cap) if (c != 0xFF) ncap++;
    if (ncap == 0) return -4;
    std::vector<uint8_t> blob;
    flbgpu::DevFx fx;
    // info2[0] < 0 on entry (with info2 given): the two-position form (-1: fx4, four write ports; -2: fx5, three)
    const int pairs = info2 && info2[0] < 0 ? (info2[0] == -2 ? 2 : 1) : 0;
    if (!flbgpu::build_fx3(p->ascii, ncap, blob, fx, pairs) || !fx.ok) return -4;
    if (info2) { info2[0] = pairs ? (int) ((fx.bytes - 2048) / (((fx.ncls1 * fx.ncls1) | 1u) * 8)) : (int) ((fx.bytes - 1024) / ((fx.ncls1 | 1u) * 8)); info2[1] = (int) fx.bytes; }
    std::vector<uint16_t> caps(fx.nslots);
    const int r = pairs ? flbgpu::simulate_fx4(blob, fx, ncap, (const uint8_t *) s, (uint32_t) len, caps.data())
                        : flbgpu::simulate_fx3(blob, fx, ncap, (const uint8_t *) s, (uint32_t) len, caps.data());
    if (r < 0) return r;
    for (int g = 0; g <= p->ngroups; g++) { beg[g] = -1; end[g] = -1; }
    beg[0] = 0; end[0] = r;
    for (int g = 1; g <= p->ngroups; g++) {
        const uint8_t cb = p->slot2cap[2 * (size_t) g], ce = p->slot2cap[2 * (size_t) g + 1];
        if (cb == 0xFF || ce == 0xFF) continue;
        const uint16_t b = caps[(size_t) cb + 1], e = caps[(size_t) ce + 1];
        if (b != 0xFFFF && e != 0xFFFF) { beg[g] = b; end[g] = e; }
    }
    return p->ngroups;
}

static int simulate_fx_tables(void *h, const char *s, int len, int *beg, int *end, bool pair, bool use_tail)
{
    auto *p = (rx::Program *) h;
    int ncap = 0;
    for (uint8_t c : p->slot2cap) if (c != 0xFF) ncap++;
    if (ncap == 0) return -4;
    std::vector<uint8_t> blob;
    flbgpu::DevFx fx;
    if (!flbgpu::build_fx(p->ascii, ncap, blob, fx, pair) || !fx.ok) return -4;
    std::vector<uint16_t> caps(fx.nslots);
    const int r = flbgpu::simulate_fx(blob, fx, ncap, (const uint8_t *) s, (uint32_t) len, caps.data(), use_tail);
    if (r < 0) return r;
    for (int g = 0; g <= p->ngroups; g++) { beg[g] = -1; end[g] = -1; }
    beg[0] = 0; end[0] = r;
    for (int g = 1; g <= p->ngroups; g++) {
        const uint8_t cb = p->slot2cap[2 * (size_t) g], ce = p->slot2cap[2 * (size_t) g + 1];
        if (cb == 0xFF || ce == 0xFF) continue;
        const uint16_t b = caps[(size_t) cb + 1], e = caps[(size_t) ce + 1];
        if (b != 0xFFFF && e != 0xFFFF) { beg[g] = b; end[g] = e; }
    }
    return p->ngroups;
}

/* diagnostics of the compact tables: out[0] rows, [1] classes (+1: end of text), [2] look-ahead rows, [3] pair entries, [4] bytes;
 * and over a text: [5] steps, [6] steps through a look-ahead cell, [7] steps through a pair cell */
int flbgpu_rx_fx_profile(void *h, const char *s, int len, long *out)
{
    auto *p = (rx::Program *) h;
    int ncap = 0;
    for (uint8_t c : p->slot2cap) if (c != 0xFF) ncap++;
    std::vector<uint8_t> b;
    flbgpu::DevFx fx;
    if (ncap == 0 || !flbgpu::build_fx(p->ascii, ncap, b, fx) || !fx.ok) return -4;
    const rx::TableSet &t = p->ascii;
    out[0] = (long) t.nX * t.NKp + 2; out[1] = t.ncls + 1; out[2] = (long) (t.ft2.size() >> t.fc_shift); out[3] = (long) ((b.size() - fx.off_p2) / 8); out[4] = (long) b.size();
    auto u32at = [&](uint32_t at) -> uint32_t { uint32_t v; memcpy(&v, b.data() + at, 4); return v; };
    auto cls_of = [&](int pos) -> uint32_t { return u32at(4 * (pos < len ? (uint8_t) s[pos] : 0xFFu)); };
    uint32_t e = fx.start_off;
    long steps = 0, look = 0, pair = 0;
    for (int j = 0; j <= len; j++) {
        e = u32at((e & flbgpu::FX_ROW_MASK) + cls_of(j));
        steps++;
        if (e & 0x80000000u) {
            if (!(e & 0x40000000u)) { look++; e = u32at((e & flbgpu::FX_ROW_MASK) + cls_of(j + 1)); }
            if (e & 0x80000000u) { pair++; e = u32at(fx.off_p2 + 8 * (e & 0x3FFFFFFFu)); }
        }
    }
    out[5] += steps; out[6] += look; out[7] += pair;
    return 0;
}

/* the tail of the compact tables (dev.hpp DevFx::tail_min): rows in it (0: none), its kill bytes; and over a text: *first = the first
 * position (a multiple of 16) from which a lane alone in its wave would skip to the end (-1: it never stands in the tail at a boundary) */
int flbgpu_rx_fx_tail(void *h, const char *s, int len, int *nkill, unsigned char *kill4, int *first)
{
    auto *p = (rx::Program *) h;
    int ncap = 0;
    for (uint8_t c : p->slot2cap) if (c != 0xFF) ncap++;
    std::vector<uint8_t> b;
    flbgpu::DevFx fx;
    if (ncap == 0 || !flbgpu::build_fx(p->ascii, ncap, b, fx) || !fx.ok) return -4;
    *nkill = (int) fx.nkill;
    for (uint32_t k = 0; k < fx.nkill && k < 4; k++) kill4[k] = fx.kill[k];
    *first = -1;
    const uint32_t stride = (((uint32_t) p->ascii.ncls + 1) | 1u) * 4;
    if (s && fx.tail_min < fx.absorb_off) {
        auto u32at = [&](uint32_t at) -> uint32_t { uint32_t v; memcpy(&v, b.data() + at, 4); return v; };
        auto cls_of = [&](int pos) -> uint32_t { return u32at(4 * (pos < len ? (uint8_t) s[pos] : 0xFFu)); };
        uint32_t e = fx.start_off;
        for (int j = 0; j <= len; j++) {
            const uint32_t row = e & flbgpu::FX_ROW_MASK;
            if (j > 0 && (j & 15) == 0 && row >= fx.tail_min && row < fx.absorb_off) { *first = j; break; }
            e = u32at(row + cls_of(j));
            if (e & 0x80000000u) {
                if (!(e & 0x40000000u)) e = u32at((e & flbgpu::FX_ROW_MASK) + cls_of(j + 1));
                if (e & 0x80000000u) e = u32at(fx.off_p2 + 8 * (e & 0x3FFFFFFFu));
            }
        }
    }
    return (int) ((fx.absorb_off - fx.tail_min) / stride);
}

/* "name=group\n" lines in onig_foreach_name order */
int flbgpu_rx_names(void *h, char *buf, int cap)
{
    auto *p = (rx::Program *) h;
    std::string o;
    for (size_t i = 0; i < p->names.size(); i++)
        for (int g : p->name_groups[i]) o += p->names[i] + "=" + std::to_string(g) + "\n";
    if ((int) o.size() + 1 > cap) return -1;
    memcpy(buf, o.c_str(), o.size() + 1);
    return (int) o.size();
}

}
