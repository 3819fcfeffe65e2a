// fx.cpp -- compact forward tables of the single-pass tile kernel (dev.hpp DevFx): builder + a host execution of the
// SAME tables with the kernel's rules (sentinel byte, result slots), so that the CPU-only tests can check the table
// transformation against the golden vectors of the real engine.  Host only; the filters never call the simulation.
#include <cstring>
#include <string>
#include <vector>
#include "host_int.hpp"

using namespace flbgpu;

// Compact forward tables of k_parser_tile (dev.hpp DevFx), derived from the kernel tables of the ascii set (rx.hpp
// `ft` / `ft2`): one column per byte class plus the end-of-text column -- the (kind, class) columns of `ft` are
// sparse, a class has exactly one kind --, every offset an LDS byte address, ONE capture write per entry, and what the
// classic walk handles as special entries (MATCH, dead end, several candidates) turned into plain steps to the
// absorbing row whose capture write says what happened.  ncap = capture columns of the parser's named fields.
bool flbgpu::build_fx(const rx::TableSet &t, int ncap, std::vector<uint8_t> &b, DevFx &out, bool pair) {
    memset(&out, 0, sizeof(out));
    if (!t.has_capture || !t.ascii_only || t.ft.empty()) return true;
    // a row holds an ODD number of dwords: lanes that read the same column of different rows -- the common case, the lanes of a
    // wave sit in different states over similar bytes -- then fall on different LDS banks (rows aligned to a power of two put
    // a column of every row on ONE bank: measured 3.5x the bank conflicts and a 25 % slower kernel)
    // pair = true (k_parser_reg<PAIR2>): behind its single-step cells a row carries one cell per PAIR of byte classes -- two steps
    // with one table read -- and every entry names the next row by the address of that pair section (the single-step cells sit
    // `pair_bias` bytes in front of it)
    const uint32_t stride = (uint32_t) t.ncls + 1;
    const uint32_t rstride = (pair ? stride + stride * stride : stride) | 1u;      // dwords between rows
    const uint32_t rowb = rstride * 4, bias = pair ? stride * 4 : 0;
    const uint32_t r2stride = stride | 1u, row2b = r2stride * 4;                   // look-ahead rows: single-step cells only
    const uint32_t nrows = (uint32_t) t.nX * (uint32_t) t.NKp;          // + absorb + poison
    const size_t W = (size_t) 1 << t.wsh, ncols2 = (size_t) 1 << t.fc_shift, nm = t.ft2.size() / ncols2;
    const uint32_t ft_at = 1024;
    const uint64_t ft2_at64 = (uint64_t) ft_at + (uint64_t) (nrows + 2) * rowb;
    const uint64_t p2_at64 = ft2_at64 + (uint64_t) nm * row2b;
    const uint32_t nslots = (uint32_t) ncap + 5;
    if (nslots > 63 || p2_at64 > 60000) return true;
    const uint32_t ft2_at = (uint32_t) ft2_at64;
    auto rowaddr = [&](uint32_t r) -> uint32_t { return ft_at + r * rowb + bias; };     // what an entry carries
    const uint32_t absorb = rowaddr(nrows), poison = rowaddr(nrows + 1);
    const uint32_t S_END_EOT = (uint32_t) ncap + FXS_END_EOT, S_END_MID = (uint32_t) ncap + FXS_END_MID,
                   S_DEAD_EOT = (uint32_t) ncap + FXS_DEAD_EOT, S_FAIL = (uint32_t) ncap + FXS_FAIL;
    std::vector<uint32_t> ft((size_t) (nrows + 2) * rstride), ft2(nm * r2stride), p2;
    auto plain = [&](uint32_t next_at, uint32_t slot) -> uint32_t { return next_at | ((slot * 128u) << FX_SLOT_SHIFT); };
    auto pairw = [&](uint32_t next_at, uint32_t a, uint32_t b) -> uint32_t {
        p2.push_back(plain(next_at, a)); p2.push_back(b * 128u);
        return FX_PAIR | (uint32_t) (p2.size() / 2 - 1);
    };
    bool fits = true;
    // e: an entry of t.ft / t.ft2 (rx.hpp encoding); eot: the cell belongs to the end-of-text column
    auto conv = [&](uint32_t e, bool eot) -> uint32_t {
        if (e & rx::FT_SPECIAL) {
            const uint32_t ty = rx::ft_type(e);
            if (ty == rx::FT_LOOK) {
                if ((e & 0xFFFFFF) >= nm) { fits = false; return plain(absorb, S_FAIL); }
                return FX_LOOK | (ft2_at + (e & 0xFFFFFF) * row2b);
            }
            if (ty == rx::FT_MATCH) {
                const uint32_t a = (e >> 12) & 63, b = (e >> 18) & 63, endslot = eot ? S_END_EOT : S_END_MID;
                if (a && b) return plain(absorb, S_FAIL);                       // three writes: the classic walk takes the record
                if (a || b) return pairw(absorb, a ? a : b, endslot);
                return plain(absorb, endslot);
            }
            if (ty == rx::FT_MULTI) return plain(absorb, S_FAIL);
            return plain(absorb, eot ? S_DEAD_EOT : 0);                         // dead end
        }
        const uint32_t next = rowaddr(e & 0xFFF), a = (e >> 12) & 63, b = (e >> 18) & 63;
        if (a && b) return pairw(next, a, b);
        return plain(next, a ? a : b);
    };
    for (uint32_t r = 0; r < nrows; r++)
        for (uint32_t c = 0; c <= (uint32_t) t.ncls; c++) {
            const bool eot = c == (uint32_t) t.ncls;
            const uint32_t kind = eot ? (uint32_t) t.kind_edge : t.kind_of_cls[c];
            const size_t colc = ((size_t) kind << t.fc_shift) | c;
            ft[(size_t) r * rstride + c] = (!eot && (int) c == t.high_cls) ? plain(poison, 0) : conv(t.ft[(size_t) r * W + colc], eot);
        }
    for (uint32_t c = 0; c < stride; c++) { ft[(size_t) nrows * rstride + c] = plain(absorb, 0); ft[(size_t) (nrows + 1) * rstride + c] = plain(poison, 0); }
    // ft2 rows: resolved entries of a LOOK cell, indexed by the class of the NEXT byte; the cell itself is never in the
    // end-of-text column (nothing follows it), so a MATCH here ends in front of the end of the text
    for (size_t m = 0; m < nm; m++)
        for (uint32_t c = 0; c < stride; c++) {
            uint32_t v;
            if (c < (uint32_t) t.ncls && (int) c == t.high_cls) v = plain(absorb, S_FAIL);   // a byte >= 0x80 follows: the UTF-8 tables decide
            else v = conv(t.ft2[m * ncols2 + c], false);
            if ((v & FX_PAIR) == FX_LOOK) v = plain(absorb, S_FAIL);
            ft2[m * r2stride + c] = v;
        }
    if (!fits) return true;
    if (pair) {
        // cell (row, c1, c2) = the two single steps folded: next row | slot of the first step << 16 | slot of the second << 24;
        // FX2_LOOK3 | slot1 << 16 | look-ahead row: the SECOND step's cell waits for the byte behind the pair;
        // FX2_SPECIAL: a step writes two captures -- the kernel takes the two single steps
        auto single = [&](uint32_t row_at, uint32_t c) -> uint32_t { return ft[(size_t) ((row_at - bias - ft_at) / rowb) * rstride + c]; };
        for (uint32_t r = 0; r < nrows + 2; r++)
            for (uint32_t c1 = 0; c1 < stride; c1++)
                for (uint32_t c2 = 0; c2 < stride; c2++) {
                    uint32_t v = FX2_SPECIAL;
                    uint32_t s1 = ft[(size_t) r * rstride + c1];
                    if ((s1 & FX_PAIR) == FX_LOOK) s1 = ft2[(size_t) (((s1 & FX_ROW_MASK) - ft2_at) / row2b) * r2stride + c2];
                    if (!(s1 & 0x80000000u)) {
                        const uint32_t slot1 = (s1 >> FX_SLOT_SHIFT) / 128u, s2 = single(s1 & FX_ROW_MASK, c2);
                        if (!(s2 & 0x80000000u)) v = (s2 & FX_ROW_MASK) | (slot1 << 16) | (((s2 >> FX_SLOT_SHIFT) / 128u) << 24);
                        else if ((s2 & FX_PAIR) == FX_LOOK) v = FX2_LOOK3 | (slot1 << 16) | (s2 & FX_ROW_MASK);
                    }
                    ft[(size_t) r * rstride + stride + c1 * stride + c2] = v;
                }
    }
    std::vector<uint32_t> cls(256);
    for (int b = 0; b < 256; b++) cls[(size_t) b] = (uint32_t) t.cls[b] * 4u;
    cls[255] = (uint32_t) t.ncls * 4u;                                          // the end-of-text sentinel
    b.assign(p2_at64 + p2.size() * 4, 0);
    memcpy(b.data(), cls.data(), 1024);
    memcpy(b.data() + ft_at, ft.data(), ft.size() * 4);
    if (!ft2.empty()) memcpy(b.data() + ft2_at, ft2.data(), ft2.size() * 4);
    if (!p2.empty()) memcpy(b.data() + p2_at64, p2.data(), p2.size() * 4);
    const size_t total = (b.size() + 15) & ~(size_t) 15;
    b.resize(total);
    if (total > 60000) return true;                                             // addresses are 16 bits; the record tiles need the rest of the LDS
    out.base = nullptr; out.bytes = (uint32_t) total;
    out.off_p2 = (uint32_t) p2_at64;
    out.start_off = rowaddr(((uint32_t) t.nX - 1) * (uint32_t) t.NKp + (uint32_t) t.kind_edge);
    out.absorb_off = absorb; out.poison_off = poison; out.nslots = nslots;
    out.pair_bias = bias; out.ncls1 = stride;
    out.ok = 1;
    return true;
}


// The walk of k_parser_tile (tile_kernels.inc fx_walk + the result rules that follow it) on the host.
// caps: nslots u16 columns (0xFFFF = unset).  Returns >= 0 end of the match, -1 the forward walk from boundary 0 does
// not settle the value (the kernel then runs the reverse pass + classic walk), -2 a byte >= 0x80 / a real 0xFF.
int flbgpu::simulate_fx(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps) {
    auto u32at = [&](uint32_t at) -> uint32_t { uint32_t v; memcpy(&v, b.data() + at, 4); return v; };
    for (uint32_t i = 0; i < fx.nslots; i++) caps[i] = 0xFFFF;
    uint32_t e = fx.start_off;
    const uint32_t bias = fx.pair_bias;
    auto cls_of = [&](uint32_t pos) -> uint32_t { return u32at(4 * (pos < len ? s[pos] : 0xFFu)); };   // the sentinel behind the value
    auto step1 = [&](uint32_t j) {
        e = u32at((e & FX_ROW_MASK) - bias + cls_of(j));
        if (e & 0x80000000u) {
            if (!(e & 0x40000000u)) e = u32at((e & FX_ROW_MASK) + cls_of(j + 1));
            if (e & 0x80000000u) {
                const uint32_t k = e & 0x3FFFFFFFu;
                caps[u32at(fx.off_p2 + 8 * k + 4) / 128] = (uint16_t) j;
                e = u32at(fx.off_p2 + 8 * k);
            }
        }
        caps[(e >> FX_SLOT_SHIFT) / 128] = (uint16_t) j;
    };
    if (!bias) for (uint32_t j = 0; j <= len; j++) step1(j);
    else {
        // k_parser_reg<PAIR2>: two positions per table read; like the kernel it walks whole pairs (positions behind the end of the
        // text meet the sentinel's column in an absorbing row)
        for (uint32_t j = 0; j <= len; j += 2) {
            const uint32_t ep = e;
            e = u32at((e & FX_ROW_MASK) + (cls_of(j) / 4) * fx.ncls1 * 4 + cls_of(j + 1));
            if (e & 0x80000000u) {
                uint32_t e2 = 0;
                const bool look3 = !(e & 0x40000000u);
                if (look3) e2 = u32at((e & FX_ROW_MASK) + cls_of(j + 2));
                if (!look3 || (e2 & 0x80000000u)) { e = ep; step1(j); step1(j + 1); continue; }
                caps[(e >> 16) & 255] = (uint16_t) j;
                caps[(e2 >> FX_SLOT_SHIFT) / 128] = (uint16_t) (j + 1);
                e = e2;
                continue;
            }
            caps[(e >> 16) & 255] = (uint16_t) j;
            caps[(e >> 24) & 63] = (uint16_t) (j + 1);
        }
    }
    const uint32_t S = e & FX_ROW_MASK;
    const uint32_t e_eot = caps[ncap + FXS_END_EOT], e_mid = caps[ncap + FXS_END_MID], d_eot = caps[ncap + FXS_DEAD_EOT], failed = caps[ncap + FXS_FAIL];
    if (S == fx.poison_off) return -2;
    if (failed != 0xFFFF) return -1;
    if (e_mid != 0xFFFF) return (int) e_mid;
    if (e_eot != 0xFFFF) return e_eot == len ? (int) len : -2;
    if (d_eot != 0xFFFF) return d_eot == len ? -1 : -2;
    return -1;
}
