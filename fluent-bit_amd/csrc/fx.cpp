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
bool flbgpu::build_fx(const rx::TableSet &t, int ncap, std::vector<uint8_t> &b, DevFx &out) {
    memset(&out, 0, sizeof(out));
    if (!t.has_capture || !t.ascii_only || t.ft.empty()) return true;
    // a row holds an ODD number of dwords: lanes that read the same column of different rows -- the common case, the lanes of a
    // wave sit in different states over similar bytes -- then fall on different LDS banks (rows aligned to a power of two put
    // a column of every row on ONE bank: measured 3.5x the bank conflicts and a 25 % slower kernel)
    const uint32_t stride = (uint32_t) t.ncls + 1;
    const uint32_t rstride = stride | 1u;                                // dwords between rows
    const uint32_t rowb = rstride * 4;
    const uint32_t nrows = (uint32_t) t.nX * (uint32_t) t.NKp;          // + absorb + poison
    const size_t W = (size_t) 1 << t.wsh, ncols2 = (size_t) 1 << t.fc_shift, nm = t.ft2.size() / ncols2;
    const uint32_t ft_at = 1024, ft2_at = ft_at + (nrows + 2) * rowb;
    const uint64_t p2_at64 = (uint64_t) ft2_at + (uint64_t) nm * rowb;
    const uint32_t absorb = ft_at + nrows * rowb, poison = ft_at + (nrows + 1) * rowb;
    const uint32_t nslots = (uint32_t) ncap + 5;
    if (nslots > 63 || p2_at64 > 60000) return true;
    const uint32_t S_END_EOT = (uint32_t) ncap + FXS_END_EOT, S_END_MID = (uint32_t) ncap + FXS_END_MID,
                   S_DEAD_EOT = (uint32_t) ncap + FXS_DEAD_EOT, S_FAIL = (uint32_t) ncap + FXS_FAIL;
    std::vector<uint32_t> ft((size_t) (nrows + 2) * rstride), ft2(nm * rstride), p2;
    auto plain = [&](uint32_t next_at, uint32_t slot) -> uint32_t { return next_at | ((slot * 128u) << FX_SLOT_SHIFT); };
    auto pair = [&](uint32_t next_at, uint32_t a, uint32_t b) -> uint32_t {
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
                return FX_LOOK | (ft2_at + (e & 0xFFFFFF) * rowb);
            }
            if (ty == rx::FT_MATCH) {
                const uint32_t a = (e >> 12) & 63, b = (e >> 18) & 63, endslot = eot ? S_END_EOT : S_END_MID;
                if (a && b) return plain(absorb, S_FAIL);                       // three writes: the classic walk takes the record
                if (a || b) return pair(absorb, a ? a : b, endslot);
                return plain(absorb, endslot);
            }
            if (ty == rx::FT_MULTI) return plain(absorb, S_FAIL);
            return plain(absorb, eot ? S_DEAD_EOT : 0);                         // dead end
        }
        const uint32_t next = ft_at + (e & 0xFFF) * rowb, a = (e >> 12) & 63, b = (e >> 18) & 63;
        if (a && b) return pair(next, a, b);
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
            ft2[m * rstride + c] = v;
        }
    if (!fits) return true;
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
    out.start_off = ft_at + (((uint32_t) t.nX - 1) * (uint32_t) t.NKp + (uint32_t) t.kind_edge) * rowb;
    out.absorb_off = absorb; out.poison_off = poison; out.nslots = nslots;
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
    auto cls_of = [&](uint32_t pos) -> uint32_t { return u32at(4 * (pos < len ? s[pos] : 0xFFu)); };   // the sentinel behind the value
    for (uint32_t j = 0; j <= len; j++) {
        e = u32at((e & FX_ROW_MASK) + cls_of(j));
        if (e & 0x80000000u) {
            if (!(e & 0x40000000u)) e = u32at((e & FX_ROW_MASK) + cls_of(j + 1));
            if (e & 0x80000000u) {
                const uint32_t k = e & 0x3FFFFFFFu;
                caps[u32at(fx.off_p2 + 8 * k + 4) / 128] = (uint16_t) j;
                e = u32at(fx.off_p2 + 8 * k);
            }
        }
        caps[(e >> FX_SLOT_SHIFT) / 128] = (uint16_t) j;
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
