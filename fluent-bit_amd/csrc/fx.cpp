// fx.cpp -- compact forward tables of the single-pass tile kernel (dev.hpp DevFx): builder + a host execution of the
// SAME tables with the kernel's rules (sentinel byte, result slots), so that the CPU-only tests can check the table
// transformation against the golden vectors of the real engine.  Host only; the filters never call the simulation.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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
    // ---- the tail (DevFx::tail_min): a set Q of rows, closed under every byte class outside a small kill set K, in which all rows hold
    // the SAME plain entry per class (the row after a byte depends on the byte alone), with a class whose end-of-text step reports a
    // match (a tail, not a field in the middle), and whose capture writes do not outlive the last byte: every class behind which the text
    // may end in a match writes the one slot the set writes at all.  Chosen: fewest kill bytes, then the larger set.  The rows of Q move
    // to the end of the table so that "row address >= tail_min" is the test.
    std::vector<uint32_t> tail_rows;
    std::vector<uint8_t> tail_kill;
    if (!pair) {
        auto row_of = [&](uint32_t at) -> uint32_t { return (at - ft_at) / rowb; };
        // slots an entry writes (a pair entry: two); `end` = it reports a match
        auto slots_of = [&](uint32_t v, uint32_t *sl) -> int {
            if (!(v & 0x80000000u)) { sl[0] = (v >> FX_SLOT_SHIFT) / 128u; return 1; }
            const uint32_t k = v & 0x3FFFFFFFu;
            sl[0] = (p2[2 * k] >> FX_SLOT_SHIFT) / 128u; sl[1] = p2[2 * k + 1] / 128u;
            return 2;
        };
        auto next_of = [&](uint32_t v) -> uint32_t { return row_of(((v & 0x80000000u) ? p2[2 * (v & 0x3FFFFFFFu)] : v) & FX_ROW_MASK); };
        auto is_look = [&](uint32_t v) -> bool { return (v & FX_PAIR) == FX_LOOK; };
        size_t best_kill = 99;
        for (uint32_t r0 = 0; r0 < nrows; r0++) {
            std::vector<uint32_t> Q{r0};
            std::vector<char> K((size_t) t.ncls, 0);
            if (t.high_cls >= 0) K[(size_t) t.high_cls] = 1;
            auto in_q = [&](uint32_t r) -> bool { for (uint32_t q : Q) if (q == r) return true; return false; };
            // a target row: false = the class that led to it becomes a kill class
            auto take = [&](uint32_t tr, bool &changed) -> bool {
                if (tr >= nrows) return false;                                         // an absorbing row: the walk leaves the set
                if (in_q(tr)) return true;
                if (Q.size() >= 8) return false;
                Q.push_back(tr); changed = true;
                return true;
            };
            bool changed = true;
            while (changed) {
                changed = false;
                for (uint32_t c = 0; c < (uint32_t) t.ncls && !changed; c++) {
                    if (K[c]) continue;
                    const uint32_t v0 = ft[(size_t) Q[0] * rstride + c];
                    bool uni = (v0 & FX_PAIR) != FX_PAIR;                              // plain or look-ahead; a double write is not taken
                    for (uint32_t q : Q) if (ft[(size_t) q * rstride + c] != v0) uni = false;
                    if (!uni) { K[c] = 1; changed = true; break; }
                    if (!is_look(v0)) { if (!take(row_of(v0 & FX_ROW_MASK), changed)) { K[c] = 1; changed = true; } continue; }
                    // a look-ahead cell: what the class of the NEXT byte selects must be a plain step inside the set too
                    const size_t m = ((v0 & FX_ROW_MASK) - ft2_at) / row2b;
                    for (uint32_t c2 = 0; c2 < (uint32_t) t.ncls && !changed; c2++) {
                        if (K[c2]) continue;
                        const uint32_t w = ft2[m * r2stride + c2];
                        if ((w & 0x80000000u) || !take(row_of(w & FX_ROW_MASK), changed)) { K[c] = 1; changed = true; }
                    }
                }
            }
            std::vector<uint8_t> kb;
            for (int b8 = 0; b8 < 128; b8++) if (K[(size_t) t.cls[b8]]) kb.push_back((uint8_t) b8);
            if (getenv("FLBGPU_FX_DEBUG") && kb.size() <= 6) {
                fprintf(stderr, "seed %u Q=%zu kill bytes:", r0, Q.size());
                for (uint8_t x : kb) fprintf(stderr, " %u", x);
                fprintf(stderr, "\n");
            }
            if (kb.size() > 3) continue;
            // the one slot written inside the set (steps that are not the last one)
            uint32_t wslot = 0;
            bool ok = true, accepting = false;
            auto inner = [&](uint32_t v) { const uint32_t sl = (v >> FX_SLOT_SHIFT) / 128u; if (sl) { if (wslot && wslot != sl) ok = false; wslot = sl; } };
            for (uint32_t c = 0; c < (uint32_t) t.ncls; c++) {
                if (K[c]) continue;
                const uint32_t v = ft[(size_t) Q[0] * rstride + c];
                if (!is_look(v)) { inner(v); continue; }
                const size_t m = ((v & FX_ROW_MASK) - ft2_at) / row2b;
                for (uint32_t c2 = 0; c2 < (uint32_t) t.ncls; c2++) if (!K[c2]) inner(ft2[m * r2stride + c2]);
            }
            // the last byte: its entry (a look-ahead cell sees the end of the text), then the end-of-text step
            for (uint32_t c = 0; c < (uint32_t) t.ncls && ok; c++) {
                if (K[c]) continue;
                uint32_t v = ft[(size_t) Q[0] * rstride + c];
                if (is_look(v)) v = ft2[((v & FX_ROW_MASK) - ft2_at) / row2b * r2stride + (uint32_t) t.ncls];
                if (is_look(v)) { ok = false; break; }
                uint32_t sl[4] = {0, 0, 0, 0};
                int ns = slots_of(v, sl);
                const uint32_t g = next_of(v);
                bool acc = false;
                for (int k = 0; k < ns; k++) acc |= sl[k] == S_END_EOT || sl[k] == S_END_MID;
                if (g < nrows) {
                    const uint32_t ve = ft[(size_t) g * rstride + (uint32_t) t.ncls];
                    if (is_look(ve)) { ok = false; break; }
                    uint32_t se[2] = {0, 0};
                    const int ne = slots_of(ve, se);
                    for (int k = 0; k < ne; k++) { acc |= se[k] == S_END_EOT || se[k] == S_END_MID; sl[ns + k] = se[k]; }
                    ns += ne;
                }
                if (!acc) continue;
                accepting = true;
                if (wslot) { bool has = false; for (int k = 0; k < ns; k++) has |= sl[k] == wslot; if (!has) ok = false; }
            }
            if (getenv("FLBGPU_FX_DEBUG")) fprintf(stderr, "  seed %u ok %d accepting %d wslot %u\n", r0, (int) ok, (int) accepting, wslot);
            if (!ok || !accepting) continue;
            if (kb.size() < best_kill || (kb.size() == best_kill && Q.size() > tail_rows.size())) { best_kill = kb.size(); tail_rows = Q; tail_kill = kb; }
        }
    }
    uint32_t tail_min = absorb;
    if (!tail_rows.empty()) {
        std::vector<uint32_t> perm(nrows + 2);
        std::vector<char> inq(nrows, 0);
        for (uint32_t q : tail_rows) inq[q] = 1;
        uint32_t at = 0;
        for (uint32_t r = 0; r < nrows; r++) if (!inq[r]) perm[r] = at++;
        tail_min = rowaddr(at);
        for (uint32_t r = 0; r < nrows; r++) if (inq[r]) perm[r] = at++;
        perm[nrows] = nrows; perm[nrows + 1] = nrows + 1;
        auto remap = [&](uint32_t v) -> uint32_t {
            if (v & 0x80000000u) return v;
            return (v & ~FX_ROW_MASK) | rowaddr(perm[(v & FX_ROW_MASK) >= ft_at ? ((v & FX_ROW_MASK) - ft_at) / rowb : 0]);
        };
        std::vector<uint32_t> nft(ft.size());
        for (uint32_t r = 0; r < nrows + 2; r++)
            for (uint32_t c = 0; c < rstride; c++) nft[(size_t) perm[r] * rstride + c] = c < stride ? remap(ft[(size_t) r * rstride + c]) : 0u;
        ft.swap(nft);
        for (auto &v : ft2) v = remap(v);
        for (size_t k = 0; k + 1 < p2.size(); k += 2) p2[k] = remap(p2[k]);
        out.start_off = 0;      // (set below, through perm)
        tail_rows.assign(perm.begin(), perm.end());          // now: logical -> physical
    }
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
    {
        const uint32_t start_row = ((uint32_t) t.nX - 1) * (uint32_t) t.NKp + (uint32_t) t.kind_edge;
        out.start_off = rowaddr(tail_min != absorb ? tail_rows[start_row] : start_row);
    }
    out.tail_min = tail_min; out.nkill = (uint32_t) tail_kill.size();
    for (size_t k = 0; k < tail_kill.size() && k < 4; k++) out.kill[k] = tail_kill[k];
    out.absorb_off = absorb; out.poison_off = poison; out.nslots = nslots;
    out.pair_bias = bias; out.ncls1 = stride;
    out.ok = 1;
    return true;
}


// ---- fx3: the compact tables WITHOUT special entries (k_parser_reg<.., FX3>, round 4).
// In the tables above a step is a table read, a test of the entry's top bit by the whole wave, and -- in two steps of five on access
// logs -- a second read (the look-ahead cell), a select and a second test (a double capture write): a compare, a scalar branch and
// often another LDS round trip on the walk's ONE dependent chain.  Here every cell is 8 bytes
//     lo = address of the next row | (slot B * 128) << 16        hi = slot A * 128
// and a step is ONE ds_read_b64 and two unconditional capture writes: slot A gets position j - 1, slot B position j.  That second
// write is what dissolves the special entries:
//   * look-ahead: the cell "the next byte decides" leads (no write) to a PENDING row P_m; its cell for class c2 holds what the
//     resolved entry ft2[m][c2] = (row n, slot s) and then the step from n on c2 give together: slot A = s (position j - 1: the byte
//     the look-ahead stood on), slot B and the next row from ft[n][c2] (which may be another pending row);
//   * two capture writes at one position: one goes out now (slot B), the other is CARRIED into the next step's slot A -- the
//     next row is a copy (n, y) of row n whose every cell has slot A = y (written with the next step's j - 1, the same position).
//     The walk runs one position past the end-of-text column so that a write carried out of that column lands too.
// What cannot be spelled with two writes per step (a look-ahead that resolves to a double write, a carry into a cell that already
// carries) goes to the absorbing row with the FAIL slot: the record takes the generic kernel.  No tail, no pair cells: a row's cells
// are all there is.  Result slots, sentinel byte and poison row are those of the tables above.
bool flbgpu::build_fx3(const rx::TableSet &t, int ncap, std::vector<uint8_t> &b, DevFx &out, int pairs) {
    memset(&out, 0, sizeof(out));
    if (!t.has_capture || !t.ascii_only || t.ft.empty() || t.stub) return true;
    const uint32_t ncol = (uint32_t) t.ncls + 1;                                 // byte classes + the end-of-text column
    const uint32_t nbase = (uint32_t) t.nX * (uint32_t) t.NKp, ABS = nbase, POI = nbase + 1;
    const size_t W = (size_t) 1 << t.wsh, ncols2 = (size_t) 1 << t.fc_shift, nm = t.ft2.size() / ncols2;
    const uint32_t nslots = (uint32_t) ncap + 5;
    if (nslots > 63) return true;
    const uint32_t S_END_EOT = (uint32_t) ncap + FXS_END_EOT, S_END_MID = (uint32_t) ncap + FXS_END_MID,
                   S_DEAD_EOT = (uint32_t) ncap + FXS_DEAD_EOT, S_FAIL = (uint32_t) ncap + FXS_FAIL;
    // an entry of the kernel tables in the abstract: kind 0 one write (s1), 1 two writes (s1, s2), 2 look-ahead row m
    struct Ent { int kind; uint32_t next, s1, s2, m; };
    auto conv = [&](uint32_t e, bool eot) -> Ent {
        if (e & rx::FT_SPECIAL) {
            const uint32_t ty = rx::ft_type(e);
            if (ty == rx::FT_LOOK) { if ((e & 0xFFFFFF) >= nm) return Ent{0, ABS, S_FAIL, 0, 0}; return Ent{2, 0, 0, 0, e & 0xFFFFFF}; }
            if (ty == rx::FT_MATCH) {
                const uint32_t a = (e >> 12) & 63, c = (e >> 18) & 63, endslot = eot ? S_END_EOT : S_END_MID;
                if (a && c) return Ent{0, ABS, S_FAIL, 0, 0};                    // three writes
                if (a || c) return Ent{1, ABS, a ? a : c, endslot, 0};
                return Ent{0, ABS, endslot, 0, 0};
            }
            if (ty == rx::FT_MULTI) return Ent{0, ABS, S_FAIL, 0, 0};
            return Ent{0, ABS, eot ? S_DEAD_EOT : 0u, 0, 0};                     // dead end
        }
        const uint32_t a = (e >> 12) & 63, c = (e >> 18) & 63;
        if (a && c) return Ent{1, e & 0xFFF, a, c, 0};
        return Ent{0, e & 0xFFF, a ? a : c, 0, 0};
    };
    auto base_ent = [&](uint32_t r, uint32_t c) -> Ent {                        // row r (base, ABS, POI), column c
        const bool eot = c == (uint32_t) t.ncls;
        if (r == ABS) return Ent{0, ABS, 0, 0, 0};
        if (r == POI) return Ent{0, POI, 0, 0, 0};
        if (!eot && (int) c == t.high_cls) return Ent{0, POI, 0, 0, 0};
        const uint32_t kind = eot ? (uint32_t) t.kind_edge : t.kind_of_cls[c];
        return conv(t.ft[(size_t) r * W + (((size_t) kind << t.fc_shift) | c)], eot);
    };
    auto look_ent = [&](uint32_t m, uint32_t c) -> Ent {                        // what look-ahead row m resolves to when the NEXT byte has class c
        if (c < (uint32_t) t.ncls && (int) c == t.high_cls) return Ent{0, ABS, S_FAIL, 0, 0};      // a byte >= 0x80 follows: the UTF-8 tables decide
        Ent v = conv(t.ft2[m * ncols2 + c], false);
        if (v.kind == 2) return Ent{0, ABS, S_FAIL, 0, 0};
        return v;
    };
    // new rows: (base row r | pending row of look m, carried slot y), made on demand
    struct Key { uint32_t pend, id, carry; bool operator<(const Key &o) const { return pend != o.pend ? pend < o.pend : id != o.id ? id < o.id : carry < o.carry; } };
    std::map<Key, uint32_t> ids;
    std::vector<Key> rows;
    auto row_id = [&](uint32_t pend, uint32_t id, uint32_t carry) -> uint32_t {
        const Key k{pend, id, carry};
        auto it = ids.find(k);
        if (it != ids.end()) return it->second;
        const uint32_t n = (uint32_t) rows.size();
        ids.emplace(k, n); rows.push_back(k);
        return n;
    };
    struct Cell { uint32_t next, sa, sb; };
    std::vector<Cell> cells;
    const uint32_t start_base = ((uint32_t) t.nX - 1) * (uint32_t) t.NKp + (uint32_t) t.kind_edge;
    row_id(0, start_base, 0); row_id(0, ABS, 0); row_id(0, POI, 0);              // rows 0, 1, 2: start, absorbing, poison
    const Cell FAILC{1, 0, S_FAIL};
    // the step `v` taken with `sa` already owed to position j - 1: where it leads and what it writes
    auto finish = [&](const Ent &v, uint32_t sa) -> Cell {
        if (v.kind == 2) return Cell{row_id(1, v.m, 0), sa, 0};
        if (v.kind == 1) return Cell{row_id(0, v.next, v.s2), sa, v.s1};
        return Cell{row_id(0, v.next, 0), sa, v.s1};
    };
    for (size_t ri = 0; ri < rows.size(); ri++) {
        if (rows.size() > 4000) return true;
        const Key k = rows[ri];
        for (uint32_t c = 0; c < ncol; c++) {
            Cell out_c;
            if (!k.pend) out_c = finish(base_ent(k.id, c), k.carry);
            else {
                // the look-ahead stood on the byte in front of this one: its resolution's write belongs to position j - 1, then the
                // step from the resolved row on this byte
                const Ent r = look_ent(k.id, c);
                if (r.kind == 1) out_c = FAILC;                                   // two writes at j - 1 and the step's own: three
                else out_c = finish(base_ent(r.next, c), r.s1);
            }
            cells.push_back(out_c);
        }
    }
    if (pairs) {
        // ---- fx4: TWO positions per cell.  The cell of (row R, class of byte j, class of byte j + 1) holds what the two fx3 steps
        // give together: the row behind both, and four capture slots -- step 1's slot A (position j - 1) and slot B (j), step 2's slot
        // A (its j - 1 = j) and slot B (j + 1).  One ds_read_b64 per two bytes on the walk's dependent chain.  Rows are the fx3 rows
        // reachable at EVEN positions; 8 bytes per cell: lo = next row | s0 << 16 | s1 << 22, hi = s2 | s3 << 6 (slot numbers).
        // Two class tables in front: even positions give class * ncol * 8, odd ones class * 8.
        std::vector<uint32_t> id4(rows.size(), 0xFFFFFFFFu), order;
        auto r4 = [&](uint32_t r3) -> uint32_t { if (id4[r3] == 0xFFFFFFFFu) { id4[r3] = (uint32_t) order.size(); order.push_back(r3); } return id4[r3]; };
        r4(0); r4(1); r4(2);
        struct C4 { uint32_t next, s0, s1, s2, s3; };
        std::vector<C4> c4;
        uint32_t n_conflict = 0;
        for (size_t i = 0; i < order.size(); i++) {
            const uint32_t R = order[i];
            for (uint32_t c0 = 0; c0 < ncol; c0++)
                for (uint32_t c1 = 0; c1 < ncol; c1++) {
                    const Cell &x = cells[(size_t) R * ncol + c0];
                    const Cell &y = cells[(size_t) x.next * ncol + c1];
                    C4 v{r4(y.next), x.sa, x.sb, y.sa, y.sb};
                    // pairs == 2 (fx5, round 5): THREE write ports per cell -- positions j - 1, j, j + 1 --: the two writes at position j
                    // (step 1's slot B, step 2's slot A) share one port.  A cell that needs both (apache2: cells no well-formed line
                    // reaches -- 0 of 258 k steps over 2 000 lines) goes to the absorbing row with the FAIL slot like everything else the
                    // cells cannot spell: the record takes the generic kernel.
                    if (pairs == 2 && v.s1 && v.s2) { n_conflict++; v = C4{r4(1), 0, S_FAIL, 0, 0}; }
                    c4.push_back(v);
                }
        }
        if (getenv("FLBGPU_FX_DEBUG")) fprintf(stderr, "fx pairs=%d: %zu rows, %u cells with two writes at one position\n", pairs, order.size(), n_conflict);
        if (nslots > 34) return true;                                            // (a lane's block of capture slots is 68 bytes: the caller keeps fx3)
        const uint32_t n4 = (uint32_t) order.size(), rs4 = (ncol * ncol) | 1u, rowb4 = rs4 * 8, at4 = 2048;
        const uint64_t tot4 = (uint64_t) at4 + (uint64_t) n4 * rowb4;
        if (tot4 > 60000) return true;                                           // (the caller keeps fx3)
        std::vector<uint32_t> ca(256), cb(256);
        for (int i = 0; i < 256; i++) { ca[(size_t) i] = (uint32_t) t.cls[i] * ncol * 8u; cb[(size_t) i] = (uint32_t) t.cls[i] * 8u; }
        ca[255] = (uint32_t) t.ncls * ncol * 8u; cb[255] = (uint32_t) t.ncls * 8u;
        b.assign((size_t) ((tot4 + 15) & ~15ull), 0);
        memcpy(b.data(), ca.data(), 1024);
        memcpy(b.data() + 1024, cb.data(), 1024);
        for (uint32_t r = 0; r < n4; r++)
            for (uint32_t c = 0; c < ncol * ncol; c++) {
                const C4 &x = c4[(size_t) r * ncol * ncol + c];
                // lo = the next row, nothing else (the next cell's address is ONE three-operand addition: row + the two class offsets);
                // hi = the four slots' BYTE offsets in a lane's block of capture slots (slot * 2): the kernel adds a selected byte to the
                // lane's block address -- one operation per write (tile_kernels.inc REG_SLOT)
                const uint32_t lo = at4 + x.next * rowb4,
                               hi = pairs == 2 ? (x.s0 * 2) | (((x.s1 ? x.s1 : x.s2) * 2) << 8) | ((x.s3 * 2) << 16)
                                               : (x.s0 * 2) | ((x.s1 * 2) << 8) | ((x.s2 * 2) << 16) | ((x.s3 * 2) << 24);
                memcpy(b.data() + at4 + (size_t) r * rowb4 + c * 8, &lo, 4);
                memcpy(b.data() + at4 + (size_t) r * rowb4 + c * 8 + 4, &hi, 4);
            }
        out.base = nullptr; out.bytes = (uint32_t) b.size();
        out.off_p2 = 0;
        out.start_off = at4; out.absorb_off = at4 + rowb4; out.poison_off = at4 + 2 * rowb4;
        out.tail_min = out.absorb_off; out.nkill = 0;
        out.nslots = nslots; out.pair_bias = pairs == 2 ? 2 : 1; out.ncls1 = ncol;     // pair_bias != 0: the two-position form (2: three write ports)
        out.ok = 1;
        return true;
    }
    const uint32_t nrows = (uint32_t) rows.size();
    const uint32_t rs = ncol | 1u;                                              // cells per row, odd: rows spread over the LDS banks
    const uint32_t rowb = rs * 8, at0 = 1024;
    const uint64_t total64 = (uint64_t) at0 + (uint64_t) nrows * rowb;
    if (total64 > 60000) return true;
    std::vector<uint32_t> cls(256);
    for (int i = 0; i < 256; i++) cls[(size_t) i] = (uint32_t) t.cls[i] * 8u;
    cls[255] = (uint32_t) t.ncls * 8u;                                          // the end-of-text sentinel
    b.assign((size_t) ((total64 + 15) & ~15ull), 0);
    memcpy(b.data(), cls.data(), 1024);
    for (uint32_t r = 0; r < nrows; r++)
        for (uint32_t c = 0; c < ncol; c++) {
            const Cell &x = cells[(size_t) r * ncol + c];
            const uint32_t lo = (at0 + x.next * rowb) | ((x.sb * 128u) << FX_SLOT_SHIFT), hi = x.sa * 128u;
            memcpy(b.data() + at0 + (size_t) r * rowb + c * 8, &lo, 4);
            memcpy(b.data() + at0 + (size_t) r * rowb + c * 8 + 4, &hi, 4);
        }
    out.base = nullptr; out.bytes = (uint32_t) b.size();
    out.off_p2 = 0;
    out.start_off = at0; out.absorb_off = at0 + rowb; out.poison_off = at0 + 2 * rowb;
    out.tail_min = out.absorb_off; out.nkill = 0;
    out.nslots = nslots; out.pair_bias = 0; out.ncls1 = ncol;
    out.ok = 1;
    return true;
}

// k_parser_reg<.., 4>'s walk on the host: two positions per cell, positions 0 .. len + 1 rounded up to whole pairs
int flbgpu::simulate_fx4(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps) {
    auto u32at = [&](uint32_t at) -> uint32_t { uint32_t v; memcpy(&v, b.data() + at, 4); return v; };
    auto byte_at = [&](uint32_t j) -> uint32_t { return j < len ? s[j] : j == len ? 0xFFu : 0u; };
    for (uint32_t i = 0; i < fx.nslots; i++) caps[i] = 0xFFFF;
    uint32_t e = fx.start_off;
    for (uint32_t j = 0; j <= len + 1; j += 2) {
        const uint32_t at = (e & FX_ROW_MASK) + u32at(4 * byte_at(j)) + u32at(1024 + 4 * byte_at(j + 1));
        const uint32_t lo = u32at(at), hi = u32at(at + 4);
        caps[(hi & 255) / 2] = (uint16_t) (j - 1);
        caps[((hi >> 8) & 255) / 2] = (uint16_t) j;
        if (fx.pair_bias == 2) caps[((hi >> 16) & 255) / 2] = (uint16_t) (j + 1);      // fx5: three ports
        else {
            caps[((hi >> 16) & 255) / 2] = (uint16_t) j;
            caps[((hi >> 24) & 255) / 2] = (uint16_t) (j + 1);
        }
        e = lo;
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

// k_parser_reg<.., FX3>'s walk on the host: positions 0 .. len + 1, one 8-byte cell per step, two writes
int flbgpu::simulate_fx3(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps) {
    auto u32at = [&](uint32_t at) -> uint32_t { uint32_t v; memcpy(&v, b.data() + at, 4); return v; };
    for (uint32_t i = 0; i < fx.nslots; i++) caps[i] = 0xFFFF;
    uint32_t e = fx.start_off;
    for (uint32_t j = 0; j <= len + 1; j++) {
        const uint32_t at = (e & FX_ROW_MASK) + u32at(4 * (j < len ? s[j] : j == len ? 0xFFu : 0u));      // (behind the sentinel: zero bytes)
        const uint32_t lo = u32at(at), hi = u32at(at + 4);
        caps[hi / 128] = (uint16_t) (j - 1);
        caps[(lo >> FX_SLOT_SHIFT) / 128] = (uint16_t) j;
        e = lo;
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


// The walk of k_parser_tile (tile_kernels.inc fx_walk + the result rules that follow it) on the host.
// caps: nslots u16 columns (0xFFFF = unset).  Returns >= 0 end of the match, -1 the forward walk from boundary 0 does
// not settle the value (the kernel then runs the reverse pass + classic walk), -2 a byte >= 0x80 / a real 0xFF.
int flbgpu::simulate_fx(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps, bool use_tail) {
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
    if (!bias) {
        // k_parser_reg: every 16 positions the wave looks whether all its lanes stand in the pattern's tail (or are absorbed); here the
        // one lane leaves at the FIRST such boundary -- the most the tail logic is ever asked to do
        const bool allow = use_tail && len < 272;            // (the kernel scans the value's registers for kill bytes)
        for (uint32_t i = 0; i < fx.nslots; i++) caps[i] = 0xFFFF;
        e = fx.start_off;
        for (uint32_t j = 0; j <= len; j++) {
            if (allow && j > 0 && (j & 15) == 0 && (e & FX_ROW_MASK) >= fx.tail_min) {
                if ((e & FX_ROW_MASK) >= fx.absorb_off) break;                            // absorbed: nothing changes any more
                bool kill = false;
                for (uint32_t q = j; q < len; q++) {
                    kill |= s[q] >= 0x80;
                    for (uint32_t k = 0; k < fx.nkill; k++) kill |= s[q] == fx.kill[k];
                }
                if (kill) return -2;                                                      // the way of the poisoned records: the complete algorithm decides
                if (len > j) step1(len - 1);              // the row it stands in is as good as the one it would have reached: the last byte decides
                step1(len);
                break;
            }
            step1(j);
        }
    }
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
