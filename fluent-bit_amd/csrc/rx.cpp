// rx.cpp -- regex front-end + table compiler (see rx.hpp).
//
// Semantics implemented are those the reference gets from Onigmo 6.2.0 under
// ONIG_SYNTAX_RUBY / ONIG_ENCODING_UTF8 / ONIG_OPTION_DEFAULT (src/flb_regex.c:142-145):
//   ^ $ are line anchors, \A \z string anchors, '.' excludes \n unless (?m);
//   \d \s \w \h and POSIX brackets are ASCII-range (ONIG_OPTION_ASCII_RANGE is forced for Ruby
//   syntax: lib/onigmo/regcomp.c:5842-5850), their negations contain every non-ASCII code point;
//   plain groups do not capture once a named group exists (lib/onigmo/regparse.c:971-984);
//   an empty iteration of an unbounded loop leaves the loop (OP_NULL_CHECK_END, regexec.c).
// Constructs that need a stack (back-references, look-around, atomic/possessive, absent, calls,
// conditionals) are rejected: the filter fails to initialise rather than fall back to a CPU path.
#include "rx.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>
#include <unordered_map>
#include <sys/mman.h>
#include <ucontext.h>

namespace rx {
namespace {

// ---------------------------------------------------------------- code point sets
using Range = std::pair<uint32_t, uint32_t>;
constexpr uint32_t MAXCP = 0x10FFFF;

struct CodeSet {
    std::vector<Range> r;
    void add(uint32_t lo, uint32_t hi) { if (lo <= hi) r.emplace_back(lo, hi); }
    void norm() {
        std::sort(r.begin(), r.end());
        std::vector<Range> o;
        for (auto &x : r) {
            if (!o.empty() && x.first <= o.back().second + 1) o.back().second = std::max(o.back().second, x.second);
            else o.push_back(x);
        }
        r.swap(o);
    }
    void negate() {
        norm();
        std::vector<Range> o;
        uint32_t next = 0;
        for (auto &x : r) {
            if (x.first > next) o.emplace_back(next, x.first - 1);
            next = x.second + 1;
        }
        if (next <= MAXCP) o.emplace_back(next, MAXCP);
        r.swap(o);
    }
    void merge(const CodeSet &s) { r.insert(r.end(), s.r.begin(), s.r.end()); }
    bool has(uint32_t c) const {
        for (auto &x : r) if (c >= x.first && c <= x.second) return true;
        return false;
    }
    void fold_ascii_case() {
        size_t n = r.size();
        for (size_t i = 0; i < n; i++) {
            uint32_t lo = r[i].first, hi = r[i].second;
            uint32_t a = std::max<uint32_t>(lo, 'a'), b = std::min<uint32_t>(hi, 'z');
            if (a <= b) add(a - 32, b - 32);
            a = std::max<uint32_t>(lo, 'A'); b = std::min<uint32_t>(hi, 'Z');
            if (a <= b) add(a + 32, b + 32);
        }
    }
};

// ---------------------------------------------------------------- character class model
// What one "character" node of the pattern accepts, kept in the two-part form the reference's engine
// compiles a class to (lib/onigmo/regcomp.c compile_cclass_node, regexec.c OP_CCLASS* :2018-2138),
// because the two parts are consulted differently on input that is not well-formed UTF-8:
//   bs   256 bits tested on the FIRST BYTE whenever the character is not a multi-byte head (ASCII, a byte
//        that cannot start or continue a sequence here, a lone lead byte at the end) -- and, when the class
//        has no code-range part at all, on the first byte of EVERY character;
//   mb   code points >= 0x80, tested on the decoded character when it is a multi-byte head: a well-formed
//        sequence, or a prefix-valid sequence cut by the end of the text, which counts as ONE character
//        that spans the rest of the text and whose code is its lead byte (enc/utf_8.c mbc_to_code with a
//        NEEDMORE length; regenc.c onigenc_mbclen);
//   neg  the class is negated as a whole (kept as a flag: [^ ] has bs = {' '}, no mb, neg).
constexpr uint32_t LASTCP = 0x7fffffffu;          // ONIG_LAST_CODE_POINT: the complement of a code-range part

struct ByteSet {
    uint64_t w[4] = {0, 0, 0, 0};
    void set(int b) { w[b >> 6] |= 1ull << (b & 63); }
    void clear(int b) { w[b >> 6] &= ~(1ull << (b & 63)); }
    void set_range(int lo, int hi) { for (int b = lo; b <= hi; b++) set(b); }
    bool has(int b) const { return (w[b >> 6] >> (b & 63)) & 1; }
    bool empty() const { return !(w[0] | w[1] | w[2] | w[3]); }
    void invert() { for (auto &x : w) x = ~x; }
    void merge(const ByteSet &o) { for (int i = 0; i < 4; i++) w[i] |= o.w[i]; }
};

// complement of a code-range part inside [0x80, LASTCP] (regparse.c not_code_range_buf)
CodeSet mb_complement(const CodeSet &in) {
    CodeSet s = in, o;
    s.norm();
    uint32_t next = 0x80;
    bool done = false;
    for (auto &x : s.r) {
        if (x.second < 0x80) continue;
        uint32_t lo = std::max<uint32_t>(x.first, 0x80);
        if (lo > next) o.add(next, lo - 1);
        if (x.second >= LASTCP) { done = true; break; }
        next = x.second + 1;
    }
    if (!done) o.add(next, LASTCP);
    return o;
}

CodeSet unicode_word_set();           // {cp >= 0x80 : unicode_word(cp)} (defined behind posix_ranges.inc)

struct CC {
    enum Kind { CLASS, ANY, WORD, LIT } kind = CLASS;
    ByteSet bs;
    CodeSet mb;
    // what the automaton is BUILT from for well-formed multi-byte characters.  Equal to mb since round 3 (the Unicode members
    // of a POSIX bracket are in: [[:alpha:]] accepts U+00E9 as the reference does; a pattern whose UTF-8 tables then need
    // more byte classes than the tables hold is refused at create).
    CodeSet mbx;
    bool neg = false;
    // (?i): the ASCII members that came in as characters, ranges, \d \s \h or POSIX brackets other than [:word:] / [:ascii:] --
    // NOT through \w.  The reference keeps this shadow class (regparse.c parse_char_class `asc_cc`, :4739-4748, :4316-4328)
    // and lets a case fold cross the ASCII boundary (k -> U+212A, s -> U+017F) only for members of it (i_apply_case_fold :5543-5553):
    // (?i)[a-z] accepts the Kelvin sign, (?i)[\w.-] does not.
    ByteSet asc;
    uint32_t lit = 0;                 // LIT: the code point (matched as its exact byte sequence)
    bool any_nl = false;              // ANY: also matches \n  ((?m))
    bool uni = false;                 // WORD under (?u): OP_WORD instead of OP_ASCII_WORD -- ONIGENC_IS_MBC_WORD on the decoded character

    bool has_mb() const { for (auto &x : mb.r) if (x.second >= 0x80 && x.first <= x.second) return true; return false; }
    // dest |= o, o's negation resolved first (regparse.c or_cclass / or_code_range_buf with not1 == 0)
    void merge_class(const CC &o) {
        ByteSet b2 = o.bs;
        if (o.neg) b2.invert();
        bs.merge(b2);
        ByteSet a2 = o.asc;
        if (o.neg) { a2.invert(); for (int b = 0x80; b < 256; b++) a2.clear(b); }
        asc.merge(a2);
        if (o.neg) { CodeSet c = mb_complement(o.mb); mb.merge(c); CodeSet cx = mb_complement(o.mbx); mbx.merge(cx); }
        else { mb.merge(o.mb); mbx.merge(o.mbx); }
        mb.norm(); mbx.norm();
    }
    void add_cp(uint32_t lo, uint32_t hi) {
        // regparse.c next_state_val: single-byte values go to the bit set, code points to the range part; a
        // range that starts single-byte and ends above sets the bits up to min(end, 0xff) AND the whole range
        if (hi < 0x80) { bs.set_range((int) lo, (int) hi); asc.set_range((int) lo, (int) hi); return; }
        if (lo < 0x80) { bs.set_range((int) lo, (int) std::min<uint32_t>(hi, 0xff)); asc.set_range((int) lo, 0x7f); }
        mb.add(lo, hi);
        mbx.add(lo, hi);
    }

    // ---- what the class accepts, by the kind of character at the position
    bool ascii(int b) const {
        switch (kind) {
        case CLASS: return bs.has(b) != neg;
        case ANY: return b != '\n' || any_nl;
        case WORD: return (((b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_')) != neg;
        case LIT: return (uint32_t) b == lit;
        }
        return false;
    }
    // a byte >= 0x80 that is a character of its own (no sequence starts here)
    bool invalid_byte(int b) const {
        switch (kind) {
        case CLASS: return bs.has(b) != neg;
        case ANY: return true;
        case WORD: return (uni && b < 0xfe && unicode_word((uint32_t) b)) != neg;
        case LIT: return false;
        }
        return false;
    }
    // a prefix-valid sequence of two or more bytes cut by the end of the text, lead byte b
    bool truncated(int b) const {
        switch (kind) {
        case CLASS: return (has_mb() ? mb.has((uint32_t) b) : bs.has(b)) != neg;
        case ANY: return true;
        case WORD: return (uni && unicode_word((uint32_t) b)) != neg;
        case LIT: return false;
        }
        return false;
    }
    // the well-formed multi-byte characters accepted, as a code point set within [0x80, MAXCP]
    CodeSet valid_multibyte() const {
        CodeSet o;
        switch (kind) {
        case CLASS:
            if (has_mb()) {
                CodeSet m = neg ? mb_complement(mbx) : mbx;
                m.norm();
                for (auto &x : m.r) if (x.second >= 0x80 && x.first <= MAXCP) o.add(std::max<uint32_t>(x.first, 0x80), std::min(x.second, MAXCP));
            }
            else {
                // no code-range part: the bit of the LEAD byte decides for the whole character
                for (int b = 0xc2; b <= 0xf4; b++) {
                    if (bs.has(b) == neg) continue;
                    if (b <= 0xdf) o.add((uint32_t) (b & 0x1f) << 6, ((uint32_t) (b & 0x1f) << 6) | 0x3f);
                    else if (b <= 0xef) o.add(std::max<uint32_t>((uint32_t) (b & 0x0f) << 12, 0x800), ((uint32_t) (b & 0x0f) << 12) | 0xfff);
                    else o.add(std::max<uint32_t>((uint32_t) (b & 0x07) << 18, 0x10000), std::min<uint32_t>(((uint32_t) (b & 0x07) << 18) | 0x3ffff, MAXCP));
                }
            }
            break;
        case ANY: o.add(0x80, MAXCP); break;
        case WORD:
            if (!uni) { if (neg) o.add(0x80, MAXCP); }
            else {
                CodeSet w = unicode_word_set();
                if (neg) { w = mb_complement(w); w.norm(); for (auto &x : w.r) if (x.first <= MAXCP) o.add(x.first, std::min(x.second, MAXCP)); }
                else o = w;
            }
            break;
        case LIT: if (lit >= 0x80) o.add(lit, lit); break;
        }
        o.norm();
        return o;
    }
};

#include "posix_ranges.inc"

CodeSet unicode_word_set() {
    CodeSet w;
    for (uint32_t c : {0xb2u, 0xb3u, 0xb9u, 0xbcu, 0xbdu, 0xbeu}) w.add(c, c);
    for (int i = 0; i < posix_u_word_n; i++) w.add(posix_u_word[i][0], posix_u_word[i][1]);
    w.norm();
    return w;
}

}  // namespace
// what \b / \B ask of a character (regexec.c OP_WORD_BOUND: ONIGENC_IS_MBC_WORD -> onigenc_unicode_is_code_ctype): below U+0100
// the encoding's Latin-1 table decides (enc/unicode.c EncUNICODE_ISO_8859_1_CtypeTable: superscripts and vulgar fractions
// U+00B2 B3 B9 BC BD BE are word characters THERE, though [[:word:]] -- built from the CR_Word ranges -- rejects them), the code
// ranges from U+0100 on
bool unicode_word(uint32_t cp) {
    if (cp < 0x80) return (cp >= '0' && cp <= '9') || (cp >= 'A' && cp <= 'Z') || (cp >= 'a' && cp <= 'z') || cp == '_';
    if (cp == 0xb2 || cp == 0xb3 || cp == 0xb9 || cp == 0xbc || cp == 0xbd || cp == 0xbe) return true;
    int lo = 0, hi = posix_u_word_n - 1;
    while (lo <= hi) {
        const int m = (lo + hi) / 2;
        if (cp < posix_u_word[m][0]) hi = m - 1;
        else if (cp > posix_u_word[m][1]) lo = m + 1;
        else return true;
    }
    return false;
}
const unsigned int (*unicode_word_ranges(int *n))[2] { *n = posix_u_word_n; return posix_u_word; }
namespace {

// (?a) / (?u) / (?d) (regparse.c:5257-5297): three option bits of the reference, kept as "off" flags so that 0 is Ruby's default
// (ONIG_OPTION_ASCII_RANGE | POSIX_BRACKET_ALL_RANGE | WORD_BOUND_ALL_RANGE, regcomp.c:5842-5850):
//   (?a)  ASCII_RANGE on,  POSIX_BRACKET_ALL_RANGE off, WORD_BOUND_ALL_RANGE off
//   (?u)  ASCII_RANGE off, POSIX_BRACKET_ALL_RANGE off, WORD_BOUND_ALL_RANGE off
//   (?d)  all three on (the default)
// \d \s \w \h are ASCII iff ASCII_RANGE; a POSIX bracket iff ASCII_RANGE && !POSIX_BRACKET_ALL_RANGE (regparse.c:4312);
// \b \B iff ASCII_RANGE && !WORD_BOUND_ALL_RANGE (regparse.c:3437)
constexpr unsigned OPT_AR_OFF = 8u, OPT_PBA_OFF = 16u, OPT_WBA_OFF = 32u;
inline bool ctype_is_ascii(unsigned o) { return !(o & OPT_AR_OFF); }
inline bool posix_is_ascii(unsigned o) { return !(o & OPT_AR_OFF) && (o & OPT_PBA_OFF); }
inline bool wordb_is_ascii(unsigned o) { return !(o & OPT_AR_OFF) && (o & OPT_WBA_OFF); }

// \d \s \w \h inside or outside brackets: ASCII-range under Ruby syntax (ONIG_OPTION_ASCII_RANGE).  The
// positive form has no code-range part; the negated form is "ASCII complement + every code point >= 0x80"
// (regparse.c add_ctype_to_cc with ascii_range)
void ctype_ascii(ByteSet &bs, char t) {
    switch (t) {
    case 'd': bs.set_range('0', '9'); break;
    case 'w': bs.set_range('0', '9'); bs.set_range('A', 'Z'); bs.set_range('a', 'z'); bs.set('_'); break;
    case 's': bs.set_range(9, 13); bs.set(' '); break;
    case 'h': bs.set_range('0', '9'); bs.set_range('A', 'F'); bs.set_range('a', 'f'); break;
    }
}
bool add_posix(CC &cc, const std::string &n, bool negated, bool ascii_range);
void add_ctype(CC &cc, char t, bool negated, bool unicode = false) {
    if (unicode) {
        // (?u): the Unicode property ranges (add_ctype_to_cc without ascii_range = what a POSIX bracket of the same type adds)
        add_posix(cc, t == 'd' ? "digit" : t == 's' ? "space" : t == 'h' ? "xdigit" : "word", negated, false);
        return;
    }
    ByteSet a;
    ctype_ascii(a, t);
    if (t != 'w') { for (int b = 0; b < 0x80; b++) if (a.has(b) != negated) cc.asc.set(b); }
    if (!negated) { cc.bs.merge(a); return; }
    for (int b = 0; b < 0x80; b++) if (!a.has(b)) cc.bs.set(b);
    cc.mb.add(0x80, LASTCP);
    cc.mb.norm();
    cc.mbx.add(0x80, LASTCP);
    cc.mbx.norm();
}

// \p{Name} beyond the POSIX bracket names: general categories, scripts, binary properties, blocks (regparse.c parse_char_property /
// parse_char_class TK_CHAR_PROPERTY -> add_ctype_to_cc(cc, ctype, not, 0): the property's code ranges, the ASCII ones included, never
// ASCII-range; inside brackets the (?i) shadow class gets them too).  Members: unicode_props.inc, generated by probing the
// reference's engine on every code point (tools/gen_unicode_props.py); `n` as the engine normalises a name.
#include "unicode_props.inc"
bool add_uniprop(CC &cc, const std::string &n, bool negated) {
    size_t lo = 0, hi = sizeof(uprop_names) / sizeof(uprop_names[0]);
    int set = -1;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const int c = strcmp(n.c_str(), uprop_names[mid].name);
        if (c == 0) { set = (int) uprop_names[mid].set; break; }
        if (c < 0) hi = mid;
        else lo = mid + 1;
    }
    if (set < 0) return false;
    ByteSet a;
    CodeSet m;
    const unsigned int first = uprop_sets[set][0], cnt = uprop_sets[set][1];
    for (unsigned int i = first; i < first + cnt; i++) {
        const unsigned int l = uprop_ranges[i][0], h = uprop_ranges[i][1];
        for (unsigned int b = l; b <= h && b < 0x80; b++) a.set((int) b);
        if (h >= 0x80) m.add(l < 0x80 ? 0x80 : l, h);
    }
    for (int b = 0; b < 0x80; b++) if (a.has(b) != negated) cc.asc.set(b);
    if (!negated) { cc.bs.merge(a); cc.mb.merge(m); cc.mbx.merge(m); }
    else {
        for (int b = 0; b < 0x80; b++) if (!a.has(b)) cc.bs.set(b);
        CodeSet c = mb_complement(m);
        cc.mb.merge(c);
        cc.mbx.merge(c);
    }
    cc.mb.norm(); cc.mbx.norm();
    return true;
}

// [[:name:]] / [[:^name:]]: NOT ASCII-range (ONIG_OPTION_POSIX_BRACKET_ALL_RANGE): the code points >= 0x80
// come from posix_ranges.inc (generated by probing the reference's engine, tools/gen_posix_ranges.py)
bool add_posix(CC &cc, const std::string &n, bool negated, bool ascii_range = false) {
    ByteSet a;
    const unsigned int (*u)[2] = nullptr;
    int un = 0;
#define PX(nm) u = posix_u_##nm; un = posix_u_##nm##_n
    if (n == "alpha") { a.set_range('A', 'Z'); a.set_range('a', 'z'); PX(alpha); }
    else if (n == "digit") { a.set_range('0', '9'); PX(digit); }
    else if (n == "alnum") { a.set_range('0', '9'); a.set_range('A', 'Z'); a.set_range('a', 'z'); PX(alnum); }
    else if (n == "upper") { a.set_range('A', 'Z'); PX(upper); }
    else if (n == "lower") { a.set_range('a', 'z'); PX(lower); }
    else if (n == "space") { a.set_range(9, 13); a.set(' '); PX(space); }
    else if (n == "blank") { a.set(9); a.set(' '); PX(blank); }
    else if (n == "cntrl") { a.set_range(0, 31); a.set(127); PX(cntrl); }
    else if (n == "punct") { a.set_range(33, 47); a.set_range(58, 64); a.set_range(91, 96); a.set_range(123, 126); PX(punct); }
    else if (n == "graph") { a.set_range(33, 126); PX(graph); }
    else if (n == "print") { a.set_range(32, 126); PX(print); }
    else if (n == "xdigit") { a.set_range('0', '9'); a.set_range('A', 'F'); a.set_range('a', 'f'); PX(xdigit); }
    else if (n == "word") { a.set_range('0', '9'); a.set_range('A', 'Z'); a.set_range('a', 'z'); a.set('_'); PX(word); }
    else if (n == "ascii") { a.set_range(0, 127); PX(ascii); }
    else return false;
#undef PX
    CodeSet m;
    for (int i = 0; i < un; i++) m.add(u[i][0], u[i][1]);
    if (ascii_range) {
        // (?a): add_ctype_to_cc with ascii_range (regparse.c:4153-4180): the members below 0x80; negated: their ASCII complement and
        // every code point from 0x80 on.  (The shadow class `asc` of (?i) does not get them: regparse.c:4320-4324 `!ascii_range`.)
        if (!negated) { cc.bs.merge(a); return true; }
        for (int b = 0; b < 0x80; b++) if (!a.has(b)) cc.bs.set(b);
        cc.mb.add(0x80, LASTCP); cc.mb.norm();
        cc.mbx.add(0x80, LASTCP); cc.mbx.norm();
        return true;
    }
    if (n != "word" && n != "ascii") { for (int b = 0; b < 0x80; b++) if (a.has(b) != negated) cc.asc.set(b); }
    if (!negated) { cc.bs.merge(a); cc.mb.merge(m); cc.mbx.merge(m); }
    else {
        for (int b = 0; b < 0x80; b++) if (!a.has(b)) cc.bs.set(b);
        CodeSet c = mb_complement(m);
        cc.mb.merge(c);
        cc.mbx.merge(c);
    }
    cc.mb.norm(); cc.mbx.norm();
    return true;
}

// (?i): the class is closed under case folding.  ASCII letters fold among themselves and 'k' / 's' also
// with U+212A KELVIN SIGN / U+017F LONG S (the only single-character folds that reach an ASCII letter);
// a class that holds part of the non-ASCII range would need the full fold tables: refused.
bool fold_case(CC &cc) {
    const bool had_k = cc.bs.has('k'), had_K = cc.bs.has('K'), had_s = cc.bs.has('s'), had_S = cc.bs.has('S');
    for (int c = 'a'; c <= 'z'; c++) {
        if (cc.bs.has(c) || cc.bs.has(c - 32)) { cc.bs.set(c); cc.bs.set(c - 32); }
    }
    CodeSet rest = mb_complement(cc.mb);
    bool partial = cc.has_mb() && !rest.r.empty();
    if (partial) {
        // only the two letters' partners may be there (added by this very function on an enclosing level)
        for (auto &x : cc.mb.r) {
            if (x.second < 0x80) continue;
            if (x.first != x.second || (x.first != 0x212a && x.first != 0x17f)) return false;
        }
    }
    // A class that holds EVERY code point from 0x80 on ([\D], [\W], a negated ASCII-range bracket ...): the reference's fold pass
    // (regparse.c i_apply_case_fold: for every fold pair (from, to) with `from` in the class it adds `to` -- into the BIT SET when
    // to < 0x100) sets the bits of the Latin-1 letters that have a case partner, and the bit set is what a stray byte >= 0x80 -- a
    // character of its own whose code is the byte -- is tested on: (?i)[\D] takes a lone 0xE9, [\D] does not (probed from the engine)
    if (cc.has_mb() && !partial) {
        cc.bs.set(0xb5);
        for (int b = 0xc0; b <= 0xff; b++) if (b != 0xd7 && b != 0xf7) cc.bs.set(b);
    }
    // (the letters were closed under ASCII case above, from the members as they were: the shadow class decides for each of the two)
    const bool kx = cc.mb.has(0x212a) || (had_k && cc.asc.has('k')) || (had_K && cc.asc.has('K'));
    const bool sx = cc.mb.has(0x17f) || (had_s && cc.asc.has('s')) || (had_S && cc.asc.has('S'));
    if (cc.mb.has(0x212a)) { cc.bs.set('k'); cc.bs.set('K'); }
    if (cc.mb.has(0x17f)) { cc.bs.set('s'); cc.bs.set('S'); }
    if (kx) { cc.mb.add(0x212a, 0x212a); cc.mbx.add(0x212a, 0x212a); }
    if (sx) { cc.mb.add(0x17f, 0x17f); cc.mbx.add(0x17f, 0x17f); }
    cc.mb.norm(); cc.mbx.norm();
    return true;
}

// [x&&y]: the members both operands have (regparse.c and_cclass over two classes without their own NOT: the bit sets and'ed, the code
// ranges and'ed)
CodeSet mb_intersect(const CodeSet &x0, const CodeSet &y0) {
    CodeSet x = x0, y = y0, o;
    x.norm(); y.norm();
    size_t i = 0, j = 0;
    while (i < x.r.size() && j < y.r.size()) {
        const uint32_t lo = std::max(x.r[i].first, y.r[j].first), hi = std::min(x.r[i].second, y.r[j].second);
        if (lo <= hi) o.add(lo, hi);
        if (x.r[i].second < y.r[j].second) i++; else j++;
    }
    return o;
}
CC cc_and(const CC &a, const CC &b) {
    CC o;
    for (int i = 0; i < 4; i++) { o.bs.w[i] = a.bs.w[i] & b.bs.w[i]; o.asc.w[i] = a.asc.w[i] & b.asc.w[i]; }
    o.mb = mb_intersect(a.mb, b.mb);
    o.mbx = mb_intersect(a.mbx, b.mbx);
    return o;
}

// ---------------------------------------------------------------- AST
// (A_WORDB_A / A_NWORDB_A: \b \B under (?a) -- OP_ASCII_WORD_BOUND: only [0-9A-Za-z_] are word characters)
// (A_SEMI_EOS \Z, A_BEGIN_POS \G and the node kinds LOOK / ATOMIC / BACKREF / KEEP exist only in trees parsed with Syntax::ext -- the
// product's backtracking matcher, rxbt.inc; the table and NFA builders never see them)
enum AnchorKind { A_BOL, A_EOL, A_BOS, A_EOS, A_WORDB, A_NWORDB, A_WORDB_A, A_NWORDB_A, A_SEMI_EOS, A_BEGIN_POS };

struct Ast {
    enum T { EMPTY, SET, CAT, ALT, GROUP, REPEAT, ANCHOR, LOOK, ATOMIC, BACKREF, KEEP, COND, ABSENT, CALL } t = EMPTY;   // COND: (?(refs)kids[0]|kids[1])
    const Ast *target = nullptr;           // CALL (\g<..>): the group it runs (refs[0] = its number, 0 = the whole pattern); set by bt_compile
    bool ahead = true, neg_look = false;   // LOOK
    std::vector<int> refs;                 // BACKREF: the groups the reference names, in group order
    bool ref_icase = false;
    CodeSet btmb;                          // SET, rxbt: the well-formed multi-byte characters it accepts
    uint64_t btasc[2] = {0, 0};            // SET, rxbt: the ASCII characters it accepts, a bit each
    uint64_t btnext[4] = {0, 0, 0, 0};     // REPEAT inside a CAT, rxbt: the bytes what FOLLOWS it in the CAT can begin with (btnext_on:
    bool btnext_on = false;                //   that rest takes at least one character, so an end of the repeat in front of another byte fails)
    CC cc;
    std::vector<std::unique_ptr<Ast>> kids;
    int cap = 0;
    int min = 0, max = 0;       // max < 0: unbounded
    bool greedy = true;
    AnchorKind anchor = A_BOL;
    uint32_t ilit = 0;          // SET made from a literal LETTER under (?i): the letter, lower case (multi-character folds, below)
    bool icase = false;         // SET parsed under (?i) (Program::corner_flags)
};
using AstP = std::unique_ptr<Ast>;

AstP mk(Ast::T t) { AstP a(new Ast); a->t = t; return a; }

// (?i) over non-ASCII characters, host matcher only (round 5): the partners of a character and the multi-character folds as the
// reference's engine applies them, probed from it (tools/gen_casefold.py)
#include "casefold.inc"
static const size_t CF_NSIMPLE = sizeof(CF_SIMPLE) / sizeof(CF_SIMPLE[0]), CF_NMULTI = sizeof(CF_MULTI) / sizeof(CF_MULTI[0]);
// first row of character c in CF_SIMPLE (rows are sorted by character), CF_NSIMPLE when it has none
static size_t cf_first(uint32_t c) {
    size_t lo = 0, hi = CF_NSIMPLE;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (CF_SIMPLE[mid][0] < c) lo = mid + 1; else hi = mid; }
    return lo < CF_NSIMPLE && CF_SIMPLE[lo][0] == c ? lo : CF_NSIMPLE;
}
// the folded form of c: what CF_MULTI's sequences are spelled in (ASCII letters: lower case)
static uint32_t cf_key(uint32_t c) {
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? c + 32 : c;
    const size_t i = cf_first(c);
    return i < CF_NSIMPLE ? CF_SIMPLE[i][2] : c;
}
static const uint32_t *cf_multi(uint32_t c) {
    for (size_t i = 0; i < CF_NMULTI; i++) if (CF_MULTI[i][0] == c) return CF_MULTI[i];
    return nullptr;
}

struct Syntax {
    const unsigned char *s, *e, *p;
    bool has_named = false;
    int ncap = 0;
    int nopen = 0;                    // groups opened so far, the plain ones next to named ones included (regparse.c env->num_mem while parsing)
    std::string err;
    std::vector<std::string> names;
    std::vector<std::vector<int>> name_groups;
    bool ext = false;             // accept what only a backtracking matcher can run: look-around, atomic groups, possessive repeats, back-references, \Z \G \K
    bool nonregular = false;      // ... such a construct was met (with ext off: the reason of the failure)
    std::vector<std::pair<Ast *, std::string>> named_refs;    // \k<name> met before the end of the pattern: resolved there
    std::vector<Ast *> calls;                                  // \g<..> nodes: their groups are looked up when the pattern is complete
    int max_ref = 0;

    bool fail(const char *m) { if (err.empty()) { err = m; err += " (offset " + std::to_string(p - s) + ")"; } return false; }
    bool fail(const std::string &m) { return fail(m.c_str()); }
    bool failed() const { return !err.empty(); }
    bool eof() const { return p >= e; }

    uint32_t take_cp() {
        uint32_t c = *p++;
        int n = 0;
        if (c < 0x80) return c;
        if (c >= 0xc2 && c <= 0xdf) { n = 1; c &= 0x1f; }
        else if (c >= 0xe0 && c <= 0xef) { n = 2; c &= 0x0f; }
        else if (c >= 0xf0 && c <= 0xf4) { n = 3; c &= 0x07; }
        else { fail("invalid UTF-8 in pattern"); return 0xfffd; }
        while (n-- > 0) {
            if (eof() || (*p & 0xc0) != 0x80) { fail("invalid UTF-8 in pattern"); return 0xfffd; }
            c = (c << 6) | (*p++ & 0x3f);
        }
        return c;
    }

    static int hexv(int c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    // \cX  \C-X  \M-X and their nestings (regparse.c:2429 fetch_escaped_value): a code point -- \M-a is U+00E1, two bytes of subject --,
    // never a raw byte.  p stands behind the backslash.
    bool escaped_value(uint32_t &out) {
        if (eof()) return fail("end pattern at escape");
        uint32_t c = take_cp();
        if (c == 'M') {
            if (eof()) return fail("end pattern at meta");
            if (take_cp() != '-') return fail("invalid meta-code syntax");
            if (eof()) return fail("end pattern at meta");
            c = take_cp();
            if (c == '\\' && !escaped_value(c)) return false;
            c = (c & 0xff) | 0x80;
        }
        else if (c == 'C' || c == 'c') {
            if (c == 'C') {
                if (eof()) return fail("end pattern at control");
                if (take_cp() != '-') return fail("invalid control-code syntax");
            }
            if (eof()) return fail("end pattern at control");
            c = take_cp();
            if (c == '?') c = 0177;
            else {
                if (c == '\\' && !escaped_value(c)) return false;
                c &= 0x9f;
            }
        }
        else {
            switch (c) {                    // (regparse.c conv_backslash_value)
            case 'n': c = 10; break; case 't': c = 9; break; case 'r': c = 13; break; case 'f': c = 12; break;
            case 'a': c = 7; break; case 'b': c = 8; break; case 'e': c = 27; break; case 'v': c = 11; break;
            }
        }
        out = c;
        return !failed();
    }

    // \1 .. \777 inside brackets: always octal there (regparse.c fetch_token_in_cc)
    bool class_octal(uint32_t &out) {
        uint32_t v = 0; int n = 0;
        while (!eof() && *p >= '0' && *p <= '7' && n < 3) { v = v * 8 + (uint32_t) (*p++ - '0'); n++; }
        if (v >= 0x80) return fail("raw byte escapes >= 0x80 are not supported");
        out = v;
        return true;
    }

    // after the backslash: escapes denoting one code point
    bool escape_cp(uint32_t &out, bool in_class) {
        int c = *p;
        switch (c) {
        case 't': p++; out = 9; return true;
        case 'n': p++; out = 10; return true;
        case 'r': p++; out = 13; return true;
        case 'f': p++; out = 12; return true;
        case 'v': p++; out = 11; return true;
        case 'a': p++; out = 7; return true;
        case 'e': p++; out = 27; return true;
        case 'x': {
            uint32_t v = 0; int n = 0;
            p++;
            if (!eof() && *p == '{') {
                p++;
                while (!eof() && hexv(*p) >= 0 && n < 8) { v = v * 16 + hexv(*p++); n++; }
                if (eof() || *p != '}' || n == 0) { fail("bad \\x{...}"); return true; }
                p++;
            } else {
                while (!eof() && hexv(*p) >= 0 && n < 2) { v = v * 16 + hexv(*p++); n++; }
                if (n == 0) { fail("bad \\x"); return true; }
                if (v >= 0x80) { fail("raw byte escapes >= 0x80 are not supported"); return true; }
            }
            out = v; return true;
        }
        case 'u': {
            uint32_t v = 0; int n = 0;
            p++;
            while (!eof() && hexv(*p) >= 0 && n < 4) { v = v * 16 + hexv(*p++); n++; }
            if (n != 4) { fail("bad \\u"); return true; }
            out = v; return true;
        }
        case '0': {
            uint32_t v = 0; int n = 0;
            while (!eof() && *p >= '0' && *p <= '7' && n < 3) { v = v * 8 + (*p++ - '0'); n++; }
            if (v >= 0x80) { fail("raw byte escapes >= 0x80 are not supported"); return true; }
            out = v; return true;
        }
        case 'c': case 'C': case 'M':
            escaped_value(out); return true;
        }
        if (in_class && c == 'b') { p++; out = 8; return true; }
        return false;
    }

    // \p{Name} \p{^Name} \P{Name} (regparse.c fetch_token 'p' -> TK_CHAR_PROPERTY; enc/unicode.c onigenc_unicode_property_name_to_ctype:
    // the name without blanks, '-' and '_', any case).  The fourteen names that are the POSIX brackets' ctypes are taken -- the same sets
    // as [[:name:]], never ASCII-range (parse_char_property passes ascii_range 0) --; scripts, general categories, ages are refused.
    // p stands on the 'p' / 'P'; on success behind the '}'.
    // as_flag (an atom outside brackets): the positive set goes in and the NOT is a flag of the class (regparse.c parse_exp
    // TK_CHAR_PROPERTY: NCCLASS_SET_NOT) -- it decides what an ill-formed byte matches; inside brackets the complement is added.
    bool property(CC &cc, bool *as_flag = nullptr) {
        const bool upper = *p == 'P';
        p++;
        if (eof() || *p != '{') return fail("invalid character property name {...}");
        p++;
        bool neg = upper;
        if (!eof() && *p == '^') { neg = !neg; p++; }
        std::string nm;
        while (!eof() && *p != '}') {
            const int ch = *p++;
            if (ch == ' ' || ch == '-' || ch == '_') continue;
            nm += (char) ((ch >= 'A' && ch <= 'Z') ? ch + 32 : ch);
        }
        if (eof()) return fail("invalid character property name {...}");
        p++;
        if (as_flag) { *as_flag = neg; neg = false; }
        // (\p{Punct} is the Unicode category P -- without $ + < = > ^ ` | ~, which [[:punct:]] has: the property table's set, not the bracket's)
        if (nm != "punct" && add_posix(cc, nm, neg, false)) return true;
        if (add_uniprop(cc, nm, neg)) return true;
        return fail(("invalid character property name {" + nm + "}").c_str());
    }

    void skip_extended(unsigned opts) {
        if (!(opts & OPT_EXTEND)) return;
        while (!eof()) {
            int c = *p;
            if (c == ' ' || (c >= 9 && c <= 13)) p++;
            else if (c == '#') { while (!eof() && *p != '\n') p++; }
            else break;
        }
    }

    void note_name(const std::string &n, int group) {
        for (size_t i = 0; i < names.size(); i++) if (names[i] == n) { name_groups[i].push_back(group); return; }
        names.push_back(n);
        name_groups.push_back({group});
    }

    // '[' already consumed; fills `out` (empty on entry) with the class up to the matching ']'
    bool char_class(CC &out, unsigned opts) {
        bool neg = false, first = true, have_acc = false;
        CC s, acc;
        if (!eof() && *p == '^') { neg = true; p++; }
        for (;;) {
            uint32_t lo = 0, hi;
            bool have = false;
            if (eof()) return fail("premature end of char-class");
            if (*p == ']') {
                if (!first) { p++; break; }
                p++; lo = ']'; have = true;
            }
            first = false;
            if (!have) {
                if (*p == '[') {
                    if (p + 1 < e && p[1] == ':') {
                        const unsigned char *q = p + 2;
                        bool pneg = false;
                        if (q < e && *q == '^') { pneg = true; q++; }
                        const unsigned char *nm = q;
                        while (q < e && *q >= 'a' && *q <= 'z') q++;
                        if (q + 1 < e && q[0] == ':' && q[1] == ']') {
                            if (!add_posix(s, std::string((const char *) nm, q - nm), pneg, posix_is_ascii(opts))) return fail("unknown POSIX bracket");
                            p = q + 2;
                            continue;
                        }
                    }
                    p++;
                    CC t;
                    if (!char_class(t, opts)) return false;
                    s.merge_class(t);
                    continue;
                }
                if (*p == '&' && p + 1 < e && p[1] == '&') {
                    // x&&y&&z: what stands left of the operator so far is one operand (regparse.c parse_char_class CC_AND)
                    p += 2;
                    s.mb.norm(); s.mbx.norm();
                    acc = have_acc ? cc_and(acc, s) : s;
                    have_acc = true;
                    s = CC();
                    continue;
                }
                if (*p == '\\') {
                    p++;
                    if (eof()) return fail("end pattern at escape");
                    int c = *p;
                    if (c == 'd' || c == 'w' || c == 's' || c == 'h') { p++; add_ctype(s, (char) c, false, !ctype_is_ascii(opts)); continue; }
                    if (c == 'D' || c == 'W' || c == 'S' || c == 'H') { p++; add_ctype(s, (char) (c + 32), true, !ctype_is_ascii(opts)); continue; }
                    if (c == 'p' || c == 'P') { if (!property(s)) return false; continue; }
                    if (c == 'R' || c == 'X') return fail("\\R / \\X are not supported");
                    if (escape_cp(lo, true)) { if (failed()) return false; }
                    else if (c >= '1' && c <= '7') { if (!class_octal(lo)) return false; }
                    else lo = take_cp();
                }
                else lo = take_cp();
                if (failed()) return false;
            }
            hi = lo;
            if (p + 1 < e && p[0] == '-' && p[1] != ']') {
                const unsigned char *save = p;
                p++;
                if (*p == '[') p = save;
                else if (*p == '\\') {
                    p++;
                    if (eof()) return fail("end pattern at escape");
                    if (strchr("dwshDWSHpP", *p)) p = save;
                    else if (escape_cp(hi, true)) { if (failed()) return false; }
                    else if (*p >= '1' && *p <= '7') { if (!class_octal(hi)) return false; }
                    else hi = take_cp();
                }
                else hi = take_cp();
                if (failed()) return false;
                if (hi < lo) return fail("empty range in char class");
            }
            s.add_cp(lo, hi);
        }
        s.mb.norm();
        if (have_acc) { s.mbx.norm(); s = cc_and(acc, s); s.mb.norm(); s.mbx.norm(); }
        if ((opts & OPT_IGNORECASE) && !fold_case_any(s)) return fail("case-insensitive classes with non-ASCII members are not supported on the GPU path");
        if (neg) cc_multi.clear();                  // (a negated class takes no sequences)
        s.neg = neg;
        out = s;
        return true;
    }

    // (?i) over a class with some non-ASCII members, host matcher only (round 5): the class's (positive) members are closed under the
    // engine's fold pairs (regparse.c i_apply_case_fold with CASE_FOLD_IS_APPLIED_INSIDE_NEGATIVE_CCLASS; a pair that crosses the ASCII
    // boundary from an ASCII letter only for members of the shadow class); the members that stand for a sequence are noted in cc_multi --
    // the class then also matches those sequences, unless it is negated
    std::vector<const uint32_t *> cc_multi;
    bool fold_case_any(CC &cc) {
        if (fold_case(cc)) return true;
        if (!ext) { nonregular = true; return false; }
        const CodeSet before = cc.mb;
        const ByteSet bs0 = cc.bs;
        auto has = [&](uint32_t c) { return c < 0x80 ? bs0.has((int) c) : before.has(c); };
        for (size_t i = 0; i < CF_NSIMPLE; i++) {
            const uint32_t c = CF_SIMPLE[i][0], m = CF_SIMPLE[i][1];
            // (a partner below 0x100 goes into the BIT SET, which a well-formed two-byte character is never tested on -- only a stray byte
            // of that value is: (?i)[\xe0-\xff] does not take U+00C0 in the reference, i_apply_case_fold's SINGLE_BYTE_SIZE branch)
            if (has(c) && !has(m)) { if (m < 0x100) cc.bs.set((int) m); else { cc.mb.add(m, m); cc.mbx.add(m, m); } }
            if (m < 0x80 && has(m) && cc.asc.has((int) m) && !has(c)) { cc.mb.add(c, c); cc.mbx.add(c, c); }
        }
        for (size_t i = 0; i < CF_NMULTI; i++) if (before.has(CF_MULTI[i][0])) cc_multi.push_back(CF_MULTI[i]);
        cc.mb.norm(); cc.mbx.norm();
        return true;
    }

    // (?:[class]|seq1|seq2 ..): the members of a folded class that stand for a sequence (cc_multi) also match it, folded (regparse.c
    // i_apply_case_fold, to_len > 1)
    AstP with_sequences(AstP a, unsigned opts) {
        if (cc_multi.empty()) return a;
        std::vector<const uint32_t *> ms;
        ms.swap(cc_multi);
        if (ms.size() > 120) { fail("pattern too large (case folds)"); return a; }
        AstP alt = mk(Ast::ALT);
        alt->kids.push_back(std::move(a));
        for (const uint32_t *mf : ms) {
            AstP seq = mk(Ast::CAT);
            for (uint32_t k = 0; k < mf[1]; k++) seq->kids.push_back(literal(mf[2 + k], opts));
            fold_strings(seq.get());                   // (the sequence is compared folded: it also takes the characters that stand for it)
            alt->kids.push_back(std::move(seq));
        }
        AstP g = mk(Ast::GROUP);
        g->kids.push_back(std::move(alt));
        return g;
    }

    AstP literal(uint32_t c, unsigned opts) {
        AstP a = mk(Ast::SET);
        if ((opts & OPT_IGNORECASE) && c >= 0x80) {
            // the host's matcher takes it (round 5): the character and its partners as a class; a character that stands for a sequence
            // (U+00DF: "ss") also matches the sequence, each of its characters folded in turn
            nonregular = true;
            if (!ext) { fail("case-insensitive non-ASCII literals are not supported on the GPU path"); return a; }
            a->cc.kind = CC::CLASS;
            a->cc.add_cp(c, c);
            for (size_t i = cf_first(c); i < CF_NSIMPLE && CF_SIMPLE[i][0] == c; i++) {
                const uint32_t m = CF_SIMPLE[i][1];
                a->cc.add_cp(m, m);
                if (m < 0x80) a->cc.asc.set((int) m);
            }
            a->cc.mb.norm(); a->cc.mbx.norm();
            a->ilit = cf_key(c);
            a->icase = true;
            if (const uint32_t *mf = cf_multi(c)) {
                if (++fold_budget > 4000) { fail("pattern too large (case folds)"); return a; }
                AstP seq = mk(Ast::CAT);
                for (uint32_t k = 0; k < mf[1]; k++) seq->kids.push_back(literal(mf[2 + k], opts));
                fold_strings(seq.get());
                AstP alt = mk(Ast::ALT);
                a->ilit = 0;                            // (not part of a run of letters: it is an alternation now)
                alt->kids.push_back(std::move(a));
                alt->kids.push_back(std::move(seq));
                return alt;
            }
            return a;
        }
        if ((opts & OPT_IGNORECASE) && ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) {
            // a folded letter is a small class (both cases; k and s also reach U+212A / U+017F)
            a->cc.kind = CC::CLASS;
            a->cc.bs.set((int) c);
            a->cc.asc.set((int) c);
            fold_case(a->cc);
            a->ilit = c | 32;
            a->icase = true;
            return a;
        }
        a->cc.kind = CC::LIT;
        a->cc.lit = c;
        return a;
    }

    int number() {
        long v = 0; int n = 0;
        while (!eof() && *p >= '0' && *p <= '9') { v = v * 10 + (*p++ - '0'); n++; if (v > 100000) v = 100001; }
        return n ? (int) v : -1;
    }

    AstP alternation(unsigned opts, int depth);

    AstP atom(unsigned &opts, int depth, bool &is_group_tail) {
        is_group_tail = false;
        skip_extended(opts);
        if (eof()) return nullptr;
        int c = *p;
        if (c == '|' || c == ')') return nullptr;
        if (c == '(') {
            p++;
            if (depth > 100) { fail("nesting too deep"); return nullptr; }
            AstP g = mk(Ast::GROUP);
            if (!eof() && *p == '?') {
                p++;
                if (eof()) { fail("end pattern in group"); return nullptr; }
                c = *p;
                if (c == '#') {
                    while (!eof() && *p != ')') p++;
                    if (eof()) { fail("end pattern in group"); return nullptr; }
                    p++;
                    return mk(Ast::EMPTY);
                }
                if (c == ':') { p++; g->kids.push_back(alternation(opts, depth + 1)); }
                else if (c == '=' || c == '!') {
                    nonregular = true;
                    if (!ext) { fail("look-ahead is not supported on the GPU path"); return nullptr; }
                    p++;
                    g->t = Ast::LOOK; g->ahead = true; g->neg_look = c == '!';
                    g->kids.push_back(alternation(opts, depth + 1));
                }
                else if (c == '>') {
                    nonregular = true;
                    if (!ext) { fail("atomic groups are not supported on the GPU path"); return nullptr; }
                    p++;
                    g->t = Ast::ATOMIC;
                    g->kids.push_back(alternation(opts, depth + 1));
                }
                else if (c == '<' && p + 1 < e && (p[1] == '=' || p[1] == '!')) {
                    nonregular = true;
                    if (!ext) { fail("look-behind is not supported on the GPU path"); return nullptr; }
                    g->t = Ast::LOOK; g->ahead = false; g->neg_look = p[1] == '!';
                    p += 2;
                    g->kids.push_back(alternation(opts, depth + 1));
                }
                else if (c == '<' || c == '\'') {
                    int term = c == '<' ? '>' : '\'';
                    p++;
                    const unsigned char *nm = p;
                    while (!eof() && *p != term) {
                        int ch = *p;
                        if (!((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_' || ch >= 0x80)) { fail("invalid group name"); return nullptr; }
                        p++;
                    }
                    if (eof() || p == nm || (*nm >= '0' && *nm <= '9')) { fail("invalid group name"); return nullptr; }
                    g->cap = ++ncap; nopen++;
                    note_name(std::string((const char *) nm, p - nm), g->cap);
                    p++;
                    g->kids.push_back(alternation(opts, depth + 1));
                }
                else if (c == '(' && ext) {
                    // (?(cond)yes|no): cond = a group number or <name> / 'name' (regparse.c parse_enclose '(': OP_CONDITION)
                    nonregular = true;
                    p++;
                    g->t = Ast::COND;
                    g->ref_icase = false;
                    if (eof()) { fail("end pattern in group"); return nullptr; }
                    if (*p >= '0' && *p <= '9') {
                        int v = 0;
                        while (!eof() && *p >= '0' && *p <= '9' && v < 1000) { v = v * 10 + (*p - '0'); p++; }
                        if (has_named) { fail("numbered backref/call is not allowed. (use name)"); return nullptr; }
                        if (v <= 0) { fail("invalid backref number/name"); return nullptr; }
                        g->refs.push_back(v);
                        max_ref = std::max(max_ref, v);
                    }
                    else if (*p == '<' || *p == '\'') {
                        const int term = *p == '<' ? '>' : '\'';
                        p++;
                        const unsigned char *nm = p;
                        while (!eof() && *p != term) p++;
                        if (eof() || p == nm) { fail("invalid backref name"); return nullptr; }
                        named_refs.emplace_back(g.get(), std::string((const char *) nm, p - nm));
                        p++;
                    }
                    else { fail("invalid conditional pattern"); return nullptr; }
                    if (eof() || *p != ')') { fail("invalid conditional pattern"); return nullptr; }
                    p++;
                    AstP body = alternation(opts, depth + 1);
                    if (failed()) return nullptr;
                    if (body && body->t == Ast::ALT) {
                        if (body->kids.size() > 2) { fail("invalid conditional pattern"); return nullptr; }
                        g->kids.push_back(std::move(body->kids[0]));
                        g->kids.push_back(std::move(body->kids[1]));
                    }
                    else { g->kids.push_back(std::move(body)); g->kids.push_back(mk(Ast::EMPTY)); }
                }
                else if (c == '~') {
                    // (?~X) the absent operator (regparse.c parse_enclose '~': OP_PUSH_ABSENT_POS / OP_ABSENT / OP_ABSENT_END)
                    nonregular = true;
                    if (!ext) { fail("the absent operator is not supported on the GPU path"); return nullptr; }
                    p++;
                    g->t = Ast::ABSENT;
                    g->kids.push_back(alternation(opts, depth + 1));
                }
                else if (c == '(' || c == '&' || c == 'P') { if (c == '(') nonregular = true; fail("unsupported group construct"); return nullptr; }
                else {
                    unsigned o = opts;
                    bool on = true;
                    for (;;) {
                        if (eof()) { fail("end pattern in group"); return nullptr; }
                        c = *p;
                        if (c == 'i') { if (on) o |= OPT_IGNORECASE; else o &= ~OPT_IGNORECASE; }
                        else if (c == 'm') { if (on) o |= OPT_MULTILINE; else o &= ~OPT_MULTILINE; }
                        else if (c == 'x') { if (on) o |= OPT_EXTEND; else o &= ~OPT_EXTEND; }
                        else if (c == '-') on = false;
                        else if (c == ')' || c == ':') break;
                        // (?a) (?u) (?d): regparse.c:5257-5297 (Ruby syntax; not negatable)
                        else if (c == 'a' && on) { o &= ~OPT_AR_OFF; o |= OPT_PBA_OFF | OPT_WBA_OFF; }
                        else if (c == 'u' && on) { o |= OPT_AR_OFF | OPT_PBA_OFF | OPT_WBA_OFF; }
                        else if (c == 'd' && on) { o &= ~(OPT_AR_OFF | OPT_PBA_OFF | OPT_WBA_OFF); }
                        else { fail("undefined group option"); return nullptr; }
                        p++;
                    }
                    p++;
                    if (c == ')') {
                        // isolated option: the REST of the enclosing group, alternation included,
                        // becomes the body (lib/onigmo/regparse.c parse_exp "option only")
                        g->kids.push_back(alternation(o, depth + 1));
                        is_group_tail = true;
                        return g;
                    }
                    g->kids.push_back(alternation(o, depth + 1));
                }
            }
            else {
                if (!has_named) g->cap = ++ncap;
                nopen++;
                g->kids.push_back(alternation(opts, depth + 1));
            }
            if (failed()) return nullptr;
            if (eof() || *p != ')') { fail("end pattern with unmatched parenthesis"); return nullptr; }
            p++;
            return g;
        }
        if (c == '[') {
            p++;
            AstP a = mk(Ast::SET);
            cc_multi.clear();
            if (!char_class(a->cc, opts)) return nullptr;
            a->icase = (opts & OPT_IGNORECASE) != 0;
            return with_sequences(std::move(a), opts);
        }
        if (c == '.') {
            p++;
            AstP a = mk(Ast::SET);
            a->cc.kind = CC::ANY;
            a->cc.any_nl = (opts & OPT_MULTILINE) != 0;
            return a;
        }
        if (c == '^') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_BOL; return a; }
        if (c == '$') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_EOL; return a; }
        if (c == '*' || c == '+' || c == '?') { fail("target of repeat operator is not specified"); return nullptr; }
        if (c == '\\') {
            p++;
            if (eof()) { fail("end pattern at escape"); return nullptr; }
            c = *p;
            if (strchr("dwshDWSH", c)) {
                p++;
                // \w \W compile to the word opcodes, \d \s \h (and negations) to a bit-set class whose NOT is
                // a flag (regparse.c parse_exp TK_CHAR_TYPE)
                AstP a = mk(Ast::SET);
                if ((c | 32) == 'w') { a->cc.kind = CC::WORD; a->cc.neg = !(c & 32); a->cc.uni = !ctype_is_ascii(opts); }
                else if (ctype_is_ascii(opts)) { ctype_ascii(a->cc.bs, (char) (c | 32)); a->cc.neg = !(c & 32); }
                else { add_ctype(a->cc, (char) (c | 32), false, true); a->cc.neg = !(c & 32); }
                return a;
            }
            if (c == 'A') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_BOS; return a; }
            if (c == 'z') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_EOS; return a; }
            if (c == 'b') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = wordb_is_ascii(opts) ? A_WORDB_A : A_WORDB; return a; }
            if (c == 'B') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = wordb_is_ascii(opts) ? A_NWORDB_A : A_NWORDB; return a; }
            if (c == 'p' || c == 'P') {
                AstP a = mk(Ast::SET);
                bool not_flag = false;
                if (!property(a->cc, &not_flag)) return nullptr;
                a->cc.neg = not_flag;
                a->cc.mb.norm(); a->cc.mbx.norm();
                cc_multi.clear();
                if ((opts & OPT_IGNORECASE) && !fold_case_any(a->cc)) { fail("case-insensitive classes with non-ASCII members are not supported on the GPU path"); return nullptr; }
                a->icase = (opts & OPT_IGNORECASE) != 0;
                if (a->cc.neg) cc_multi.clear();
                return with_sequences(std::move(a), opts);
            }
            if (c == 'Z') {
                nonregular = true;
                if (!ext) { fail("\\Z is not supported on the GPU path"); return nullptr; }
                p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_SEMI_EOS; return a;
            }
            if (ext && c == 'G') { p++; AstP a = mk(Ast::ANCHOR); a->anchor = A_BEGIN_POS; return a; }
            if (ext && c == 'K') { p++; return mk(Ast::KEEP); }
            if (ext && c == 'k' && p + 1 < e && (p[1] == '<' || p[1] == '\'')) {
                // \k<name> \k<n> \k<-n>  (regparse.c fetch_token 'k')
                const int term = p[1] == '<' ? '>' : '\'';
                p += 2;
                const unsigned char *nm = p;
                while (!eof() && *p != term) p++;
                if (eof() || p == nm) { fail("invalid backref name"); return nullptr; }
                std::string name((const char *) nm, p - nm);
                p++;
                AstP a = mk(Ast::BACKREF);
                a->ref_icase = (opts & OPT_IGNORECASE) != 0;
                const bool numeric = (name[0] >= '0' && name[0] <= '9') || (name[0] == '-' && name.size() > 1);
                if (numeric) {
                    if (has_named) { fail("numbered backref/call is not allowed. (use name)"); return nullptr; }
                    int v = 0;
                    for (size_t i = name[0] == '-' ? 1 : 0; i < name.size(); i++) { if (name[i] < '0' || name[i] > '9' || v > 1000) { fail("invalid backref number/name"); return nullptr; } v = v * 10 + (name[i] - '0'); }
                    if (name[0] == '-') v = ncap + 1 - v;
                    if (v <= 0) { fail("invalid backref number/name"); return nullptr; }
                    a->refs.push_back(v);
                    max_ref = std::max(max_ref, v);
                }
                else named_refs.emplace_back(a.get(), name);
                return a;
            }
            if (ext && c == 'g' && p + 1 < e && (p[1] == '<' || p[1] == '\'')) {
                // \g<name> \g<n> \g<-n> \g<+n> \g<0>: a subexpression call (regparse.c fetch_token 'g', TK_CALL)
                const int term = p[1] == '<' ? '>' : '\'';
                p += 2;
                const unsigned char *nm = p;
                while (!eof() && *p != term) p++;
                if (eof() || p == nm) { fail("invalid group name <>"); return nullptr; }
                std::string name((const char *) nm, p - nm);
                p++;
                AstP a = mk(Ast::CALL);
                const bool numeric = (name[0] >= '0' && name[0] <= '9') || ((name[0] == '-' || name[0] == '+') && name.size() > 1);
                if (numeric) {
                    int v = 0;
                    for (size_t i = (name[0] == '-' || name[0] == '+') ? 1 : 0; i < name.size(); i++) { if (name[i] < '0' || name[i] > '9' || v > 1000) { fail("invalid group name <" + name + ">"); return nullptr; } v = v * 10 + (name[i] - '0'); }
                    if (has_named && !(v == 0 && name[0] != '-' && name[0] != '+')) { fail("numbered backref/call is not allowed. (use name)"); return nullptr; }
                    if (name[0] == '-') { v = ncap + 1 - v; if (v <= 0) { fail("invalid backref number/name"); return nullptr; } }
                    else if (name[0] == '+') { if (v <= 0) { fail("invalid backref number/name"); return nullptr; } v = ncap + v; }
                    a->refs.push_back(v);
                    calls.push_back(a.get());
                }
                else { named_refs.emplace_back(a.get(), name); calls.push_back(a.get()); }
                nonregular = true;
                return a;
            }
            if (c == 'X' && ext) {
                // \X, an extended grapheme cluster: the reference builds it from property classes (regparse.c node_extended_grapheme_cluster,
                // UAX #29 as of Unicode 11) -- CR LF | a control | Prepend* core postcore* | any character, atomic, case folding off --;
                // spelled here as a pattern over the same properties (their members: unicode_props.inc, probed from the engine) and parsed
                // by this front end
                static const char XGC[] =
                    "(?>\\x0D\\x0A|[\\p{Grapheme_Cluster_Break=Control}\\x0A\\x0D]|"
                    "\\p{Grapheme_Cluster_Break=Prepend}*"
                    "(?:\\p{Grapheme_Cluster_Break=L}*(?:\\p{Grapheme_Cluster_Break=V}+|\\p{Grapheme_Cluster_Break=LV}\\p{Grapheme_Cluster_Break=V}*|\\p{Grapheme_Cluster_Break=LVT})\\p{Grapheme_Cluster_Break=T}*"
                    "|\\p{Grapheme_Cluster_Break=L}+|\\p{Grapheme_Cluster_Break=T}+|\\p{Regional_Indicator}{2}"
                    "|\\p{Extended_Pictographic}(?:\\p{Grapheme_Cluster_Break=Extend}*\\x{200D}\\p{Extended_Pictographic})*"
                    "|[\\P{Grapheme_Cluster_Break=Control}&&[^\\x0A\\x0D]])"      // (a POSITIVE class of the complement, as the reference adds it: a cut sequence is no member)
                    "[\\p{Grapheme_Cluster_Break=Extend}\\p{Grapheme_Cluster_Break=SpacingMark}\\x{200D}]*"
                    "|(?m:.))";
                p++;
                nonregular = true;
                Syntax sub;
                sub.ext = true;
                sub.s = sub.p = (const unsigned char *) XGC;
                sub.e = sub.s + sizeof(XGC) - 1;
                AstP n = sub.alternation(opts & ~(unsigned) OPT_IGNORECASE, depth + 1);
                if (sub.failed() || !sub.eof()) { fail("\\X: " + sub.err); return nullptr; }
                return n;
            }
            if (c == 'R' && ext) {
                // \R: (?>\x0D\x0A|[\x0A-\x0D\x{85}\x{2028}\x{2029}])  (regparse.c node_linebreak)
                p++;
                AstP crlf = mk(Ast::CAT);
                crlf->kids.push_back(literal(0x0D, 0));
                crlf->kids.push_back(literal(0x0A, 0));
                AstP one = mk(Ast::SET);
                one->cc.add_cp(0x0A, 0x0D); one->cc.add_cp(0x85, 0x85); one->cc.add_cp(0x2028, 0x2029);
                one->cc.mb.norm(); one->cc.mbx.norm();
                AstP alt = mk(Ast::ALT);
                alt->kids.push_back(std::move(crlf));
                alt->kids.push_back(std::move(one));
                AstP at = mk(Ast::ATOMIC);
                at->kids.push_back(std::move(alt));
                nonregular = true;
                return at;
            }
            if (c == 'R') nonregular = true;
            if (strchr("GKRXkg", c)) { if (c == 'G' || c == 'K' || c == 'k' || c == 'g' || c == 'X') nonregular = true; fail("unsupported escape"); return nullptr; }
            if (c >= '1' && c <= '9') {
                // a decimal number is a back-reference while it is at most 9 or at most the groups opened so far; otherwise \8 \9 are
                // the digits themselves and the rest an octal escape of up to three digits (regparse.c fetch_token '1'..'9')
                const unsigned char *prev = p;
                int v = 0;
                while (!eof() && *p >= '0' && *p <= '9') { if (v <= 1000) v = v * 10 + (*p - '0'); p++; }
                if (!(v <= 1000 && (v <= nopen || v <= 9))) {
                    p = prev;
                    if (c == '8' || c == '9') { p++; return literal((uint32_t) c, opts); }
                    uint32_t o = 0; int n = 0;
                    while (!eof() && *p >= '0' && *p <= '7' && n < 3) { o = o * 8 + (uint32_t) (*p++ - '0'); n++; }
                    if (o > 0xff) { fail("too big number"); return nullptr; }
                    if (o >= 0x80) { fail("raw byte escapes >= 0x80 are not supported"); return nullptr; }
                    return literal(o, opts);
                }
                nonregular = true;
                if (!ext) { fail("back-references are not supported on the GPU path"); return nullptr; }
                if (has_named) { fail("numbered backref/call is not allowed. (use name)"); return nullptr; }
                AstP a = mk(Ast::BACKREF);
                a->ref_icase = (opts & OPT_IGNORECASE) != 0;
                a->refs.push_back(v);
                max_ref = std::max(max_ref, v);
                return a;
            }
            uint32_t v = 0;
            if (escape_cp(v, false)) { if (failed()) return nullptr; return literal(v, opts); }
            v = take_cp();
            return literal(v, opts);
        }
        return literal(take_cp(), opts);
    }

    AstP piece(unsigned &opts, int depth, bool &tail) {
        AstP a = atom(opts, depth, tail);
        if (!a || failed() || tail) return a;
        for (;;) {
            skip_extended(opts);
            if (eof()) break;
            int c = *p, lo, hi;
            bool brace = false;
            const unsigned char *save = p;
            if (c == '*') { lo = 0; hi = -1; p++; }
            else if (c == '+') { lo = 1; hi = -1; p++; }
            else if (c == '?') { lo = 0; hi = 1; p++; }
            else if (c == '{') {
                p++;
                lo = number();
                if (!eof() && *p == ',') {
                    p++;
                    hi = number();
                    if (lo < 0 && hi < 0) { p = save; break; }
                    if (lo < 0) lo = 0;
                }
                else {
                    if (lo < 0) { p = save; break; }
                    hi = lo;
                }
                if (eof() || *p != '}') { p = save; break; }
                p++;
                if (lo > 100000 || hi > 100000) { fail("too big number for repeat range"); return nullptr; }
                if (hi >= 0 && lo > hi) { fail("upper is smaller than lower in repeat range"); return nullptr; }
                brace = true;
            }
            else break;
            if (a->t == Ast::ANCHOR || a->t == Ast::KEEP) { fail("target of repeat operator is invalid"); return nullptr; }
            AstP r = mk(Ast::REPEAT);
            r->min = lo; r->max = hi;
            if (!eof() && *p == '?') { p++; r->greedy = false; }
            bool possessive = false;
            if (r->greedy && !brace && !eof() && *p == '+') {
                nonregular = true;
                if (!ext) { fail("possessive repeats are not supported on the GPU path"); return nullptr; }
                p++;
                possessive = true;
            }
            r->kids.push_back(std::move(a));
            a = std::move(r);
            if (possessive) { AstP at = mk(Ast::ATOMIC); at->kids.push_back(std::move(a)); a = std::move(at); }
        }
        return a;
    }

    // ---- (?i): the multi-character case folds that reach ASCII letters.  The reference folds with
    // INTERNAL_ONIGENC_CASE_FOLD_MULTI_CHAR on (ONIGENC_CASE_FOLD_DEFAULT): inside ONE literal string of the pattern,
    // "ss" also matches U+00DF / U+1E9E, "st" U+FB05 / U+FB06, "ff" "fi" "fl" "ffi" "ffl" U+FB00 .. U+FB04
    // (lib/onigmo/regcomp.c:3575 expand_case_fold_string, enc/unicode.c onigenc_unicode_get_case_fold_codes_by_str:
    // alternatives per position, the rest of the string compared case-folded at run time -- every way of cutting the run
    // into letters and ligatures).  A string is a run of literal characters that no quantifier, group or class
    // interrupts (a quantifier takes the last character out of the run: regparse.c parse_exp).
    static AstP clone_set(const Ast *a) { AstP c = mk(Ast::SET); c->cc = a->cc; c->ilit = a->ilit; return c; }
    static AstP cp_class(std::initializer_list<uint32_t> cps) {
        AstP a = mk(Ast::SET);
        a->cc.kind = CC::CLASS;
        for (uint32_t c : cps) a->cc.add_cp(c, c);
        a->cc.mb.norm(); a->cc.mbx.norm();
        return a;
    }
    int fold_budget = 0;
    AstP fold_run(const std::vector<AstP> &run, size_t i) {
        if (i >= run.size()) return mk(Ast::EMPTY);
        if (++fold_budget > 4000) { fail("pattern too large for the GPU tables (case folds)"); return mk(Ast::EMPTY); }
        auto at = [&](size_t k, uint32_t c) { return k < run.size() && run[k]->ilit == c; };
        std::vector<AstP> alts;
        auto add = [&](AstP head, size_t used) {
            AstP c = mk(Ast::CAT);
            c->kids.push_back(std::move(head));
            c->kids.push_back(fold_run(run, i + used));
            alts.push_back(std::move(c));
        };
        add(clone_set(run[i].get()), 1);
        if (ext) {
            // every multi-character fold whose sequence starts here: the characters that stand for it, as one class per sequence
            for (size_t m = 0; m < CF_NMULTI; m++) {
                const uint32_t *mf = CF_MULTI[m];
                bool same = true, first = true;
                for (uint32_t k = 0; same && k < mf[1]; k++) same = at(i + k, mf[2 + k]);
                if (!same) continue;
                for (size_t m2 = 0; m2 < m; m2++)            // (one alternative per sequence, for all its characters)
                    if (CF_MULTI[m2][1] == mf[1] && CF_MULTI[m2][2] == mf[2] && CF_MULTI[m2][3] == mf[3] && CF_MULTI[m2][4] == mf[4]) first = false;
                if (!first) continue;
                AstP cl = mk(Ast::SET);
                cl->cc.kind = CC::CLASS;
                for (size_t m2 = m; m2 < CF_NMULTI; m2++)
                    if (CF_MULTI[m2][1] == mf[1] && CF_MULTI[m2][2] == mf[2] && CF_MULTI[m2][3] == mf[3] && CF_MULTI[m2][4] == mf[4]) {
                        const uint32_t c0 = CF_MULTI[m2][0];
                        cl->cc.add_cp(c0, c0);
                        for (size_t q = cf_first(c0); q < CF_NSIMPLE && CF_SIMPLE[q][0] == c0; q++) cl->cc.add_cp(CF_SIMPLE[q][1], CF_SIMPLE[q][1]);
                    }
                cl->cc.mb.norm(); cl->cc.mbx.norm();
                add(std::move(cl), mf[1]);
            }
            if (alts.size() == 1) return std::move(alts[0]);
            AstP alt = mk(Ast::ALT);
            for (auto &x : alts) alt->kids.push_back(std::move(x));
            return alt;
        }
        if (at(i, 's') && at(i + 1, 's')) add(cp_class({0xDF, 0x1E9E}), 2);
        if (at(i, 's') && at(i + 1, 't')) add(cp_class({0xFB05, 0xFB06}), 2);
        if (at(i, 'f') && at(i + 1, 'f')) {
            add(cp_class({0xFB00}), 2);
            if (at(i + 2, 'i')) add(cp_class({0xFB03}), 3);
            if (at(i + 2, 'l')) add(cp_class({0xFB04}), 3);
        }
        if (at(i, 'f') && at(i + 1, 'i')) add(cp_class({0xFB01}), 2);
        if (at(i, 'f') && at(i + 1, 'l')) add(cp_class({0xFB02}), 2);
        if (alts.size() == 1) return std::move(alts[0]);
        AstP alt = mk(Ast::ALT);
        for (auto &x : alts) alt->kids.push_back(std::move(x));
        return alt;
    }
    void fold_strings(Ast *cat) {
        std::vector<AstP> out;
        size_t i = 0;
        auto &k = cat->kids;
        while (i < k.size()) {
            size_t j = i;
            while (j < k.size() && k[j]->t == Ast::SET && k[j]->ilit) j++;
            bool hit = false;
            for (size_t q = i; q + 1 < j && !hit; q++) {
                const uint32_t a = k[q]->ilit, b = k[q + 1]->ilit;
                hit = (a == 's' && (b == 's' || b == 't')) || (a == 'f' && (b == 'f' || b == 'i' || b == 'l'));
                if (ext && !hit) for (size_t m = 0; m < CF_NMULTI && !hit; m++) hit = CF_MULTI[m][2] == a && CF_MULTI[m][3] == b;
            }
            if (!hit) { for (size_t q = i; q < (j > i ? j : i + 1); q++) out.push_back(std::move(k[q])); i = j > i ? j : i + 1; continue; }
            std::vector<AstP> run;
            for (size_t q = i; q < j; q++) run.push_back(std::move(k[q]));
            out.push_back(fold_run(run, 0));
            i = j;
        }
        k = std::move(out);
    }

    AstP concat(unsigned &opts, int depth) {
        AstP cat = mk(Ast::CAT);
        for (;;) {
            bool tail = false;
            AstP r = piece(opts, depth, tail);
            if (failed()) return cat;
            if (!r) break;
            cat->kids.push_back(std::move(r));
            if (tail) break;
        }
        fold_strings(cat.get());
        return cat;
    }
};

AstP Syntax::alternation(unsigned opts, int depth) {
    unsigned o = opts;
    AstP first = concat(o, depth);
    if (failed() || eof() || *p != '|') return first;
    AstP alt = mk(Ast::ALT);
    alt->kids.push_back(std::move(first));
    while (!eof() && *p == '|') {
        p++;
        alt->kids.push_back(concat(o, depth));
        if (failed()) break;
    }
    return alt;
}

bool scan_named(const unsigned char *s, const unsigned char *e) {
    int in_class = 0;
    for (const unsigned char *p = s; p < e;) {
        if (*p == '\\') { p += 2; continue; }
        if (in_class) {
            if (*p == '[') in_class++;
            else if (*p == ']') in_class--;
            p++;
            continue;
        }
        if (*p == '[') { in_class = 1; p++; if (p < e && *p == '^') p++; if (p < e && *p == ']') p++; continue; }
        if (*p == '(' && p + 2 < e && p[1] == '?' &&
            ((p[2] == '<' && p + 3 < e && p[3] != '=' && p[3] != '!') || p[2] == '\'')) return true;
        p++;
    }
    return false;
}

// ---------------------------------------------------------------- byte-level NFA
enum NType { N_CONSUME, N_SPLIT, N_ASSERT, N_SAVE, N_MATCH };

struct NNode {
    NType t;
    ByteSet set;              // CONSUME
    int next = -1;            // CONSUME / ASSERT / SAVE
    std::vector<int> outs;    // SPLIT (priority order)
    AnchorKind akind = A_BOL; // ASSERT
    int slot = 0;             // SAVE
    bool is_loop = false;     // SPLIT heading an unbounded repeat
    int exit = -1;            // loop: the edge leaving the loop
    int pos = -1;             // CONSUME: position id
};

constexpr int MAX_NODES = 6000;

struct Nfa {
    std::vector<NNode> n;
    int npos = 0;
    int start = -1;
    bool uses_nl = false, uses_word = false;      // uses_word: a Unicode-aware \b / \B (the default)
    bool uses_aword = false;                      // an ASCII-only \b / \B ((?a))
    std::vector<const void *> pos_ast;            // character-level automaton (rx_nfa.inc): position -> its character node
    std::string err;

    int add(NNode x) {
        if ((int) n.size() >= MAX_NODES) { if (err.empty()) err = "pattern too large for the GPU tables"; return 0; }
        n.push_back(std::move(x));
        return (int) n.size() - 1;
    }
    std::vector<char> pos_hi;                     // ascii set: the position's character node also accepts some character >= 0x80
    int consume(const ByteSet &s, int next) {
        NNode x; x.t = N_CONSUME; x.set = s; x.next = next; x.pos = npos++;
        pos_hi.push_back(0);
        return add(std::move(x));
    }
    int split(std::vector<int> outs) { NNode x; x.t = N_SPLIT; x.outs = std::move(outs); return add(std::move(x)); }
};

// UTF-8 encoding of a code point set as alternatives of byte-range sequences
struct Seq { int n; uint8_t lo[4], hi[4]; };

void utf8_split(uint32_t lo, uint32_t hi, std::vector<Seq> &out) {
    if (lo > hi) return;
    static const uint32_t lim[3] = {0x7f, 0x7ff, 0xffff};
    for (uint32_t l : lim) if (lo <= l && l < hi) { utf8_split(lo, l, out); utf8_split(l + 1, hi, out); return; }
    if (hi < 0x80) { Seq s; s.n = 1; s.lo[0] = (uint8_t) lo; s.hi[0] = (uint8_t) hi; out.push_back(s); return; }
    int n = lo < 0x800 ? 2 : lo < 0x10000 ? 3 : 4;
    for (int i = 1; i < n; i++) {
        uint32_t m = (1u << (6 * i)) - 1;
        if ((lo & ~m) != (hi & ~m)) {
            if ((lo & m) != 0) { utf8_split(lo, lo | m, out); utf8_split((lo | m) + 1, hi, out); return; }
            if ((hi & m) != m) { utf8_split(lo, (hi & ~m) - 1, out); utf8_split(hi & ~m, hi, out); return; }
        }
    }
    auto enc = [n](uint32_t c, uint8_t *b) {
        if (n == 2) { b[0] = 0xc0 | (c >> 6); b[1] = 0x80 | (c & 0x3f); }
        else if (n == 3) { b[0] = 0xe0 | (c >> 12); b[1] = 0x80 | ((c >> 6) & 0x3f); b[2] = 0x80 | (c & 0x3f); }
        else { b[0] = 0xf0 | (c >> 18); b[1] = 0x80 | ((c >> 12) & 0x3f); b[2] = 0x80 | ((c >> 6) & 0x3f); b[3] = 0x80 | (c & 0x3f); }
    };
    Seq s; s.n = n;
    enc(lo, s.lo); enc(hi, s.hi);
    out.push_back(s);
}

// ---- input that is not well-formed UTF-8 (utf8 table set only)
// The walkers do not step on raw bytes there but on SYMBOLS: every byte of a well-formed sequence is itself;
// a byte >= 0x80 that is a character of its own (never-valid lead, stray continuation, lead whose sequence
// breaks off -- regenc.c onigenc_mbclen gives such a byte the length 1) and the lead of a prefix-valid
// sequence cut by the end of the text (one character that spans the rest of the text; the walkers shorten the
// text to end right after that lead) are replaced by one of the eleven byte values that never occur in
// well-formed UTF-8.  Two bytes share a symbol when every character node of the pattern treats them alike,
// so the symbol alone tells the automaton what the byte matches.
const uint8_t SPARE_BYTES[11] = {0xc0, 0xc1, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa, 0xfb, 0xfc, 0xfd};

struct SymbolMap {
    uint8_t xl[512];                                   // [b]: stray byte b, [256 + b]: truncated sequence with lead b
    SymbolMap() { for (int b = 0; b < 256; b++) { xl[b] = (uint8_t) b; xl[256 + b] = (uint8_t) b; } }
    std::map<const Ast *, ByteSet> accepts;            // per character node: the symbols it accepts
    bool sym_word[11] = {false, false, false, false, false, false, false, false, false, false, false};   // uses_word: the symbol's byte is a word character
    static bool has_word_anchor(const Ast *a) {
        if (a->t == Ast::ANCHOR && (a->anchor == A_WORDB || a->anchor == A_NWORDB)) return true;
        for (auto &k : a->kids) if (has_word_anchor(k.get())) return true;
        return false;
    }
    bool build(const Ast *root, std::string &err) {
        std::vector<const Ast *> atoms;
        collect(root, atoms);
        // \b / \B: a stray byte b (a character of its own whose code is b) and the lead of a cut sequence (code = the lead) are
        // word characters when U+00xx is one (regexec.c OP_WORD_BOUND: ONIGENC_IS_MBC_WORD on the decoded character): the kind
        // is part of what tells symbols apart
        const bool uses_word = has_word_anchor(root);
        std::map<std::vector<bool>, int> ids;
        std::vector<std::vector<bool>> sigs;
        auto sym_of = [&](const std::vector<bool> &sig) -> int {
            auto it = ids.find(sig);
            if (it != ids.end()) return it->second;
            int id = (int) sigs.size();
            ids[sig] = id;
            sigs.push_back(sig);
            return id;
        };
        memset(xl, 0, sizeof(xl));
        for (int b = 0; b < 256; b++) { xl[b] = (uint8_t) b; xl[256 + b] = (uint8_t) b; }
        for (int b = 0x80; b < 256; b++) {
            const bool w = uses_word && b < 0xfe && unicode_word((uint32_t) b);     // (0xFE / 0xFF decode to the engine's INVALID_CODE_*: no word)
            std::vector<bool> sig(atoms.size() + 1);
            for (size_t i = 0; i < atoms.size(); i++) sig[i] = atoms[i]->cc.invalid_byte(b);
            sig[atoms.size()] = w;
            int id = sym_of(sig);
            if (id >= 11) { err = "too many kinds of ill-formed UTF-8 bytes for the GPU tables"; return false; }
            xl[b] = SPARE_BYTES[id];
            sym_word[id] = w;
            if (b >= 0xc2 && b <= 0xf4) {
                for (size_t i = 0; i < atoms.size(); i++) sig[i] = atoms[i]->cc.truncated(b);
                id = sym_of(sig);
                if (id >= 11) { err = "too many kinds of ill-formed UTF-8 bytes for the GPU tables"; return false; }
                xl[256 + b] = SPARE_BYTES[id];
                sym_word[id] = w;
            }
            else xl[256 + b] = xl[b];
        }
        for (size_t i = 0; i < atoms.size(); i++) {
            ByteSet bs;
            for (size_t k = 0; k < sigs.size(); k++) if (sigs[k][i]) bs.set(SPARE_BYTES[k]);
            accepts[atoms[i]] = bs;
        }
        return true;
    }
    static void collect(const Ast *a, std::vector<const Ast *> &out) {
        if (a->t == Ast::SET) out.push_back(a);
        for (auto &k : a->kids) collect(k.get(), out);
    }
};

struct Builder {
    Nfa &nfa;
    bool ascii_only;
    const SymbolMap *syms;
    bool char_level = false;              // rx_nfa.inc: ONE position per character node, whatever the character's encoded length
    Builder(Nfa &n, bool ascii, const SymbolMap *sm) : nfa(n), ascii_only(ascii), syms(sm) {}

    // node matching one character accepted by the character node `a`, continuing at `next`
    int build_set(const Ast *a, int next) {
        const CC &cc = a->cc;
        if (char_level) {
            ByteSet none;
            const int nd = nfa.consume(none, next);
            nfa.pos_ast.push_back(a);
            return nd;
        }
        ByteSet single;
        for (int b = 0; b < 0x80; b++) if (cc.ascii(b)) single.set(b);
        std::vector<Seq> seqs;
        if (!ascii_only) {
            auto it = syms->accepts.find(a);
            if (it != syms->accepts.end()) single.merge(it->second);
            CodeSet v = cc.valid_multibyte();
            for (auto &r : v.r) {
                uint32_t lo = r.first, hi = std::min(r.second, MAXCP);
                // surrogates are not encodable
                if (lo <= 0xd7ff) utf8_split(lo, std::min<uint32_t>(hi, 0xd7ff), seqs);
                if (hi >= 0xe000) utf8_split(std::max<uint32_t>(lo, 0xe000), hi, seqs);
            }
        }
        std::vector<int> alts;
        if (!single.empty()) alts.push_back(nfa.consume(single, next));
        if (ascii_only && !alts.empty()) {
            bool hi = !cc.valid_multibyte().r.empty();
            for (int b = 0x80; b < 256 && !hi; b++) hi = cc.invalid_byte(b) || (b >= 0xc2 && b <= 0xf4 && cc.truncated(b));
            nfa.pos_hi.back() = hi ? 1 : 0;
        }
        // share continuation chains between sequences with identical tails
        std::map<std::vector<uint8_t>, int> tails;
        // group sequences by (tail signature) so leads with the same tail merge into one CONSUME
        std::map<std::vector<uint8_t>, ByteSet> lead_by_tail;
        for (auto &s : seqs) {
            std::vector<uint8_t> sig;
            for (int i = 1; i < s.n; i++) { sig.push_back(s.lo[i]); sig.push_back(s.hi[i]); }
            ByteSet &ls = lead_by_tail[sig];
            ls.set_range(s.lo[0], s.hi[0]);
        }
        for (auto &kv : lead_by_tail) {
            const std::vector<uint8_t> &sig = kv.first;
            int cont = next;
            for (int i = (int) sig.size() / 2 - 1; i >= 0; i--) {
                std::vector<uint8_t> tsig(sig.begin() + 2 * i, sig.end());
                auto it = tails.find(tsig);
                if (it != tails.end()) { cont = it->second; }
                else {
                    ByteSet bs; bs.set_range(sig[2 * i], sig[2 * i + 1]);
                    cont = nfa.consume(bs, cont);
                    tails[tsig] = cont;
                }
            }
            alts.push_back(nfa.consume(kv.second, cont));
        }
        if (alts.empty()) {
            ByteSet none;
            const int nd = nfa.consume(none, next);      // matches nothing (ascii set: nothing below 0x80)
            if (ascii_only) {
                bool hi = !cc.valid_multibyte().r.empty();
                for (int b = 0x80; b < 256 && !hi; b++) hi = cc.invalid_byte(b) || (b >= 0xc2 && b <= 0xf4 && cc.truncated(b));
                nfa.pos_hi.back() = hi ? 1 : 0;
            }
            return nd;
        }
        if (alts.size() == 1) return alts[0];
        return nfa.split(alts);
    }

    int build(const Ast *a, int next) {
        if (!nfa.err.empty()) return next;
        switch (a->t) {
        case Ast::EMPTY: return next;
        case Ast::SET: return build_set(a, next);
        case Ast::CAT: {
            int cur = next;
            for (int i = (int) a->kids.size() - 1; i >= 0; i--) cur = build(a->kids[i].get(), cur);
            return cur;
        }
        case Ast::ALT: {
            std::vector<int> outs;
            for (auto &k : a->kids) outs.push_back(build(k.get(), next));
            return nfa.split(outs);
        }
        case Ast::GROUP: {
            if (!a->cap) return build(a->kids[0].get(), next);
            NNode close; close.t = N_SAVE; close.slot = 2 * a->cap + 1; close.next = next;
            int c = nfa.add(close);
            int body = build(a->kids[0].get(), c);
            NNode open; open.t = N_SAVE; open.slot = 2 * a->cap; open.next = body;
            return nfa.add(open);
        }
        case Ast::ANCHOR: {
            NNode x; x.t = N_ASSERT; x.akind = a->anchor; x.next = next;
            if (a->anchor == A_BOL || a->anchor == A_EOL) nfa.uses_nl = true;
            if (a->anchor == A_WORDB || a->anchor == A_NWORDB) nfa.uses_word = true;
            if (a->anchor == A_WORDB_A || a->anchor == A_NWORDB_A) nfa.uses_aword = true;
            return nfa.add(x);
        }
        case Ast::REPEAT: {
            const Ast *body = a->kids[0].get();
            int cur;
            if (a->max < 0) {
                // unbounded tail: L = SPLIT(body -> L, exit)
                NNode l; l.t = N_SPLIT; l.is_loop = true; l.exit = next;
                int L = nfa.add(l);
                int b = build(body, L);
                if (a->greedy) nfa.n[L].outs = {b, next};
                else nfa.n[L].outs = {next, b};
                cur = L;
            }
            else {
                cur = next;
                for (int i = a->min; i < a->max; i++) {
                    int b = build(body, cur);
                    cur = a->greedy ? nfa.split({b, next}) : nfa.split({next, b});
                }
            }
            for (int i = 0; i < a->min; i++) cur = build(body, cur);
            return cur;
        }
        default:
            // (look-around, atomic groups, back-references, \K, conditionals: compile() refuses them in front of every table builder)
            nfa.err = "not a regular expression";
            return next;
        }
        return next;
    }
};

// ---------------------------------------------------------------- closure lists
// K_WORD: an ASCII word byte; K_UWORD: a non-ASCII character that is a Unicode word character (told apart from K_WORD only when the
// pattern holds both a Unicode-aware and an ASCII-only (?a) word anchor)
enum Kind { K_OTHER = 0, K_NL = 1, K_WORD = 2, K_EDGE = 3, K_UWORD = 4, NKIND = 5 };
// canonical kind -> compact index of the kinds this pattern tells apart; returns their number
int kind_map(const Nfa &nfa, bool any_assert, int *kmap) {
    int n = 0;
    kmap[K_OTHER] = n++;
    kmap[K_NL] = nfa.uses_nl ? n++ : kmap[K_OTHER];
    kmap[K_WORD] = (nfa.uses_word || nfa.uses_aword) ? n++ : kmap[K_OTHER];
    kmap[K_EDGE] = any_assert ? n++ : kmap[K_OTHER];
    kmap[K_UWORD] = (nfa.uses_word && nfa.uses_aword) ? n++ : nfa.uses_word ? kmap[K_WORD] : kmap[K_OTHER];
    return n;
}
constexpr int T_MATCH = -1;

struct Target { int pos; int tagseq; };   // pos == T_MATCH for MATCH

struct Tables {
    const Nfa &nfa;
    std::vector<int> core_node;                       // core x -> NFA node to start the closure at
    std::vector<int> pos_node;                        // position -> CONSUME node
    std::map<std::vector<uint8_t>, int> tag_ids;
    std::vector<std::vector<uint8_t>> tag_seqs;
    std::vector<std::vector<Target>> lists;           // [(x * NKIND + pk) * NKIND + nk]
    std::vector<char> have;

    explicit Tables(const Nfa &n) : nfa(n) {
        pos_node.resize(nfa.npos);
        for (size_t i = 0; i < nfa.n.size(); i++) if (nfa.n[i].t == N_CONSUME) pos_node[nfa.n[i].pos] = (int) i;
        for (int p = 0; p < nfa.npos; p++) core_node.push_back(nfa.n[pos_node[p]].next);
        core_node.push_back(nfa.start);               // START core
        lists.resize(core_node.size() * NKIND * NKIND);
        have.assign(lists.size(), 0);
        intern({});
    }
    int ncores() const { return (int) core_node.size(); }
    int start_core() const { return (int) core_node.size() - 1; }

    int intern(const std::vector<uint8_t> &t) {
        auto it = tag_ids.find(t);
        if (it != tag_ids.end()) return it->second;
        int id = (int) tag_seqs.size();
        tag_ids[t] = id;
        tag_seqs.push_back(t);
        return id;
    }

    static bool assert_ok(AnchorKind a, int pk, int nk) {
        switch (a) {
        default: break;
        case A_BOL: return pk == K_EDGE || (pk == K_NL && nk != K_EDGE);   // OP_BEGIN_LINE
        case A_EOL: return nk == K_EDGE || nk == K_NL;                     // OP_END_LINE
        case A_BOS: return pk == K_EDGE;
        case A_EOS: return nk == K_EDGE;
        case A_WORDB: return (pk == K_WORD || pk == K_UWORD) != (nk == K_WORD || nk == K_UWORD);
        case A_NWORDB: return (pk == K_WORD || pk == K_UWORD) == (nk == K_WORD || nk == K_UWORD);
        case A_WORDB_A: return (pk == K_WORD) != (nk == K_WORD);
        case A_NWORDB_A: return (pk == K_WORD) == (nk == K_WORD);
        }
        return false;
    }

    long budget = 0;

    void dfs(int node, int pk, int nk, std::vector<char> &visited, std::vector<char> &onstack,
             std::vector<uint8_t> &tags, std::vector<Target> &out) {
        const NNode &nd = nfa.n[node];
        if (++budget > 4000000) return;
        if (nd.t == N_SPLIT && nd.is_loop && onstack[node]) {
            // the loop body matched the empty string: leave the loop (OP_NULL_CHECK_END)
            dfs(nd.exit, pk, nk, visited, onstack, tags, out);
            return;
        }
        if (visited[node]) return;
        visited[node] = 1;
        switch (nd.t) {
        case N_CONSUME: out.push_back({nd.pos, intern(tags)}); break;
        case N_MATCH: out.push_back({T_MATCH, intern(tags)}); break;
        case N_SAVE:
            tags.push_back((uint8_t) nd.slot);
            dfs(nd.next, pk, nk, visited, onstack, tags, out);
            tags.pop_back();
            break;
        case N_ASSERT:
            if (assert_ok(nd.akind, pk, nk)) dfs(nd.next, pk, nk, visited, onstack, tags, out);
            break;
        case N_SPLIT:
            if (!nd.is_loop) {
                for (int o : nd.outs) dfs(o, pk, nk, visited, onstack, tags, out);
                break;
            }
            onstack[node] = 1;
            for (int o : nd.outs) {
                if (o == nd.exit) { dfs(o, pk, nk, visited, onstack, tags, out); continue; }
                // A new iteration entered at this boundary may legitimately pass through nodes an
                // earlier (higher-priority) path already visited: a backtracking engine would try
                // them again with this iteration's captures, and the empty-iteration exit above
                // makes the outcome path-dependent.  Explore the body with a fresh visited set.
                std::vector<char> saved = visited;
                std::fill(visited.begin(), visited.end(), 0);
                visited[node] = 1;
                dfs(o, pk, nk, visited, onstack, tags, out);
                for (size_t i = 0; i < visited.size(); i++) visited[i] |= saved[i];
            }
            onstack[node] = 0;
            break;
        }
    }

    const std::vector<Target> &list(int x, int pk, int nk) {
        size_t idx = ((size_t) x * NKIND + pk) * NKIND + nk;
        if (!have[idx]) {
            std::vector<char> visited(nfa.n.size(), 0), onstack(nfa.n.size(), 0);
            std::vector<uint8_t> tags;
            std::vector<Target> raw;
            dfs(core_node[x], pk, nk, visited, onstack, tags, raw);
            std::vector<char> seen(nfa.npos + 1, 0);
            for (const Target &t : raw) {           // only the first (highest priority) occurrence
                int k = t.pos == T_MATCH ? nfa.npos : t.pos;
                if (seen[k]) continue;
                seen[k] = 1;
                lists[idx].push_back(t);
                if (t.pos == T_MATCH) break;        // nothing after MATCH can ever be chosen
            }
            have[idx] = 1;
        }
        return lists[idx];
    }
};

struct VecHash {
    size_t operator()(const std::vector<uint64_t> &v) const {
        uint64_t h = 0xcbf29ce484222325ull;
        for (uint64_t x : v) { h ^= x; h *= 0x100000001b3ull; h ^= h >> 29; }
        return (size_t) h;
    }
};

inline bool bit(const std::vector<uint64_t> &v, int i) { return (v[i >> 6] >> (i & 63)) & 1; }
inline void setbit(std::vector<uint64_t> &v, int i) { v[i >> 6] |= 1ull << (i & 63); }

constexpr int MAX_DFA_STATES = 20000;
constexpr int MAX_WIDE_STATES = 200000;    // utf8 set's reverse automaton with 32-bit entries (rx.hpp: `wide`)

}  // namespace

// ---------------------------------------------------------------- public: /pat/flags splitting
void split_flb_pattern(const char *pattern, const char **start, const char **end, unsigned *options) {
    // src/flb_regex.c:60-152 check_option() + str_to_regex()
    size_t len = strlen(pattern);
    const char *s = pattern, *e = pattern + len, *new_end = nullptr;
    unsigned opt = 0;
    if (s[0] == '/') {
        const char *chr = strrchr(s, '/');
        if (chr && chr != s && chr != e) {
            bool ok = true;
            new_end = chr;
            for (chr++; chr != e && *chr; chr++) {
                if (*chr == 'm') opt |= OPT_MULTILINE;
                else if (*chr == 'i') opt |= OPT_IGNORECASE;
                else if (*chr == 'x') opt |= OPT_EXTEND;
                else if (*chr == 'o') { }
                else { ok = false; break; }
            }
            if (!ok || opt == 0) { new_end = nullptr; opt = 0; }
        }
    }
    if (len > 1 && pattern[0] == '/' && pattern[len - 1] == '/') { s++; e--; }
    if (new_end) { s = pattern + 1; e = new_end; }
    *start = s; *end = e; *options = opt;
}

// ---------------------------------------------------------------- compile
namespace {

bool build_tables(const Ast *root, bool ascii_only, bool want_match_dfa, bool want_capture, const std::vector<uint8_t> &slot2cap, TableSet &out, std::string &err) {
    SymbolMap syms;
    if (!ascii_only && !syms.build(root, err)) return false;
    Nfa nfa;
    {
        NNode m; m.t = N_MATCH;
        int match = nfa.add(m);
        NNode close; close.t = N_SAVE; close.slot = 1; close.next = match;     // group 0 end
        int c = nfa.add(close);
        Builder b(nfa, ascii_only, &syms);
        nfa.start = b.build(root, c);           // group 0 begin is the start boundary itself
    }
    if (!nfa.err.empty()) { err = nfa.err; return false; }
    if (nfa.npos > 4000) { err = "pattern too large for the GPU tables (positions)"; return false; }
    out = TableSet();
    out.ascii_only = ascii_only;
    memcpy(out.xl, syms.xl, sizeof(out.xl));

    // ---- context kinds actually distinguished by this pattern
    bool any_assert = false;
    for (auto &nd : nfa.n) if (nd.t == N_ASSERT) any_assert = true;
    int kmap[NKIND];                         // canonical kind -> compact index
    out.NK = kind_map(nfa, any_assert, kmap);
    out.kind_edge = kmap[K_EDGE];
    if (out.NK > 4) { err = "pattern mixes (?a) and Unicode word anchors with line anchors: more context kinds than the byte tables hold (needs the NFA engine)"; return false; }
    int kinv[NKIND];                         // compact index -> a canonical kind
    for (int k = NKIND - 1; k >= 0; k--) kinv[kmap[k]] = k;
    // kind of a SYMBOL (rx.hpp TableSet::cls): an ASCII byte by itself; utf8 set with \b / \B: a stray byte / the lead of a cut
    // sequence by the code it stands for (SymbolMap::sym_word), a byte of a well-formed multi-byte character by the half of
    // the symbol space the walker found it in (256 + b: the character is a word character)
    const int NSYM = (!ascii_only && nfa.uses_word) ? 512 : 256;
    out.word_variants = NSYM == 512;
    auto kind_of_byte = [&](int sym) -> int {
        const int b = sym & 255;
        if (nfa.uses_nl && b == '\n') return K_NL;
        if (!nfa.uses_word && !nfa.uses_aword) return K_OTHER;
        if (b < 0x80) return ((b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_') ? K_WORD : K_OTHER;
        if (ascii_only || !nfa.uses_word) return K_OTHER;
        for (int k = 0; k < 11; k++) if (SPARE_BYTES[k] == b) return syms.sym_word[k] ? K_UWORD : K_OTHER;
        return sym >= 256 ? K_UWORD : K_OTHER;
    };

    Tables tb(nfa);
    const int P = nfa.npos;
    {
        std::map<std::vector<uint64_t>, int> sig2cls;
        std::vector<uint64_t> sig((P + 63) / 64 + 1);
        for (int sy = 0; sy < NSYM; sy++) {
            const int b = sy & 255;
            std::fill(sig.begin(), sig.end(), 0);
            for (int p = 0; p < P; p++) if (nfa.n[tb.pos_node[p]].set.has(b)) setbit(sig, p);
            sig.back() = (uint64_t) kind_of_byte(sy) | ((ascii_only && b >= 0x80) ? 16u : 0u);
            auto it = sig2cls.find(sig);
            int id;
            if (it == sig2cls.end()) { id = (int) sig2cls.size(); sig2cls[sig] = id; }
            else id = it->second;
            out.cls[sy] = (uint8_t) id;
            if (ascii_only && b >= 0x80) out.high_cls = id;
        }
        if (NSYM == 256) for (int sy = 256; sy < 512; sy++) out.cls[sy] = out.cls[sy - 256];
        out.ncls = (int) sig2cls.size();
        if (out.ncls > 63) { err = "too many byte classes for the GPU tables"; return false; }
    }
    std::vector<int> rep(out.ncls, -1), rsym(out.ncls, -1);      // a byte of the class / a symbol of the class
    for (int sy = NSYM - 1; sy >= 0; sy--) { rsym[out.cls[sy]] = sy; rep[out.cls[sy]] = sy & 255; }
    std::vector<std::vector<int>> pos_of_cls(out.ncls);
    for (int c = 0; c < out.ncls; c++)
        for (int p = 0; p < P; p++) if (nfa.n[tb.pos_node[p]].set.has(rep[c])) pos_of_cls[c].push_back(p);
    std::vector<int> kind_cls(out.ncls);     // canonical kinds
    for (int c = 0; c < out.ncls; c++) kind_cls[c] = kind_of_byte(rsym[c]);

    const int X = tb.ncores(), START = tb.start_core();
    const int W = (X + 63) / 64;

    // ---- match-only forward DFA over sets of cores
    if (want_match_dfa) {
        std::unordered_map<std::vector<uint64_t>, int, VecHash> ids;
        std::vector<std::vector<uint64_t>> states;
        auto intern = [&](std::vector<uint64_t> &k) -> int {
            auto it = ids.find(k);
            if (it != ids.end()) return it->second;
            int id = (int) states.size();
            ids.emplace(k, id);
            states.push_back(k);
            return id;
        };
        std::vector<uint64_t> k0(W + 1, 0);
        setbit(k0, START);
        k0[W] = (uint64_t) kinv[kmap[K_EDGE]];
        out.d_init = intern(k0);
        for (size_t si = 0; si < states.size(); si++) {
            if ((int) states.size() > MAX_DFA_STATES) { err = "match DFA exceeds the state budget"; return false; }
            std::vector<uint64_t> S = states[si];
            int pk = (int) S[W];
            out.ddelta.resize((si + 1) * out.ncls);
            for (int c = 0; c < out.ncls; c++) {
                if (c == out.high_cls) { out.ddelta[si * out.ncls + c] = D_POISON; continue; }
                int nk = kind_cls[c];
                bool accept = false;
                std::vector<uint64_t> N(W + 1, 0);
                setbit(N, START);
                for (int x = 0; x < X && !accept; x++) {
                    if (!bit(S, x)) continue;
                    for (const Target &t : tb.list(x, pk, nk)) {
                        if (t.pos == T_MATCH) { accept = true; break; }
                        if (nfa.n[tb.pos_node[t.pos]].set.has(rep[c])) setbit(N, t.pos);
                    }
                }
                if (accept) { out.ddelta[si * out.ncls + c] = D_ACCEPT; continue; }
                N[W] = (uint64_t) kinv[kmap[nk]];
                int id = intern(N);
                if (id >= 0xFFF0) { err = "match DFA exceeds the state budget"; return false; }
                out.ddelta[si * out.ncls + c] = (uint16_t) id;
            }
            bool fin = false;
            for (int x = 0; x < X && !fin; x++) {
                if (!bit(S, x)) continue;
                for (const Target &t : tb.list(x, pk, K_EDGE)) if (t.pos == T_MATCH) { fin = true; break; }
            }
            out.d_final.push_back(fin ? 1 : 0);
        }
        out.nD = (int) states.size();
        // ---- d_live: can a match still follow from this state on a text WITHOUT line feeds, whatever its characters -- the ones
        // >= 0x80 included, which this set's transitions do not describe?  (The product automaton of the multiline rules,
        // ml.cpp build_product, may stop reading a line only in states that are dead for real: a rule like /^\s+原因/ has no accepting
        // path over ASCII bytes at all.)  Over-approximated per (core, previous kind): some next kind has a list entry that is MATCH or a
        // position that takes a character of that kind and is live behind it.
        {
            auto ascii_word = [](int b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_'; };
            const bool any_word = nfa.uses_word || nfa.uses_aword;
            std::vector<int> kinds = {K_OTHER};
            if (any_word) kinds.push_back(K_WORD);
            if (nfa.uses_word) kinds.push_back(K_UWORD);
            auto takes = [&](int p, int k) -> bool {
                const NNode &nd = nfa.n[tb.pos_node[p]];
                if (nfa.pos_hi[p] && (k == K_OTHER || k == K_UWORD)) return true;
                for (int b = 0; b < 0x80; b++) {
                    if (b == '\n' || !nd.set.has(b)) continue;
                    const int kb = (any_word && ascii_word(b)) ? K_WORD : K_OTHER;
                    if (kb == k) return true;
                }
                return false;
            };
            std::vector<char> live((size_t) X * NKIND, 0);
            for (bool changed = true; changed;) {
                changed = false;
                for (int x = 0; x < X; x++)
                    for (int pk = 0; pk < NKIND; pk++) {
                        if (live[(size_t) x * NKIND + pk]) continue;
                        bool l = false;
                        for (const Target &t : tb.list(x, pk, K_EDGE)) if (t.pos == T_MATCH) l = true;
                        for (size_t ki = 0; ki < kinds.size() && !l; ki++)
                            for (const Target &t : tb.list(x, pk, kinds[ki])) {
                                if (t.pos == T_MATCH) { l = true; break; }
                                if (takes(t.pos, kinds[ki]) && live[(size_t) t.pos * NKIND + kinds[ki]]) { l = true; break; }
                            }
                        if (l) { live[(size_t) x * NKIND + pk] = 1; changed = true; }
                    }
            }
            bool restart = false;                                 // the search may start again behind any later character
            for (int k : kinds) if (live[(size_t) START * NKIND + k]) restart = true;
            out.d_live.assign((size_t) out.nD, 0);
            for (int si = 0; si < out.nD; si++) {
                const int pk = (int) states[(size_t) si][W];
                bool l = restart;
                for (int x = 0; x < X && !l; x++) if (bit(states[(size_t) si], x) && live[(size_t) x * NKIND + pk]) l = true;
                out.d_live[(size_t) si] = l ? 1 : 0;
            }
        }
    }
    if (tb.budget > 4000000) { err = "pattern too complex (nested empty loops)"; return false; }
    if (!want_capture) return true;

    // ---- reverse DFA over sets of viable positions
    const int WP = (P + 63) / 64 + 1;          // last word: next-kind (canonical)
    std::unordered_map<std::vector<uint64_t>, int, VecHash> ids;
    std::vector<std::vector<uint64_t>> states;
    auto intern = [&](std::vector<uint64_t> &k) -> int {
        auto it = ids.find(k);
        if (it != ids.end()) return it->second;
        int id = (int) states.size();
        ids.emplace(k, id);
        states.push_back(k);
        return id;
    };
    auto ok = [&](int x, int pk, int nk, const std::vector<uint64_t> &V) -> bool {
        for (const Target &t : tb.list(x, pk, nk)) if (t.pos == T_MATCH || bit(V, t.pos)) return true;
        return false;
    };
    std::vector<uint64_t> k0(WP, 0);
    k0[WP - 1] = (uint64_t) kinv[kmap[K_EDGE]];
    out.r_init = intern(k0);
    // transitions are collected as 32-bit entries (id | start << 31) and narrowed below when the ids fit 15 bits
    std::vector<uint32_t> rd32;
    const int state_cap = ascii_only ? 0x7FF0 : MAX_WIDE_STATES;
    for (size_t si = 0; si < states.size(); si++) {
        if ((int) states.size() > state_cap) { err = "capture automaton exceeds the state budget"; return false; }
        std::vector<uint64_t> S = states[si];
        int nk = (int) S[WP - 1];
        rd32.resize((si + 1) * out.ncls);
        for (int c = 0; c < out.ncls; c++) {
            if (c == out.high_cls) { rd32[si * out.ncls + c] = R_POISON; continue; }
            int pk = kind_cls[c];
            std::vector<uint64_t> N(WP, 0);
            for (int p : pos_of_cls[c]) if (ok(p, pk, nk, S)) setbit(N, p);
            N[WP - 1] = (uint64_t) kinv[kmap[kind_cls[c]]];
            bool startok = ok(START, pk, nk, S);
            int id = intern(N);
            rd32[si * out.ncls + c] = (uint32_t) id | (startok ? 0x80000000u : 0u);
        }
        out.r_info.push_back((uint8_t) (kmap[nk] | (ok(START, K_EDGE, nk, S) ? 0x80 : 0)));
    }
    if ((int) states.size() > state_cap) { err = "capture automaton exceeds the state budget"; return false; }
    out.wide = (int) states.size() > 0x7FF0 || (!ascii_only && std::getenv("FLBGPU_RX_FORCE_WIDE") != nullptr);
    if (out.wide) out.rdelta32.swap(rd32);
    else {
        out.rdelta.resize(rd32.size());
        for (size_t i = 0; i < rd32.size(); i++)
            out.rdelta[i] = rd32[i] == R_POISON ? R_POISON : (uint16_t) ((rd32[i] & 0x7FFF) | ((rd32[i] >> 31) ? 0x8000 : 0));
    }
    out.nR = (int) states.size();
    out.P = P;
    out.VW = (P + 31) / 32;
    if (out.VW == 0) out.VW = 1;
    out.vmask.assign((size_t) out.nR * out.VW, 0);
    for (int r = 0; r < out.nR; r++)
        for (int p = 0; p < P; p++) if (bit(states[r], p)) out.vmask[(size_t) r * out.VW + (p >> 5)] |= 1u << (p & 31);

    // ---- forward candidate lists per (core, prev kind, next kind)
    out.nX = X;
    out.kind_of_cls.resize(out.ncls);
    for (int c = 0; c < out.ncls; c++) out.kind_of_cls[c] = (uint8_t) kmap[kind_cls[c]];
    out.list_off.push_back(0);
    for (int x = 0; x < X; x++)
        for (int pk = 0; pk < out.NK; pk++)
            for (int nk = 0; nk < out.NK; nk++) {
                for (const Target &t : tb.list(x, kinv[pk], kinv[nk]))
                    out.list_ent.push_back((t.pos == T_MATCH ? (uint32_t) F_MATCH : (uint32_t) t.pos) | ((uint32_t) t.tagseq << 16));
                out.list_off.push_back((uint32_t) out.list_ent.size());
            }
    // byte-resolved fast path (packed entries, see rx.hpp)
    {
        size_t nl = out.list_off.size() - 1;
        out.fc_shift = 0;
        while ((1 << out.fc_shift) < out.ncls + 1) out.fc_shift++;
        const size_t ncols = (size_t) 1 << out.fc_shift;
        out.fastc.assign(nl * ncols, FC_DEAD);
        const bool targets_fit = X < (int) FC_TMATCH;
        for (size_t li = 0; li < nl; li++) {
            for (int c = 0; c <= out.ncls; c++) {
                if (c == out.high_cls) continue;
                uint32_t pick = FC_DEAD;
                int cnt = 0;
                for (uint32_t k = out.list_off[li]; k < out.list_off[li + 1]; k++) {
                    uint32_t ent = out.list_ent[k], tg = ent & 0xFFFF;
                    bool possible = tg == F_MATCH || (c < out.ncls && nfa.n[tb.pos_node[tg]].set.has(rep[c]));
                    if (!possible) continue;
                    if (cnt == 0) pick = ent;
                    cnt++;
                    if (tg == F_MATCH) break;     // nothing after MATCH can be chosen
                }
                uint32_t enc;
                if (cnt == 0) enc = FC_DEAD;
                else if (cnt > 1 && targets_fit && c < out.ncls) {
                    // one byte of lookahead: keep the candidates that can still make a step (or
                    // finish) when the following byte has class c2 / the text ends
                    std::vector<uint32_t> row(ncols, FC_DEAD);
                    bool useful = false;
                    const int pk2 = kind_cls[c];
                    int ctx_pk = (int) ((li / out.NK) % out.NK), ctx_nk = (int) (li % out.NK);
                    (void) ctx_pk; (void) ctx_nk;
                    for (int c2 = 0; c2 <= out.ncls; c2++) {
                        if (c2 == out.high_cls) continue;
                        const int nk2 = c2 < out.ncls ? kind_cls[c2] : K_EDGE;
                        uint32_t pick2 = FC_DEAD;
                        int cnt2 = 0;
                        for (uint32_t k = out.list_off[li]; k < out.list_off[li + 1]; k++) {
                            uint32_t ent = out.list_ent[k], tg = ent & 0xFFFF;
                            if (tg == F_MATCH) { if (cnt2 == 0) pick2 = ent; cnt2++; break; }
                            if (!nfa.n[tb.pos_node[tg]].set.has(rep[c])) continue;
                            bool alive = false;
                            for (const Target &t2 : tb.list((int) tg, pk2, nk2)) {
                                if (t2.pos == T_MATCH) { alive = true; break; }
                                if (c2 < out.ncls && nfa.n[tb.pos_node[t2.pos]].set.has(rep[c2])) { alive = true; break; }
                            }
                            if (!alive) continue;
                            if (cnt2 == 0) pick2 = ent;
                            cnt2++;
                        }
                        uint32_t e2;
                        if (cnt2 == 0) e2 = FC_DEAD;
                        else if (cnt2 > 1) e2 = FC_MULTI;
                        else {
                            const std::vector<uint8_t> &tags = tb.tag_seqs[pick2 >> 16];
                            uint32_t caps[2] = {0, 0};
                            int nc = 0;
                            bool ok = true;
                            for (uint8_t slot : tags) {
                                uint8_t ci = slot < slot2cap.size() ? slot2cap[slot] : 0xFF;
                                if (ci == 0xFF) continue;
                                if (nc == 2 || ci >= 62) { ok = false; break; }
                                for (int q = 0; q < nc; q++) if (caps[q] == (uint32_t) ci + 1) ok = false;
                                caps[nc++] = (uint32_t) ci + 1;
                            }
                            uint32_t tg = pick2 & 0xFFFF;
                            e2 = ok ? ((tg == F_MATCH ? FC_TMATCH : tg) | (caps[0] << 12) | (caps[1] << 18)) : FC_MULTI;
                            if (ok) useful = true;
                        }
                        row[c2] = e2;
                    }
                    if (useful && out.fast2.size() / ncols < 0xFFFFFF) {
                        enc = FC_LOOK | (uint32_t) (out.fast2.size() / ncols);
                        out.fast2.insert(out.fast2.end(), row.begin(), row.end());
                    }
                    else enc = FC_MULTI;
                }
                else if (cnt > 1 || !targets_fit) enc = FC_MULTI;
                else {
                    // pack the named-group capture writes of the tag sequence
                    const std::vector<uint8_t> &tags = tb.tag_seqs[pick >> 16];
                    uint32_t caps[2] = {0, 0};
                    int nc = 0;
                    bool ok = true;
                    for (uint8_t slot : tags) {
                        uint8_t ci = slot < slot2cap.size() ? slot2cap[slot] : 0xFF;
                        if (ci == 0xFF) continue;
                        if (nc == 2 || ci >= 62) { ok = false; break; }
                        for (int q = 0; q < nc; q++) if (caps[q] == (uint32_t) ci + 1) ok = false;   // repeated slot: keep order semantics in the slow path
                        caps[nc++] = (uint32_t) ci + 1;
                    }
                    uint32_t tg = pick & 0xFFFF;
                    enc = ok ? ((tg == F_MATCH ? FC_TMATCH : tg) | (caps[0] << 12) | (caps[1] << 18)) : FC_MULTI;
                }
                out.fastc[li * ncols + c] = enc;
            }
        }
        // padded reverse table and the fused class|kind byte table
        out.cls_shift = 0;
        while ((1 << out.cls_shift) < out.ncls) out.cls_shift++;
        const size_t rc = (size_t) 1 << out.cls_shift;
        // row nR is an absorbing POISON state (entered on a byte >= 0x80 in the ASCII set), so
        // the kernels can test for it once per unrolled group and never index out of the table
        if (out.wide) {
            // (no POISON in the utf8 set; row nR is kept so both layouts have nR + 1 rows)
            out.rdelta32_p.assign((size_t) (out.nR + 1) * rc, (uint32_t) out.nR);
            for (int r = 0; r < out.nR; r++)
                for (int c = 0; c < out.ncls; c++) out.rdelta32_p[(size_t) r * rc + c] = out.rdelta32[(size_t) r * out.ncls + c];
        }
        else {
            out.rdelta_p.assign((size_t) (out.nR + 1) * rc, (uint16_t) out.nR);
            for (int r = 0; r < out.nR; r++)
                for (int c = 0; c < out.ncls; c++) {
                    uint16_t e = out.rdelta[(size_t) r * out.ncls + c];
                    out.rdelta_p[(size_t) r * rc + c] = (e & 0x7FFF) == R_POISON ? (uint16_t) out.nR : e;
                }
        }
        out.ck.resize(512);
        for (int b = 0; b < 512; b++) out.ck[b] = (uint8_t) (out.cls[b] | (out.kind_of_cls[out.cls[b]] << 6));
    }
    // kernel encoding of the fast tables
    {
        out.NKp = out.NK <= 1 ? 1 : out.NK <= 2 ? 2 : 4;
        const int ncols = 1 << out.fc_shift;
        out.wsh = out.fc_shift + (out.NKp == 1 ? 0 : out.NKp == 2 ? 1 : 2);
        const size_t W = (size_t) 1 << out.wsh;
        if ((size_t) X * out.NKp >= 4096 || out.NKp * ncols > 256) { err = "pattern too large for the GPU fast tables"; return false; }
        if ((((size_t) X * out.NKp + 1) << out.wsh) > (1u << 19)) { err = "pattern too large for the GPU fast tables (forward table over 2 MiB)"; return false; }
        out.col.resize(512);
        for (int b = 0; b < 512; b++) out.col[b] = (uint8_t) ((out.kind_of_cls[out.cls[b]] << out.fc_shift) | out.cls[b]);
        out.col_eot = (out.kind_edge << out.fc_shift) | out.ncls;
        auto enc = [&](uint32_t e, int nk) -> uint32_t {
            if (e == FC_DEAD) return FT_SPECIAL | (FT_DEAD << 28);
            if (e == FC_MULTI) return FT_SPECIAL | (FT_MULTI << 28);
            if ((e >> 24) == (FC_LOOK >> 24)) return FT_SPECIAL | (FT_LOOK << 28) | (e & 0xFFFFFF);
            uint32_t tg = e & 0xFFF, caps = e & 0xFFF000;
            if (tg == FC_TMATCH) return FT_SPECIAL | (FT_MATCH << 28) | caps;
            // a plain step: next row | capture writes (slot 0 = the dummy "no write" column)
            return (tg * (uint32_t) out.NKp + (uint32_t) nk) | caps;
        };
        // one more row than there are (core, previous kind) pairs: the absorbing row the forward walk
        // parks in after MATCH / a dead end (every column -> itself, no capture writes)
        out.ft.assign(((size_t) X * out.NKp + 1) * W, FT_SPECIAL | (FT_DEAD << 28));
        for (size_t c = 0; c < W; c++) out.ft[(size_t) X * out.NKp * W + c] = (uint32_t) X * (uint32_t) out.NKp;
        out.ft2.assign(out.fast2.size(), FT_SPECIAL | (FT_DEAD << 28));
        for (int x = 0; x < X; x++)
            for (int pk = 0; pk < out.NK; pk++)
                for (int nk = 0; nk < out.NK; nk++) {
                    size_t li = ((size_t) x * out.NK + pk) * out.NK + nk;
                    for (int c = 0; c <= out.ncls; c++) {
                        // a real byte class has exactly one kind; the end-of-text column has kind_edge
                        int colkind = c < out.ncls ? out.kind_of_cls[c] : out.kind_edge;
                        if (colkind != nk) continue;
                        uint32_t e = out.fastc[(li << out.fc_shift) + c];
                        size_t row = (size_t) x * out.NKp + pk, colc = ((size_t) nk << out.fc_shift) | (size_t) c;
                        out.ft[row * W + colc] = enc(e, nk);
                        if ((e >> 24) == (FC_LOOK >> 24) && e != FC_MULTI && e != FC_DEAD) {
                            size_t m = e & 0xFFFFFF;
                            for (int c2 = 0; c2 < ncols; c2++) out.ft2[m * ncols + c2] = enc(out.fast2[m * ncols + c2], nk);
                        }
                    }
                }
    }
    if (tb.budget > 4000000) { err = "pattern too complex (nested empty loops)"; return false; }
    if (tb.tag_seqs.size() > 0xFFFF) { err = "too many tag sequences"; return false; }
    out.tag_off.push_back(0);
    for (auto &t : tb.tag_seqs) {
        out.tag_data.insert(out.tag_data.end(), t.begin(), t.end());
        out.tag_off.push_back((uint32_t) out.tag_data.size());
    }
    out.has_capture = true;
    return true;
}

#include "rx_nfa.inc"

// The ascii table set of a pattern whose ASCII automata do not fit either: ONE class, one state, every byte poisons -- the kernels
// hand every value on to the second engine exactly as they hand on a value with a byte >= 0x80.  (An EMPTY value has no byte to
// poison the walk: d_final[0] = 2 and TableSet::stub tell the walkers; kdev.inc.)
void make_ascii_stub(TableSet &t, bool want_capture) {
    t = TableSet();
    t.ascii_only = true;
    t.stub = true;
    memset(t.cls, 0, sizeof(t.cls));
    for (int b = 0; b < 256; b++) { t.xl[b] = (uint8_t) b; t.xl[256 + b] = (uint8_t) b; }
    t.ncls = 1; t.high_cls = 0;
    t.nD = 1; t.d_init = 0;
    t.ddelta.assign(1, D_POISON);
    t.d_final.assign(1, 2);
    if (!want_capture) return;
    t.nR = 1; t.r_init = 0;
    t.rdelta.assign(1, R_POISON);
    t.r_info.assign(1, 0);
    t.P = 0; t.VW = 1;
    t.vmask.assign(1, 0);
    t.nX = 1; t.NK = 1; t.kind_edge = 0;
    t.kind_of_cls.assign(1, 0);
    t.list_off.assign(2, 0);
    t.fc_shift = 1; t.cls_shift = 0;
    t.fastc.assign(2, FC_DEAD);
    t.rdelta_p.assign(2, 1);
    t.ck.assign(512, 0);
    t.NKp = 1; t.wsh = 1;
    t.col.assign(512, 0);
    t.col_eot = 1;
    t.ft.assign(4, FT_SPECIAL | (FT_DEAD << 28));
    t.ft[2] = t.ft[3] = 1;                      // the absorbing row
    t.tag_off.assign(2, 0);
    t.has_capture = true;
}

}  // namespace

bool compile(const char *pattern, size_t len, unsigned options, bool want_captures, Program &out, std::string &err) {
    Syntax sx;
    sx.s = sx.p = (const unsigned char *) pattern;
    sx.e = sx.s + len;
    sx.has_named = scan_named(sx.s, sx.e);
    AstP root = sx.alternation(options & (OPT_IGNORECASE | OPT_EXTEND | OPT_MULTILINE), 0);
    if (!sx.failed() && !sx.eof()) sx.fail(*sx.p == ')' ? "unmatched close parenthesis" : "trailing garbage");
    if (sx.failed()) { err = sx.err; out = Program(); out.nonregular = sx.nonregular; return false; }
    // (the tables' capture programs keep a group set in 32 bits; the reference takes 32 767 groups, lib/onigmo/onigmo.h:441: the pattern goes
    // to the host's matcher, which has no such limit -- round 5)
    if (sx.ncap > 31) { err = "more than 31 capture groups are not supported on the GPU path"; out = Program(); out.nonregular = true; return false; }
    out = Program();
    {
        // the two corners where the reference's own answer depends on its search optimizer (DESIGN: deviations): which of them this
        // pattern can meet at all -- the walkers count the values that could (rx::corner), nothing is answered silently
        std::vector<const Ast *> st{root.get()};
        while (!st.empty()) {
            const Ast *a = st.back(); st.pop_back();
            if (a->t == Ast::ANCHOR) {
                if (a->anchor == A_BOL) out.corner_flags |= CF_NL_LOOKBACK;
                if (a->anchor == A_WORDB || a->anchor == A_NWORDB || a->anchor == A_WORDB_A || a->anchor == A_NWORDB_A) out.corner_flags |= CF_WORD_LOOKBACK;
            }
            if (a->t == Ast::SET && a->icase) out.corner_flags |= CF_ICASE_FOLD;
            for (auto &k : a->kids) st.push_back(k.get());
        }
    }
    const unsigned corner_flags = out.corner_flags;
    out.ngroups = sx.ncap;
    out.names = sx.names;
    out.name_groups = sx.name_groups;
    // caps row layout: one (begin, end) pair per (name, group) in onig_foreach_name order
    out.slot2cap.assign(2 * (size_t) (sx.ncap + 1), 0xFF);
    {
        int f = 0;
        for (size_t i = 0; i < sx.names.size(); i++)
            for (int g : sx.name_groups[i]) {
                if (f < 120) { out.slot2cap[2 * g] = (uint8_t) (2 * f); out.slot2cap[2 * g + 1] = (uint8_t) (2 * f + 1); }
                f++;
            }
    }
    // FLBGPU_RX_FORCE_NFA (tests): 1 = the NFA engine stands in for the utf8 table set even when that one builds; 2 = for both sets
    const char *force_env = std::getenv("FLBGPU_RX_FORCE_NFA");
    const int force = force_env ? atoi(force_env) : 0;
    std::string why;
    bool ascii_ok = force < 2, utf8_ok = force < 1;
    if (ascii_ok && !build_tables(root.get(), true, true, want_captures, out.slot2cap, out.ascii, why)) {
        // a parser only walks the capture program: the match-only DFA (forward subset construction) may be left out
        // when it alone is over the budget (stock parser `ambassador`); grep rules (no captures) still need it
        ascii_ok = false;
        if (want_captures && why == "match DFA exceeds the state budget") {
            std::string e2;
            out.ascii = TableSet();
            ascii_ok = build_tables(root.get(), true, false, true, out.slot2cap, out.ascii, e2);
            if (!ascii_ok) why = e2;
        }
    }
    // (the utf8 set only grows on the ascii one: not attempted when that one is over the budget)
    if (ascii_ok && utf8_ok && !build_tables(root.get(), false, false, true, out.slot2cap, out.utf8, why)) utf8_ok = false;
    (void) corner_flags;
    if (ascii_ok && utf8_ok) return true;
    // ---- the second engine (rx.hpp NfaSet)
    std::string e3;
    out.utf8 = TableSet();
    if (!build_nfa(root.get(), out.nfa, e3)) {
        err = why.empty() ? e3 : why + "; " + e3;
        return false;
    }
    out.utf8_nfa = true;
    out.why_nfa = why;
    if (!ascii_ok) { make_ascii_stub(out.ascii, want_captures); out.ascii_stub = true; }
    return true;
}

// ---------------------------------------------------------------- the optimizer-dependent corners: which texts can meet them
// (1) `^` / \b / \B looking back at a match start that sits right behind stray continuation bytes: the reference's matcher takes
//     the stray byte as the previous character, unless its forward search jumped to the start -- then onigenc_get_prev_char_head
//     left-adjusts over every 10xxxxxx byte (regexec.c:3559, regenc.c:107); which happens depends on the optimisation chosen.
//     `^` only sees a difference behind "\n" + continuation bytes; a word anchor behind any stray continuation byte.
// (2) (?i) and a text character whose case fold has another UTF-8 length (U+212A, U+017F, U+00DF, U+1E9E, U+FB00 .. U+FB06): the
//     engine bounds where a match may start by the byte length of the pattern's prefix; its answer there is not leftmost.
// The tables / the NFA engine give the leftmost-first answer in both; a value for which the reference MAY answer otherwise is
// counted (kdev.inc rx_corner, flbgpu_filter_regex_corners) -- never a silent difference.
bool corner(unsigned flags, const uint8_t *s, int len) {
    if (!flags) return false;
    bool hi = false;
    for (int i = 0; i < len && !hi; i++) hi = s[i] >= 0x80;
    if (!hi) return false;
    if (flags & CF_NL_LOOKBACK) for (int i = 0; i + 1 < len; i++) if (s[i] == '\n' && s[i + 1] >= 0x80 && s[i + 1] <= 0xbf) return true;
    if (flags & CF_WORD_LOOKBACK) {
        for (int i = 0; i < len;) {
            const int b = s[i];
            if (b >= 0xc2 && b <= 0xf4) { const int L = utf8_seq_len(s, i, len); i += (L > 1 && i + L <= len) ? L : 1; continue; }
            if (b >= 0x80 && b <= 0xbf) return true;
            i++;
        }
    }
    if (flags & CF_ICASE_FOLD) {
        for (int i = 0; i + 1 < len; i++) {
            const int b = s[i], c = s[i + 1], d = i + 2 < len ? s[i + 2] : -1;
            if ((b == 0xc5 && c == 0xbf) || (b == 0xc3 && c == 0x9f) || (b == 0xe2 && c == 0x84 && d == 0xaa) || (b == 0xe1 && c == 0xba && d == 0x9e) ||
                (b == 0xef && c == 0xac && d >= 0x80 && d <= 0x86)) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------- test aid: random texts drawn from a pattern
namespace {
struct Sampler {
    uint64_t st;
    uint32_t next() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t) (st >> 33); }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
    static void put_cp(std::string &o, uint32_t c) {
        if (c < 0x80) o.push_back((char) c);
        else if (c < 0x800) { o.push_back((char) (0xc0 | (c >> 6))); o.push_back((char) (0x80 | (c & 0x3f))); }
        else if (c < 0x10000) { o.push_back((char) (0xe0 | (c >> 12))); o.push_back((char) (0x80 | ((c >> 6) & 0x3f))); o.push_back((char) (0x80 | (c & 0x3f))); }
        else { o.push_back((char) (0xf0 | (c >> 18))); o.push_back((char) (0x80 | ((c >> 12) & 0x3f))); o.push_back((char) (0x80 | ((c >> 6) & 0x3f))); o.push_back((char) (0x80 | (c & 0x3f))); }
    }
    void set(const CC &cc, std::string &o) {
        // mostly a printable ASCII member; now and then a non-ASCII one, a control character, or any member at all
        const uint32_t r = below(16);
        if (r == 0) {
            CodeSet v = cc.valid_multibyte();
            if (!v.r.empty()) {
                const Range &x = v.r[below((uint32_t) v.r.size())];
                uint32_t c = x.first + below(std::min<uint32_t>(x.second - x.first + 1, 64));
                if (c >= 0xd800 && c <= 0xdfff) c = 0xe000;
                if (c <= MAXCP && v.has(c)) { put_cp(o, c); return; }
            }
        }
        int cand[128], n = 0;
        const int lo = r == 1 ? 0 : 32, hi = r == 1 ? 128 : 127;
        for (int b = lo; b < hi; b++) if (cc.ascii(b)) cand[n++] = b;
        if (!n) for (int b = 0; b < 128; b++) if (cc.ascii(b)) cand[n++] = b;
        if (n) { o.push_back((char) cand[below((uint32_t) n)]); return; }
        CodeSet v = cc.valid_multibyte();
        if (!v.r.empty()) { uint32_t c = v.r[0].first; if (c >= 0xd800 && c <= 0xdfff) c = 0xe000; put_cp(o, c); }
    }
    void walk(const Ast *a, std::string &o, int depth) {
        if (o.size() > 4000) return;
        switch (a->t) {
        case Ast::EMPTY: case Ast::ANCHOR: return;
        case Ast::SET: set(a->cc, o); return;
        case Ast::CAT: for (auto &k : a->kids) walk(k.get(), o, depth + 1); return;
        case Ast::ALT: walk(a->kids[below((uint32_t) a->kids.size())].get(), o, depth + 1); return;
        case Ast::GROUP: case Ast::ATOMIC: walk(a->kids[0].get(), o, depth + 1); return;
        case Ast::LOOK: case Ast::KEEP: return;
        case Ast::COND: walk(a->kids[below(2)].get(), o, depth + 1); return;
        case Ast::BACKREF: if (!o.empty() && below(2)) o.push_back(o[below((uint32_t) o.size())]); return;
        case Ast::REPEAT: {
            int span = a->max < 0 ? (below(4) == 0 ? 12 : 4) : std::min(a->max - a->min, 6);
            const int n = a->min + (int) below((uint32_t) span + 1);
            for (int i = 0; i < n; i++) walk(a->kids[0].get(), o, depth + 1);
            return;
        }
        }
    }
};
}  // namespace

bool sample(const char *pattern, size_t len, unsigned options, uint64_t seed, std::string &out, std::string &err) {
    Syntax sx;
    sx.ext = true;
    sx.s = sx.p = (const unsigned char *) pattern;
    sx.e = sx.s + len;
    sx.has_named = scan_named(sx.s, sx.e);
    AstP root = sx.alternation(options & (OPT_IGNORECASE | OPT_EXTEND | OPT_MULTILINE), 0);
    if (!sx.failed() && !sx.eof()) sx.fail("trailing garbage");
    if (sx.failed()) { err = sx.err; return false; }
    Sampler sm;
    sm.st = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    out.clear();
    sm.walk(root.get(), out, 0);
    return true;
}

// ---------------------------------------------------------------- host simulation of the tables
int utf8_seq_len(const uint8_t *s, int i, int len) {
    // well-formed UTF-8 (the transition table of lib/onigmo/enc/utf_8.c); a prefix-valid
    // sequence cut by the end of the text counts with its full length (NEEDMORE)
    int b0 = s[i], rem = len - i - 1;
    int need, lo1 = 0x80, hi1 = 0xbf;
    if (b0 >= 0xc2 && b0 <= 0xdf) need = 1;
    else if (b0 >= 0xe0 && b0 <= 0xef) { need = 2; if (b0 == 0xe0) lo1 = 0xa0; if (b0 == 0xed) hi1 = 0x9f; }
    else if (b0 >= 0xf0 && b0 <= 0xf4) { need = 3; if (b0 == 0xf0) lo1 = 0x90; if (b0 == 0xf4) hi1 = 0x8f; }
    else return 1;
    for (int k = 1; k <= need; k++) {
        if (k > rem) return need + 1;                 // truncated by end of text
        int b = s[i + k];
        if (k == 1) { if (b < lo1 || b > hi1) return 1; }
        else if (b < 0x80 || b > 0xbf) return 1;
    }
    return need + 1;
}

// the text the UTF-8 tables walk: its length (a prefix-valid sequence of two or more bytes cut by the end
// of the text is ONE symbol, the text ends right behind its lead) ...
int utf8_walk_len(const uint8_t *s, int len) {
    for (int k = 2; k <= 3 && k <= len; k++) {
        int j = len - k, b = s[j];
        if (b >= 0xc2 && b <= 0xf4 && utf8_seq_len(s, j, len) > k) return j + 1;
    }
    return len;
}
// ... and the symbol at position i (i < walk length; len = the real length)
// (word_variants: + 256 for a byte of a well-formed multi-byte character that is a word character)
static int utf8_word_half(const uint8_t *s, int lead, int L) {
    uint32_t cp = L == 2 ? (uint32_t) (s[lead] & 0x1f) : L == 3 ? (uint32_t) (s[lead] & 0x0f) : (uint32_t) (s[lead] & 0x07);
    for (int k = 1; k < L; k++) cp = (cp << 6) | (uint32_t) (s[lead + k] & 0x3f);
    return unicode_word(cp) ? 256 : 0;
}
int utf8_symbol(const uint8_t *xl, const uint8_t *s, int i, int len, int *seqlen, bool word_variants) {
    int b = s[i];
    if (seqlen) *seqlen = 1;
    if (b < 0x80) return b;
    if (b >= 0xc2 && b <= 0xf4) {
        int L = utf8_seq_len(s, i, len);
        if (L == 1) return xl[b];                          // the sequence breaks off: a character of its own
        if (i + L > len) return len - i >= 2 ? xl[256 + b] : xl[b];      // cut by the end of the text
        if (seqlen) *seqlen = L;
        return b + (word_variants ? utf8_word_half(s, i, L) : 0);
    }
    if (b <= 0xbf) {
        // a continuation byte belongs to the sequence whose lead is the nearest non-continuation byte
        // to its left, when that sequence is well-formed and reaches this far
        for (int d = 1; d <= 3 && d <= i; d++) {
            int c = s[i - d];
            if (c >= 0x80 && c <= 0xbf) continue;
            if (c >= 0xc2 && c <= 0xf4) {
                int L = utf8_seq_len(s, i - d, len);
                if (L > d && i - d + L <= len) return b + (word_variants ? utf8_word_half(s, i - d, L) : 0);
            }
            break;
        }
    }
    return xl[b];
}

namespace {

// returns 1 match, 0 no match, -3 poisoned (needs the utf8 tables).  Mirrors the kernels:
// reverse pass keeps one state id every CHK boundaries (counted from the end of the text); the
// forward walk uses the byte-resolved fast table and only rebuilds the reverse state of a
// boundary (from the nearest checkpoint to its right) when several candidates remain.
constexpr int CHK = 16;
thread_local long g_stat_fast = 0, g_stat_look = 0, g_stat_multi = 0;

int run_capture(const TableSet &t, int ngroups, const uint8_t *s, int olen, int *beg, int *end) {
    // the utf8 set walks symbols over a possibly shortened text (see utf8_walk_len)
    const int len = t.ascii_only ? olen : utf8_walk_len(s, olen);
    auto sym = [&](int i, int *L) -> int { if (t.ascii_only) { if (L) *L = 1; return s[i]; } return utf8_symbol(t.xl, s, i, olen, L, t.word_variants); };
    // the wide layout mirrors the kernels too: 32-bit entries, a checkpoint every 32 boundaries kept as two
    // 16-bit halves in the slots the narrow layout would use for boundaries 32k and 32k + 16
    const int CHKW = t.wide ? 2 * CHK : CHK;
    std::vector<uint16_t> chk(len / CHK + 2);
    auto chk_put = [&](int tt, int R) {
        if (!t.wide) chk[tt / CHK] = (uint16_t) R;
        else { chk[(tt / CHKW) * 2] = (uint16_t) (R & 0xFFFF); chk[(tt / CHKW) * 2 + 1] = (uint16_t) ((uint32_t) R >> 16); }
    };
    auto chk_get = [&](int tt) -> int {
        return !t.wide ? (int) chk[tt / CHK] : (int) ((uint32_t) chk[(tt / CHKW) * 2] | ((uint32_t) chk[(tt / CHKW) * 2 + 1] << 16));
    };
    int R = t.r_init, best = -1;
    int h1 = -1, h2 = -1;                     // best as it was 1/2 boundaries to the right
    chk_put(0, R);
    for (int i = len - 1; i >= 0; i--) {
        int L = 1;
        bool st;
        if (t.wide) {
            uint32_t e = t.rdelta32[(size_t) R * t.ncls + t.cls[sym(i, &L)]];
            st = (e >> 31) != 0; R = (int) (e & 0x7FFFFFFFu);
        }
        else {
            uint16_t e = t.rdelta[(size_t) R * t.ncls + t.cls[sym(i, &L)]];
            if ((e & 0x7FFF) == R_POISON) return -3;
            st = (e & 0x8000) != 0; R = e & 0x7FFF;
        }
        int before = best;
        if (st) best = i + 1;
        int tt = len - i;                     // distance of boundary i from the end
        if (tt % CHKW == 0) chk_put(tt, R);
        if (!t.ascii_only) {
            // a match may only start on a character boundary (onig_search advances by enclen):
            // boundaries strictly inside the well-formed sequence that starts here are not start candidates
            if (L == 2) best = before;
            else if (L == 3) best = h1;
            else if (L == 4) best = h2;
            h2 = h1; h1 = before;
        }
    }
    if (t.r_info[R] & 0x80) best = 0;
    if (best < 0) return 0;
    std::vector<int> slot(2 * (ngroups + 1), -1);
    slot[0] = best;
    int j = best;
    int pk0 = j == 0 ? t.kind_edge : t.kind_of_cls[t.cls[sym(j - 1, nullptr)]];
    uint32_t S = (uint32_t) ((t.nX - 1) * t.NKp + pk0);
    const int ncols = 1 << t.fc_shift;
    for (;;) {
        int colc = j < len ? t.col[sym(j, nullptr)] : t.col_eot;
        uint32_t e = t.ft[((size_t) S << t.wsh) + colc];
        if ((e & FT_SPECIAL) && ft_type(e) == FT_LOOK) {
            g_stat_look++;
            int c2 = j + 1 < len ? t.cls[sym(j + 1, nullptr)] : t.ncls;
            e = t.ft2[(size_t) (e & 0xFFFFFF) * ncols + c2];
        }
        int x = (int) (S / t.NKp), pk = (int) (S % t.NKp), nk = colc >> t.fc_shift;
        size_t li = ((size_t) x * t.NK + pk) * t.NK + nk;
        uint32_t ty = (e & FT_SPECIAL) ? ft_type(e) : 0;
        if (ty == FT_DEAD) return -2;            // table inconsistency (must never happen)
        uint32_t pick = 0xFFFFFFFFu;
        if (ty != FT_MULTI) {
            g_stat_fast++;
            // the kernel applies the inline capture writes of named groups; the host simulation
            // also reports unnamed groups, so it looks the tag sequence up in the candidate list
            bool is_match = ty == FT_MATCH;
            uint32_t tgt = is_match ? F_MATCH : ((e & 0xFFF) / (uint32_t) t.NKp);
            if (!is_match && (int) ((e & 0xFFF) % (uint32_t) t.NKp) != nk) return -2;
            for (uint32_t k = t.list_off[li]; k < t.list_off[li + 1]; k++)
                if ((t.list_ent[k] & 0xFFFF) == tgt) { pick = t.list_ent[k]; break; }
            if (pick == 0xFFFFFFFFu) return -2;
            // (the packed capture writes of named groups are exercised by the GPU parity tests)
        }
        else {
            g_stat_multi++;
            int t0 = ((len - j) / CHKW) * CHKW, b0 = len - t0;
            int r = chk_get(t0);
            for (int i = b0 - 1; i >= j; i--)                                   // never poisoned here
                r = t.wide ? (int) (t.rdelta32_p[((size_t) r << t.cls_shift) + t.cls[sym(i, nullptr)]] & 0x7FFFFFFFu)
                           : (int) (t.rdelta_p[((size_t) r << t.cls_shift) + t.cls[sym(i, nullptr)]] & 0x7FFF);
            for (uint32_t k = t.list_off[li]; k < t.list_off[li + 1]; k++) {
                uint32_t ent = t.list_ent[k];
                uint32_t tg = ent & 0xFFFF;
                if (tg == F_MATCH || ((t.vmask[(size_t) r * t.VW + (tg >> 5)] >> (tg & 31)) & 1)) { pick = ent; break; }
            }
            if (pick == 0xFFFFFFFFu) return -2;
        }
        uint32_t ts = pick >> 16;
        for (uint32_t k = t.tag_off[ts]; k < t.tag_off[ts + 1]; k++) slot[t.tag_data[k]] = j;
        if ((pick & 0xFFFF) == F_MATCH) break;
        S = (pick & 0xFFFF) * (uint32_t) t.NKp + (uint32_t) nk;
        j++;
        if (j > len) return -2;
    }
    for (int g = 0; g <= ngroups; g++) {
        if (slot[2 * g] >= 0 && slot[2 * g + 1] >= 0) {
            // the boundary behind a truncated last character is the end of the real text
            beg[g] = slot[2 * g] == len ? olen : slot[2 * g];
            end[g] = slot[2 * g + 1] == len ? olen : slot[2 * g + 1];
        }
        else { beg[g] = -1; end[g] = -1; }
    }
    return 1;
}

}  // namespace

void debug_stats(long *out) { out[0] = g_stat_fast; out[1] = g_stat_look; out[2] = g_stat_multi; g_stat_fast = g_stat_look = g_stat_multi = 0; }

// ---------------------------------------------------------------- host execution of the NFA set (what kdev.inc nfa_* do)
namespace {

struct NfaCh { int cls, L; };

int nfa_mb_class(const NfaSet &t, uint32_t cp) {
    size_t lo = 0, hi = t.mb_lo.size();               // the last interval that starts at or below cp
    while (hi - lo > 1) {
        const size_t m = (lo + hi) / 2;
        if (t.mb_lo[m] <= cp) lo = m; else hi = m;
    }
    return t.mb_cls[lo];
}
uint32_t nfa_decode(const uint8_t *s, int i, int L) {
    uint32_t cp = L == 2 ? (uint32_t) (s[i] & 0x1f) : L == 3 ? (uint32_t) (s[i] & 0x0f) : (uint32_t) (s[i] & 0x07);
    for (int k = 1; k < L; k++) cp = (cp << 6) | (uint32_t) (s[i + k] & 0x3f);
    return cp;
}
// the character that starts at byte i of the walked text (olen: the real length; the walked text ends behind the lead of a cut sequence)
NfaCh nfa_char_at(const NfaSet &t, const uint8_t *s, int i, int olen) {
    const int b = s[i];
    if (b < 0x80) return {t.cls_byte[b], 1};
    if (b >= 0xc2 && b <= 0xf4) {
        const int L = utf8_seq_len(s, i, olen);
        if (L > 1) {
            if (i + L > olen) return {olen - i >= 2 ? t.cls_byte[256 + b] : t.cls_byte[b], 1};
            return {nfa_mb_class(t, nfa_decode(s, i, L)), L};
        }
    }
    return {t.cls_byte[b], 1};
}
// the character that ends at boundary e of the walked text [0, wlen)
NfaCh nfa_char_before(const NfaSet &t, const uint8_t *s, int e, int wlen, int olen) {
    const int i = e - 1, b = s[i];
    if (b < 0x80) return {t.cls_byte[b], 1};
    if (b <= 0xbf) {
        for (int d = 1; d <= 3 && d <= i; d++) {
            const int c = s[i - d];
            if (c >= 0x80 && c <= 0xbf) continue;
            if (c >= 0xc2 && c <= 0xf4) {
                const int L = utf8_seq_len(s, i - d, olen);
                if (L == d + 1 && i - d + L <= olen) return {nfa_mb_class(t, nfa_decode(s, i - d, L)), L};
            }
            break;
        }
        return {t.cls_byte[b], 1};
    }
    if (i == wlen - 1 && wlen < olen) return {t.cls_byte[256 + b], 1};      // the lead of the sequence the end of the text cuts
    return {t.cls_byte[b], 1};
}

// one step of the reverse walk: the state (V, nk) of boundary e becomes the one of the boundary in front of the character `ch`;
// *start = a match may start at boundary e (with ch as the character in front of it)
void nfa_rev_step(const NfaSet &t, const NfaCh &ch, uint32_t *V, int &nk, bool *start) {
    const int VW = t.VW, P = t.P, k = t.ckind[ch.cls];
    const size_t kk = (size_t) k * t.NK + nk;
    const uint32_t *rows = t.pred.data() + kk * (size_t) (P + 2) * VW;
    uint32_t N[NFA_MAXP / 32];
    bool st = t.mstart[kk] != 0;
    for (int w = 0; w < VW; w++) { N[w] = rows[(size_t) P * VW + w]; if (rows[(size_t) (P + 1) * VW + w] & V[w]) st = true; }
    for (int w = 0; w < VW; w++)
        for (uint32_t m = V[w]; m; m &= m - 1) {
            const uint32_t *r = rows + (size_t) (32 * w + __builtin_ctz(m)) * VW;
            for (int u = 0; u < VW; u++) N[u] |= r[u];
        }
    for (int w = 0; w < VW; w++) V[w] = N[w] & t.amask[(size_t) ch.cls * VW + w];
    nk = k;
    if (start) *start = st;
}

}  // namespace

int nfa_run(const NfaSet &t, int ngroups, const uint8_t *s, int olen, int *beg, int *end) {
    const int VW = t.VW, NK = t.NK, P = t.P;
    const int wlen = utf8_walk_len(s, olen);
    // ---- reverse pass: leftmost viable start; the state of the first boundary of every block of NFA_CHK boundaries (from the end) is kept
    const int CW = VW + 1;
    std::vector<uint32_t> chk((size_t) (wlen / NFA_CHK + 2) * CW, 0u);
    uint32_t V[NFA_MAXP / 32];
    for (int w = 0; w < VW; w++) V[w] = 0;
    int nk = t.kind_edge, best = -1, e = wlen, last_blk = -1;
    for (;;) {
        const int tt = wlen - e, blk = tt / NFA_CHK;
        if (blk != last_blk) {
            for (int w = 0; w < VW; w++) chk[(size_t) blk * CW + w] = V[w];
            chk[(size_t) blk * CW + VW] = (uint32_t) nk | ((uint32_t) (tt % NFA_CHK) << 8);
            last_blk = blk;
        }
        if (e == 0) break;
        const NfaCh ch = nfa_char_before(t, s, e, wlen, olen);
        bool st;
        nfa_rev_step(t, ch, V, nk, &st);
        if (st) best = e;
        e -= ch.L;
    }
    {
        // a start at boundary 0: the text edge in front of it
        const size_t kk = (size_t) t.kind_edge * NK + nk;
        const uint32_t *rows = t.pred.data() + kk * (size_t) (P + 2) * VW;
        bool st = t.mstart[kk] != 0;
        for (int w = 0; w < VW; w++) if (rows[(size_t) (P + 1) * VW + w] & V[w]) st = true;
        if (st) best = 0;
    }
    if (best < 0) return 0;
    if (!beg) return 1;
    // ---- forward pass: the first candidate that is viable
    std::vector<int> slot(2 * (ngroups + 1), -1);
    slot[0] = best;
    int j = best, x = P;
    int pk = j == 0 ? t.kind_edge : t.ckind[nfa_char_before(t, s, j, wlen, olen).cls];
    for (;;) {
        NfaCh ch = {0, 1};
        int nk2 = t.kind_edge;
        if (j < wlen) { ch = nfa_char_at(t, s, j, olen); nk2 = t.ckind[ch.cls]; }
        const size_t li = ((size_t) x * NK + pk) * NK + nk2;
        const uint32_t *am = t.amask.data() + (size_t) ch.cls * VW;
        uint32_t pick = 0xFFFFFFFFu;
        int cnt = 0;
        for (uint32_t q = t.list_off[li]; q < t.list_off[li + 1]; q++) {
            const uint32_t ent = t.list_ent[q], tg = ent & 0xFFFF;
            if (tg == NFA_MATCH || (j < wlen && ((am[tg >> 5] >> (tg & 31)) & 1))) { if (!cnt) pick = ent; cnt++; }
            if (tg == NFA_MATCH) break;
        }
        if (cnt == 0) return -2;
        if (cnt > 1) {
            // several candidates accept the character: the set V of this boundary decides (replayed from its block's checkpoint)
            const int blk = (wlen - j) / NFA_CHK;
            uint32_t Vj[NFA_MAXP / 32];
            for (int w = 0; w < VW; w++) Vj[w] = chk[(size_t) blk * CW + w];
            const uint32_t info = chk[(size_t) blk * CW + VW];
            int nkc = (int) (info & 0xFF), ec = wlen - (blk * NFA_CHK + (int) (info >> 8));
            while (ec > j) {
                const NfaCh c2 = nfa_char_before(t, s, ec, wlen, olen);
                nfa_rev_step(t, c2, Vj, nkc, nullptr);
                ec -= c2.L;
            }
            if (ec != j) return -2;
            pick = 0xFFFFFFFFu;
            for (uint32_t q = t.list_off[li]; q < t.list_off[li + 1]; q++) {
                const uint32_t ent = t.list_ent[q], tg = ent & 0xFFFF;
                if (tg == NFA_MATCH || ((Vj[tg >> 5] >> (tg & 31)) & 1)) { pick = ent; break; }
            }
            if (pick == 0xFFFFFFFFu) return -2;
        }
        const uint32_t ts = pick >> 16;
        for (uint32_t q = t.tag_off[ts]; q < t.tag_off[ts + 1]; q++) slot[t.tag_data[q]] = j;
        if ((pick & 0xFFFF) == NFA_MATCH) break;
        x = (int) (pick & 0xFFFF);
        pk = nk2;
        j += ch.L;
        if (j > wlen) return -2;
    }
    for (int g = 0; g <= ngroups; g++) {
        if (slot[2 * g] >= 0 && slot[2 * g + 1] >= 0) {
            beg[g] = slot[2 * g] == wlen ? olen : slot[2 * g];
            end[g] = slot[2 * g + 1] == wlen ? olen : slot[2 * g + 1];
        }
        else { beg[g] = -1; end[g] = -1; }
    }
    return 1;
}

#include "rxbt.inc"

int simulate_match(const Program &p, const uint8_t *s, int len) {
    const TableSet &t = p.ascii;
    if (p.ascii_stub) return nfa_run(p.nfa, p.ngroups, s, len, nullptr, nullptr);
    if (t.nD == 0) {                          // no match-only DFA (see compile): the capture program answers
        std::vector<int> b(p.ngroups + 1), e(p.ngroups + 1);
        int r = simulate_capture(p, s, len, b.data(), e.data());
        return r >= 0 ? 1 : r == -1 ? 0 : r;
    }
    int st = t.d_init;
    for (int i = 0; i < len; i++) {
        uint16_t n = t.ddelta[(size_t) st * t.ncls + t.cls[s[i]]];
        if (n == D_ACCEPT) return 1;
        if (n == D_POISON) {
            if (p.utf8_nfa) return nfa_run(p.nfa, p.ngroups, s, len, nullptr, nullptr);
            std::vector<int> b(p.ngroups + 1), e(p.ngroups + 1);
            int r = run_capture(p.utf8, p.ngroups, s, len, b.data(), e.data());
            return r < 0 ? r : r;
        }
        st = n;
    }
    return t.d_final[st];
}

int simulate_capture(const Program &p, const uint8_t *s, int len, int *beg, int *end) {
    int r = (p.ascii.has_capture && !p.ascii_stub) ? run_capture(p.ascii, p.ngroups, s, len, beg, end) : -3;
    if (r == -3) r = p.utf8_nfa ? nfa_run(p.nfa, p.ngroups, s, len, beg, end) : run_capture(p.utf8, p.ngroups, s, len, beg, end);
    if (r == 1) return p.ngroups + 1;
    if (r == 0) return -1;
    return r;
}

}  // namespace rx
