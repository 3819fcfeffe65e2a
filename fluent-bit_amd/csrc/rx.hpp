// rx.hpp -- host-side regex compiler for the MI355X filter path.
//
// Turns an Onigmo/Ruby-syntax pattern (the dialect src/flb_regex.c:116-152 hands to
// onig_new with ONIG_ENCODING_UTF8 / ONIG_SYNTAX_RUBY) into flat tables the HIP kernels step
// one input byte per lane:
//
//   * byte classes            cls[256]
//   * match-only forward DFA  ddelta[nD][ncls]            (filter_grep / log_to_metrics gate:
//                                                          flb_regex_match, src/flb_regex.c:270)
//   * capture program         rdelta[nR][ncls]  reverse DFA over "viable position" bit sets
//                             vmask[nR][VW]     the bit sets themselves (the bit-NFA state)
//                             list_ent[]        per (core, context) priority-ordered targets
//                             tag sequences     group open/close slots crossed per step
//                                                         (flb_regex_do + flb_regex_parse,
//                                                          src/flb_regex.c:182-231,294-313)
//
// The capture program reproduces backtracking (leftmost-first) semantics exactly without a
// stack: pass 1 runs the reversed automaton from the end of the value and records, per byte
// boundary, which NFA positions can still reach a match; pass 2 walks forward from the leftmost
// viable start always taking the highest-priority transition whose target is viable, so the
// first choice is the one a backtracking engine would have settled on.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rx {

constexpr unsigned OPT_IGNORECASE = 1u;   // ONIG_OPTION_IGNORECASE
constexpr unsigned OPT_EXTEND = 2u;       // ONIG_OPTION_EXTEND
constexpr unsigned OPT_MULTILINE = 4u;    // ONIG_OPTION_MULTILINE ('.' matches \n)

constexpr uint16_t D_ACCEPT = 0xFFFF;     // match-only DFA: sticky accept

constexpr uint16_t D_POISON = 0xFFFE;     // ASCII tables: a byte >= 0x80 was seen -> UTF-8 tables
constexpr uint16_t R_POISON = 0x7FFF;     // same, reverse automaton (low 15 bits of the entry)
constexpr uint16_t F_MATCH = 0xFFFF;      // forward list entry: low half == MATCH reached

// One set of tables.  Two sets are built per pattern: `ascii` assumes every input byte is
// < 0x80 (any other byte maps to the HIGH class whose transitions are POISON; the kernels then
// re-run that record on the `utf8` set), `utf8` expands every character class to well-formed
// UTF-8 byte sequences (+ the never-valid lead bytes as 1-byte characters, as Onigmo's
// mbc_enc_len does).  Logs are overwhelmingly ASCII, and the ASCII automata are 3-6x smaller,
// small enough to be staged in LDS.
struct TableSet {
    bool ascii_only = false;
    bool stub = false;                        // ascii set that hands EVERY value on (rx.cpp make_ascii_stub: Program::ascii_stub)
    // [symbol] -> class.  Symbols 0 .. 255 are bytes (utf8 set: after the SymbolMap translation); the utf8 set of a pattern
    // with \b / \B has a second half, 256 + b: byte b of a well-formed multi-byte character that IS a word character
    // (the reference's \b is Unicode-aware, ONIG_OPTION_WORD_BOUND_ALL_RANGE: every byte of such a character carries the
    // character's kind, so a class comes in a word and a non-word variant and the walkers pick the column by decoding the
    // character -- `word_variants`, rx.cpp utf8_symbol)
    uint8_t cls[512];
    bool word_variants = false;
    int ncls = 0;
    int high_cls = -1;
    // utf8 set only: symbol of a byte >= 0x80 that is a character of its own ([b]) / of the lead of a
    // sequence cut by the end of the text ([256 + b]); see SymbolMap in rx.cpp
    uint8_t xl[512];

    // match-only forward DFA (built for the ascii set only)
    int nD = 0, d_init = 0;
    std::vector<uint16_t> ddelta;             // [nD][ncls]
    std::vector<uint8_t> d_final;             // [nD] (a stub: 2 = hand the value on, dev.hpp DevDfa)
    std::vector<uint8_t> d_live;              // [nD] a match can still follow on a one-line text of ANY characters (rx.cpp; ml.cpp build_product)

    // capture program
    bool has_capture = false;
    int nR = 0, r_init = 0;
    std::vector<uint16_t> rdelta;             // [nR][ncls]; bit15 = a match may START at the
                                              // boundary right of the consumed byte
    std::vector<uint8_t> r_info;              // [nR] bits0-2: kind of the byte right of the
                                              // boundary (ctx next-kind), bit7: start viable at
                                              // boundary 0 (beginning of text)
    int P = 0, VW = 0;
    std::vector<uint32_t> vmask;              // [nR][VW] viable-position bit sets
    int nX = 0;                               // cores: positions 0..P-1, START = P
    int NK = 1;                               // number of context kinds in use
    int kind_edge = 0;                        // kind index of beginning/end of text
    std::vector<uint8_t> kind_of_cls;         // [ncls]
    std::vector<uint32_t> list_off;           // [(x*NK + pk)*NK + nk] -> range in list_ent
    std::vector<uint32_t> list_ent;           // target core | tagseq << 16, priority order
    std::vector<uint32_t> tag_off;            // [ntagseq+1]
    std::vector<uint8_t> tag_data;            // capture slots (2*g, 2*g+1)
    // byte-resolved fast path of the forward walk: fastc[list][c] for byte class c (column ncls =
    // end of text) is the ONLY candidate of that list whose byte set contains c (or MATCH), so
    // no viability test is needed; FC_MULTI = several candidates remain (resolve with vmask),
    // FC_DEAD = none.
    std::vector<uint32_t> fastc;              // [nX*NK*NK][1 << fc_shift], column ncls = end of text
    // Kernel-friendly encodings of the above (what is actually uploaded):
    //   rdelta_p  rows padded to 1 << cls_shift entries, so the row offset is a shift; row nR is
    //             the absorbing POISON state (replaces R_POISON)
    //   ck[256]   byte -> class | kind << 6          (one lookup instead of two; ncls <= 64)
    //   fastc     entry = target (12 bit, FC_TMATCH = match) | (capA+1) << 12 | (capB+1) << 18:
    //             up to two capture-span writes of NAMED groups are carried inline, so the walk
    //             needs no tag table; entries whose tag sequence does not fit are FC_MULTI
    //   fast2     cells that stay ambiguous after one byte are usually decided by the NEXT byte
    //             (delimiter-driven log patterns: "\S*\"" vs the closing quote): such a cell
    //             holds FC_LOOK | m and fast2[m][class of byte j+1] is consulted (same encoding)
    int cls_shift = 0, fc_shift = 0;
    std::vector<uint16_t> rdelta_p;           // [nR + 1][1 << cls_shift]
    // WIDE reverse automaton (utf8 set only: the ASCII set is what the hot kernels step and stays narrow).
    // A capture automaton of more than 0x7FF0 states (stock parsers `envoy`, `ambassador`: 33 k / 96 k states
    // once every class is spelled out in UTF-8 sequences) keeps its transitions as 32-bit entries, bit 31 = a
    // match may start here, no POISON row in use.  rdelta / rdelta_p are empty then; the generic kernels walk
    // rdelta32_p straight from HBM and keep a checkpoint every 32 boundaries in two 16-bit halves.
    bool wide = false;
    std::vector<uint32_t> rdelta32;           // [nR][ncls]
    std::vector<uint32_t> rdelta32_p;         // [nR + 1][1 << cls_shift]
    std::vector<uint8_t> ck;                  // [256]
    std::vector<uint32_t> fast2;              // [nmulti][1 << fc_shift]
    // ---- what the kernels step through (derived from fastc/fast2 by encode_kernel_tables):
    //   col[256]  byte -> column code  kind << fc_shift | class        (one lookup per byte)
    //   ft        [nX * NKp][NKp << fc_shift]  row S = core * NKp + prev kind; a plain entry IS the
    //             next row number (bits 0-11) | (capA+1) << 12 | (capB+1) << 18  (capture span
    //             writes of named groups, 0 = none); entries with bit 31 are special, type in
    //             bits 28-30:
    //               2 MATCH (+ capture writes)   3 look at next byte: ft2[m][its class]
    //               4 several candidates (resolve with vmask)   5 dead
    //   ft2       [nmulti][1 << fc_shift], same encoding
    int NKp = 1, wsh = 0;                     // NK rounded up to a power of two; row width shift
    int col_eot = 0;                          // column code of "end of text"
    std::vector<uint8_t> col;                 // [256]
    std::vector<uint32_t> ft, ft2;
};

constexpr uint32_t FT_SPECIAL = 0x80000000u;
constexpr uint32_t FT_CAPS = 1, FT_MATCH = 2, FT_LOOK = 3, FT_MULTI = 4, FT_DEAD = 5;
inline uint32_t ft_type(uint32_t e) { return (e >> 28) & 7; }

constexpr uint32_t FC_MULTI = 0xFFFFFFFEu;
constexpr uint32_t FC_DEAD = 0xFFFFFFFFu;
constexpr uint32_t FC_TMATCH = 0xFFFu;        // target field value meaning MATCH
constexpr uint32_t FC_LOOK = 0xFE000000u;     // | m: resolve with fast2[m][next byte class]

// ---- the second engine: a bit-parallel walk over the CHARACTER-level position automaton (rx_nfa.inc).
// Where the table compiler gives up -- a reverse automaton over the state budget (stock parser `istio-envoy-proxy`), classes whose
// UTF-8 spelling needs more than 63 byte classes or thousands of byte positions ([[:alpha:]], [[:alnum:]] ...: `http_statement`),
// bounded repeats like .{0,300} -- the same two passes run on position SETS instead of state ids: the reverse pass keeps the set V
// of positions from which the rest of the text still matches as VW 32-bit words and steps it per CHARACTER
//     V' = accept(c) & (pred[k][nk][MATCH] | OR over q in V of pred[k][nk][q])
// (k, nk: the context kinds of the consumed character and of the one right of it), the forward pass takes the first candidate of
// list(core, pk, nk) that is in V.  One position per character node of the pattern whatever the character's encoded length: a
// non-ASCII character is decoded and classified by a search in the merged range table (mb_lo / mb_cls), so [[:alpha:]] costs one
// position and one range search, not 730 ranges spelled out in UTF-8 bytes.  Ill-formed input follows the engine exactly like the
// byte tables do (a stray byte / the lead of a sequence cut by the end of the text are characters of their own: cls_byte).
constexpr int NFA_MAXP = 320;                 // positions (ten 32-bit words per set)
constexpr int NFA_CHK = 8;                    // the reverse walk keeps its state once per block of 8 byte boundaries (counted from the end)
constexpr uint32_t NFA_MATCH = 0xFFFFu;       // list entry target: MATCH
struct NfaSet {
    bool ok = false;
    int P = 0, VW = 0;                        // positions, words per set
    int NK = 1, kind_edge = 0;                // context kinds told apart (compact indices), the one of "no character" (text edge)
    int ncls = 0;                             // character classes: (accepting positions, kind) signatures
    std::vector<uint16_t> cls_byte;           // [512]: [b] an ASCII byte or a byte >= 0x80 that is a character of its own;
                                              //        [256 + b] the lead b of a prefix-valid sequence cut by the end of the text
    std::vector<uint32_t> mb_lo;              // first code points of the intervals that partition [0x80, 0x10FFFF], ascending
    std::vector<uint16_t> mb_cls;             // their classes
    std::vector<uint32_t> amask;              // [ncls][VW] positions that accept a character of the class
    std::vector<uint8_t> ckind;               // [ncls] its context kind
    // per context kk = k * NK + nk: rows q < P = the positions whose continuation reaches q, row P = those that reach MATCH,
    // row P + 1 = what the START core reaches (the forward view of one row, for "may a match start here")
    std::vector<uint32_t> pred;               // [NK * NK][P + 2][VW]
    std::vector<uint8_t> mstart;              // [NK * NK] START reaches MATCH
    std::vector<uint32_t> list_off;           // [((P + 1) * NK + pk) * NK + nk .. + 1] core P = START
    std::vector<uint32_t> list_ent;           // target position (NFA_MATCH) | tag sequence << 16, priority order
    std::vector<uint32_t> tag_off;
    std::vector<uint8_t> tag_data;
};

struct Program {
    int ngroups = 0;                          // capture groups excluding group 0
    std::vector<std::string> names;           // first-appearance order (onig_foreach_name)
    std::vector<std::vector<int>> name_groups;
    std::vector<uint8_t> slot2cap;            // capture slot (2g, 2g+1) -> index in the caps row of
                                              // the named fields (0xFF: not a named group's slot)
    TableSet ascii;                           // match DFA (+ capture program when requested)
    TableSet utf8;                            // capture program (also answers match-only)
    // the NFA engine stands in for a table set the compiler could not build:
    bool utf8_nfa = false;                    // values with a byte >= 0x80 (everything the ascii set hands on) walk `nfa`; utf8 is empty
    bool ascii_stub = false;                  // the ascii set is a stub that hands EVERY value on (always-poison tables)
    NfaSet nfa;
    std::string why_nfa;                      // what the table compiler said when it gave up (diagnostics)
    bool nonregular = false;                  // compile() failed on a construct only a backtracking matcher runs (bt_compile takes the pattern)
    unsigned corner_flags = 0;                // CF_*: the optimizer-dependent corners of the reference this pattern can meet (rx::corner)
};
constexpr unsigned CF_NL_LOOKBACK = 1u, CF_WORD_LOOKBACK = 2u, CF_ICASE_FOLD = 4u;
// a text for which the reference's answer MAY differ from the leftmost-first one (rx.cpp): counted by the walkers, never silent
bool corner(unsigned flags, const uint8_t *s, int len);

// Compiles `pattern` (already stripped of the /../flags wrapper).  want_captures=false skips the
// capture program (grep rules).  Returns false and fills err when the pattern uses a construct
// the tables cannot express (back-references, look-around, atomic groups, ...) or exceeds the
// table budget.  The caller must then FAIL LOUDLY: there is no CPU fallback in the product.
bool compile(const char *pattern, size_t len, unsigned options, bool want_captures, Program &out,
             std::string &err);

// src/flb_regex.c:60-152: "/pat/imx" option syntax.  Returns the inner pattern range and options.
void split_flb_pattern(const char *pattern, const char **start, const char **end, unsigned *options);

// Host execution of the SAME tables the kernels use (debug/self-test aid, exercised by the CPU
// unit tests; never called by the filters).
//   returns -1 on mismatch else number of registers; beg/end sized ngroups+1.
//   Both try the ascii set first and re-run on the utf8 set when a byte >= 0x80 poisons it,
//   exactly as the kernels do.
int simulate_capture(const Program &p, const uint8_t *s, int len, int *beg, int *end);
int simulate_match(const Program &p, const uint8_t *s, int len);
// host execution of the NFA set alone (the algorithm the kernels run: kdev.inc nfa_*): 1 match (spans filled when beg != nullptr),
// 0 no match, -2 inconsistency
// ---- the backtracking matcher for what is not a regular expression (rxbt.inc): look-around, atomic groups, possessive repeats,
// back-references, \Z \G \K.  Host only; the filters give it the values of the rules / parsers whose compile() failed with
// Program::nonregular set.  bt_search: groups + 1 on a match (beg / end may be NULL), -1 no match, -4 the backtrack budget was spent.
struct BtProgram;
BtProgram *bt_compile(const char *pattern, size_t len, unsigned options, std::string &err);
void bt_free(BtProgram *p);
int bt_ngroups(const BtProgram *p);
const std::vector<std::string> &bt_names(const BtProgram *p);
const std::vector<std::vector<int>> &bt_name_groups(const BtProgram *p);
int bt_search(const BtProgram *p, const uint8_t *s, int len, int *beg, int *end);

int nfa_run(const NfaSet &t, int ngroups, const uint8_t *s, int len, int *beg, int *end);
// UTF-8 sequence length rule shared with the kernels: length (2..4) of the well-formed or
// end-truncated sequence starting at s[i], else 1
int utf8_seq_len(const uint8_t *s, int i, int len);
// what the utf8 table set steps on: the length of the walked text and the symbol at position i
int utf8_walk_len(const uint8_t *s, int len);
int utf8_symbol(const uint8_t *xl, const uint8_t *s, int i, int len, int *seqlen, bool word_variants = false);
// Unicode word character (ONIGENC_CTYPE_WORD of the reference's UTF-8 encoding: posix_ranges.inc, probed from the real engine)
bool unicode_word(uint32_t cp);
// the ranges behind unicode_word for code points >= 0x80: {lo, hi} pairs, ascending (uploaded for the device walkers)
const unsigned int (*unicode_word_ranges(int *n))[2];
// test aid: a random text drawn from the pattern itself (every alternative / repeat count / class member by a seeded generator;
// anchors are not looked at, so a text need not match) -- the differentials against the real engine feed on it
bool sample(const char *pattern, size_t len, unsigned options, uint64_t seed, std::string &out, std::string &err);
// forward-walk step counters of simulate_capture since the last call: {fast, lookahead, slow}
void debug_stats(long *out);

}  // namespace rx
