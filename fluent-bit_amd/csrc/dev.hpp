// dev.hpp -- structures shared by the host side (flbgpu.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <cstdint>
#include <atomic>
#include "tzif.hpp"

namespace flbgpu {

// ---- the NFA engine's tables (rx.hpp NfaSet; walked by nfa_dev.inc), as device pointers
struct DevNfa {
    const uint16_t *cls_byte;      // [512] class of an ASCII byte / a stray byte (b), of the lead of a cut sequence (256 + b)
    const uint32_t *mb_lo;         // [nmb] first code points of the intervals over [0x80, 0x10FFFF]
    const uint16_t *mb_cls;        // [nmb]
    const uint32_t *amask;         // [ncls][VW]
    const uint8_t *ckind;          // [ncls]
    const uint32_t *pred;          // [NK * NK][P + 2][VW]
    const uint8_t *mstart;         // [NK * NK]
    const uint32_t *list_off;      // [(P + 1) * NK * NK + 1]
    const uint32_t *list_ent;
    const uint32_t *tag_off;
    const uint8_t *tag_data;
    int P, VW, NK, kind_edge, ncls, nmb;
};

// ---- capture tables of one rx::TableSet, as device pointers
struct DevCap {
    // hot tables (touched once per input byte); stored back to back so that a workgroup can
    // stage them into LDS with one coalesced copy: [hot_base, hot_base + hot_bytes)
    const uint16_t *rdelta;        // [nR + 1][1 << cls_shift]; bit15 = a match may start here (`wide`: see below)
    const uint32_t *ft;            // [nX * NKp][1 << wsh] forward table (encoding: rx.hpp)
    const uint32_t *ft2;           // [nmulti][1 << fc_shift] one-byte-lookahead rows
    const uint8_t *cls;            // [512] symbol -> class (symbols 256 + b: rx.hpp TableSet::cls, word variants)
    const uint8_t *col;            // [512] symbol -> kind << fc_shift | class
    const uint8_t *xl;             // [512] utf8 set: symbol of a stray byte / of a truncated sequence's lead (rx.cpp SymbolMap)
    // cold tables (only when several candidates remain for a byte)
    const uint8_t *r_info;         // [nR]
    const uint32_t *vmask;         // [nR][VW]
    const uint32_t *list_off;      // [nX*NK*NK + 1]
    const uint32_t *list_ent;
    const uint32_t *tag_off;
    const uint8_t *tag_data;
    int ncls, nR, r_init, VW, nX, NK, NKp, kind_edge, ascii_only, cls_shift, fc_shift, wsh, col_eot;
    int word_variants;             // utf8 set of a pattern with \b / \B: a byte of a well-formed multi-byte WORD character is symbol 256 + b
    const uint32_t *wr;            // the Unicode word ranges {lo, hi} the walkers decide that with (rx::unicode_word_ranges), nwr pairs
    int nwr;
    int wide;                      // utf8 set only (rx.hpp `wide`): rdelta points at 32-bit entries, bit31 = a match may
                                   // start here; walked from HBM by the generic kernels, never staged
    const uint8_t *hot_base;
    uint32_t hot_bytes;
    uint32_t off_rdelta, off_ft, off_ft2, off_cls, off_col;   // byte offsets inside the hot block
    int stub;                      // ascii set only: a stub that hands EVERY value on, the empty one too (rx.cpp make_ascii_stub)
    // utf8 slot: the optimizer-dependent corners of the reference this pattern can meet (rx::CF_*) and where the walkers count the
    // values that do (kdev.inc rx_corner_note; one u64 in device memory, read by flbgpu_filter_regex_corners)
    int corner_flags;
    unsigned long long *corner_count;
    int nfa_on;                    // utf8 slot only: the NFA engine stands in for the table set (rx::Program::utf8_nfa); the
    DevNfa nfa;                    // table pointers above are null then
};

// ---- compact forward tables of the single-pass tile kernel (k_parser_tile, tile_kernels.inc), built from the ascii
// table set by upload_fx (flbgpu.cpp) and staged at offset 0 of the kernel's dynamic LDS: cls | ft | ft2 | p2.
// cls[256] (u32) maps a byte to class * 4; ft / ft2 hold one dword per (row, byte class | end of text).  Every offset
// is an LDS BYTE address, so that a step is: address = (entry & 0xFFFF) + class * 4, one ds_read_b32:
//   plain   bit31 = 0 | capture slot * 128 << 16 | address of the next row           (ONE capture write, slot 0 = none)
//   LOOK    bit31 = 1, bit30 = 0 | address of the ft2 row indexed by the class of the NEXT byte
//   PAIR    bit31 = 1, bit30 = 1 | index of a p2 entry {plain entry, second slot * 128}  (two capture writes: rare)
// MATCH, dead ends and multi-candidate cells are plain entries into the absorbing row whose capture write records
// what happened (slots ncap+1 ..: END_EOT, END_MID, DEAD_EOT, FAIL); a byte >= 0x80 leads to the POISON row.  Byte
// 0xFF is the END-OF-TEXT column: the kernel writes it behind the value in its LDS tile, a real 0xFF in the text is
// recognised by the position the slot holds (!= length) and sends the record to the UTF-8 tables.
struct DevFx {
    const uint8_t *base;           // cls | ft | ft2 | p2 (staged into LDS as one piece)
    uint32_t bytes;
    uint32_t off_p2;               // LDS address of p2
    uint32_t start_off, absorb_off, poison_off;   // row addresses
    uint32_t nslots;               // capture columns per lane: dummy + 2 * fields + 4
    int ok;                        // 0: the pattern has no compact tables (classic kernels only)
    uint32_t pair_bias, ncls1;     // pair tables (fx.cpp build_fx pair = true): an entry names a row by the address of its pair section, the
                                   // single-step cells sit pair_bias = ncls1 * 4 bytes in front of it; 0: single-step rows only
    // The tail of the pattern (fx.cpp find_tail): rows at addresses >= tail_min (they sit last, in front of the two absorbing rows) form a
    // set the walk cannot leave except through a "kill" byte (>= 0x80, or one of kill[0 .. nkill-1]) and in which the row after a byte
    // depends on that byte alone -- `(?<message>.*)$`, `"(?<agent>.*)")?$`.  Once every lane of a wave stands there (or in an absorbing
    // row) the rest of the text needs no steps: no kill byte in it, then the last byte and the end-of-text column decide.
    // tail_min == absorb_off: no such set (the walk still ends early when every lane is absorbed).
    uint32_t tail_min, nkill;
    uint8_t kill[4];
};
constexpr uint32_t FX_SLOT_SHIFT = 16, FX_ROW_MASK = 0xFFFFu, FX_LOOK = 0x80000000u, FX_PAIR = 0xC0000000u;
constexpr uint32_t FX2_LOOK3 = 0x80000000u, FX2_SPECIAL = 0xC0000000u;      // kinds of a pair cell (bit 31 clear: next row | slot 1 << 16 | slot 2 << 24)
enum { FXS_END_EOT = 1, FXS_END_MID = 2, FXS_DEAD_EOT = 3, FXS_FAIL = 4 };   // + ncap

// ---- match-only DFA
struct DevDfa {
    const uint8_t *cls;            // [256]
    const uint16_t *ddelta;        // [nD][ncls]
    const uint8_t *d_final;        // [nD] the answer at the end of the text AS AN RX_* CODE: 0 no match, 1 match, 2 hand the value on (the stub
                                   // of a pattern whose ASCII automaton does not fit: an empty text has no byte to poison the walk)
    int ncls, nD, d_init;
    uint32_t lds_bytes;
};

constexpr int MAX_RULES_PG = 64;         // (= MAX_RULES of filter_grep)
constexpr int MAX_GROUPS = 32;           // capture registers incl. group 0
constexpr int MAX_NAMES = 32;
constexpr int MAX_TIMEFMT = 96;
constexpr int MAX_KEY = 128;
constexpr int MAX_SUBKEYS = 8;

// Types cast (src/flb_parser.c:2067-2164)
enum { TY_NONE = 0, TY_INT = 1, TY_FLOAT = 2, TY_BOOL = 3, TY_STRING = 4, TY_HEX = 5 };

// Fixed-layout plan of a Time_Format (no %L): when every directive of the format has exactly one
// width the text can be checked and read at fixed offsets; anything else about the input -- another
// length, a one-digit day, a full month name -- sends the record to the strptime interpreter.
enum { TP_LIT = 1, TP_SPACE, TP_NUM2, TP_YEAR4, TP_MON3, TP_TZ5 };
enum { TPF_MDAY = 0, TPF_HOUR, TPF_MIN, TPF_SEC, TPF_MON1 };
struct TimeOp { uint8_t kind, off, a, b; };   // LIT: a = the character; NUM2: a = TPF_* field, b = upper limit
struct TimePlan {
    int ok, len, nops;
    TimeOp ops[32];
    uint8_t lo[32];                      // NUM2: lower limit
};

// ---- Decode_Field / Decode_Field_As of a parser (src/flb_parser_decoder.c; include/fluent-bit/flb_parser_decoder.h:27-43): one
// entry per key with its rules in configuration order (get_decoder_key_context :556-591)
enum { DEC_T_DEFAULT = 0, DEC_T_AS = 1 };                    // FLB_PARSER_DEC_DEFAULT (Decode_Field) / _AS (Decode_Field_As)
enum { DEC_A_NONE = 0, DEC_A_TRY_NEXT = 1, DEC_A_DO_NEXT = 2 };
constexpr int MAX_DEC_KEYS = 8, MAX_DEC_RULES = 8;
struct DevDecRule { uint8_t type, backend, action, pad; };   // backend: dec::BK_* (csrc/dec.hpp)
struct DevDecoder {
    uint8_t key[64];
    uint32_t key_len;
    uint32_t add_extra_keys;                                 // a Decode_Field rule exists for the key: its decoded object's pairs are appended
    uint32_t nrules;
    DevDecRule rules[MAX_DEC_RULES];
};
struct DevDecoders { uint32_t n; DevDecoder d[MAX_DEC_KEYS]; };
constexpr uint32_t DEC_REGIONS = 6;                          // scratch regions per lane of k_parser_dec (dec_dev.inc)

// ---- one regex parser (struct flb_parser, include/fluent-bit/flb_parser.h:41-70)
struct DevParser {
    DevCap ascii, utf8;
    int ngroups;
    // named groups in onig_foreach_name order, one entry per (name, group) pair
    int nfields;
    int field_group[MAX_NAMES];
    int field_name_off[MAX_NAMES];       // into names[]
    int field_name_len[MAX_NAMES];
    int field_is_time[MAX_NAMES];        // strcmp(name, time_key) == 0 and a time format exists
    int field_type[MAX_NAMES];           // TY_*
    char names[1024];
    int nregs_minus1;                    // flb_regex_do return value (num_regs - 1)
    int skip_empty;
    int has_time;
    int time_keep, time_strict, time_with_tz, time_offset;
    int has_frac;                        // %L present: fmt2 is the part after it
    char fmt1[MAX_TIMEFMT];              // expanded to primitive directives, NUL terminated
    char fmt2[MAX_TIMEFMT];
    uint8_t slot2cap[2 * MAX_GROUPS];    // capture slot -> index in the caps row (0xFF: not a named field)
    uint32_t keywords[MAX_NAMES + 1024 / 4 + MAX_NAMES];   // per field: msgpack str header + name, zero padded to a dword multiple
    int kw_off[MAX_NAMES];               // first dword of field f in keywords[]
    int kw_bytes[MAX_NAMES];             // header + name bytes
    int is_json;                         // Format json (src/flb_parser_json.c): no regex, the value is a JSON object
                                         // (also set for the other walker formats below: same kernels)
    int kv_format;                       // 1 Format logfmt, 2 Format ltsv (pkv_dev.inc); 0 otherwise
    int no_bare_keys;                    // Logfmt_No_Bare_Keys
    int tkey_len;                        // time key of a json parser (default "time")
    char tkey[64];
    int fwd_first;                       // the pattern is anchored at the start: try the forward walk from boundary 0 before
                                         // paying for the reverse pass (any match that starts at 0 is the leftmost one)
    int time_field;                      // the ONE named field that is the time key, -1 if none or several
    int plain_types;                     // no Types cast changes a value's encoded size (all string / none)
    TimePlan plan;                       // fast path of fmt1 (ok == 0: interpreter only)
    // Time_Format without %Y / %y / %s (src/flb_parser.c:922-941): every lookup reads "<current year> " in front
    // of the text and starts from today's month and day (:1945-2001).  The three numbers are written into the
    // filter's device copy of this struct at the start of every run (time(NULL), UTC).
    int yearless;
    int now_year, now_mon, now_mday;     // 4-digit year, tm_mon (0..11), tm_mday
    // Types of a logfmt / ltsv parser (keys come from the text, so the casts are looked up by name per pair:
    // flb_parser_typecast, src/flb_parser.c:2067-2164); names live in names[]
    int nkvtypes;
    int kvtype_off[MAX_NAMES], kvtype_len[MAX_NAMES], kvtype_kind[MAX_NAMES];
    DevFx fx;
    DevFx fx2;                     // the same tables with a pair section per row (two steps per read); ok = 0: does not fit                            // compact forward tables of the tile kernel (ok == 0: none)
    const DevDecoders *decs;             // Decode_Field / Decode_Field_As rules (device memory; nullptr: none) -- dec_dev.inc
    // Time_Zone / Time_System_Timezone (src/flb_parser.c:685-696 flb_parser_tm2time_parser; tzif.hpp).  zone_mode 0: timegm - gmtoff;
    // 1: the zone's TZif table decides (when the format carries no zone of its own); 2: the process's zone, which is UTC
    // (mktime == timegm, whatever offset the text named).  Parsers with a mode take the time interpreter (plan.ok == 0).
    int zone_mode, time_offset_given;
    int tz_timecnt, tz_typecnt, tz_default;
    const int64_t *tz_trans;             // device memory, owned by the flbgpu_parser
    const int32_t *tz_gmtoff;
    const uint8_t *tz_ttype;
    // a HOST parser (the Regex is not a regular expression: the host's backtracking matcher answers, rxbt.inc): no tables above; in a
    // list of parsers its answers for the chunk's values stand in ParserMatchArgs::host_res[host_slot] (k_parser_generic reads them)
    int host_only, host_slot;
    // %Z's last resort (src/flb_strptime.c:611-650): a zone text that is in neither of flb_strptime's tables is compared, without case and
    // as a prefix, with the two names of the PROCESS's zone -- tzname[0], tzname[1] -- and either one means -timezone; read at create
    // (tzset).  tz_names == 0: a process without a zone (both names "UTC")
    int tz_names, tzn_gmtoff;
    int tzn_len[2];
    char tzn[2][16];
};
constexpr int MAX_HOST_PARSERS = 4;      // host parsers in one list of parsers

// ---- record accessor / key
struct DevKey {
    int is_ra;                           // '$key['a'][1]' form (backward lookup, STR keys only)
    int key_len;
    char key[MAX_KEY];
    int nsub;
    int sub_is_index[MAX_SUBKEYS];
    int sub_index[MAX_SUBKEYS];
    int sub_off[MAX_SUBKEYS], sub_len[MAX_SUBKEYS];
    char sub_str[256];
};

// per-record result of the parser match pass
struct alignas(16) RecInfo {
    uint32_t flags;
    uint32_t val_off;                    // value offset relative to record start
    uint32_t val_len;
    uint32_t key_index;                  // index of the parsed kv in the body map
    uint32_t ts_sec, ts_nsec;            // timestamp to emit
    uint32_t body_off, body_len, meta_off, meta_len;   // relative to record start; meta_len 0 => {}
    int32_t parser_idx;
    uint32_t nkept;                      // fields that will be packed (map count after skips)
    uint32_t drop_mask;                  // bit f: named field f is not packed (empty+skip_empty,
                                         // unparsable time, or time consumed and !time_keep)
    uint32_t meta_canon;                 // size of the canonically re-packed metadata (1 when there is none)
    uint32_t pad_[2];                    // 64 bytes
};
constexpr int REC_NCOLS = 14;
// capture spans live in a separate column: caps[rec][2*field + {0,1}] (begin/end relative to the
// value, 0xFFFFFFFF = group did not participate), field = index in DevParser::field_group

enum {
    RF_VALID = 1,          // well-formed log event
    RF_SKIP = 2,           // group marker / negative timestamp: hidden by the decoder
    RF_PARSED = 4,         // a parser matched
    RF_BADTS = 8,          // timestamp outside the EventTime range: encoder error, record dropped
    RF_BAD = 16,           // decoder error: processing stops here
    RF_CAND = 32,          // exactly one candidate value located (fast phases)
    RF_RXOK = 64,          // parser 0's regex matched, spans published
    RF_GENERIC = 128,      // needs k_parser_generic
    RF_EXACT = 256,        // a Types float value needs the exact decimal conversion: k_parser_emit_exact rewrites the record
    RF_PGDONE = 512,       // pair [filter_parser, filter_grep]: grep's rules were evaluated on the spans by k_parser_rx ...
    RF_PGKEEP = 1024,      // ... and keep the record
    RF_NEEDLOC = 4096,     // k_parser_reg<false> left the row to the fix-up launch (another layout, the end of the chunk)
    RF_DEC = 8192,         // the winning parser has decoders: sized and written by k_parser_dec (dec_dev.inc), skipped by k_parser_emit
    RF_DESC = 2048,        // pair mode: the kept record's fields are in its descriptor (PgEmitArgs::desc), not in the columns
    RF_TIMEPEND = 16384,   // pair mode, RF_DESC rows: the time text has not been looked up yet -- k_pg_emit does it for the records that are
                           // kept (TileCfg::defer_time); the descriptor carries the event's own time
};
constexpr uint32_t PG_UNDECIDED = 0xFFFFFFFFu;   // keep_len of a row whose rules k_pg_decide still has to evaluate

struct GrepRule;
// grep's rules handed to filter_parser's kernels when the two filters run as a pair (device memory)
struct PgInline {
    const GrepRule *rules;
    int nrules, logical_op;
    uint32_t rule_fmask[MAX_RULES_PG];    // per rule: the parser's named fields its key names
    uint32_t rule_lds_off[MAX_RULES_PG], rule_lds_bytes[MAX_RULES_PG];   // match-only DFA block in LDS, relative to pg_lds_off (0xFFFFFFFF: global)
    uint32_t static_drop;                 // named fields never packed: time fields consumed when Time_Keep is off
    uint32_t time_fields;                 // named fields whose presence depends on the time lookup
};

constexpr uint32_t CAP_UNSET = 0xFFFFFFFFu;
constexpr int TBUF_WORDS = 8;
constexpr int CHK_STEP = 16;             // one reverse-DFA state id is kept every CHK_STEP boundaries
constexpr uint32_t FT_SPECIAL = 0x80000000u, FT_CAPS = 1, FT_MATCH = 2, FT_LOOK = 3, FT_MULTI = 4, FT_DEAD = 5;
constexpr uint32_t TG_MATCH = 0xFFFFFFFDu, TG_DEAD = 0xFFFFFFFFu;   // results of the slow-path resolver

// ---- filter_parser configuration (plugins/filter_parser/filter_parser.c:460-489)
struct FParserCfg {
    DevKey key;
    int reserve_data, preserve_key;
    int nparsers;
    // Key_Name entries a parser succeeded on are remembered per record in a 64-bit mask (body map index
    // 0..63); successes at index 64 and up go to this side list of (record, index) pairs, filled by the size
    // pass and read by both passes (set per run by the host; normally empty)
    unsigned long long *ov_pairs;
    unsigned int *ov_count;          // entries appended (may exceed ov_cap: the run then fails loudly)
    unsigned int ov_cap;
};

// What k_parser_reg / k_parser_tile read per record of the parser's and the rules' configuration, passed by value in
// the kernel arguments: a uniform load from the argument segment is a scalar load the compiler may hoist, while the same
// field read through a pointer into global memory is a vector load per use (and the uses are serialised by the branches
// that depend on them: ~100 loads of several hundred cycles per record).
struct TileRule { uint32_t type, fmask, lds_off, ncls, d_init, o_dd, o_df; };   // a grep rule whose match-only DFA is staged in the LDS
constexpr int TILE_RULES = 8;
struct TileCfg {
    int nfields, skip_empty, time_field, time_keep, nregs_minus1, time_with_tz, time_offset, plain_types;
    uint32_t is_time_mask;               // bit f: named field f is the time key
    uint8_t name_cost[MAX_NAMES];        // bytes of field f's key in the output map (str header + name)
    TimePlan plan;
    int pg_on, pg_nrules, pg_logical_op;
    int pg_fast;                         // every rule is in rule[] (at most TILE_RULES, DFAs staged in the LDS)
    uint32_t pg_static_drop, pg_time_fields, pg_named;
    TileRule rule[TILE_RULES];
    // pair mode (round 5): the time lookup of a record is left to k_pg_emit, which only sees the records grep keeps.  What the pass
    // still has to know of a record it drops is whether the encoder would have refused its parsed time (the record then never
    // reached grep: flb_filter_do's counts) -- the four year digits at their place in the fixed layout, 1971 .. 2105, rule that out;
    // any other text takes the lookup here, as before.
    int defer_time;
    uint32_t year_off;                   // offset of the plan's %Y digits in the time text
};

constexpr uint32_t STAGE_MAXK = 20;              // 1 KiB pieces of a staging buffer (ParserMatchArgs::stage_bytes <= 20 KiB)
constexpr uint32_t STAGE_SLACK = 48;             // bytes behind a buffer that the lanes' 16-byte reads may touch

struct ParserMatchArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    FParserCfg cfg;
    const DevParser *parsers;
    uint32_t *info;             // record columns: REC_NCOLS arrays of n u32 (structure of arrays:
                                // a wave reads 64 consecutive words of a field, not 64 rows)
    uint32_t *caps;             // [caps_stride][n] capture spans, one column per span end
    uint32_t caps_stride;
    uint64_t *null_mask;        // [n]
    uint32_t *out_len;          // [n]
    uint32_t *tbuf;             // [TBUF_WORDS][n]: the first 32 bytes of the time field's text, copied by k_parser_rx
                                // while they are cache-hot so that k_parser_finish never touches the chunk
    uint16_t *chk;              // scratch: reverse-DFA state checkpoints [slots][chk_len][64]
    uint32_t chk_len;           // checkpoints per lane (max value length / CHK_STEP + 2)
    uint32_t chk_nfa_off;       // k_parser_generic: the 16-bit slot the NFA engine's kept states start at (behind the table walkers' slots)
    uint32_t lds_bytes;         // dynamic LDS: parser 0's hot ASCII tables are staged when > 0
    uint32_t caps_lds_off;      // byte offset of the per-thread capture columns inside the dynamic LDS
    uint32_t caps_in_lds;       // 0: spans are written straight to the global row
    uint32_t lds_total;         // dynamic LDS bytes to request (tables + capture columns)
    uint32_t debug_skip;        // reserved (0)
    unsigned long long *first_bad;   // min index of a record that stops the decoder loop
    unsigned long long *counts;      // [0] decoded log records, [1] records emitted, [2] records for the generic kernel,
                                     // [3] records for k_parser_emit_exact, [8] records k_parser_tile leaves to k_parser_finish
    uint64_t bytes;                  // chunk size (bounds the coalesced tile loads)
    // pair mode (fused_kernels.inc): grep's rules evaluated inline; nullptr otherwise
    const PgInline *pg;
    uint32_t pg_lds_off;             // k_parser_rx: where the rules' DFA blocks are staged in its dynamic LDS
    uint32_t *pg_keep_len;           // [n] written by k_parser_finish: out_len when kept, 0, or PG_UNDECIDED
    // k_parser_tile: dynamic LDS layout = fx tables | rule DFAs (pg_lds_off) | per wave: record tile + capture columns
    uint32_t tile_lds_off;           // first wave's area
    uint32_t use_fx2;                // parsers[0].fx2 (pair cells) instead of .fx: k_parser_reg<.., PAIR2>
    uint32_t tile_wave_bytes;        // bytes per wave (tile + capture columns)
    // pair mode: one descriptor of dstride dwords per ROW, written for the rows the single pass keeps -- ONE aligned store
    // of whole 64-byte sectors instead of 34 column stores of 4 bytes (each of which costs a sector write):
    //   [0] ts_sec [1] ts_nsec [2] val_off [3] drop_mask [4] meta_off [5] meta_len [6] nkept, then the capture spans as u16 pairs
    uint32_t *desc;
    uint32_t dstride;
    uint64_t fix_first;              // k_parser_reg<true>: first flagged row
    // the rows k_parser_reg<false> leaves to the fix-up launch (another layout: several keys, metadata, legacy events ...), as a LIST:
    // the fix-up launch takes 64 listed rows per wave instead of looking for flagged rows among all of them (10 % such rows scattered
    // over the chunk had every wave of it run with six lanes: as long as the main pass, round 4)
    uint32_t *fix_list;              // [n] row numbers (nullptr: none -- n >= 2^32)
    unsigned long long *fix_count;   // how many
    // k_parser_reg, staged ingest: a workgroup's waves share stage_nbuf LDS buffers of stage_bytes (+ 32 of slack) each at
    // stage_lds_off; a wave takes one, copies its 64 records (one contiguous range of the chunk) into it with coalesced
    // 16 B / lane loads, every lane pulls header + value into registers, and the buffer goes back (0 buffers: per-lane loads)
    uint32_t stage_lds_off, stage_bytes, stage_nbuf;
    // diagnostics (FLBGPU_TRACE=<iterations>): s_memtime stamps of the phases of a wave's first iterations, [wave][iteration][8]
    unsigned long long *trace;
    uint32_t trace_iters;
    const uint8_t *tail_buf;         // the chunk's bytes from tail_start on, followed by 512 zero bytes (wide loads of the last records)
    uint64_t tail_start;
    TileCfg tc;
    // a list of parsers with host parsers in it: per host parser [1 + 2 * nfields][n] -- column 0 = 1 + the offset in the record of the
    // value the matcher took (0: it did not match / the row is no candidate), then the capture spans as k_parser_rx writes them
    const uint32_t *host_res[MAX_HOST_PARSERS];
    // k_parser_reg<false>: the row a wave's lane takes is perm[its place] instead of its place -- the rows ordered by length class
    // (kernels_perm.hip), so that the position-synchronous walk of a wave is as long as ITS rows and not as the chunk's longest line;
    // nullptr: chunk order.  len_stat (chunk order only): += longest row x rows of a wave-iteration, [1] += the rows' bytes, from one
    // wave in eight -- what the host decides the next call's order by
    const uint32_t *perm;
    unsigned long long *len_stat;
    const ParserMatchArgs *self;     // this structure in device memory: what the out-of-line slow paths read (taking the address of a
                                     // kernel argument makes the compiler keep the whole argument block in scratch memory)
};

// What the record writer reads per field of parser 0, by value in the kernel arguments (scalar loads; through the DevParser
// pointer each is a dependent vector load per field per record).  ok: parser 0 is a regex parser without Types casts and the
// filter neither reserves nor preserves keys -- the plain "parsed fields only" record.
struct EmitCfg {
    int ok, nfields, nregs_minus1;
    uint8_t kw_off[MAX_NAMES];           // first dword of field f's packed key in keywords[]
    uint8_t kw_bytes[MAX_NAMES];         // str header + name bytes
    uint32_t keywords[96];
};

struct ParserEmitArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    FParserCfg cfg;
    const DevParser *parsers;
    uint64_t n_cols;            // column length of info/caps (the n of the match pass, >= n)
    const uint32_t *info;       // record columns (REC_NCOLS x n_cols)
    const uint32_t *caps;       // [caps_stride][n]
    uint32_t caps_stride;
    const uint64_t *null_mask;
    const uint32_t *out_len;
    const uint64_t *out_off;    // exclusive scan of out_len, [n+1]
    uint8_t *out;
    uint64_t bytes;             // chunk size (bounds the wide tail loads)
    uint64_t out_cap;           // launched ahead of the size (flbgpu.cpp SpecCall): room behind `out`; a larger out_off[n] ends the kernel (0: not checked)
    EmitCfg ec;
};

// k_parser_dec: the rows whose winning parser has decoders (dec_dev.inc).  mode 0 = size pass (out_len, flags, timestamps are
// rewritten), 1 = emit pass (the record at out_off).  scratch: lanes x DEC_REGIONS x cap bytes, one slab per launched lane.
struct DecArgs {
    ParserEmitArgs e;
    uint32_t *info_w;           // = e.info, writable (size pass)
    uint32_t *out_len_w;        // = e.out_len, writable (size pass)
    uint8_t *scratch;
    uint32_t cap;               // bytes per region
    int mode;
    unsigned long long *err;    // rows whose scratch overflowed (the call fails)
    unsigned long long *ndec;   // rows handled (size pass)
};
void launch_parser_dec(const DecArgs &a, int blocks, hipStream_t st);

// ---- filter_grep (plugins/filter_grep/grep.c)
constexpr int MAX_RULES = 64;
enum { GREP_REGEX = 1, GREP_EXCLUDE = 2 };
enum { OP_LEGACY = 0, OP_OR = 1, OP_AND = 2 };

struct GrepRule {
    int type;
    DevKey key;
    DevDfa dfa;
    DevCap utf8;
    // a HOST rule: the pattern is not a regular expression (csrc/rxbt.inc answers it on the host, flbgpu.cpp "host rules").  host_mode 1:
    // k_grep_match notes the value's place per row (host_io[2r] offset in the row, [2r+1] length; 0xFFFFFFFF: no string value);
    // 2: host_io is the host's answers, one bit per row.  0: a device rule.
    uint32_t *host_io;
    int host_mode;
};

struct GrepArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    uint64_t bytes;             // chunk size (bounds the coalesced tile loads)
    const GrepRule *rules;
    int nrules;
    int logical_op;
    uint32_t *keep_len;         // [n] record length when kept, else 0
    uint32_t *status;           // [n] RF_* flags
    unsigned long long *first_bad;
    unsigned long long *counts; // [0] = decoded (non-skipped) records, [1] = kept records
    int host_collect;           // pass 1 of a filter with host rules (GrepRule::host_mode 1)
    // the rules' match-only DFA blobs (cls | ddelta | d_final: upload_dfa) staged behind the record tiles in LDS: two dependent
    // table reads per byte of a tested value, ~100 cycles from LDS against ~400 through L2
    uint32_t rule_lds_off[MAX_RULES], rule_lds_bytes[MAX_RULES];   // offset 0xFFFFFFFF: not staged
    uint32_t rules_lds_total;
    // the rules' top-level keys, each once: ONE walk over the body map finds the value of every key (the last entry of that name, as
    // flb_ra_key.c's backward lookup does) instead of one walk per rule.  nslots == 0: more than GREP_SLOTS distinct keys -- every rule
    // looks its key up itself
    int nslots;
    uint8_t slot_rule[8];       // a rule that carries the slot's key
    uint8_t rule_slot[MAX_RULES];
};
constexpr int GREP_SLOTS = 8;


// ---- msgpack -> JSON output formatter (fmt_dev.inc / kernels_fmt.hip): flb_pack_msgpack_to_json_format
struct JsonFmtCfg {
    int json_format;            // FLB_PACK_JSON_FORMAT_JSON 1 / STREAM 2 / LINES 3 (include/fluent-bit/flb_pack.h:57-60)
    int date_format;            // FLB_PACK_JSON_DATE_* (:38-42)
    int escape_unicode;
    int nan_to_null;            // json.convert_nan_to_null (src/flb_pack.c:54-61)
    int has_date;               // date_key != NULL
    int date_key_is_internal;   // date_key == "__internal__"
    uint32_t date_key_len;
    const uint8_t *date_key;    // device copy
};
struct JsonFmtArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    JsonFmtCfg cfg;
    uint64_t bytes;             // size of the chunk (16-byte loads stop in front of its end)
    uint32_t *len;              // [n] bytes the row contributes to the output (0: markers, rows past the end)
    uint64_t *slow;             // [(n + 63) / 64] bit r % 64: row r holds a key twice in one map (printed by the look-ahead walker)
    const uint32_t *g_row;      // [n] 1 + row of the group opener that governs the row, 0 = none; nullptr: chunk without groups
    unsigned long long *first_bad;   // first row the decoder refuses
    unsigned long long *first_fail;  // first row whose temporary map would not unpack (the reference returns NULL)
    unsigned long long *counts;      // [0] records, [1] group markers (-1 / -2), [2] skipped rows (all negative times)
    const uint64_t *out_off;    // [n + 1] exclusive scan of len
    uint8_t *out;
};
struct JsonGroupArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    uint32_t *g_row;
    unsigned long long *skip_limit;  // first row that follows 1000 consecutive skipped rows (decoder :27,:389-394), else ~0
};
void launch_fmt_size(const JsonFmtArgs &a, int cus, hipStream_t st);
void launch_fmt_emit(const JsonFmtArgs &a, int cus, hipStream_t st);
void launch_fmt_groups(const JsonGroupArgs &a, hipStream_t st);

// ---- flb_filter_do over [filter_parser, filter_grep] without materialising the parsed chunk (fused_kernels.inc)
struct PgDecideArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n, n_cols;              // rows in front of the first decoder error / column length
    const uint32_t *info;            // record columns of filter_parser's pass 1
    const uint32_t *caps;            // capture span columns
    const uint32_t *out_len;         // [n] size of the record filter_parser would emit
    const GrepRule *rules;
    int nrules, logical_op;
    uint32_t rule_fmask[MAX_RULES];  // per rule: the parser's named fields its key names (bit f)
    uint32_t rule_lds_off[MAX_RULES], rule_lds_bytes[MAX_RULES];   // the rule's match-only DFA block in LDS (off 0xFFFFFFFF: not staged)
    uint32_t lds_total;
    uint32_t *keep_len;              // [n] out_len when filter_grep keeps the record, else 0
    unsigned long long *counts;      // [4] bytes filter_parser emits, [5] records filter_grep keeps, [6] records filter_parser emits
};
struct PgEmitArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    FParserCfg cfg;
    const DevParser *parsers;
    uint64_t n_cols;
    const uint32_t *info;
    const uint32_t *caps;
    const uint64_t *null_mask;
    const uint32_t *keep_len;
    uint64_t n;
    const uint64_t *out_off;
    uint8_t *out;
    const uint32_t *desc;            // row descriptors of the single pass (ParserMatchArgs::desc), nullptr: columns only
    uint32_t dstride;
    uint64_t bytes;                  // chunk size (bounds the wide tail loads)
    uint64_t out_cap;                // as in ParserEmitArgs
    EmitCfg ec;
    unsigned long long *counts;      // [13]: RF_TIMEPEND rows whose time text the fixed-layout plan does not settle (the host repeats the
                                     // call with the lookup inside the single pass, and keeps it there for this filter)
};

// ---- filter_log_to_metrics (plugins/filter_log_to_metrics/log_to_metrics.c)
enum { L2M_COUNTER = 0, L2M_GAUGE = 1, L2M_HISTOGRAM = 2 };
constexpr int L2M_MAX_LABELS = 128;          // MAX_LABEL_COUNT (log_to_metrics.h:50)
constexpr int L2M_LABEL_MAX = 251;           // snprintf(buf, MAX_LABEL_LENGTH - 1, ...): at most 251 characters
// One row of 64-bit words per series.  Every word merges across chunks, workgroups and GPUs with an
// associative integer operation (max or add), which is what makes the result independent of the
// order in which records are visited:
//   [W_FIRST]    max of ~(global index of the record that created the series)   -> snapshot order
//   [W_LASTIDX]  max of (global index + 1) of the record that last set the gauge
//   [W_LASTVAL]  binary64 bits of the value that record carried (follows W_LASTIDX)
//   [W_COUNT]    add: records counted (counter mode)
//   [W_SPECIAL+0..2]  add: NaN / +Inf / -Inf observations (they do not fit the fixed-point sum)
//   [W_LIMB + j] add: sum of the observed values as a fixed-point integer in 32-bit digits,
//                digit j has weight 2^(32 j - 1074); exact, so the f64 sum is rounded once
//   [W_BUCKET + b] add: observations whose smallest containing bucket is b (b == nb: above all bounds)
constexpr int L2M_NLIMB = 68;
constexpr int L2M_W_FIRST = 0, L2M_W_LASTIDX = 1, L2M_W_LASTVAL = 2, L2M_W_COUNT = 3, L2M_W_SPECIAL = 4,
              L2M_W_LIMB = 7, L2M_W_BUCKET = L2M_W_LIMB + L2M_NLIMB;
inline int l2m_row_words(int mode, int nb) { return mode == L2M_HISTOGRAM ? L2M_W_BUCKET + nb + 1 : L2M_W_SPECIAL; }

constexpr uint32_t L2M_SID_NONE = 0xFFFFFFFFu;     // record makes no observation
constexpr uint32_t L2M_SID_DEFER = 0xFFFFFFFEu;    // needs k_l2m_generic (float label, doubtful decimal)
constexpr uint32_t L2M_SID_STALE = 0x80000000u;    // flag: value is the previous assigned value of this call

// series dictionary: open addressing on the 64-bit hash of the label tuple; the tuple's bytes live
// in an append-only arena and every lookup compares them (no reliance on the hash being unique)
struct L2mTable {
    unsigned long long *slot_hash;   // [cap]: 0 empty, 1 being published, else hash (>= 2)
    uint32_t *slot_sid;              // [cap]
    uint64_t cap_mask;
    uint32_t max_series;
    unsigned int *n_series;
    uint8_t *arena;
    unsigned long long *arena_used;
    uint64_t arena_cap;
    unsigned long long *key_off;     // [max_series]
    uint32_t *key_len;               // [max_series]
    unsigned long long *series_hash; // [max_series]
    unsigned int *overflow;
};

struct L2mArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    uint64_t bytes;
    const GrepRule *rules;
    int nrules;
    const DevKey *labels;            // key_len < 0: accessor without a key ($TAG, $0): always empty
    int nlabels;
    const DevKey *value_key;
    int mode;
    L2mTable t;
    uint32_t *sid_col;               // [n]
    uint64_t *val_col;               // [n] binary64 bits
    unsigned long long *first_bad;
    unsigned long long *counts;      // [0] observations, [1] deferred rows, [2] stale rows
};

struct L2mAggArgs {
    const uint32_t *sid_col;
    const uint64_t *val_col;
    uint64_t n;
    const unsigned long long *first_bad;
    unsigned long long *rows;        // [max_series][W]
    int W, mode, nb;
    const double *bounds;            // [nb] ascending
    uint64_t idx_base;               // global index of row 0
    const unsigned int *n_series;
};

// ---- stream processor: GROUP BY aggregation (src/stream_processor/flb_sp.c:1280-1601), sp_kernels.inc / sp.cpp
constexpr int SP_MAX_KEYS = 12;       // distinct key references: GROUP BY columns, aggregated keys, WHERE operands
constexpr int SP_MAX_GB = 4;
constexpr int SP_MAX_SRC = 6;         // distinct aggregated keys
constexpr int SP_MAX_SUB = 3;
constexpr int SP_MAX_LEAF = 20;
constexpr int SP_MAX_OPS = 24;
constexpr int SP_BLOB = 448;
struct SpKeyRef { uint16_t name_off, name_len, nsub, pad; uint16_t sub_off[SP_MAX_SUB], sub_len[SP_MAX_SUB]; };
enum { SPL_KEY = 0, SPL_INT, SPL_FLOAT, SPL_STR, SPL_BOOL, SPL_NULL, SPL_TIME, SPL_CONTAINS, SPL_NONE };
struct SpLeaf { uint8_t kind, key; uint16_t str_off, str_len, pad; uint64_t v; };
enum { SPO_EQ = 0, SPO_LT, SPO_LTE, SPO_GT, SPO_GTE, SPO_TRUTH, SPO_NOT, SPO_AND, SPO_OR };
struct SpOp { uint8_t op, l, r, pad; };
struct SpPlan {
    int nkeys, ngb, nsrc, nleaf, nops, str_conv;
    SpKeyRef keys[SP_MAX_KEYS];
    uint8_t gb_key[SP_MAX_GB], src_key[SP_MAX_SRC];
    SpLeaf leaf[SP_MAX_LEAF];
    SpOp ops[SP_MAX_OPS];          // postfix; the bool stack is a bit mask
    char blob[SP_BLOB];            // key names, sub-key names, string constants
};
// One row of 64-bit words per group; like the log_to_metrics rows every word merges with max or add, so the state does
// not depend on the order records (or GPUs) are visited in:
//   [0]                      max of ~(global index of the record that created the group)  -> first-seen order of package_results
//   [1 + 4 s + 0..3]         max: ~ord(min int), ord(max int), ~ord(min float), ord(max float) of aggregated key s
//   [A] (A = 1 + 4 nsrc)     add: records of the group (aggr_node->records)
//   [A + 1 + s SP_SRC_ADD +] add: ints seen, non-zero floats seen, wrapping int64 sum, NaN / +Inf / -Inf counts, then the
//                            exact fixed-point sum digits of every value (L2M_NLIMB words, digit j has weight 2^(32 j - 1074))
constexpr int SP_SRC_MAX = 4;
constexpr int SP_A_NINT = 0, SP_A_NFLT = 1, SP_A_ISUM = 2, SP_A_NAN = 3, SP_A_PINF = 4, SP_A_NINF = 5, SP_A_LIMB = 6;
constexpr int SP_SRC_ADD = SP_A_LIMB + 68;
inline int sp_row_words(int nsrc) { return 1 + SP_SRC_MAX * nsrc + 1 + SP_SRC_ADD * nsrc; }
constexpr uint32_t SP_GID_NONE = 0xFFFFFFFFu;
constexpr uint32_t SP_GID_DEFER = 0xFFFFFFFEu;     // needs k_sp_generic (decimal string -> binary64)
// status bits (SpArgs::flags): inputs the reference itself does not treat in an order-independent way -> the call fails
enum { SPF_BAD_RECORD = 1, SPF_KEY_NUL = 2, SPF_KEY_NAN = 4, SPF_FLOAT_RANGE = 8 };
struct SpArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n, bytes;
    const SpPlan *plan;             // in device memory (its blob is read through pointers)
    L2mTable t;
    uint32_t *gid_col;               // [n]
    uint64_t *val_col;               // [nsrc][n] int64 / binary64 bits
    uint8_t *vt_col;                 // [nsrc][n] class (0 none, 1 int, 2 float) | occurrences << 2
    unsigned long long *first_bad;
    unsigned long long *counts;      // [0] records that entered a group, [1] records deferred to k_sp_generic
    unsigned int *col_class;         // [SP_MAX_GB] OR of (1 << class) seen in each GROUP BY column (1 int, 2 float, 4 string)
    unsigned int *flags;
};
struct SpAggArgs {
    const uint32_t *gid_col;
    const uint64_t *val_col;
    const uint8_t *vt_col;
    uint64_t n;
    const unsigned long long *first_bad;
    unsigned long long *rows;
    int W, nsrc;
    uint64_t idx_base;
    const unsigned int *n_series;
};
void launch_sp_extract(const SpArgs &a, int cus, hipStream_t st);
void launch_sp_generic(const SpArgs &a, hipStream_t st);
void launch_sp_aggregate(const SpAggArgs &a, int cus, hipStream_t st);

// ---- JSON text -> msgpack (src/flb_pack.c:389-508)
struct JsonArgs {
    const uint8_t *text;
    const uint64_t *row_off;
    uint64_t n;
    uint32_t *out_len;          // [n] msgpack bytes of the row (events mode: event bytes, 0 when not one object)
    uint32_t *records;          // [n] values parsed (0 => flb_pack_json returns -1 unless the row was blank)
    uint32_t *consumed;         // [n]
    uint8_t *root_type;         // [n] jsmn type of the first value: 1 object 2 array 3 string 4 primitive
    uint8_t *status;            // [n] 0 ok, 1 error (no value parsed), 2 deferred to the generic kernels
    const uint64_t *out_off;    // [n + 1] (emit)
    uint8_t *out;
    uint32_t *cnt;              // [8][n] element counts of each row's first containers (size pass -> emit pass)
    int events;                 // wrap single-object rows as V2 log events with the timestamp below
    uint32_t ts_sec, ts_nsec;
    unsigned long long *counts; // [0] rows deferred, [1] values parsed, [2] rows in error; the one-pass kernel (jlane_kernels.inc): [3] rows it left
                                // to the row-per-lane kernels, [4] bytes it wrote, [5] workgroups without room in `out`, [6] tokens, [7] its ticket counter
    int tile_mode;              // the one-pass kernel ran first: the row-per-lane kernels take only the rows it marked JS_TDEFER
};

// the NDJSON one-pass kernel (jlane_kernels.inc): a row per lane in LDS, text read once, output placed by a look-back over the workgroups
struct JtArgs {
    JsonArgs j;                 // text, row_off, n, the row columns, out, events, ts, counts
    uint64_t *off_out;          // [n + 1] row offsets of the output (written by the pass)
    unsigned long long *tile_state;   // [ntiles] look-back words (zeroed by the host)
    unsigned long long *ticket;       // tile counter (zeroed)
    uint64_t ntiles;
    uint32_t rows_per_tile;     // <= 64
    uint64_t out_cap;           // room in j.out; a tile that would pass it writes nothing and raises counts[5]
    unsigned long long *prof;   // [8] cycles per phase summed over the waves (nullptr: no stamps); tools/perf_json.py
    int lb_off;                 // timing experiments only: no look-back, every workgroup writes at a place of its own
};


// ---- in_tail's line packing (tail_kernels.inc)
struct TailArgs {
    const uint8_t *text;
    uint64_t bytes;
    const unsigned long long *lead;  // leading NUL bytes
    const uint64_t *nl_pos;          // [nl] positions of the newlines
    uint64_t nl;
    uint32_t *out_len;               // [nl]
    const uint64_t *out_off;         // [nl + 1]
    uint8_t *out;
    unsigned long long *lines;       // records produced
    uint64_t stream_offset;
    uint32_t ts_sec, ts_nsec;
    int skip_empty_lines;
    uint32_t la, lb, lc;             // bytes of [path_key path], [offset_key], [key] in pre[]
    uint8_t pre[1024];               // the packed strings every record carries
};
void launch_tl_lead(const uint8_t *text, uint64_t bytes, unsigned long long *lead, hipStream_t st);
size_t tl_tiles(uint64_t bytes);
void launch_tl_count(const uint8_t *text, uint64_t bytes, uint64_t *masks, uint32_t *tile_cnt, hipStream_t st);
void launch_tl_fill(const uint64_t *masks, uint64_t bytes, const uint64_t *tile_off, uint64_t *nl_pos, hipStream_t st);
void launch_tl_size(const TailArgs &a, hipStream_t st);
void launch_tl_emit(const TailArgs &a, int cus, hipStream_t st);


// ---- multiline in front of the path (ml_kernels.inc; src/multiline/flb_ml.c, flb_ml_rule.c, flb_ml_group.c): text lines of one stream,
// ONE multiline parser.  Items: k = 0 is the buffer the stream carries from the previous call, k = 1.. the lines that reach
// flb_ml_append_text (Skip_Empty_Lines applied).  rule_to_state: 0 none, r + 1 = rule r -- a line's transition is a function over
// these <= 16 states, packed as 16 nibbles.
constexpr int ML_MAX_RULES = 15;
enum { ML_REGEX = 0, ML_ENDSWITH = 1, ML_EQ = 2 };                         // flb_ml.h FLB_ML_REGEX / ENDSWITH / EQ
enum { MLK_CARRY = 0, MLK_CONT = 1, MLK_START = 2, MLK_ALONE = 3, MLK_APPEND = 4, MLK_DROPPED = 5, MLK_MASK = 7, MLK_BA = 8,
       MLK_SEP = 64, MLK_TRUNC = 128 };   // act[]: kind | flush after | bits 4-5 MLT_* | a '\n' goes in front | flb_ml_group_cat cut the item
enum { MLT_EMPTY = 0, MLT_NL = 1, MLT_OTHER = 2 };                         // how the group buffer ends (act[] bits 4-5: after the item)
enum { MLP_FIRST = 1, MLP_SEP = 2, MLP_TRAIL = 4, MLP_CARRY_TIME = 8, MLP_NLBODY = 16, MLP_OPEN = 32, MLP_TRUNC = 64 };   // pk[]
struct MlParserDev {
    int type, negate, nrules;
    uint32_t start_mask;                 // rules with a start_state among their from_states
    uint32_t cont_mask[16];              // by rule_to_state: the non-start rules its to_state leads to (rule order = to_state_map order)
    uint32_t flush_after;                // rules whose to_state leads to a start rule (try_flushing_buffer)
    int8_t look_idx[16];                 // rule "^(?!A)B" / "^(?=A)B": index of the automaton of ^(?:A) in rules[] (behind the nrules B parts), -1 none
    uint32_t look_neg;                   // ... the look-ahead is negative
    uint32_t match_len;
    uint8_t match_str[256];              // ENDSWITH / EQ
    uint32_t has_key_content, key_len;
    uint8_t key[264];                    // the packed msgpack string of key_content | "log"
    uint64_t buffer_limit;               // 0: none
};
struct MlMisc {                          // device words of one call
    unsigned long long lead, total, records, truncated, trunc_k;
    unsigned int final_state, new_carry_len, new_tail, first_reg, refused, last_ba, has_open, open_first, new_carry_trunc, nslow;
};
struct MlArgs {
    MlParserDev p;
    const GrepRule *rules;               // [nrules] match-only DFA + UTF-8 tables per rule (key unused)
    // product of the rules' match-only DFAs (ml.cpp build_product): ONE table walk per line answers which rules match.
    // blob = joint class of every byte [256] | rules matching if the line ended in the state, u16 [nS] | next state u16 [nS][nj]
    // (0xFFFF: a byte >= 0x80 -- the per-rule UTF-8 tables decide); states >= prod_T are absorbing (every rule has matched or cannot any more)
    const uint8_t *prod; uint32_t prod_bytes, prod_nj, prod_nS, prod_T, prod_init;
    const uint8_t *text; uint64_t bytes;
    const uint64_t *nl_pos; uint64_t nl; // raw lines
    int skip_empty_lines, flush_all;
    uint64_t NB;                         // allocated items (nl + 1); the real count is koff[nl] + 1
    uint32_t *keep; const uint64_t *koff;
    uint64_t *ls; uint32_t *ll;          // [NB] line start / length (CR of a CR LF dropped)
    uint32_t *info;                      // [NB] rule match bits | last byte << 16 | non-empty << 24
    uint64_t *F;                         // [NB] transition functions
    uint8_t *sin;                        // [NB] state entering the item
    uint8_t *act;                        // [NB]
    uint32_t *c; const uint64_t *coff;   // [NB] bytes the item adds to its group's buffer, their scan
    uint32_t *head; const uint64_t *gidx;// [NB] first item of a group, scan = group index
    uint64_t *ghead;                     // [groups + 1] first item of every group
    uint32_t *plen; const uint64_t *po;  // [NB] bytes the item writes into the output, their scan
    uint32_t *pk; uint32_t *gC;          // [NB] MLP_* / content length of the group (first items)
    uint32_t *slow;                      // [NB] lines the product automaton left to the per-rule tables (a byte >= 0x80 before all rules were decided)
    uint32_t *ovr;                       // [NB] continuations pinned as truncating: clipped bytes << 1 | separator (0xFFFFFFFF: not pinned)
    const uint8_t *carry; uint32_t carry_len, carry_tail, carry_state, carry_trunc;
    uint32_t carry_sec, carry_nsec, ts_sec, ts_nsec;
    uint8_t *carry_out;
    uint8_t *out;
    MlMisc *misc;
};
void launch_ml_keep(const MlArgs &a, hipStream_t st);
void launch_ml_compact(const MlArgs &a, hipStream_t st);
void launch_ml_match(const MlArgs &a, int cus, hipStream_t st);
size_t ml_fscan_tmp_bytes(uint64_t n);
void launch_ml_fscan(const uint64_t *F, uint64_t n, uint32_t init, uint8_t *sin, void *tmp, unsigned int *final_state, hipStream_t st);
void launch_ml_act(const MlArgs &a, hipStream_t st);
void launch_ml_ghead(const MlArgs &a, hipStream_t st);
void launch_ml_reset(MlMisc *m, hipStream_t st);
void launch_ml_trunc(const MlArgs &a, hipStream_t st);
void launch_ml_override(const MlArgs &a, uint64_t k, hipStream_t st);
void launch_ml_piece(const MlArgs &a, hipStream_t st);
void launch_ml_rows(const MlArgs &a, uint64_t *row_off, hipStream_t st);
void launch_ml_emit(const MlArgs &a, int cus, hipStream_t st);
void launch_ml_carry(const MlArgs &a, hipStream_t st);

struct GatherArgs {
    const uint8_t *data;
    const uint64_t *row_off;
    uint64_t n;
    const uint32_t *keep_len;
    const uint64_t *out_off;
    uint8_t *out;
    uint64_t out_cap;                // as in ParserEmitArgs
};

// progress of the device-side record indexer (index_kernels.inc)
struct IdxState {
    unsigned long long start_idx;        // chain kernels: candidate the chain (re)starts from
    unsigned long long pos;              // k_idx_one: byte position to walk from
    unsigned long long consumed;         // bytes covered by whole records once status is final
    unsigned int kind;                   // how the chain ended (IDX_EOF / LAND / INVALID / LONG), IDX_NONE while open
    unsigned int term;                   // candidate index it ended on
    unsigned int n_extras;               // rows that are not candidates
    unsigned int status;                 // 0 chain runs from start_idx, 1 finished (consumed valid), 2 side list full, 3 k_idx_one runs from pos
};

}  // namespace flbgpu

// launchers implemented in kernels.hip
#include <hip/hip_runtime_api.h>
namespace flbgpu {
bool upload_time_tables();
void launch_parser_locate(const ParserMatchArgs &a, int cus, hipStream_t st);
void launch_parser_rx(const ParserMatchArgs &a, int grid, int threads, hipStream_t st);
void launch_parser_finish(const ParserMatchArgs &a, int cus, hipStream_t st);
constexpr int TILE_BYTES = 17920;         // k_parser_tile: LDS bytes of a wave's record tile (64 records of 277 B + slack)
void launch_parser_tile(const ParserMatchArgs &a, int grid, int threads, hipStream_t st);
void launch_parser_reg(const ParserMatchArgs &a, int grid, int threads, bool fixup, hipStream_t st);
void launch_parser_generic(const ParserMatchArgs &a, int grid, hipStream_t st);
void launch_count_nonzero(const uint32_t *len, uint64_t n, unsigned long long *out, hipStream_t st);
constexpr int MATCH_BLOCK = 1024;         // threads per workgroup of k_parser_match
void launch_parser_emit(const ParserEmitArgs &a, int cus, hipStream_t st);
void launch_parser_emit_exact(const ParserEmitArgs &a, hipStream_t st);
void launch_pg_decide(const PgDecideArgs &a, int cus, hipStream_t st);
void launch_pg_emit(const PgEmitArgs &a, int nfields, int cus, hipStream_t st);
void launch_pjson_size(const ParserMatchArgs &a, int cus, hipStream_t st);
void launch_pjson_size_generic(const ParserMatchArgs &a, hipStream_t st);
void launch_grep_match(const GrepArgs &a, int cus, hipStream_t st);
void launch_gather(const GatherArgs &a, hipStream_t st);
// the end of a call launched ahead of its sizes: counters + output size into page-locked host words and, when `sink` is given and the
// output fits, the output into the page-locked slab -- written by the device, no copy command that would need the size on the host
void launch_call_prep(void *words, uint32_t words_bytes, uint8_t *tail, uint32_t tail_room, const uint8_t *data, uint64_t bytes, uint32_t tail_bytes,
                      uint32_t *keep_len, uint64_t n, hipStream_t st);
void launch_finish_to_host(const uint8_t *out, uint64_t out_cap, const uint64_t *total, uint8_t *sink, uint64_t sink_cap, const void *words, void *host_words,
                           uint32_t nwords_bytes, uint64_t *host_total, hipStream_t st);
size_t scan_tmp_elems(uint64_t n);
// exclusive scan u32 -> u64 (out[n] = total); nonzero (optional) receives the number of non-zero inputs
void launch_scan(const uint32_t *in, uint64_t n, uint64_t *tmp, uint64_t *out, hipStream_t st, unsigned long long *nonzero = nullptr);
void launch_max_row_len(const uint64_t *row_off, uint64_t n, unsigned long long *out, hipStream_t st);
void launch_l2m_extract(const L2mArgs &a, int cus, hipStream_t st);
void launch_l2m_generic(const L2mArgs &a, hipStream_t st);
void launch_l2m_stale(uint32_t *sid_col, uint64_t *val_col, uint64_t n, const unsigned long long *first_bad, uint64_t *tmp, hipStream_t st);
size_t l2m_stale_tmp_elems(uint64_t n);
void launch_l2m_aggregate(const L2mAggArgs &a, int cus, hipStream_t st);
void launch_l2m_seqsum(const uint32_t *sid_col, const uint64_t *val_col, uint64_t n, const unsigned long long *first_bad, double *seq, uint32_t nseries,
                       hipStream_t st);
void launch_l2m_rehash(const L2mTable &t, uint32_t nseries, hipStream_t st);
void launch_json_size(const JsonArgs &a, int cus, hipStream_t st);
void launch_json_emit(const JsonArgs &a, int cus, hipStream_t st);
void launch_json_generic(const JsonArgs &a, bool emit, hipStream_t st);
size_t idx_tiles(uint64_t bytes);
size_t idx_blocks(uint64_t nc);
void launch_idx_count(const uint8_t *data, uint64_t bytes, uint64_t *masks, uint32_t *tile_cnt, hipStream_t st);
void launch_idx_fill(const uint64_t *masks, uint64_t bytes, const uint64_t *tile_off, uint64_t *cand_pos, hipStream_t st);
void launch_idx_walk(const uint8_t *data, uint64_t bytes, const uint64_t *cand_pos, uint64_t nc, uint32_t *rec_len, uint32_t *succ, hipStream_t st);
void launch_idx_exit(const uint32_t *succ, uint64_t nc, uint32_t *exitp, hipStream_t st);
void launch_idx_chain_mark(const uint32_t *succ, const uint32_t *rec_len, const uint64_t *cand_pos, uint64_t nc, const uint32_t *exitp,
                           uint32_t *entry, uint32_t *flags, uint64_t bytes, IdxState *state, hipStream_t st);
void launch_idx_one(const uint8_t *data, uint64_t bytes, const uint64_t *cand_pos, uint64_t nc, uint64_t *extras, uint32_t extras_cap, IdxState *state,
                    hipStream_t st);
void launch_idx_emit(const uint64_t *cand_pos, uint64_t nc, const uint32_t *flags, const uint64_t *off, const uint64_t *extras, uint32_t n_extras,
                     uint64_t consumed, uint64_t *row_off, hipStream_t st);
bool l2m_test_numconv(const char *strs, const uint32_t *off, uint32_t n, int mode, uint64_t *bits, int *status);

}  // namespace flbgpu
