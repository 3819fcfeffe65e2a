// kernels.hip -- hand-written HIP kernels (gfx950 / MI355X) for the Fluent Bit filter hot path.
//
// Layout in HBM: a chunk is the reference wire format itself (concatenated msgpack log events,
// src/flb_log_event_encoder.c:195-217) as one contiguous byte column plus a row-offset column
// (u64 [n+1]).  One record is processed per lane; the automaton tables (byte classes, reverse
// DFA, viable-position bit sets, priority lists) are staged in LDS and stepped one input byte
// per lane.  Nothing here is a dense contraction, so there is no MFMA: the bound is HBM/LDS.
//
//   k_parser_locate / k_parser_rx / k_parser_finish (/ k_parser_generic)
//                    filter_parser pass 1 in phases: decode event + locate Key_Name, run the
//                    capture program, parse the time field and compute the output size
//                    (plugins/filter_parser/filter_parser.c:226-323, src/flb_parser_regex.c:114-227)
//   k_parser_emit    filter_parser pass 2: write the V2 record at its scanned offset
//                    (filter_parser.c:325-413, src/flb_log_event_encoder.c:195-217)
//   k_grep_match     filter_grep: rule evaluation -> keep flag / kept length
//                    (plugins/filter_grep/grep.c:167-194,250-284, src/flb_ra_key.c:374-434)
//   k_gather         copy of the kept records (flb_log_event_encoder_emit_raw_record)
//   k_scan_*         exclusive prefix sums (u32 -> u64) used for the write offsets
//   k_index_*        record boundary discovery helpers
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "dec.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "dec_dev.inc"

// ------------------------------------------------------------------------------------------
// filter_parser pass 1 is split into phase kernels so that every wave of a launch runs the same
// small piece of code (register pressure and instruction-cache footprint of one phase only):
//
//   k_parser_locate   decode the event, find the value Key_Name designates, size the record as
//                     if it stayed unparsed (records staged through LDS tiles, coalesced reads)
//   k_parser_rx       parser 0's capture program on the located value (tables + spans in LDS)
//   k_parser_finish   named fields, time lookup, size of the parsed record
//   k_parser_generic  everything the fast phases do not cover (UTF-8 input, several parsers or
//                     candidate keys, huge values): the complete per-record algorithm, run only
//                     on the records flagged RF_GENERIC (normally none)
// ------------------------------------------------------------------------------------------
constexpr int LOC_BLOCK = 256;
constexpr int LOC_TILE = 18432;             // LDS bytes per wave (64 records of 277 B + slack)


DEV uint32_t locate_one(const ParserMatchArgs &a, uint64_t r, const uint8_t *rec, const uint8_t *rec_end, uint32_t &n_dec) {
    RecInfo ri;
    const uint32_t out_len = locate_core(a, r, rec, rec_end, n_dec, ri);
    rec_store(a.info, a.n, r, ri);
    return out_len;
}

__global__ void __launch_bounds__(LOC_BLOCK) k_parser_locate(ParserMatchArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *tile = (LDS_AS uint8_t *) g_lds + (size_t) wave * LOC_TILE;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    const uint8_t *data_end = a.data + a.bytes;
    uint32_t n_dec = 0;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        const uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.row_off[r]; o1 = a.row_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            const uint64_t g0 = __shfl(o0, (int) lo, 64);
            const uint32_t align = (uint32_t) (g0 & 15);
            const bool fit = lane >= lo && lane < cnt && (o1 - g0 + align) <= (uint64_t) LOC_TILE;
            const uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            const bool direct = (m == 0);                 // one record larger than the tile: parse it in place
            if (direct) m = 1;
            if (!direct) {
                const uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - g0) + align;
                const uint8_t *src = a.data + (g0 - align);
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                for (uint32_t u = lane; u * 16 < total; u += 64) {
                    const uint8_t *p = src + (size_t) u * 16;
                    v4 v;
                    if (p + 16 <= data_end) v = *(const v4 *) p;
                    else {
                        uint32_t t4[4] = {0, 0, 0, 0};
                        for (int q = 0; q < 16 && p + q < data_end; q++) t4[q >> 2] |= (uint32_t) p[q] << (8 * (q & 3));
                        v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
                    }
                    *(LDS_AS v4 *) (tile + (size_t) u * 16) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            if (lane >= lo && lane < lo + m) {
                const uint8_t *rec, *rec_end;
                if (direct) { rec = a.data + o0; rec_end = a.data + o1; }
                else { rec = (const uint8_t *) (tile + align + (uint32_t) (o0 - g0)); rec_end = rec + (o1 - o0); }
                a.out_len[r] = locate_one(a, r, rec, rec_end, n_dec);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lo += m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_dec += __shfl_down(n_dec, o, 64);
    if (lane == 0 && n_dec) atomicAdd(&a.counts[0], (unsigned long long) n_dec);
}

// parser 0's capture program on the located values
template <bool LDS>
__global__ void __launch_bounds__(MATCH_BLOCK) k_parser_rx(ParserMatchArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint16_t *chk = a.chk + ((size_t) wave_slot * a.chk_len) * 64 + lane;
    const DevParser &ps = a.parsers[0];
    // stage the hot ASCII tables into LDS (one coalesced 16 B/lane copy per workgroup)
    HotTabs<LDS> hot0;
    if constexpr (LDS) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u *src = (const v4u *) ps.ascii.hot_base;
        LDS_AS v4u *dst = (LDS_AS v4u *) g_lds;
        for (uint32_t i = threadIdx.x; i < a.lds_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        hot0 = hot_lds(ps.ascii, (LDS_AS uint8_t *) g_lds);
    }
    else hot0 = hot_global(ps.ascii);
    CapLds capl;
    capl.base = (LDS_AS uint8_t *) (g_lds + a.caps_lds_off) + 2 * threadIdx.x;
    capl.stride_b = 2 * blockDim.x;
    const int ncap = 2 * ps.nfields;
    uint32_t n_gen = 0;
    // pair mode: the match-only DFAs of grep's rules next to the tables
    LDS_AS uint8_t *pg_lds = (LDS_AS uint8_t *) g_lds + a.pg_lds_off;
    if (a.pg) {
        for (int i = 0; i < a.pg->nrules; i++) {
            if (a.pg->rule_lds_off[i] == 0xFFFFFFFFu) continue;
            const uint8_t *src = a.pg->rules[i].dfa.cls;
            for (uint32_t k = threadIdx.x; k < a.pg->rule_lds_bytes[i]; k += blockDim.x) pg_lds[a.pg->rule_lds_off[i] + k] = src[k];
        }
        __syncthreads();
    }
    for (uint64_t base = (uint64_t) wave_slot * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        if (r >= a.n) continue;
        const uint32_t flags = a.info[r];                     // column 0
        if (!(flags & RF_CAND)) continue;
        const uint32_t vlen = a.info[2 * a.n + r];
        const uint8_t *val = a.data + a.row_off[r] + a.info[1 * a.n + r];
        if (ps.ascii.stub) {
            // no ASCII tables for this pattern (rx.cpp make_ascii_stub): every value, the empty one too, is the second engine's
            a.info[r] = flags | RF_GENERIC;
            n_gen++;
            continue;
        }
        // phase 0: forward walk from boundary 0 with no reverse pass (start-anchored patterns: a match
        // that starts at 0 is the leftmost one, and every choice the walk makes is forced by the
        // byte / the next byte whenever a match exists); phase 1: reverse pass; phase 2: forward
        // walk from the leftmost viable start.  One call site each keeps the kernel small.
        int phase = (ps.fwd_first && ps.nregs_minus1 > 0) ? 0 : 1;
        int best = 0, endb = -1;
        for (;;) {
            if (phase == 1) {
                best = rx_reverse(hot0, ps.ascii.r_info, val, vlen, chk);
                if (best < 0 || ps.nregs_minus1 <= 0) break;
                phase = 2;
            }
            for (int c = 0; c < ncap; c++) capl.set((uint32_t) c, CAP_UNSET);
            endb = rx_forward(ps.ascii, hot0, val, vlen, phase == 0 ? 0 : best, phase == 0 ? (const uint16_t *) nullptr : chk, ps.slot2cap, capl);
            if (endb >= 0 || phase == 2) break;
            phase = 1;
        }
        if (endb >= 0) {
            // publish the spans: [span][record] columns, a wave stores 64 consecutive words
            for (int c = 0; c < ncap; c++) a.caps[(uint64_t) c * a.n + r] = capl.get((uint32_t) c);
            uint32_t pgbits = 0;
            if (a.pg) {
                // filter_grep's rules on the record filter_parser will emit, while the spans are in LDS and the value
                // bytes in cache.  Fields dropped as empty (Skip_Empty_Values) or consumed (the time field without
                // Time_Keep) are known here; a rule that names a time field that is kept waits for the time lookup.
                uint32_t drop = a.pg->static_drop, any = 0;
                for (int f = 0; f < ps.nfields; f++) {
                    const uint32_t b = capl.get((uint32_t) (2 * f)), e = capl.get((uint32_t) (2 * f + 1));
                    const bool set = b != CAP_UNSET && e != CAP_UNSET;
                    any |= set ? 1u : 0u;
                    if ((!set || e == b) && ps.skip_empty) drop |= 1u << f;
                }
                uint32_t named = 0;
                for (int i = 0; i < a.pg->nrules; i++) named |= a.pg->rule_fmask[i];
                if (any && !(named & a.pg->time_fields & ~drop)) {
                    struct { const CapLds *c; DEV uint32_t operator[](uint32_t i) const { return c->get(i); } } cv{&capl};
                    const bool keep = pg_grep_parsed<true>(a.pg->rules, a.pg->nrules, a.pg->logical_op, a.pg->rule_fmask, a.pg->rule_lds_off, drop, val, cv,
                                                     (LDS_AS const uint8_t *) pg_lds);
                    pgbits = RF_PGDONE | (keep ? RF_PGKEEP : 0u);
                }
            }
            a.info[r] = flags | RF_RXOK | pgbits;
            if (ps.time_field >= 0) {
                // the time text (cache-hot here) goes to its own coalesced column
                const uint32_t tb = capl.get((uint32_t) (2 * ps.time_field)), te = capl.get((uint32_t) (2 * ps.time_field + 1));
                if (tb != CAP_UNSET && te != CAP_UNSET && te - tb <= 4 * TBUF_WORDS) {
                    const uint32_t tl = te - tb;
                    v4u32 w0 = load16(val + tb, 0, tl), w1 = load16(val + tb + 16, 0, tl > 16 ? tl - 16 : 0);
                    a.tbuf[0 * a.n + r] = w0.x; a.tbuf[1 * a.n + r] = w0.y; a.tbuf[2 * a.n + r] = w0.z; a.tbuf[3 * a.n + r] = w0.w;
                    a.tbuf[4 * a.n + r] = w1.x; a.tbuf[5 * a.n + r] = w1.y; a.tbuf[6 * a.n + r] = w1.z; a.tbuf[7 * a.n + r] = w1.w;
                }
            }
        }
        else if (best == -2 || a.cfg.nparsers > 1) {
            // a byte >= 0x80 (UTF-8 tables) or more parsers to try: the generic kernel takes over
            a.info[r] = flags | RF_GENERIC;
            n_gen++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_gen += __shfl_down(n_gen, o, 64);
    if (lane == 0 && n_gen) atomicAdd(&a.counts[2], (unsigned long long) n_gen);
}

// named fields, time lookup and size of the records parser 0 matched
constexpr int FIN_SLOT = 4 * TBUF_WORDS + 4;      // per-lane LDS slot for the time text (+4: spreads the banks)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_parser_finish(ParserMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t tls_mem[256 * FIN_SLOT];
    __shared__ char fmt_mem[2 * MAX_TIMEFMT];
    LDS_AS uint8_t *tls = (LDS_AS uint8_t *) tls_mem;
    const DevParser &ps = a.parsers[0];
    // the format strings are walked character by character for every record: from LDS, not through
    // a chain of dependent global loads
    for (uint32_t i = threadIdx.x; i < 2 * MAX_TIMEFMT; i += blockDim.x) fmt_mem[i] = i < MAX_TIMEFMT ? ps.fmt1[i] : ps.fmt2[i - MAX_TIMEFMT];
    __syncthreads();
    LDS_AS const char *lfmt1 = (LDS_AS const char *) fmt_mem, *lfmt2 = lfmt1 + MAX_TIMEFMT;
    uint32_t n_gen = 0, pg_out = 0, pg_keep = 0, pg_pending = 0;
    unsigned long long pg_bytes = 0;
    // When the record's size does not depend on bytes of the chunk (no reserved / preserved kvs, no
    // Types cast) and the time text sits in the tbuf column, this kernel reads and writes nothing
    // but coalesced columns.
    const bool size_from_columns = !a.cfg.reserve_data && !(a.cfg.preserve_key && !a.cfg.key.is_ra) && ps.plain_types;
    const uint64_t n = a.n;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t fl0 = a.info[r];
        if (fl0 & RF_PARSED) continue;                         // completed by k_parser_tile (nothing else sets the flag before this kernel)
        if (!(fl0 & RF_RXOK) || (fl0 & RF_GENERIC)) {
            if (!(fl0 & RF_GENERIC)) a.null_mask[r] = 0;
            if (a.pg_keep_len) { a.pg_keep_len[r] = PG_UNDECIDED; pg_pending++; }
            continue;
        }
        CapsView caps;
        caps.base = a.caps; caps.n = n; caps.r = r;
        bool any = false;
        uint32_t kept = 0, drop = 0, body_bytes = 0;
        int64_t sec = 0; double frac = 0;
        for (int f = 0; f < ps.nfields; f++) {
            uint32_t b = caps[2 * f], e = caps[2 * f + 1];
            bool set = (b != CAP_UNSET && e != CAP_UNSET);
            if (set) any = true;                               // last_pos (src/flb_regex.c:52-54)
            uint32_t fl = set ? e - b : 0;
            if (fl == 0 && ps.skip_empty) { drop |= 1u << f; continue; }
            if (ps.field_is_time[f]) {
                int64_t s2; double f2;
                TStr in;
                const uint8_t *tv;
                if (f == ps.time_field && fl <= 4 * TBUF_WORDS) {
                    LDS_AS uint32_t *slot = (LDS_AS uint32_t *) (tls + threadIdx.x * FIN_SLOT);
                    #pragma unroll
                    for (int k = 0; k < TBUF_WORDS; k++) slot[k] = fl > (uint32_t) (4 * k) ? a.tbuf[(uint64_t) k * n + r] : 0;
                    in.lds = (LDS_AS const uint8_t *) slot; in.in_lds = true;
                    tv = (const uint8_t *) 4096;               // position origin only, never dereferenced
                }
                else {
                    const uint8_t *val = a.data + a.row_off[r] + a.info[1 * n + r];
                    tv = set ? val + b : val;
                }
                if (in.in_lds && ps.plan.ok && fl == (uint32_t) ps.plan.len && time_fast(ps, in.lds, &s2)) f2 = 0;
                else if (time_lookup_in(ps, in, tv, fl, &s2, &f2, lfmt1, lfmt2) == -1) { drop |= 1u << f; continue; }
                sec = s2; frac = f2;
                if (!ps.time_keep) { drop |= 1u << f; continue; }
            }
            kept++;
            const uint32_t nl = (uint32_t) ps.field_name_len[f];
            body_bytes += (nl < 32 ? 1 : nl < 256 ? 2 : nl < 65536 ? 3 : 5) + nl + (fl < 32 ? 1 : fl < 256 ? 2 : fl < 65536 ? 3 : 5) + fl;
        }
        if (!any) {
            // flb_regex_parse found no participating named group: the parser fails
            if (a.cfg.nparsers > 1) { a.info[r] = fl0 | RF_GENERIC; n_gen++; }
            else a.null_mask[r] = 0;
            if (a.pg_keep_len) { a.pg_keep_len[r] = PG_UNDECIDED; pg_pending++; }
            continue;
        }
        uint32_t flags = (fl0 | RF_PARSED) & ~(uint32_t) RF_BADTS;
        const uint32_t key_index = a.info[3 * n + r];
        uint64_t null_mask = 0;
        if (!a.cfg.key.is_ra) null_note(a.cfg, r, key_index, null_mask);
        int64_t tsec = a.info[4 * n + r], tnsec = a.info[5 * n + r];
        if (fl0 & RF_BADTS) { tsec = -3; tnsec = 0; }                        // the event time was out of range
        int64_t psec = sec, pnsec = (int64_t) (frac * 1000000000);
        bool have_parsed_time = ((uint64_t) psec * 1000000000ull + (uint64_t) pnsec) != 0;
        if (have_parsed_time) { tsec = psec; tnsec = pnsec; }
        a.null_mask[r] = null_mask;
        a.info[10 * n + r] = 0;                                // parser_idx
        a.info[11 * n + r] = kept;
        a.info[12 * n + r] = drop;
        // encoder timestamp check (src/flb_log_event_encoder.c:345-363)
        if (encoder_refuses_time(tsec, tnsec)) {
            a.info[r] = flags | RF_BADTS;
            a.out_len[r] = 0;
            if (a.pg_keep_len) a.pg_keep_len[r] = 0;
            continue;
        }
        a.info[r] = flags;
        a.info[4 * n + r] = (uint32_t) tsec; a.info[5 * n + r] = (uint32_t) tnsec;
        // a parsed time the next decoder reads as a group marker (tile_kernels.inc k_parser_reg): counted
        if ((uint32_t) tsec >= 0xfffffffeu) atomicAdd(&a.counts[14], 1ull);
        if (size_from_columns) {
            // 92 92 d7 00 ts(8) + metadata + map header (width of the ORIGINAL count) + fields
            const uint32_t n0 = (uint32_t) ps.nregs_minus1;
            a.out_len[r] = 12 + a.info[13 * n + r] + (n0 < 16 ? 1 : n0 < 65536 ? 3 : 5) + body_bytes;
        }
        else {
            RecInfo ri = rec_load(a.info, n, r);
            const uint8_t *rec = a.data + a.row_off[r];
            CountSink cs;
            write_record(cs, a.cfg, a.parsers, rec, a.data + a.row_off[r + 1], ri, caps, null_mask);
            a.out_len[r] = (uint32_t) cs.n;
            if (cs.need_exact) { a.info[r] = flags | RF_EXACT; atomicAdd(&a.counts[3], 1ull); }
        }
        if (a.pg_keep_len) {
            // pair mode: the record's fate under filter_grep, settled by k_parser_rx on the spans (else k_pg_decide's)
            const uint32_t ol = a.out_len[r];
            if (fl0 & RF_PGDONE) {
                const bool keep = (fl0 & RF_PGKEEP) != 0;
                a.pg_keep_len[r] = keep ? ol : 0;
                pg_out++; pg_bytes += ol; pg_keep += keep ? 1u : 0u;
            }
            else { a.pg_keep_len[r] = PG_UNDECIDED; pg_pending++; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        n_gen += __shfl_down(n_gen, o, 64);
        pg_out += __shfl_down(pg_out, o, 64); pg_keep += __shfl_down(pg_keep, o, 64); pg_pending += __shfl_down(pg_pending, o, 64);
        pg_bytes += __shfl_down(pg_bytes, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_gen) atomicAdd(&a.counts[2], (unsigned long long) n_gen);
        if (pg_bytes) atomicAdd(&a.counts[4], pg_bytes);
        if (pg_keep) atomicAdd(&a.counts[5], (unsigned long long) pg_keep);
        if (pg_out) atomicAdd(&a.counts[6], (unsigned long long) pg_out);
        if (pg_pending) atomicAdd(&a.counts[7], (unsigned long long) pg_pending);
    }
}

// The complete per-record algorithm (every candidate key, every parser, UTF-8 tables, values of
// any length), for the records the fast phases flagged RF_GENERIC.
// The wide json walker (4096-level stack, exact decimals) is big: one out-of-line copy serves the
// slow-path kernels instead of one inlined copy per call site.
__device__ __noinline__ void pjson_try_wide(const DevParser *ps, const uint8_t *v, uint32_t vlen, uint32_t *caps_base, uint64_t n, uint64_t r,
                                            LDS_AS uint8_t *slot, PjsonTry *out) {
    JsonCounts cc;
    cc.col = caps_base; cc.n = n; cc.r = r; cc.store = true; cc.next = 0;
    *out = pjson_try<true, JSON_GENERIC_WORDS>(*ps, v, vlen, cc, slot);
}
__device__ __noinline__ uint32_t size_record_wide(const FParserCfg *cfg, const DevParser *parsers, const uint8_t *rec, const uint8_t *rec_end,
                                                  const RecInfo *ri, const uint32_t *caps_base, uint64_t n, uint64_t r, uint64_t null_mask) {
    CapsView caps;
    caps.base = caps_base; caps.n = n; caps.r = r;
    CountSink cs;
    write_record<true, JSON_GENERIC_WORDS>(cs, *cfg, parsers, rec, rec_end, *ri, caps, null_mask);
    return (uint32_t) cs.n;
}

__global__ void __launch_bounds__(256) k_parser_generic(ParserMatchArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t gen_slots[256 * 64];       // per-lane scratch of a json parser's time text
    LDS_AS uint8_t *slot = (LDS_AS uint8_t *) gen_slots + threadIdx.x * 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint16_t *chk = a.chk + ((size_t) wave_slot * a.chk_len) * 64 + lane;
    for (uint64_t base = (uint64_t) wave_slot * 64; base < a.n; base += nwaves * 64) {
        uint64_t r = base + lane;
        if (r >= a.n) continue;
        if (!(a.info[r] & RF_GENERIC)) continue;
        const uint8_t *rec = a.data + a.row_off[r];
        const uint8_t *rec_end = a.data + a.row_off[r + 1];
        RecInfo ri;
        recinfo_init(ri);
        uint64_t null_mask = 0;
        CapsView caps;
        caps.base = a.caps; caps.n = a.n; caps.r = r;
        Event ev = decode_event(rec, rec_end);
        ri.flags = ev.flags;                                  // valid, not skipped (checked by locate)
        ri.body_off = (uint32_t) (ev.body - rec); ri.body_len = (uint32_t) (ev.body_end - ev.body);
        if (ev.meta) { ri.meta_off = (uint32_t) (ev.meta - rec); ri.meta_len = (uint32_t) (ev.meta_end - ev.meta); }
        int64_t tsec = ev.sec, tnsec = ev.nsec;
        bool have_out = false, last_ok = false;
        Tok bm = mp_tok(ev.body, ev.body_end);
        const uint8_t *p = bm.next;
        uint32_t nkv = a.cfg.key.is_ra ? 1 : bm.len;
        for (uint32_t i = 0; i < nkv; i++) {
            const uint8_t *vptr = nullptr;
            uint32_t vlen = 0;
            if (a.cfg.key.is_ra) {
                const uint8_t *v = ra_resolve(a.cfg.key, ev.body, ev.body_end);
                if (v) {
                    Tok t = mp_tok(v, ev.body_end);
                    if (t.type == T_STR || t.type == T_BIN) { vptr = t.next; vlen = t.len; }
                }
            }
            else {
                Tok kt = mp_tok(p, ev.body_end);
                const uint8_t *kend = mp_skip(p, ev.body_end);
                Tok vt = mp_tok(kend, ev.body_end);
                p = mp_skip(kend, ev.body_end);
                if ((kt.type == T_STR || kt.type == T_BIN) && kt.len == (uint32_t) a.cfg.key.key_len &&
                    bytes_eq(kt.next, a.cfg.key.key, kt.len) && (vt.type == T_STR || vt.type == T_BIN)) { vptr = vt.next; vlen = vt.len; }
            }
            if (!vptr) continue;
            for (int q = 0; q < a.cfg.nparsers; q++) {
                int64_t ps = 0, pn = 0; uint32_t nk = 0, dm = 0;
                CapGlobal capg;
                capg.base = a.caps; capg.n = a.n; capg.r = r;
                if (a.parsers[q].is_json) {
                    // Format json / logfmt / ltsv in a list of parsers: the walkers of pjson_dev.inc / pkv_dev.inc
                    // (json with the wide stack and the exact decimals: such a record is emitted by k_parser_emit_exact)
                    PjsonTry t;
                    if (a.parsers[q].kv_format) t = pkv_try(a.parsers[q], vptr, vlen);
                    else pjson_try_wide(&a.parsers[q], vptr, vlen, a.caps, a.n, r, slot, &t);
                    last_ok = t.ok;
                    ps = t.sec; pn = t.nsec; nk = t.npairs; dm = t.skip;
                }
                else if (a.parsers[q].host_only)
                    last_ok = host_parser_answer(a.parsers[q], a.host_res[a.parsers[q].host_slot & (MAX_HOST_PARSERS - 1)], a.n, r, vptr, (uint32_t) (vptr - rec), capg, &ps, &pn, &nk, &dm);
                else
                last_ok = try_parser(a.parsers[q], hot_global(a.parsers[q].ascii), vptr, vlen, chk, a.chk_len, a.chk_nfa_off, capg, &ps, &pn, &nk, &dm, 0u);
                if (last_ok) {
                    have_out = true;
                    ri.val_off = (uint32_t) (vptr - rec); ri.val_len = vlen; ri.parser_idx = q; ri.nkept = nk; ri.drop_mask = dm;
                    if (!a.cfg.key.is_ra) {
                        ri.key_index = i;
                        null_note(a.cfg, r, i, null_mask);
                    }
                    if ((uint64_t) ps * 1000000000ull + (uint64_t) pn != 0) { tsec = ps; tnsec = pn; }
                    break;
                }
            }
        }
        if (have_out && last_ok) ri.flags |= RF_PARSED;
        // encoder timestamp check (src/flb_log_event_encoder.c:345-363)
        if (encoder_refuses_time(tsec, tnsec)) {
            ri.flags |= RF_BADTS;
            rec_store(a.info, a.n, r, ri); a.out_len[r] = 0; a.null_mask[r] = null_mask;
            continue;
        }
        ri.ts_sec = (uint32_t) tsec; ri.ts_nsec = (uint32_t) tnsec;
        if ((ri.flags & RF_PARSED) && (uint32_t) tsec >= 0xfffffffeu) atomicAdd(&a.counts[14], 1ull);     // (a group marker to the next decoder: tile_kernels.inc k_parser_reg)
        CountSink cs;
        const bool json_won = (ri.flags & RF_PARSED) && a.parsers[ri.parser_idx].is_json && !a.parsers[ri.parser_idx].kv_format;
        if (json_won) {
            cs.n = size_record_wide(&a.cfg, a.parsers, rec, rec_end, &ri, a.caps, a.n, r, null_mask);
            if (cs.n) cs.need_exact = true;                    // sized with the wide walker: emitted by the same one
        }
        else write_record(cs, a.cfg, a.parsers, rec, rec_end, ri, caps, null_mask);
        if (cs.need_exact) { ri.flags |= RF_EXACT; atomicAdd(&a.counts[3], 1ull); }
        rec_store(a.info, a.n, r, ri);
        a.null_mask[r] = null_mask;
        a.out_len[r] = (uint32_t) cs.n;
    }
}

// records with a non-empty output (flb_mp_count_log_records of the result)
__global__ void __launch_bounds__(256) k_count_nonzero(const uint32_t *len, uint64_t n, unsigned long long *out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) c += len[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long) c);
}

// Pass 2.  Each lane produces its record into a per-wave LDS staging area at the record's
// offset inside the wave's (contiguous) output range; the wave then flushes the staged bytes with
// 16 B per lane coalesced stores.  A record that does not fit the staging area is written directly.
constexpr int EMIT_BLOCK = 256;
constexpr int EMIT_STG = 18944;             // staging bytes per wave (64 records x 275 B + slack)

__global__ void __launch_bounds__(EMIT_BLOCK) k_parser_emit(ParserEmitArgs a) {
    if (a.out_cap && a.out_off[a.n] > a.out_cap) return;            // launched ahead of the size and the room was too small: the host runs the call again
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *stg = (LDS_AS uint8_t *) g_lds + (size_t) wave * EMIT_STG;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        uint64_t r = base + lane;
        uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.out_off[r]; o1 = a.out_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            uint64_t batch_base = __shfl(o0, (int) lo, 64);
            uint32_t align = (uint32_t) (batch_base & 15);
            bool fit = lane >= lo && lane < cnt && (o1 - batch_base + align) <= (uint64_t) EMIT_STG;
            uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            if (m == 0) {
                // one record larger than the staging area: straight to global memory
                if (lane == lo && o1 > o0 && !(a.info[r] & RF_DEC)) {
                    ByteSink s(a.out + o0);
                    CapsView cv;
                    cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
                    write_record(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], rec_load(a.info, a.n_cols, r),
                                 cv, a.null_mask[r]);
                }
                lo += 1;
                continue;
            }
            if (lane >= lo && lane < lo + m && o1 > o0 && !(a.info[r] & RF_DEC)) {      // (RF_DEC: k_parser_dec writes the row afterwards)
                LdsSink s(stg + align + (uint32_t) (o0 - batch_base));
                s.src_end = a.data + a.bytes;
                s.limit = s.p + (uint32_t) (o1 - o0);
                CapsView cv;
                cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
                const RecInfo ri = rec_load(a.info, a.n_cols, r);
                if (a.ec.ok && (ri.flags & RF_PARSED) && ri.parser_idx == 0) write_record_plain(s, a.ec, a.data + a.row_off[r], ri, cv);
                else write_record(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], ri, cv, a.null_mask[r]);
            }
            uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - batch_base);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staged bytes visible to the wave
            // flush [align, align + total) of the staging area to out[batch_base ...]
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            uint8_t *dst = a.out + (batch_base - align);               // 16 B aligned
            uint32_t n16 = (align + total + 15) / 16;
            for (uint32_t u = lane; u < n16; u += 64) {
                uint32_t b0 = u * 16, b1 = b0 + 16;
                if (b0 >= align && b1 <= align + total) *(v4u *) (dst + b0) = *(LDS_AS v4u *) (stg + b0);
                else {
                    uint32_t s0 = b0 < align ? align : b0, s1 = b1 > align + total ? align + total : b1;
                    for (uint32_t q = s0; q < s1; q++) dst[q] = stg[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staging area reusable
            lo += m;
        }
    }
}

// rows whose winning parser has decoders (dec_dev.inc): one row per lane, straight from / to global memory
__global__ void __launch_bounds__(64) k_parser_dec(DecArgs a) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t) gridDim.x * blockDim.x;
    uint8_t *rg = a.scratch + tid * (uint64_t) DEC_REGIONS * a.cap;
    for (uint64_t r = tid; r < a.e.n; r += nthreads) {
        const uint32_t fl = a.e.info[r];
        if (!(fl & RF_PARSED) || (fl & RF_BADTS)) continue;
        if (a.mode == 1 && (!(fl & RF_DEC) || a.e.out_off[r + 1] == a.e.out_off[r])) continue;
        RecInfo ri = rec_load(a.e.info, a.e.n_cols, r);
        if (!a.e.parsers[ri.parser_idx].decs) continue;
        CapsView cv;
        cv.base = a.e.caps; cv.n = a.e.n_cols; cv.r = r;
        const uint8_t *rec = a.e.data + a.e.row_off[r], *rec_end = a.e.data + a.e.row_off[r + 1];
        int64_t sec = ri.ts_sec, nsec = ri.ts_nsec;
        bool badts = false, ok;
        if (a.mode == 0) {
            CountSink cs;
            ok = dec_record(cs, a.e.cfg, a.e.parsers, rec, rec_end, ri, cv, a.e.null_mask[r], rg, a.cap, &sec, &nsec, &badts);
            if (!ok) { atomicAdd(a.err, 1ull); continue; }
            atomicAdd(a.ndec, 1ull);
            if (badts) { a.info_w[r] = fl | RF_BADTS; a.out_len_w[r] = 0; continue; }
            a.info_w[r] = fl | RF_DEC;
            a.info_w[4 * a.e.n_cols + r] = (uint32_t) sec; a.info_w[5 * a.e.n_cols + r] = (uint32_t) nsec;
            a.out_len_w[r] = (uint32_t) cs.n;
        }
        else {
            ByteSink s(a.e.out + a.e.out_off[r]);
            ok = dec_record(s, a.e.cfg, a.e.parsers, rec, rec_end, ri, cv, a.e.null_mask[r], rg, a.cap, &sec, &nsec, &badts);
            if (!ok) atomicAdd(a.err, 1ull);
        }
    }
}
void launch_parser_dec(const DecArgs &a, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_parser_dec, dim3((unsigned) blocks), dim3(64), 0, st, a);
}

// records whose Types float literal is a hard rounding case (RF_EXACT): rewritten in place with the
// big-integer conversion, one record per lane straight to global memory (rare)
__global__ void __launch_bounds__(64) k_parser_emit_exact(ParserEmitArgs a) {
    if (a.out_cap && a.out_off[a.n] > a.out_cap) return;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += (uint64_t) gridDim.x * blockDim.x) {
        if (!(a.info[r] & RF_EXACT) || a.out_off[r + 1] == a.out_off[r]) continue;
        ByteSink s(a.out + a.out_off[r]);
        CapsView cv;
        cv.base = a.caps; cv.n = a.n_cols; cv.r = r;
        write_record<true, JSON_GENERIC_WORDS>(s, a.cfg, a.parsers, a.data + a.row_off[r], a.data + a.row_off[r + 1], rec_load(a.info, a.n_cols, r), cv,
                           a.null_mask[r]);
    }
}

// ------------------------------------------------------------------------------------------
// filter_grep
// ------------------------------------------------------------------------------------------

constexpr int GREP_BLOCK = 256;
constexpr int GREP_TILE = 18432;            // LDS bytes per wave (64 records of 277 B + slack)

__global__ void __launch_bounds__(GREP_BLOCK) k_grep_match(GrepArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *tile = (LDS_AS uint8_t *) g_lds + (size_t) wave * GREP_TILE;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    const uint8_t *data_end = a.data + a.bytes;
    uint32_t n_dec = 0, n_keep = 0;
    // the rules' DFA blobs behind the tiles (one copy per workgroup)
    LDS_AS uint8_t *lds_rules = (LDS_AS uint8_t *) g_lds + (size_t) (GREP_BLOCK / 64) * GREP_TILE;
    if (a.rules_lds_total) {
        for (int i = 0; i < a.nrules; i++) {
            if (a.rule_lds_off[i] == 0xFFFFFFFFu) continue;
            const uint8_t *src = a.rules[i].dfa.cls;
            for (uint32_t k = threadIdx.x; k < a.rule_lds_bytes[i]; k += blockDim.x) lds_rules[a.rule_lds_off[i] + k] = src[k];
        }
        __syncthreads();
    }
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        const uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.row_off[r]; o1 = a.row_off[r + 1]; }
        uint32_t lo = 0;
        while (lo < cnt) {
            const uint64_t g0 = __shfl(o0, (int) lo, 64);
            const uint32_t align = (uint32_t) (g0 & 15);
            const bool fit = lane >= lo && lane < cnt && (o1 - g0 + align) <= (uint64_t) GREP_TILE;
            const uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            const bool direct = (m == 0);                 // one record larger than the tile: parse it in place
            if (direct) m = 1;
            if (!direct) {
                const uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - g0) + align;
                const uint8_t *src = a.data + (g0 - align);
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                for (uint32_t u = lane; u * 16 < total; u += 64) {
                    const uint8_t *p = src + (size_t) u * 16;
                    v4 v;
                    if (p + 16 <= data_end) v = *(const v4 *) p;
                    else {
                        uint32_t t4[4] = {0, 0, 0, 0};
                        for (int q = 0; q < 16 && p + q < data_end; q++) t4[q >> 2] |= (uint32_t) p[q] << (8 * (q & 3));
                        v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
                    }
                    *(LDS_AS v4 *) (tile + (size_t) u * 16) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            if (lane >= lo && lane < lo + m) {
                const uint8_t *rec, *rec_end;
                if (direct) { rec = a.data + o0; rec_end = a.data + o1; }
                else {
                    // generic pointer into the LDS aperture: the msgpack walkers are shared with
                    // the kernels that read from global memory
                    rec = (const uint8_t *) (tile + align + (uint32_t) (o0 - g0));
                    rec_end = rec + (o1 - o0);
                }
                Event ev = decode_event(rec, rec_end, true);
                bool keep = false;
                if (!(ev.flags & (RF_BAD | RF_SKIP))) {
                    bool valid = false;
                    keep = grep_decide(a, ev, &valid, a.rules_lds_total ? (LDS_AS const uint8_t *) lds_rules : (LDS_AS const uint8_t *) nullptr, r, rec);
                    if (!valid) valid = mp_skip(ev.body, rec_end, 1) == rec_end;  // no rule walked the map
                    if (!valid) ev.flags = RF_BAD;
                }
                else if ((ev.flags & RF_SKIP) && rec != rec_end && mp_skip(ev.body, rec_end, 1) != rec_end) ev.flags = RF_BAD;
                a.status[r] = ev.flags;
                if (ev.flags & RF_BAD) { atomicMin(a.first_bad, (unsigned long long) r); a.keep_len[r] = 0; }
                else if (ev.flags & RF_SKIP) a.keep_len[r] = 0;
                else {
                    a.keep_len[r] = keep ? (uint32_t) (o1 - o0) : 0;
                    n_dec++;
                    if (keep) n_keep++;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lo += m;
        }
    }
    // one pair of atomics per wave (same-address atomics serialise at ~12 ns each)
    for (int o = 32; o > 0; o >>= 1) { n_dec += __shfl_down(n_dec, o, 64); n_keep += __shfl_down(n_keep, o, 64); }
    if (lane == 0) {
        if (n_dec) atomicAdd(&a.counts[0], (unsigned long long) n_dec);
        if (n_keep) atomicAdd(&a.counts[1], (unsigned long long) n_keep);
    }
}

// copy of the kept records: one wave per record, byte granular

// One wave per 64 rows; the kept rows of the tile are copied FOUR at a time, each by a quarter of the wave with 16 B per lane
// (unaligned vector loads / stores: a 275 B record is two steps of its sixteen lanes).  Round 1 copied them one after the other with
// the whole wave -- twelve lanes busy for a 180 B event and up to 64 dependent load -> store round trips per tile (85 % of the wave
// cycles waiting); now a quarter of the trips, the four rows' loads in flight together.
__global__ void __launch_bounds__(256) k_gather(GatherArgs a) {
    if (a.out_cap && a.out_off[a.n] > a.out_cap) return;
    __shared__ uint8_t sel[4][64];                 // per wave: the lane that holds the j-th kept row of the tile
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, quarter = lane >> 4, ql = lane & 15;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef v4 v4un __attribute__((aligned(1)));
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        uint32_t len = 0;
        uint64_t so = 0, dof = 0;
        if (r < a.n) { len = a.keep_len[r]; if (len) { so = a.row_off[r]; dof = a.out_off[r]; } }
        const uint64_t mask = __ballot(len != 0);
        const uint32_t nk = (uint32_t) __builtin_popcountll(mask);
        if (len) sel[wave][__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint8_t) lane;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (uint32_t j0 = 0; j0 < nk; j0 += 4) {
            const uint32_t j = j0 + quarter;
            const bool on = j < nk;
            const int src_lane = on ? (int) sel[wave][j] : 0;
            const uint32_t l = __shfl(len, src_lane, 64);
            const uint8_t *src = a.data + __shfl(so, src_lane, 64);
            uint8_t *dst = a.out + __shfl(dof, src_lane, 64);
            if (on) {
                for (uint32_t o = ql * 16; o < l; o += 16 * 16) {
                    if (o + 16 <= l) *(v4un *) (dst + o) = *(const v4un *) (src + o);
                    else for (uint32_t q = o; q < l; q++) dst[q] = src[q];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// ------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64 (three small kernels; n up to 2^32)
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;            // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// tile_sums[b] = sum of tile b; tile_cnt (optional) [b] = its number of non-zero inputs
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile_sums(const uint32_t *in, uint64_t n, uint64_t *tile_sums, uint32_t *tile_cnt) {
    __shared__ uint64_t sh[SCAN_BLOCK / 64];
    uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
    uint32_t nz = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + (uint64_t) k * SCAN_BLOCK + threadIdx.x;
        if (i < n) { const uint32_t v = in[i]; s += v; nz += v != 0; }
    }
    if (tile_cnt) {
        uint32_t t = nz;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&tile_cnt[blockIdx.x], t);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) t += sh[w];
        tile_sums[blockIdx.x] = t;
    }
}

// n <= SCAN_SMALL: the whole scan in one workgroup (one launch instead of four stream commands: what a 2 MB call is made of)
constexpr uint32_t SCAN_SMALL = 32768;
__global__ void __launch_bounds__(1024) k_scan_small(const uint32_t *in, uint32_t n, uint64_t *out, unsigned long long *nonzero) {
    __shared__ uint64_t part[1024];
    __shared__ uint32_t cnt[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023) / 1024, b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
    uint64_t s = 0;
    uint32_t c = 0;
    for (uint32_t i = b; i < e; i++) { const uint32_t v = in[i]; s += v; c += v != 0; }
    part[t] = s; cnt[t] = c;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint64_t v = t >= o ? part[t - o] : 0;
        const uint32_t w = t >= o ? cnt[t - o] : 0;
        __syncthreads();
        part[t] += v; cnt[t] += w;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint32_t i = b; i < e; i++) { out[i] = run; run += in[i]; }
    if (t == 1023) { out[n] = part[1023]; if (nonzero && cnt[1023]) atomicAdd(nonzero, (unsigned long long) cnt[1023]); }
}

// counters, size and (when it fits) the output itself to page-locked host memory (launch_finish_to_host)
__global__ void __launch_bounds__(256) k_finish_to_host(const uint8_t *out, uint64_t out_cap, const uint64_t *total, uint8_t *sink, uint64_t sink_cap,
                                                        const uint32_t *words, uint32_t *host_words, uint32_t nwords, uint64_t *host_total) {
    const uint64_t t = *total;
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) host_words[i] = words[i];
        if (threadIdx.x == 0) *host_total = t;
    }
    if (!sink || t > sink_cap || t > out_cap) return;         // (over out_cap: the writer ended without writing)
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint64_t n16 = (t + 15) / 16;                 // (both buffers end on a whole 16 bytes)
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * blockDim.x)
        __builtin_nontemporal_store(((const v4 *) out)[i], (v4 *) sink + i);
}

// single block: exclusive scan of the tile sums in place; total written to tile_sums[ntiles]
__global__ void __launch_bounds__(1024) k_scan_spine(uint64_t *tile_sums, uint64_t ntiles, const uint32_t *tile_cnt, unsigned long long *nonzero) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    if (tile_cnt) {
        unsigned long long c = 0;
        for (uint64_t i = threadIdx.x; i < ntiles; i += 1024) c += tile_cnt[i];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(nonzero, c);
    }
    for (uint64_t base = 0; base < ntiles; base += 1024) {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < ntiles ? tile_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint64_t t = threadIdx.x >= (unsigned) o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        if (i < ntiles) tile_sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t *in, uint64_t n, const uint64_t *tile_sums, uint64_t *out) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { uint64_t i = base + k; v[k] = i < n ? in[i] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        uint64_t t = threadIdx.x >= (unsigned) o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = tile_sums[blockIdx.x] + sh[threadIdx.x] - s;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = tile_sums[gridDim.x];
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__global__ void k_max_row_len(const uint64_t *row_off, uint64_t n, unsigned long long *out) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        unsigned long long l = row_off[i + 1] - row_off[i];
        if (l > m) m = l;
    }
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(m, o, 64); if (t > m) m = t; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------
// the strptime directive table is a compile-time constant of every unit (kdev.inc)
bool upload_time_tables() { return true; }

void launch_parser_locate(const ParserMatchArgs &a, int cus, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    const size_t lds = (size_t) (LOC_BLOCK / 64) * LOC_TILE;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_locate, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + LOC_BLOCK / 64 - 1) / (LOC_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_locate, dim3((unsigned) blocks), dim3(LOC_BLOCK), lds, st, a);
}
void launch_parser_rx(const ParserMatchArgs &a, int grid, int threads, hipStream_t st) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_rx<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void) hipFuncSetAttribute((const void *) k_parser_rx<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (a.lds_bytes) hipLaunchKernelGGL(k_parser_rx<true>, dim3(grid), dim3(threads), a.lds_total, st, a);
    else hipLaunchKernelGGL(k_parser_rx<false>, dim3(grid), dim3(threads), a.lds_total, st, a);
}
void launch_parser_finish(const ParserMatchArgs &a, int cus, hipStream_t st) {
    uint64_t blocks = (a.n + 255) / 256, cap = (uint64_t) cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_finish, dim3((unsigned) blocks), dim3(256), 0, st, a);
}
void launch_parser_generic(const ParserMatchArgs &a, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_parser_generic, dim3(grid), dim3(256), 0, st, a);
}
void launch_count_nonzero(const uint32_t *len, uint64_t n, unsigned long long *out, hipStream_t st) {
    if (n == 0) return;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_count_nonzero, dim3((unsigned) blocks), dim3(256), 0, st, len, n, out);
}
void launch_parser_emit(const ParserEmitArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    static std::atomic<bool> attr_set{false};
    const size_t lds = (size_t) (EMIT_BLOCK / 64) * EMIT_STG;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_parser_emit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + EMIT_BLOCK / 64 - 1) / (EMIT_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;                 // a few waves of tiles per CU, grid-stride over the rest
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_parser_emit, dim3((unsigned) blocks), dim3(EMIT_BLOCK), lds, st, a);
}
void launch_grep_match(const GrepArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    static std::atomic<bool> attr_set{false};
    const size_t lds = (size_t) (GREP_BLOCK / 64) * GREP_TILE + a.rules_lds_total;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_grep_match, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + GREP_BLOCK / 64 - 1) / (GREP_BLOCK / 64);
    uint64_t cap = (uint64_t) cus * 2 * 4;                 // 2 resident workgroups per CU, a few rounds each
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_grep_match, dim3((unsigned) blocks), dim3(GREP_BLOCK), lds, st, a);
}
void launch_parser_emit_exact(const ParserEmitArgs &a, hipStream_t st) {
    uint64_t grid = (a.n + 63) / 64;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_parser_emit_exact, dim3((unsigned) grid), dim3(64), 0, st, a);
}
void launch_gather(const GatherArgs &a, hipStream_t st) {
    if (a.n == 0) return;
    uint64_t tiles = (a.n + 63) / 64, blocks = (tiles + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_gather, dim3((unsigned) blocks), dim3(256), 0, st, a);
}
// exclusive scan: out[0..n] (n+1 entries); tmp must hold ntiles+1 u64
// tmp: ntiles + 1 tile sums, then (u32) ntiles tile counts
size_t scan_tmp_elems(uint64_t n) { const size_t t = (size_t) ((n + SCAN_TILE - 1) / SCAN_TILE); return t + 2 + (t + 1) / 2 + 1; }
void launch_scan(const uint32_t *in, uint64_t n, uint64_t *tmp, uint64_t *out, hipStream_t st, unsigned long long *nonzero) {
    uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles == 0) { (void) hipMemsetAsync(out, 0, sizeof(uint64_t), st); return; }
    if (n <= SCAN_SMALL) { hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, in, (uint32_t) n, out, nonzero); return; }
    uint32_t *tile_cnt = nullptr;
    if (nonzero) {
        tile_cnt = (uint32_t *) (tmp + ntiles + 2);
        (void) hipMemsetAsync(tile_cnt, 0, ntiles * sizeof(uint32_t), st);
    }
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned) ntiles), dim3(SCAN_BLOCK), 0, st, in, n, tmp, tile_cnt);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, tmp, ntiles, (const uint32_t *) tile_cnt, nonzero);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned) ntiles), dim3(SCAN_BLOCK), 0, st, in, n, tmp, out);
}
// the start of a call launched ahead (flbgpu.cpp SpecCall), one launch for what are otherwise four or five copy / fill commands: the
// counter block zeroed (words[0..1] = first_bad = ~0), the zero-padded copy of the chunk's last bytes the register kernel reads its
// last rows from, the pair's keep_len column at "undecided"
__global__ void __launch_bounds__(256) k_call_prep(uint32_t *words, uint32_t nwords, uint8_t *tail, uint32_t tail_room, const uint8_t *data, uint64_t bytes,
                                                   uint32_t tail_bytes, uint32_t *keep_len, uint64_t n) {
    const uint64_t gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, gsz = (uint64_t) gridDim.x * blockDim.x;
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) words[i] = i < 2 ? 0xFFFFFFFFu : 0u;
    if (tail) for (uint64_t i = gid; i < tail_room; i += gsz) tail[i] = i < tail_bytes ? data[bytes - tail_bytes + i] : (uint8_t) 0;
    if (keep_len) for (uint64_t i = gid; i < n; i += gsz) keep_len[i] = 0xFFFFFFFFu;
}
void launch_call_prep(void *words, uint32_t words_bytes, uint8_t *tail, uint32_t tail_room, const uint8_t *data, uint64_t bytes, uint32_t tail_bytes,
                      uint32_t *keep_len, uint64_t n, hipStream_t st) {
    uint64_t blocks = keep_len ? (n + 1023) / 1024 : 1;
    if (blocks < 4) blocks = 4;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_call_prep, dim3((unsigned) blocks), dim3(256), 0, st, (uint32_t *) words, words_bytes / 4, tail, tail_room, data, bytes, tail_bytes, keep_len, n);
}
void launch_finish_to_host(const uint8_t *out, uint64_t out_cap, const uint64_t *total, uint8_t *sink, uint64_t sink_cap, const void *words, void *host_words,
                           uint32_t nwords_bytes, uint64_t *host_total, hipStream_t st) {
    // (the grid for the slab's size: the output's is not known here)
    unsigned blocks = sink ? 1024u : 1u;
    hipLaunchKernelGGL(k_finish_to_host, dim3(blocks), dim3(256), 0, st, out, out_cap, total, sink, sink_cap, (const uint32_t *) words, (uint32_t *) host_words,
                       nwords_bytes / 4, host_total);
}
void launch_max_row_len(const uint64_t *row_off, uint64_t n, unsigned long long *out, hipStream_t st) {
    (void) hipMemsetAsync(out, 0, sizeof(unsigned long long), st);
    if (n == 0) return;
    unsigned grid = (unsigned) ((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_max_row_len, dim3(grid), dim3(256), 0, st, row_off, n, out);
}

}  // namespace flbgpu
