// kernels_fmt.hip -- msgpack -> JSON output formatter on the device (fmt_dev.inc; shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "fmt_dev.inc"

// the group opener's body map for row r (nullptr outside groups)
DEV const uint8_t *fmt_group_attrs(const JsonFmtArgs &a, uint64_t r, const uint8_t **ga_end) {
    if (!a.g_row) return nullptr;
    const uint32_t g = a.g_row[r];
    if (!g) return nullptr;
    const uint8_t *rec = a.data + a.row_off[g - 1], *end = a.data + a.row_off[g];
    Event ge = decode_event(rec, end, true);
    *ga_end = end;
    return ge.body;
}

// bytes a row adds around its JSON object: '[' or ',' in front (json), '\n' behind (lines)
DEV uint32_t fmt_frame(const JsonFmtCfg &cfg) { return cfg.json_format == 1 || cfg.json_format == 3 ? 1 : 0; }

// pass 1: one row per lane (a wave takes 64 consecutive rows).  Decodes the event (src/flb_log_event_decoder.c), sizes
// its JSON object, reports the first row the decoder refuses and the first row the reference would fail on.
__global__ void __launch_bounds__(256) k_fmt_size(JsonFmtArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint8_t *src_end = a.data + a.bytes;
    uint32_t n_rec = 0, n_mark = 0, n_skip = 0;
    const uint64_t n_up = (a.n + 63) & ~63ull;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < n_up; r += (uint64_t) gridDim.x * blockDim.x) {
        uint32_t len = 0;
        bool slow = false;
        if (r < a.n) {
            const uint8_t *rec = a.data + a.row_off[r], *end = a.data + a.row_off[r + 1];
            Event ev = decode_event(rec, end, true);
            if (!(ev.flags & RF_VALID)) atomicMin(a.first_bad, (unsigned long long) r);
            else if (ev.flags & RF_SKIP) {
                // markers and skipped rows print nothing, but they are events the decoder has to accept
                if (rec != end) {
                    const uint8_t *be = mp_skip(ev.body, end, 1);
                    if (be != end) atomicMin(a.first_bad, (unsigned long long) r);
                    else { n_skip++; if (ev.sec == -1 || ev.sec == -2) n_mark++; }
                }
            }
            else {
                CountSink cs;
                const uint8_t *ga_end = nullptr;
                const uint8_t *ga = fmt_group_attrs(a, r, &ga_end);
                int rc = j_record<DUP_DETECT>(cs, a.cfg, ev, end, src_end, ga, ga_end);
                if (rc == JW_DUP) {
                    CountSink c2;
                    rc = j_record<DUP_EXACT>(c2, a.cfg, ev, end, src_end, ga, ga_end);
                    cs.n = c2.n;
                    slow = true;
                }
                if (rc == JW_BAD) atomicMin(a.first_bad, (unsigned long long) r);
                else {
                    if (rc == JW_DEPTH) atomicMin(a.first_fail, (unsigned long long) r);
                    else len = (uint32_t) cs.n + fmt_frame(a.cfg);
                    n_rec++;
                }
            }
            a.len[r] = len;
        }
        const uint64_t sm = __ballot(slow);
        if (lane == 0) a.slow[r >> 6] = sm;
    }
    for (int o = 32; o > 0; o >>= 1) { n_rec += __shfl_down(n_rec, o, 64); n_mark += __shfl_down(n_mark, o, 64); n_skip += __shfl_down(n_skip, o, 64); }
    if (lane == 0) {
        if (n_rec) atomicAdd(&a.counts[0], (unsigned long long) n_rec);
        if (n_mark) atomicAdd(&a.counts[1], (unsigned long long) n_mark);
        if (n_skip) atomicAdd(&a.counts[2], (unsigned long long) n_skip);
    }
}

// pass 2: the same walk writing.  Each lane prints its record into a per-wave LDS staging area at the record's offset
// inside the wave's (contiguous) output range; the wave then flushes the staged bytes with 16 B per lane coalesced
// stores (the scheme of k_parser_emit).  A record that does not fit the staging area is written directly.
constexpr int FMT_STG = 19968;              // staging bytes per wave (64 records x 312 B); 4 waves x 2 blocks fill a CU's LDS

template <class S>
DEV void fmt_emit_row(S &s, const JsonFmtArgs &a, uint64_t r, uint64_t o0, bool slow) {
    const uint8_t *rec = a.data + a.row_off[r], *end = a.data + a.row_off[r + 1];
    const uint8_t *src_end = a.data + a.bytes;
    Event ev = decode_event(rec, end, true);
    if (a.cfg.json_format == 1) s.put(o0 == 0 ? '[' : ',');
    const uint8_t *ga_end = nullptr;
    const uint8_t *ga = fmt_group_attrs(a, r, &ga_end);
    if (slow) j_record<DUP_EXACT>(s, a.cfg, ev, end, src_end, ga, ga_end);
    else j_record<DUP_NONE>(s, a.cfg, ev, end, src_end, ga, ga_end);
    if (a.cfg.json_format == 3) s.put('\n');
}

__global__ void __launch_bounds__(256) k_fmt_emit(JsonFmtArgs a) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LDS_AS uint8_t *stg = (LDS_AS uint8_t *) g_lds + (size_t) wave * FMT_STG;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    uint32_t mism = 0;
    for (uint64_t base = wave_id * 64; base < a.n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        const uint32_t cnt = (uint32_t) ((a.n - base) < 64 ? (a.n - base) : 64);
        uint64_t o0 = 0, o1 = 0;
        if (lane < cnt) { o0 = a.out_off[r]; o1 = a.out_off[r + 1]; }
        const bool slow = (a.slow[base >> 6] >> lane) & 1;
        uint32_t lo = 0;
        while (lo < cnt) {
            const uint64_t batch_base = __shfl(o0, (int) lo, 64);
            const uint32_t align = (uint32_t) (batch_base & 15);
            const bool fit = lane >= lo && lane < cnt && (o1 - batch_base + align) <= (uint64_t) FMT_STG;
            const uint64_t mask = __ballot(fit) >> lo;
            uint32_t m = (~mask == 0) ? 64 - lo : (uint32_t) __builtin_ctzll(~mask);
            if (m > cnt - lo) m = cnt - lo;
            if (m == 0) {
                // one record larger than the staging area: straight to global memory
                if (lane == lo && o1 > o0) {
                    ByteSink s(a.out + o0);
                    s.limit = a.out + o1;
                    fmt_emit_row(s, a, r, o0, true);
                    if (s.p != a.out + o1) mism++;
                }
                lo += 1;
                continue;
            }
            if (lane >= lo && lane < lo + m && o1 > o0) {
                LdsSink s(stg + align + (uint32_t) (o0 - batch_base));
                LDS_AS uint8_t *p0 = s.p;
                s.src_end = a.data + a.bytes;
                s.limit = s.p + (uint32_t) (o1 - o0);
                fmt_emit_row(s, a, r, o0, slow);
                // both passes walk the same bytes: the writer has to stop exactly where the sizes said
                if (s.p != p0 + (uint32_t) (o1 - o0)) mism++;
            }
            const uint32_t total = (uint32_t) (__shfl(o1, (int) (lo + m - 1), 64) - batch_base);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staged bytes visible to the wave
            // flush [align, align + total) of the staging area to out[batch_base ...]
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            uint8_t *dst = a.out + (batch_base - align);               // 16 B aligned
            const uint32_t n16 = (align + total + 15) / 16;
            for (uint32_t u = lane; u < n16; u += 64) {
                const uint32_t b0 = u * 16, b1 = b0 + 16;
                if (b0 >= align && b1 <= align + total) *(v4u *) (dst + b0) = *(LDS_AS v4u *) (stg + b0);
                else {
                    const uint32_t s0 = b0 < align ? align : b0, s1 = b1 > align + total ? align + total : b1;
                    for (uint32_t q = s0; q < s1; q++) dst[q] = stg[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // staging area reusable
            lo += m;
        }
    }
    if (mism) atomicAdd(&a.counts[3], (unsigned long long) mism);
}

// inclusive scans over the 1024 threads of a workgroup (wave scan by shuffles, wave totals through LDS)
DEV uint64_t block_scan_max(uint64_t v, uint64_t *wtot) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(v, o, 64); if (lane >= (uint32_t) o && t > v) v = t; }
    __syncthreads();
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) if (wtot[w] > v) v = wtot[w];
    return v;
}
DEV uint64_t block_scan_sum(uint64_t v, uint64_t *wtot) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(v, o, 64); if (lane >= (uint32_t) o) v += t; }
    __syncthreads();
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) v += wtot[w];
    return v;
}

// Group state of the decoder (src/flb_log_event_decoder.c:416-485), only run for chunks that hold markers or many
// skipped rows: every row gets the row of the opener that governs it.  One workgroup walks the rows in order, 1024
// at a time: "the last marker in front of a row" is a running maximum carried from tile to tile.
// Also the decoder's recursion guard: it skips markers and negative times by calling itself and refuses to decode
// at depth 1000 (:27,:389-394), so a row that directly follows 1000 skipped rows ends the walk.
__global__ void __launch_bounds__(1024) k_fmt_groups(JsonGroupArgs a) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry[3];        // marker maximum, break maximum, skipped rows so far
    __shared__ uint64_t sh[1024];
    const uint32_t tid = threadIdx.x;
    if (tid < 3) carry[tid] = 0;
    __syncthreads();
    for (uint64_t base = 0; base < a.n; base += 1024) {
        const uint64_t r = base + tid;
        // marker: (row + 1) << 1 | is_opener for a marker row; skipped: the decoder recursed over this row;
        // present: the row holds an event the decoder accepts (rows a filter emptied are not there at all)
        uint64_t marker = 0;
        bool skipped = false, present = false;
        if (r < a.n) {
            const uint8_t *rec = a.data + a.row_off[r], *end = a.data + a.row_off[r + 1];
            if (rec != end) {
                Event ev = decode_event(rec, end, true);
                if (ev.flags & RF_VALID) {
                    present = true;
                    if (ev.flags & RF_SKIP) {
                        skipped = true;
                        if (ev.sec == -1) marker = ((r + 1) << 1) | 1;
                        else if (ev.sec == -2) marker = (r + 1) << 1;
                    }
                }
            }
        }
        const uint64_t c_marker = carry[0], c_break = carry[1], c_skip = carry[2];
        uint64_t m = block_scan_max(marker, wtot);
        if (c_marker > m) m = c_marker;
        const uint64_t s_incl = block_scan_sum(skipped ? 1 : 0, wtot) + c_skip;      // skipped rows in [0, r]
        // last row at or before r that is present and not skipped, with the skipped count at that row
        uint64_t brk = block_scan_max((present && !skipped) ? (((r + 1) << 32) | (s_incl & 0xffffffffu)) : 0, wtot);
        if (c_break > brk) brk = c_break;
        sh[tid] = brk;
        __syncthreads();
        if (r < a.n) {
            a.g_row[r] = (present && !skipped && (m & 1)) ? (uint32_t) (m >> 1) : 0;
            if (present) {
                // skipped rows directly in front of r: skipped rows in [0, r) minus those up to the last normal row before r
                const uint64_t pb = tid ? sh[tid - 1] : c_break;
                const uint32_t s_before = (uint32_t) s_incl - (skipped ? 1u : 0u);
                const uint32_t before = s_before - (uint32_t) (pb & 0xffffffffu);
                if (before >= 1000) atomicMin(a.skip_limit, (unsigned long long) r);
            }
        }
        __syncthreads();
        if (tid == 1023) { carry[0] = m; carry[1] = brk; carry[2] = s_incl; }
        __syncthreads();
    }
}

void launch_fmt_size(const JsonFmtArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    uint64_t blocks = (a.n + 255) / 256, cap = (uint64_t) cus * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_fmt_size, dim3((unsigned) blocks), dim3(256), 0, st, a);
}
void launch_fmt_emit(const JsonFmtArgs &a, int cus, hipStream_t st) {
    if (a.n == 0) return;
    uint64_t blocks = (a.n + 255) / 256, cap = (uint64_t) cus * 8;
    if (blocks > cap) blocks = cap;
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) k_fmt_emit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fmt_emit, dim3((unsigned) blocks), dim3(256), 4 * FMT_STG, st, a);
}
void launch_fmt_groups(const JsonGroupArgs &a, hipStream_t st) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_fmt_groups, dim3(1), dim3(1024), 0, st, a);
}

}  // namespace flbgpu
