// host_int.hpp -- internals shared by the host-side translation units of libflbgpu.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <strings.h>
#include <vector>
#include <unordered_map>

#include "../../include/flb_gpu.h"
#include "dev.hpp"
#include "rx.hpp"

namespace flbgpu {

void set_err(const char *fmt, ...);
int device_cus();

#define HIPOK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ::flbgpu::set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return false;                                                                        \
        }                                                                                        \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) { (void) hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIPOK(hipMalloc(&p, want));
        cap = want;
        return true;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *) p; }
};
// a function-local device buffer: released on every return path (DevBuf itself is a plain member type that its owner releases)
struct ScopedDevBuf : DevBuf {
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf &) = delete;
    ScopedDevBuf &operator=(const ScopedDevBuf &) = delete;
    ~ScopedDevBuf() { release(); }
};

// page-locked host memory (staging slabs and the record-offset column of the host-level call)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    // keeps the first `keep` bytes when it has to move
    bool ensure(size_t bytes, size_t keep = 0) {
        if (bytes <= cap) return true;
        size_t want = bytes + bytes / 2 + 4096;
        void *q = nullptr;
        HIPOK(hipHostMalloc(&q, want, hipHostMallocDefault));
        if (p) { if (keep) memcpy(q, p, keep); (void) hipHostFree(p); }
        p = q; cap = want;
        return true;
    }
    void release() { if (p) (void) hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *) p; }
};

// uploads vectors of one table set into a single device allocation
struct TableBlob {
    void *dev = nullptr;
    ~TableBlob() { if (dev) (void) hipFree(dev); }
};

bool upload_cap(const rx::TableSet &t, TableBlob &blob, DevCap &out);
bool upload_utf8(const rx::Program &prog, TableBlob &blob, DevCap &out);
bool upload_dfa(const rx::TableSet &t, TableBlob &blob, DevDfa &out);
bool upload_fx(const rx::TableSet &t, int ncap, TableBlob &blob, DevFx &out, bool pair = false);
bool build_fx(const rx::TableSet &t, int ncap, std::vector<uint8_t> &b, DevFx &out, bool pair = false);
int simulate_fx(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps, bool use_tail = true);
// fx3: the same tables without special entries (8-byte cells, two capture writes per step; fx.cpp)
// (pairs: fx4 -- a cell per (row, class of byte j, class of byte j + 1), two positions per table read; out.ok == 0 when that does not fit)
bool build_fx3(const rx::TableSet &t, int ncap, std::vector<uint8_t> &b, DevFx &out, int pairs = 0);      // pairs: 0 fx3, 1 fx4 (four write ports), 2 fx5 (three)
int simulate_fx3(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps);
int simulate_fx4(const std::vector<uint8_t> &b, const DevFx &fx, int ncap, const uint8_t *s, uint32_t len, uint16_t *caps);
// grammar: src/record_accessor/ra.l:54-67, ra.y:60-99
bool parse_ra(const char *pat, DevKey &k, std::string &why);
// "<field> <regex>" rule of filter_grep / filter_log_to_metrics -> device rule (tables uploaded into blobs)
bool compile_rule(const std::string &ra_field, const char *pattern, GrepRule &r, std::vector<TableBlob *> &blobs, std::string &why, bool *nonregular = nullptr);

}  // namespace flbgpu

// series / group dictionary + rows in HBM (l2m.cpp; the stream processor's groups reuse the table part: sp.cpp)
struct L2mCtr { unsigned long long arena_used; unsigned int n_series; unsigned int overflow; };

struct L2mState {
    int mode = 0, discard_logs = 0, nb = 0, W = 0;
    std::vector<double> bounds;
    std::vector<std::string> label_keys;
    std::vector<flbgpu::DevKey> labels;
    flbgpu::DevKey value_key;
    flbgpu::DevBuf d_labels, d_value_key, d_bounds;
    // series dictionary + rows
    uint64_t cap = 0, arena_cap = 0;
    uint32_t max_series = 0;
    flbgpu::DevBuf d_slot_hash, d_slot_sid, d_arena, d_key_off, d_key_len, d_series_hash, d_rows, d_ctr;
    // per-call columns
    flbgpu::DevBuf d_sid, d_val, d_tmp, d_misc;
    uint64_t idx_base = 0;
    uint64_t last_obs = 0, last_deferred = 0, last_stale = 0, grows = 0;
    // sum_order reference (flbgpu_l2m_set_sum_order): next to the exact sum, the histogram sum as cmetrics builds it -- one f64
    // addition per observation in record order (lib/cmetrics/src/cmt_metric_histogram.c:124-137) --, one binary64 per series id
    // 2 (round 6) = the same across ranks: every rank keeps its observations since the last flush (series id + value, in record order)
    // and the flush folds them rank after rank, each rank continuing from the sums the rank in front ended on (l2m.cpp "the chain"):
    // the bits of ONE process that was fed rank 0's records of the interval, then rank 1's ... -- for any number of ranks
    int sum_order_ref = 1;                                   // (round 6: the reference's order is the default of the C ABI too, as it is the plugin shim's)
    flbgpu::DevBuf d_seq;
    uint32_t seq_cap = 0;
    flbgpu::DevBuf d_log_sid, d_log_val, d_nobad, d_seqwork;
    uint64_t log_n = 0, log_cap = 0;
    std::unordered_map<std::string, double> chain_sums;      // per label tuple: the sum the last flush ended on (the same on every rank)
    std::vector<double> last_chain;                          // the last flush's sums in the order of its output
};

void l2m_state_destroy(L2mState *);
flbgpu::L2mTable l2m_table_of(L2mState *s);
bool l2m_table_init(L2mState *s);
bool l2m_table_grow(L2mState *s, hipStream_t st);
// variable-size all-gather over RCCL through device staging buffers (l2m.cpp): all[r] = rank r's bytes
bool rccl_all_gather_bytes(void *rccl_comm, hipStream_t st, const void *mine, size_t bytes, std::vector<std::vector<uint8_t>> &all);
// fixed-point sum digits (L2M_NLIMB words, carries not yet propagated) + special counts -> binary64 bits, rounded once
uint64_t l2m_limbs_bits(const uint64_t *limbs, uint64_t n_nan, uint64_t n_pinf, uint64_t n_ninf);

struct KernelProf { const char *name; double ms = 0; uint64_t launches = 0; };
struct ProfPending { const char *name; hipEvent_t e0, e1; };

enum { F_PARSER = 1, F_GREP = 2, F_L2M = 3, F_JSONFMT = 4 };

struct flbgpu_filter {
    int kind = 0;
    hipStream_t stream = nullptr;
    // filter_parser
    flbgpu::FParserCfg pcfg;
    std::vector<flbgpu_parser *> parsers;
    flbgpu::DevBuf d_parsers;
    uint32_t caps_stride = 0;
    // The choices between a fast build and the build that takes everything are made per call from what the last calls showed -- no
    // choice is for good (round 5's latches: one odd chunk moved a filter to the slower build for the life of the process):
    //   tile   the single pass (k_parser_reg / k_parser_tile) or the phase kernels: bad when the pass left more than 1 value in 8 to its
    //          reverse-pass fallback or more than 1 row in 4 to the fix-up
    //   fx5    the three-port pair tables or the four-port ones: bad when the three-port walk handed on more than 1 row in 64
    //   defer  the kept records' time looked up by k_pg_emit or inside the single pass: bad when k_pg_emit met a text its plan does not settle
    //   plain  k_pg_emit's plain build or the general one: bad when the plain build left more than 1 kept record in 16 alone
    struct Probe {
        bool off = false, trying = false;
        uint32_t quiet = 0, interval = 16;
        uint64_t asked = ~0ull;         // the call the answer below belongs to (a call asks more than once: repeated stages)
        bool answer = true;
        uint64_t probes = 0, returns = 0;
        // may this call run the fast build?  While it is off the other build runs `interval` calls, then one call tries again.
        bool use(uint64_t call) {
            if (call == asked) return answer;
            asked = call; trying = false;
            if (!off) answer = true;
            else if (++quiet >= interval) { trying = true; probes++; answer = true; }
            else answer = false;
            return answer;
        }
        void bad() {                    // the fast build did badly on this call's data
            if (off && trying) interval = interval < 1024 ? interval * 2 : 1024;      // a try that failed: the next one later
            else if (!off) interval = 16;
            off = true; quiet = 0; trying = false; answer = false;
        }
        void good() { if (off && trying) { off = false; trying = false; returns++; } }
    };
    Probe tile, fx5, defer, plain;
    uint64_t calls = 0;               // device-level calls so far (the Probes' clock)
    bool sort_rows = false;           // the register kernel walks the rows in the order of their lengths (flbgpu.cpp note_lengths)
    bool again = false;               // parser_size_pass: "this call again from the top" (it has set a Probe aside: the next choice differs)
    bool last_fx5 = false;            // the last launch of the register kernel walked the three-port tables
    int fx_on_device = 0;             // which pair tables this filter's device copy of parser 0 holds: 0 three-port (as created), 1 four-port
    uint32_t last_path = 0;           // what the last call ran: bit 0 single pass, 1 three-port tables, 2 time lookup in k_pg_emit, 3 plain emit build
    bool has_decoders = false;        // a parser of the list has Decode_Field / Decode_Field_As rules (dec_dev.inc: k_parser_dec)
    // filter_grep (and the rule gate of filter_log_to_metrics)
    std::vector<flbgpu::GrepRule> rules;
    std::vector<flbgpu::TableBlob *> rule_blobs;
    // host rules (flbgpu.cpp): rules[i] whose pattern is not a regular expression -- host_rx[i] is the backtracking matcher's program
    // (rx.hpp bt_compile), nullptr for a device rule; values the matcher gave up on (its backtrack budget); filter_parser: records the
    // host parser did not take (duplicate Key_Name entries, values of 64 KB and more)
    std::vector<rx::BtProgram *> host_rx;
    bool has_host_rules = false;
    flbgpu::DevBuf d_hspans, d_hbits;
    uint64_t host_budget_over = 0, host_unhandled = 0, host_values = 0;
    flbgpu::DevBuf d_rules;
    int logical_op = 0;
    // filter_log_to_metrics
    L2mState *l2m = nullptr;
    // filter_log_to_metrics whose regex / exclude rules hold a pattern that is not a regular expression: the rules live in this hidden
    // filter_grep (host rules, flbgpu.cpp) and run in FRONT of the metric kernels, which then see the kept records and no rules
    flbgpu_filter *l2m_gate = nullptr;
    bool host_list = false;                     // filter_parser: a list of several parsers with host parsers in it (flbgpu.cpp host_list_rx)
    flbgpu::DevBuf d_hres[flbgpu::MAX_HOST_PARSERS];   // their answers for the chunk at hand
    // msgpack -> JSON output formatter (packfmt.cpp)
    flbgpu::JsonFmtCfg jcfg = {};
    flbgpu::DevBuf d_datekey, d_grow;
    // working buffers
    flbgpu::DevBuf d_info, d_caps, d_null, d_len, d_off, d_scan_tmp, d_out, d_rid, d_rid2, d_misc, d_status, d_out_off, d_ov, d_kept, d_keep, d_pg, d_args, d_desc, d_tail, d_dec, d_fix, d_units, d_perm, d_permwork;
    flbgpu::DevBuf h_in_data, h_in_off;        // device copies of host input (flbgpu_filter_run)
    flbgpu::PinnedBuf hp_misc, hp_args, hp_off, hp_stage[2];    // pinned record offsets / two staging slabs
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    flbgpu_indexer *indexer = nullptr;        // device record indexer of the host-level call (large chunks)
    uint64_t last_in = 0, last_out = 0;
    // profiling
    bool prof = false;
    std::vector<KernelProf> kp;
    std::vector<ProfPending> pending;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~flbgpu_filter() {
        for (auto &p : pending) { (void) hipEventDestroy(p.e0); (void) hipEventDestroy(p.e1); }
        if (l2m) l2m_state_destroy(l2m);
        for (auto *b : rule_blobs) delete b;
        for (auto *b : host_rx) if (b) rx::bt_free(b);
        delete l2m_gate;
        d_hspans.release(); d_hbits.release();
        for (auto &b : d_hres) b.release();
        flbgpu::DevBuf *all[] = {&d_parsers, &d_rules, &d_info, &d_caps, &d_null, &d_len, &d_off, &d_scan_tmp, &d_out, &d_rid, &d_rid2,
                                 &d_misc, &d_status, &d_out_off, &d_ov, &d_kept, &d_keep, &d_pg, &h_in_data, &h_in_off, &d_datekey, &d_grow, &d_args, &d_desc, &d_tail, &d_dec, &d_fix, &d_units, &d_perm, &d_permwork};
        for (auto *b : all) b->release();
        if (indexer) flbgpu_indexer_destroy(indexer);
        hp_misc.release(); hp_args.release(); hp_off.release(); hp_stage[0].release(); hp_stage[1].release();
        for (auto &e : ev_stage) if (e) (void) hipEventDestroy(e);
        if (ev0) (void) hipEventDestroy(ev0);
        if (ev1) (void) hipEventDestroy(ev1);
        if (stream) (void) hipStreamDestroy(stream);
    }
};

bool filter_common_init(flbgpu_filter *f);

// Kernel timing: an event pair per launch, recorded on the stream the kernel runs on and resolved
// only when the totals are read (no host synchronisation inside the timed region).
struct ProfScope {
    flbgpu_filter *f; hipStream_t st; const char *name; bool on;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(flbgpu_filter *f_, hipStream_t st_, const char *n) : f(f_), st(st_), name(n), on(f_->prof) {
        if (!on) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
        (void) hipEventRecord(e0, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void) hipEventRecord(e1, st);
        f->pending.push_back(ProfPending{name, e0, e1});
    }
};
void prof_resolve(flbgpu_filter *f);

namespace flbgpu {
// host chunk -> device (pinned slabs, record boundaries found on the host or the device); device -> host buffer
int64_t staged_upload(flbgpu_filter *f, const uint8_t *d, size_t bytes, size_t *consumed, const uint64_t **row_off, bool no_wait = false);
bool staged_download(flbgpu_filter *f, void *dst, const void *src, size_t bytes);
// row_off == NULL ("raw chunk bytes" in HBM): the records are found on the device
bool resolve_raw_chunk(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *resolved, bool *garbage);
}

// filter_log_to_metrics entry used by flbgpu_filter_run / flbgpu_filter_run_dev
bool run_l2m_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, hipStream_t st, int *ret);
// (flbgpu.cpp) the hidden gate of a log_to_metrics filter: grep's legacy rule loop over `in`; *kept = the chunk the metric kernels take
// (`in` itself when every record passes); false: failure, or a decoder error inside the chunk (flbgpu_last_error says which)
bool l2m_gate_dev(flbgpu_filter *gate, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *kept, hipStream_t st);
