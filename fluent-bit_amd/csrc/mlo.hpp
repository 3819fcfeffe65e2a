// mlo.hpp -- multiline with a parser in front (mlo_kernels.inc, ml.cpp): rows, records, kernel arguments.  Shared by the kernel
// unit and the host side; see mlo_kernels.inc for what the fields mean.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>

namespace flbgpu {

constexpr int MLO_G = 4;                                   // groups per stream on the GPU path: "_default" + 3 names (the reference: 6)
enum { MLO_SKIP = 0, MLO_OK = 1, MLO_FLUSH = 2, MLO_UNPROC = 3, MLO_FLUSHALL = 4, MLO_REFUSE = 5 };
constexpr uint32_t MLO_CARRY_ROW = 0xFFFFFFFFu;

struct MloState {                                           // per group
    uint32_t cnt[MLO_G], sum[MLO_G], rank[MLO_G];
    uint32_t reset;                                         // bit g: a flush of g since the start of the scanned range
};

struct MloRow {                                             // what k_mlo_extract finds in a row
    uint32_t kind, gid;
    uint32_t c_len, g_len, map_len, raw_len;
    uint64_t c_off, g_off, map_off, raw_off;                // byte offsets into P.data (raw_off: into T.data)
    uint32_t sec, nsec;
};
struct MloRec {                                             // one output record
    uint32_t kind;                                          // 0 group record, 1 the refused line alone
    uint32_t g, row, first_row;                             // first_row: MLO_CARRY_ROW = the carried first line
    uint32_t rank_lo, nlines, content_len, with_carry;
    uint32_t size, content_at, trunc, pad;
};
struct MloGroupCarry { const uint8_t *content; const uint8_t *map; uint32_t content_len, map_len, sec, nsec; };
struct MloArgs {
    const uint8_t *tdata; const uint64_t *trow;             // T: in_tail's records
    const uint8_t *pdata; const uint64_t *prow;             // P: the parsed records
    const uint32_t *pinfo;                                  // RF_* of the parser filter, one per row
    uint64_t n;                                             // rows (lines); row n = the flush timer's virtual row when flush_all
    int flush_all;
    int type, negate; uint32_t match_len; uint8_t match_str[64];
    uint32_t kc_len, kp_len, kg_len; uint8_t kc[64], kp[64], kg[64];    // key_content / key_pattern / key_group names
    uint32_t key_len; uint8_t key[72];                      // key_content as a packed msgpack string (the refused line's record)
    uint32_t ngroups; uint32_t name_len[MLO_G]; uint8_t names[MLO_G][32];
    MloGroupCarry carry[MLO_G];
    uint64_t buffer_limit;
    uint32_t ts_sec, ts_nsec;
    MloRow *rows;                                           // [n + 1]
    MloState *st;                                           // [n + 1] state before the row
    uint32_t *nrec; const uint64_t *rec_base;               // [n + 1], its scan
    MloRec *recs; uint32_t *rec_size; const uint64_t *rec_off;
    uint32_t *idx;                                          // [MLO_G][n] row of the r-th line of the group
    uint8_t *out;
    uint8_t *carry_out_content[MLO_G]; uint8_t *carry_out_map[MLO_G];
    unsigned int *misc;                                     // [0] first row with an unknown group name, [1] refused, [2..5] new carry content_len,
                                                            // [6..9] new carry map_len, [10..13] sec, [14..17] nsec, [18] records of refused lines
};


void launch_mlo_extract(const MloArgs &a, hipStream_t st);
void launch_mlo_kinds(const MloArgs &a, hipStream_t st);
void launch_mlo_gid(const MloArgs &a, hipStream_t st);
size_t mlo_vscan_tmp_bytes(uint64_t n);
void launch_mlo_vscan(const MloArgs &a, void *tmp, hipStream_t st);
void launch_mlo_count(const MloArgs &a, hipStream_t st);
void launch_mlo_idx(const MloArgs &a, hipStream_t st);
void launch_mlo_recs(const MloArgs &a, hipStream_t st);
void launch_mlo_size(const MloArgs &a, uint64_t nrecs, hipStream_t st);
void launch_mlo_emit(const MloArgs &a, uint64_t nrecs, hipStream_t st);
void launch_mlo_carry(const MloArgs &a, int mode, hipStream_t st);

}  // namespace flbgpu
