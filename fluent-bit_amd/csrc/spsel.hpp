// spsel.hpp -- the stream processor's plain SELECTs (no aggregation function): src/stream_processor/flb_sp.c:1607-1850 sp_process_data.
// Shared by sp.cpp and the kernel unit (sp_select.inc).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "dev.hpp"

namespace flbgpu {

// star 0: a named key (key: index into SpPlan::keys; alias: in SpPlan::blob) / 1: `*` / 2: a pair that is constant for the call (NOW(),
// UNIX_TIMESTAMP(), RECORD_TAG(): packed key + value at consts[alias_off..]) / 3: RECORD_TIME() (packed key at consts[alias_off..], then
// the record's own time as a float64)
struct SpSelKey { uint8_t star, key, has_alias, pad; uint16_t alias_off, alias_len; };
struct SpSelArgs {
    const uint8_t *data; const uint64_t *row_off; uint64_t n, bytes;
    const SpPlan *plan;                 // keys + the WHERE program (device memory)
    int nsel;
    SpSelKey sel[SP_MAX_KEYS];
    const uint8_t *consts;
    uint32_t *out_len;                  // [n] bytes the record leaves (0: filtered out, or no selected key found in it)
    const uint64_t *out_off;
    uint8_t *out;
    unsigned long long *first_bad;      // the first row that does not decode (msgpack_unpack_next stops there)
    unsigned long long *records;        // records that passed WHERE (sp_process_data's return value)
    unsigned int *flags;                // SPF_*
};
void launch_sp_select(const SpSelArgs &a, bool emit, hipStream_t st);

}  // namespace flbgpu
