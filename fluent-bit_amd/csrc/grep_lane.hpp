// grep_lane.hpp -- filter_grep in one pass (kernels_glane.hip: glane_kernels.inc): argument block and launcher
#pragma once
#include "dev.hpp"
namespace flbgpu {
struct GrepLaneArgs {
    GrepArgs g;                         // chunk, rules, logical_op, keep_len / status columns, first_bad, counts, the rules' LDS offsets, key slots
    uint64_t *off_out;                  // [n + 1] row offsets of the output (a dropped record is an empty row)
    uint8_t *out;
    uint64_t out_cap;
    unsigned long long *unit_state;     // [units] look-back words, zeroed
    unsigned long long *ticket;         // zeroed
    unsigned long long *words;          // [0] bytes written, [1] workgroups without room / that gave up waiting (<< 32)
    uint64_t ntiles;
    uint32_t rows_per_tile;             // <= 64
    unsigned long long *prof;           // measurement only (FLBGPU_GREP_PROF): [8] shader cycles per phase, summed over the waves
    uint32_t text_cap;                  // LDS bytes of a wave's records (a multiple of 16, <= grep_lane_text_max())
    // two filter_grep instances that follow each other in a chain run as ONE pass (flb_filter_do hands the second what the first keeps, and
    // grep changes no record): g2 = the second filter's rules (its table offsets count on behind the first's; counts[2] = what it keeps,
    // counts[3] = the bytes the first keeps); the slots below are the names of both
    int two;
    GrepArgs g2;
    int nslots;                         // names in slot_kw / slot_klen (GrepArgs::rule_slot of both filters index them)
    // the slots' keys as the dwords a lane reads them (little endian, zero padded): a name of up to 32 bytes is compared with eight
    // masked dword tests, no byte loop
    uint32_t slot_kw[GREP_SLOTS][8];
    uint8_t slot_klen[GREP_SLOTS];
};
void launch_grep_lane(const GrepLaneArgs &a, int cus, hipStream_t st);
void launch_tile_max(const uint64_t *row_off, uint64_t n, uint32_t R, unsigned long long *out, hipStream_t st);   // *out (zeroed) = bytes of the longest run of R rows + 15
int grep_lane_text_max();
uint64_t grep_lane_units(uint64_t ntiles);
uint32_t grep_lane_table_bytes(uint32_t nD, uint32_t ncls);    // LDS bytes of a rule's automaton in the lanes' layout
uint32_t grep_lane_table_room();
}  // namespace flbgpu
