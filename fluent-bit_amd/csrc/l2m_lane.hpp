// l2m_lane.hpp -- filter_log_to_metrics' extraction with a lean walk on LDS (kernels_l2mlane.hip: l2mlane_kernels.inc)
#pragma once
#include "dev.hpp"
namespace flbgpu {
constexpr int L2L_SLOTS = 6;            // the value field and up to L2M_LV labels, each a top-level name of at most 32 bytes
struct L2mLaneArgs {
    L2mArgs a;                          // as k_l2m_extract's (no rules: the host takes the other kernel when the filter has any)
    uint32_t text_cap;                  // LDS bytes of a wave's records (a multiple of 16)
    uint32_t rows_per_tile;             // <= 64
    int nslots;                         // slot 0..nlabels-1: the labels (in order); slot value_slot: the value field (-1: none / counter)
    int value_slot;
    uint32_t slot_kw[L2L_SLOTS][8];     // the names as the dwords a lane reads them (zero padded)
    uint8_t slot_klen[L2L_SLOTS];
};
void launch_l2m_lane(const L2mLaneArgs &a, int cus, hipStream_t st);
int l2m_lane_text_max();
}  // namespace flbgpu
