// jtile.hpp -- launchers of the one-pass NDJSON kernels (kernels_jtile.hip: jlane_kernels.inc, jtile_kernels.inc); JtArgs is dev.hpp's
#pragma once
#include "dev.hpp"
namespace flbgpu {
void launch_json_lane(const JtArgs &a, int cus, hipStream_t st);
int json_lane_text_bytes();
uint64_t json_lane_units(uint64_t ntiles);
}  // namespace flbgpu
