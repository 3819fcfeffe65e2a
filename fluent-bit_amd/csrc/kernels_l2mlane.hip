// kernels_l2mlane.hip -- filter_log_to_metrics' extraction, a record per lane staged in LDS and walked once (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "l2m_lane.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "lookback.inc"
#include "lane_dev.inc"
#include "l2m_dev.inc"
#include "l2mlane_kernels.inc"

}  // namespace flbgpu
