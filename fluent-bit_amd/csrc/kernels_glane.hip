// kernels_glane.hip -- filter_grep in one pass (a record per lane, staged in LDS, kept records placed by a look-back; shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "dec.hpp"
#include "grep_lane.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "lookback.inc"
#include "glane_kernels.inc"

}  // namespace flbgpu
