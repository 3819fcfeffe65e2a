// kernels_fused.hip -- flb_filter_do over [filter_parser, filter_grep] in one pass (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "fused_kernels.inc"

}  // namespace flbgpu
