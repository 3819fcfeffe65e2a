// kernels_jtile.hip -- the one-pass NDJSON kernel (a row per lane, rewritten in place in LDS; shares kdev.inc / json_dev.inc with kernels_misc.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "jtile.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "jlane_kernels.inc"

}  // namespace flbgpu
