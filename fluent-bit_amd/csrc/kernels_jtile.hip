// kernels_jtile.hip -- the NDJSON tile pass (a wave per tile of rows; shares kdev.inc / json_dev.inc with kernels_misc.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"
#include "jtile_kernels.inc"

}  // namespace flbgpu
