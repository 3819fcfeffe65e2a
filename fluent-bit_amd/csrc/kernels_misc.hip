// kernels_misc.hip -- JSON rows -> msgpack and the device record indexer (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "json_kernels.inc"
#include "index_kernels.inc"

}  // namespace flbgpu
