// kernels_tile.hip -- filter_parser's single-pass tile kernel (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "tile_kernels.inc"

}  // namespace flbgpu
