// kernels_tail.hip -- in_tail's line packing and the multiline core behind it (shares kdev.inc with the other kernel units)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "mlo.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "tail_kernels.inc"
#include "ml_kernels.inc"
#include "mlo_kernels.inc"

}  // namespace flbgpu
