// kernels_l2m.hip -- filter_log_to_metrics kernels (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"
#include "spsel.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "l2m_dev.inc"
#include "l2m_kernels.inc"

#include "sp_kernels.inc"

#include "sp_select.inc"

}  // namespace flbgpu
