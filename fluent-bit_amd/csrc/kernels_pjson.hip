// kernels_pjson.hip -- filter_parser with Format json / logfmt / ltsv parsers: the size pass (shares kdev.inc with kernels.hip)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "dev.hpp"
#include "numconv.hpp"

namespace flbgpu {

#include "kdev.inc"

#include "pjson_kernels.inc"

}  // namespace flbgpu
