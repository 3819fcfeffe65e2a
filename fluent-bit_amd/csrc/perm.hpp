// perm.hpp -- the rows of a chunk in the order of their lengths (kernels_perm.hip): what the register kernel walks when a chunk's lines
// differ a lot in length (its walk is position-synchronous: a wave steps as far as its LONGEST record)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stddef.h>
namespace flbgpu {
size_t row_perm_work_bytes(uint64_t n);
// perm[0 .. n): the row numbers ordered by length class (32-byte steps, everything from 8 KB on in the last class), the rows of a class
// in chunk order.  stat[0] += over the runs of 64 CONSECUTIVE rows: longest row x rows of the run; stat[1] += the rows' bytes -- what the
// walk steps through in chunk order against what it has to.  work: row_perm_work_bytes(n) bytes of device memory.  false: a launch failed.
bool launch_row_perm(const uint64_t *row_off, uint64_t n, uint32_t *perm, void *work, size_t work_bytes, unsigned long long *stat, hipStream_t st);
}  // namespace flbgpu
