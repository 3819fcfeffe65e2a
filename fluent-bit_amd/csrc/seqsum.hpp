// seqsum.hpp -- the histogram sum in the reference's order (kernels_seqsum.hip)
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stddef.h>
namespace flbgpu {
// seq[s] += the values of series s among (sid[i], val[i]), i < n, ONE AFTER THE OTHER in the order of i (binary64 additions, the bits of
// cmt_metric_hist_sum_add called once per observation).  sid entries >= nseries are no observations.  work: a device buffer of at least
// seqsum_work_bytes(n, nseries) bytes.  false: a launch failed.
size_t seqsum_work_bytes(uint64_t n, uint32_t nseries);
bool launch_seqsum_sorted(const uint32_t *sid, const uint64_t *val, uint64_t n, double *seq, uint32_t nseries, void *work, size_t work_bytes, hipStream_t st);
}  // namespace flbgpu
