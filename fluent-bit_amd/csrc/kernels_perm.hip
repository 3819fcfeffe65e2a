// kernels_perm.hip -- row numbers ordered by length class (perm.hpp): keys + one stable radix pass (rocPRIM) over eight bits
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "perm.hpp"

namespace flbgpu {

namespace {
__global__ void __launch_bounds__(256) k_perm_keys(const uint64_t *row_off, uint64_t n, uint32_t *keys, uint32_t *rows, unsigned long long *stat) {
    __shared__ unsigned long long sh[2][4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_id = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    unsigned long long acc_max = 0, acc_len = 0;
    for (uint64_t base = wave_id * 64; base < n; base += nwaves * 64) {
        const uint64_t r = base + lane;
        uint32_t len = 0;
        if (r < n) {
            const uint64_t l = row_off[r + 1] - row_off[r];
            len = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) l;
            keys[r] = (len >> 5) > 255u ? 255u : (len >> 5);
            rows[r] = (uint32_t) r;
        }
        uint32_t m = len;
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t) __shfl_xor((int) m, o, 64); m = x > m ? x : m; }
        const uint64_t cnt = n - base < 64 ? n - base : 64;
        if (lane == 0) acc_max += (unsigned long long) m * cnt;
        acc_len += len;
    }
    for (int o = 32; o > 0; o >>= 1) acc_len += __shfl_down(acc_len, o, 64);
    if (lane == 0) { sh[0][wave] = acc_max; sh[1][wave] = acc_len; }
    __syncthreads();
    if (threadIdx.x == 0 && stat) {
        atomicAdd(&stat[0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&stat[1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}
struct PermLayout { size_t keys_in, keys_out, rows_in, sort_tmp, sort_bytes, total; };
PermLayout perm_layout(uint64_t n) {
    PermLayout l;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t) 255; return o; };
    l.keys_in = take(n * 4); l.keys_out = take(n * 4); l.rows_in = take(n * 4);
    size_t tmp = 0;
    (void) rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) nullptr, (uint32_t *) nullptr, (size_t) n, 0u, 8u);
    l.sort_bytes = tmp;
    l.sort_tmp = take(tmp);
    l.total = at;
    return l;
}
}  // namespace

size_t row_perm_work_bytes(uint64_t n) { return perm_layout(n).total; }

bool launch_row_perm(const uint64_t *row_off, uint64_t n, uint32_t *perm, void *work, size_t work_bytes, unsigned long long *stat, hipStream_t st) {
    if (n == 0) return true;
    const PermLayout l = perm_layout(n);
    if (work_bytes < l.total || n > 0xFFFFFFF0ull) return false;
    uint8_t *w = (uint8_t *) work;
    uint32_t *keys_in = (uint32_t *) (w + l.keys_in), *keys_out = (uint32_t *) (w + l.keys_out), *rows_in = (uint32_t *) (w + l.rows_in);
    const unsigned blocks = (unsigned) ((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_perm_keys, dim3(blocks), dim3(256), 0, st, row_off, n, keys_in, rows_in, stat);
    size_t tmp = l.sort_bytes;
    return rocprim::radix_sort_pairs(w + l.sort_tmp, tmp, (const uint32_t *) keys_in, keys_out, (const uint32_t *) rows_in, perm, (size_t) n, 0u, 8u, st) == hipSuccess;
}

}  // namespace flbgpu
