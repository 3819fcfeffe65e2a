// calib.hip -- kernels with a KNOWN number of HBM bytes, one per request shape the filter kernels use, to calibrate
// what rocprofv3's FETCH_SIZE / WRITE_SIZE count on gfx950 for that shape (tools/calib_counters.py).  The buffer is
// far larger than the L2s and every kernel touches each 128-byte line at most once, so the bytes that must cross
// HBM are known independently of any cache behaviour:
//   0 coalesced16   a wave reads 1 KiB contiguous per instruction (16 B / lane)        -> n bytes
//   1 column4       a wave reads 256 B contiguous per instruction (4 B / lane)         -> n bytes
//   2 lane_line128  every lane reads one whole 128-byte line of its own (8 x 16 B)     -> n bytes
//   3 lane_sector16 every lane reads 16 B of a 128-byte line of its own                -> 16 B useful per line; the
//                   counter tells what a 16-byte request fetches (a 32 / 64 / 128 B sector)
//   4 write16       a wave writes 1 KiB contiguous per instruction                     -> n bytes written
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_calib(const uint8_t *in, uint8_t *out, uint64_t n, uint32_t *sink) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t) gridDim.x * blockDim.x;
    uint32_t acc = 0;
    if (MODE == 0) for (uint64_t o = tid * 16; o + 16 <= n; o += nthreads * 16) { const v4 w = *(const v4 *) (in + o); acc += w.x ^ w.y ^ w.z ^ w.w; }
    if (MODE == 1) for (uint64_t o = tid * 4; o + 4 <= n; o += nthreads * 4) acc += *(const uint32_t *) (in + o);
    if (MODE == 2) for (uint64_t o = tid * 128; o + 128 <= n; o += nthreads * 128) {
        v4 w[8];
        #pragma unroll
        for (int k = 0; k < 8; k++) w[k] = *(const v4 *) (in + o + 16 * k);
        #pragma unroll
        for (int k = 0; k < 8; k++) acc += w[k].x ^ w[k].w;
    }
    if (MODE == 3) for (uint64_t o = tid * 128; o + 128 <= n; o += nthreads * 128) { const v4 w = *(const v4 *) (in + o + 48); acc += w.x ^ w.w; }
    if (MODE == 4) for (uint64_t o = tid * 16; o + 16 <= n; o += nthreads * 16) { v4 w; w.x = (uint32_t) o; w.y = w.z = w.w = (uint32_t) tid; *(v4 *) (out + o) = w; }
    if (acc == 0x12345678u) sink[0] = acc;          // (keeps the loads alive)
}

extern "C" int flbgpu_calib_run(int mode, void *dev_in, void *dev_out, uint64_t n, int cus) {
    static uint32_t *sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return -1;
    const dim3 grid((unsigned) (cus > 0 ? cus * 8 : 2048)), block(256);
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_calib<0>, grid, block, 0, 0, (const uint8_t *) dev_in, (uint8_t *) dev_out, n, sink); break;
    case 1: hipLaunchKernelGGL(k_calib<1>, grid, block, 0, 0, (const uint8_t *) dev_in, (uint8_t *) dev_out, n, sink); break;
    case 2: hipLaunchKernelGGL(k_calib<2>, grid, block, 0, 0, (const uint8_t *) dev_in, (uint8_t *) dev_out, n, sink); break;
    case 3: hipLaunchKernelGGL(k_calib<3>, grid, block, 0, 0, (const uint8_t *) dev_in, (uint8_t *) dev_out, n, sink); break;
    case 4: hipLaunchKernelGGL(k_calib<4>, grid, block, 0, 0, (const uint8_t *) dev_in, (uint8_t *) dev_out, n, sink); break;
    default: return -1;
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
