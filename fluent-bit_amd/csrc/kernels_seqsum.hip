// kernels_seqsum.hip -- the histogram sum as the reference builds it, without a wave per series walking the whole call
//
// cmt_metric_hist_sum_add (lib/cmetrics/src/cmt_metric_histogram.c:124-137) adds every observation to a binary64, in record order, per
// series.  That fold is not associative: its bits come from the chain of additions itself.  Round 4's kernel gave every series a wave that
// read ALL the call's observations and picked its own out by ballot -- series x observations reads, 345 ms for 10 M observations on ten
// series, and no end for a dictionary of millions.  Here:
//   1. the observations are brought into (series, record order) with ONE stable radix sort on the series id (rocPRIM, as many bits as the
//      dictionary needs: a single pass for up to 256 series);
//   2. a series whose values of this call are ALL integers, with the running sum an integer too and |sum| + sum of |values| below 2^53 --
//      byte counts, milliseconds, status codes: what log metrics mostly are --, has only exact partial sums, so its fold is one integer
//      sum: a parallel pass over the sorted values (a block per 2048, one atomic per block and series);
//   3. the others: a lane folds a short run (<= 256 observations) by itself, a wave folds a long one -- 512 values at a time, the next 512
//      on their way while these are added (the same integer test per 512, else v_readlane + v_add_f64 one after the other, eight cycles
//      each: the reference's bits have that price).
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "seqsum.hpp"

namespace flbgpu {

namespace {
constexpr uint32_t SS_SMALL = 256;          // runs a lane folds by itself

__global__ void __launch_bounds__(256) k_ss_keys(const uint32_t *sid, uint64_t n, uint32_t nseries, uint32_t *keys) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t s = sid[i];
        keys[i] = s < nseries ? s : nseries;
    }
}
// the runs of the sorted keys: start[s], end[s] (both 0 for a series without observations in this call)
__global__ void __launch_bounds__(256) k_ss_runs(const uint32_t *keys, uint64_t n, uint32_t nseries, unsigned long long *start, unsigned long long *end) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        if (i == 0 || keys[i - 1] != k) { if (k < nseries) start[k] = i; if (i && keys[i - 1] < nseries) end[keys[i - 1]] = i; }
        if (i == n - 1 && k < nseries) end[k] = n;
    }
}
// per series: are this call's values all integers (nonint[s] stays 0), their sum and the sum of their magnitudes (64-bit integers)
__global__ void __launch_bounds__(256) k_ss_ints(const uint32_t *keys, const uint64_t *vals, uint64_t n, uint32_t nseries, unsigned int *nonint, long long *isum,
                                                 unsigned long long *imag) {
    __shared__ long long sh_sum[4];
    __shared__ unsigned long long sh_mag[4];
    __shared__ unsigned int sh_bad[4];
    for (uint64_t base = (uint64_t) blockIdx.x * 2048; base < n; base += (uint64_t) gridDim.x * 2048) {
        const uint64_t last = base + 2048 <= n ? base + 2047 : n - 1;
        const uint32_t k0 = keys[base], k1 = keys[last];
        long long sum = 0;
        unsigned long long mag = 0;
        unsigned int bad = 0;
        for (int u = 0; u < 8; u++) {
            const uint64_t i = base + 256 * (uint64_t) u + threadIdx.x;
            if (i > last) break;
            const double v = __longlong_as_double((long long) vals[i]), a = fabs(v);
            const bool ok = a < 4503599627370496.0 && a == floor(a);
            const long long q = ok ? (long long) v : 0;
            if (k0 == k1) { sum += q; mag += (unsigned long long) (q < 0 ? -q : q); bad |= ok ? 0u : 1u; }
            else {
                const uint32_t k = keys[i];                                     // (a block that holds the end of a run: its own atomics)
                if (k < nseries) {
                    if (!ok) atomicOr(&nonint[k], 1u);
                    else { atomicAdd((unsigned long long *) &isum[k], (unsigned long long) q); atomicAdd(&imag[k], (unsigned long long) (q < 0 ? -q : q)); }
                }
            }
        }
        if (k0 == k1 && k0 < nseries) {
            for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); mag += __shfl_xor(mag, o, 64); bad |= __shfl_xor(bad, o, 64); }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { sh_sum[threadIdx.x >> 6] = sum; sh_mag[threadIdx.x >> 6] = mag; sh_bad[threadIdx.x >> 6] = bad; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned int b = sh_bad[0] | sh_bad[1] | sh_bad[2] | sh_bad[3];
                if (b) atomicOr(&nonint[k0], 1u);
                else {
                    atomicAdd((unsigned long long *) &isum[k0], (unsigned long long) (sh_sum[0] + sh_sum[1] + sh_sum[2] + sh_sum[3]));
                    atomicAdd(&imag[k0], sh_mag[0] + sh_mag[1] + sh_mag[2] + sh_mag[3]);
                }
            }
        }
    }
}
// a series of integers is done with one addition; short runs by a lane each; long ones are listed for the wave kernel
__global__ void __launch_bounds__(256) k_ss_fold_small(const uint64_t *vals, const unsigned long long *start, const unsigned long long *end, uint32_t nseries,
                                                       double *seq, uint32_t *heavy, unsigned int *nheavy, const unsigned int *nonint, const long long *isum,
                                                       const unsigned long long *imag) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseries; s += gridDim.x * blockDim.x) {
        const unsigned long long b = start[s], e = end[s];
        if (e <= b) continue;
        {
            const double a0 = seq[s];
            if (!nonint[s] && fabs(a0) < 4503599627370496.0 && a0 == floor(a0) && imag[s] < 4503599627370496ull) {
                const long long q0 = (long long) a0;
                if ((unsigned long long) (q0 < 0 ? -q0 : q0) + imag[s] < 9007199254740992ull) { seq[s] = (double) (q0 + isum[s]); continue; }      // every partial sum exact
            }
        }
        if (e - b > SS_SMALL) { heavy[atomicAdd(nheavy, 1u)] = s; continue; }
        double acc = seq[s];
        for (unsigned long long i = b; i < e; i++) acc += __longlong_as_double((long long) vals[i]);
        seq[s] = acc;
    }
}
__device__ __forceinline__ long long ss_wave_sum(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// a wave per long run
__global__ void __launch_bounds__(64) k_ss_fold_heavy(const uint64_t *vals, const unsigned long long *start, const unsigned long long *end, const uint32_t *heavy,
                                                      const unsigned int *nheavy, double *seq) {
    const uint32_t lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) double blk[512];
    for (uint32_t h = blockIdx.x; h < *nheavy; h += gridDim.x) {
        const uint32_t s = heavy[h];
        const unsigned long long b = start[s], e = end[s];
        double acc = seq[s];
        unsigned long long i = b;
        double nx[8];
        if (i + 512 <= e) {
            #pragma unroll
            for (int u = 0; u < 8; u++) nx[u] = __longlong_as_double((long long) vals[i + 64 * u + lane]);
        }
        for (; i + 512 <= e; i += 512) {
            double v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) v[u] = nx[u];
            if (i + 1024 <= e) {
                #pragma unroll
                for (int u = 0; u < 8; u++) nx[u] = __longlong_as_double((long long) vals[i + 512 + 64 * u + lane]);    // (on their way while these are added)
            }
            // all integers, and small enough that no partial sum leaves the integers a binary64 holds exactly?
            bool ints = true;
            long long mine = 0, mag = 0;
            #pragma unroll
            for (int u = 0; u < 8; u++) {
                const double a = fabs(v[u]);
                ints = ints && a < 4503599627370496.0 && a == floor(a);             // (NaN and the infinities fail the first test)
                const long long q = ints ? (long long) v[u] : 0;
                mine += q; mag += q < 0 ? -q : q;
            }
            const bool acc_int = fabs(acc) < 4503599627370496.0 && acc == floor(acc);
            if (__ballot(ints) == ~0ull && acc_int) {
                const long long total_mag = ss_wave_sum(mag), a0 = (long long) acc;
                if ((a0 < 0 ? -a0 : a0) + total_mag < 9007199254740992ll) {
                    acc = (double) (a0 + ss_wave_sum(mine));                           // exact at every step of the chain it stands for
                    continue;
                }
            }
            // The chain itself.  Round 6: ONE lane adds, reading the 512 values back from LDS two at a time -- an observation is one
            // v_add_f64 (the chain's latency) and half a ds_read; handing every value to the chain by two v_readlane cost three vector
            // operations an observation, 21 cycles against the addition's own 8 (53 -> ~20 ms for 10 M decimals on five series)
            #pragma unroll
            for (int u = 0; u < 8; u++) blk[64 * u + lane] = v[u];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                // (sixteen values in registers while the sixteen behind them are on their way from LDS: the reads' latency off the chain)
                double2 qa[8], qb[8];
                #pragma unroll
                for (int j = 0; j < 8; j++) qa[j] = *(const double2 *) &blk[2 * j];
                for (int k = 0; k < 512; k += 32) {
                    #pragma unroll
                    for (int j = 0; j < 8; j++) qb[j] = *(const double2 *) &blk[k + 16 + 2 * j];
                    #pragma unroll
                    for (int j = 0; j < 8; j++) { acc += qa[j].x; acc += qa[j].y; }
                    if (k + 32 < 512) {
                        #pragma unroll
                        for (int j = 0; j < 8; j++) qa[j] = *(const double2 *) &blk[k + 32 + 2 * j];
                    }
                    #pragma unroll
                    for (int j = 0; j < 8; j++) { acc += qb[j].x; acc += qb[j].y; }
                }
            }
            acc = __shfl(acc, 0, 64);
            __builtin_amdgcn_wave_barrier();
        }
        // the tail: 64 at a time
        for (; i < e; i += 64) {
            const unsigned long long left = e - i < 64 ? e - i : 64;
            blk[lane] = lane < left ? __longlong_as_double((long long) vals[i + lane]) : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) for (uint32_t k = 0; k < left; k++) acc += blk[k];
            acc = __shfl(acc, 0, 64);
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) seq[s] = acc;
    }
}

struct Layout { size_t keys_in, keys_out, vals_out, start, end, heavy, nheavy, nonint, isum, imag, sort_tmp, total, sort_bytes; };
Layout layout(uint64_t n, uint32_t nseries) {
    Layout l;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t) 255; return o; };
    l.keys_in = take(n * 4); l.keys_out = take(n * 4); l.vals_out = take(n * 8);
    l.start = take((size_t) (nseries + 1) * 8); l.end = take((size_t) (nseries + 1) * 8); l.heavy = take((size_t) (nseries + 1) * 4); l.nheavy = take(16);
    l.nonint = take((size_t) (nseries + 1) * 4); l.isum = take((size_t) (nseries + 1) * 8); l.imag = take((size_t) (nseries + 1) * 8);
    size_t tmp = 0;
    (void) rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t *) nullptr, (uint32_t *) nullptr, (const uint64_t *) nullptr, (uint64_t *) nullptr, (size_t) n, 0u, 32u);
    l.sort_bytes = tmp;
    l.sort_tmp = take(tmp);
    l.total = at;
    return l;
}
}  // namespace

size_t seqsum_work_bytes(uint64_t n, uint32_t nseries) { return layout(n, nseries).total; }

bool launch_seqsum_sorted(const uint32_t *sid, const uint64_t *val, uint64_t n, double *seq, uint32_t nseries, void *work, size_t work_bytes, hipStream_t st) {
    if (n == 0 || nseries == 0) return true;
    const Layout l = layout(n, nseries);
    if (work_bytes < l.total) return false;
    uint8_t *w = (uint8_t *) work;
    uint32_t *keys_in = (uint32_t *) (w + l.keys_in), *keys_out = (uint32_t *) (w + l.keys_out), *heavy = (uint32_t *) (w + l.heavy);
    uint64_t *vals_out = (uint64_t *) (w + l.vals_out);
    unsigned long long *start = (unsigned long long *) (w + l.start), *end = (unsigned long long *) (w + l.end);
    unsigned int *nheavy = (unsigned int *) (w + l.nheavy);
    const unsigned blocks = (unsigned) ((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_ss_keys, dim3(blocks), dim3(256), 0, st, sid, n, nseries, keys_in);
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) <= (unsigned long long) nseries) bits++;         // keys 0 .. nseries
    size_t tmp = l.sort_bytes;
    if (rocprim::radix_sort_pairs(w + l.sort_tmp, tmp, (const uint32_t *) keys_in, keys_out, val, vals_out, (size_t) n, 0u, bits, st) != hipSuccess) return false;
    if (hipMemsetAsync(start, 0, (size_t) (nseries + 1) * 8, st) != hipSuccess || hipMemsetAsync(end, 0, (size_t) (nseries + 1) * 8, st) != hipSuccess ||
        hipMemsetAsync(nheavy, 0, 16, st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_ss_runs, dim3(blocks), dim3(256), 0, st, keys_out, n, nseries, start, end);
    unsigned int *nonint = (unsigned int *) (w + l.nonint);
    long long *isum = (long long *) (w + l.isum);
    unsigned long long *imag = (unsigned long long *) (w + l.imag);
    if (hipMemsetAsync(nonint, 0, (size_t) (nseries + 1) * 4, st) != hipSuccess || hipMemsetAsync(isum, 0, (size_t) (nseries + 1) * 8, st) != hipSuccess ||
        hipMemsetAsync(imag, 0, (size_t) (nseries + 1) * 8, st) != hipSuccess) return false;
    const unsigned iblocks = (unsigned) ((n + 2047) / 2048 < 4096 ? (n + 2047) / 2048 : 4096);
    hipLaunchKernelGGL(k_ss_ints, dim3(iblocks), dim3(256), 0, st, keys_out, vals_out, n, nseries, nonint, isum, imag);
    const unsigned sblocks = (unsigned) (((uint64_t) nseries + 255) / 256 < 4096 ? ((uint64_t) nseries + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_ss_fold_small, dim3(sblocks), dim3(256), 0, st, vals_out, start, end, nseries, seq, heavy, nheavy, nonint, isum, imag);
    hipLaunchKernelGGL(k_ss_fold_heavy, dim3(nseries < 1024 ? nseries : 1024), dim3(64), 0, st, vals_out, start, end, heavy, nheavy, seq);
    return hipGetLastError() == hipSuccess;
}

}  // namespace flbgpu
