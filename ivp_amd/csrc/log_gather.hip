// log_gather.hip -- second half of the one-pass accepted-step log (gfx950): page chains -> CSR.
//
// The stepping kernels leave every trajectory's records in a chain of pool pages (ivp_kargs.h, so_push_log in rk_core.h)
// and its record count in n_log.  Here:
//   ivp_log_scan    offsets[b] = sum of n_log[0 .. b), offsets[B] = total          (three small launches)
//   ivp_log_gather  one wavefront per trajectory walks its chain from the last page back to the first and copies each
//                   page's t block and y block to their place in the CSR log -- two contiguous, coalesced copies per page:
//                   t_log[offsets[b] + k], y_log[(offsets[b] + k) * n + c]  (time-major like the reference's
//                   Solution.t / Solution.y, src/solve/solout.rs:387-428, src/solve/solve_ivp.rs:288-312).
// Both are pure data movement: HBM-bound, 2 x 8 (n + 1) bytes per record.
#include <hip/hip_runtime.h>

#include "ivp_kargs.h"
#include "log_gather.h"

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                       // consecutive counts per thread
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ unsigned long long wave_incl_scan(unsigned long long v)
{
    const int lane = (int)__lane_id();
#pragma unroll
    for (int o = 1; o < IVP_WAVE; o <<= 1) {
        const unsigned long long u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (kScanThreads threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *total)
{
    __shared__ unsigned long long wsum[kScanThreads / IVP_WAVE];
    const int lane = (int)__lane_id(), w = (int)threadIdx.x / IVP_WAVE;
    const unsigned long long incl = wave_incl_scan(v);
    __syncthreads();   // a previous use of wsum is over
    if (lane == IVP_WAVE - 1) wsum[w] = incl;
    __syncthreads();
    unsigned long long base = 0, all = 0;
#pragma unroll
    for (int q = 0; q < kScanThreads / IVP_WAVE; ++q) {
        const unsigned long long s = wsum[q];
        base += q < w ? s : 0ull;
        all += s;
    }
    *total = all;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void log_tile_sums(const uint32_t *n_log, uint32_t B, unsigned long long *tile_sum)
{
    const size_t i0 = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += (i0 + k < B) ? n_log[i0 + k] : 0u;
    unsigned long long total;
    (void)block_excl_scan(s, &total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// one block: exclusive scan of the tile sums in place, offsets[B] = grand total
__global__ __launch_bounds__(kScanThreads) void log_scan_tile_sums(unsigned long long *tile_sum, uint32_t tiles, unsigned long long *offsets, uint32_t B)
{
    unsigned long long carry = 0;
    for (uint32_t t0 = 0; t0 < tiles; t0 += kScanThreads) {
        const uint32_t t = t0 + threadIdx.x;
        const unsigned long long v = t < tiles ? tile_sum[t] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, &total);
        if (t < tiles) tile_sum[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) offsets[B] = carry;
}

__global__ __launch_bounds__(kScanThreads) void log_scan_apply(const uint32_t *n_log, uint32_t B, const unsigned long long *tile_base, unsigned long long *offsets)
{
    const size_t i0 = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    uint32_t c[kScanItems];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { c[k] = (i0 + k < B) ? n_log[i0 + k] : 0u; s += c[k]; }
    unsigned long long total;
    unsigned long long run = tile_base[blockIdx.x] + block_excl_scan(s, &total);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (i0 + k < B) offsets[i0 + k] = run;
        run += c[k];
    }
}

// One wavefront per trajectory.  `capacity` = records the destination holds: a log that does not fit is left alone (the
// host reports it; nothing is written out of bounds).
__global__ __launch_bounds__(IVP_WAVE) void log_gather_kernel(const double *pool, const uint32_t *log_cur, const uint32_t *n_log,
                                                              const unsigned long long *offsets, uint32_t B, uint32_t n, uint32_t shift,
                                                              unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log)
{
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    const uint32_t cnt = n_log[j];
    if (cnt == 0u || offsets[B] > capacity) return;
    const unsigned long long off = offsets[j] + dst_base;
    const uint32_t R = 1u << shift;
    const size_t page_doubles = 1u + ((size_t)(n + 1u) << shift);
    const uint32_t pages = (cnt + R - 1u) >> shift;
    uint32_t page = log_cur[j];
    for (uint32_t p = pages; p-- > 0u;) {
        const double *src = pool + (size_t)page * page_doubles;
        const uint32_t prev = *(const uint32_t *)src;             // header: the trajectory's previous page
        const uint32_t recs = (p + 1u == pages) ? cnt - (p << shift) : R;
        const unsigned long long q0 = off + ((unsigned long long)p << shift);
        const double *st = src + 1;
        for (uint32_t i = lane; i < recs; i += IVP_WAVE) t_log[q0 + i] = st[i];
        const double *sy = st + R;
        double *dy = y_log + q0 * n;
        const uint32_t m = recs * n;
        for (uint32_t i = lane; i < m; i += IVP_WAVE) dy[i] = sy[i];
        page = prev;
    }
}

}  // namespace

size_t ivp_log_scan_scratch_bytes(size_t B) { return sizeof(unsigned long long) * ((B + kScanTile - 1) / kScanTile + 1); }

hipError_t ivp_log_scan(const uint32_t *n_log, size_t B, unsigned long long *offsets, void *scratch, hipStream_t s)
{
    const uint32_t tiles = (uint32_t)((B + kScanTile - 1) / kScanTile);
    if (tiles == 0) return hipMemsetAsync(offsets, 0, sizeof(unsigned long long), s);
    (void)hipGetLastError();
    unsigned long long *tsum = (unsigned long long *)scratch;
    hipLaunchKernelGGL(log_tile_sums, dim3(tiles), dim3(kScanThreads), 0, s, n_log, (uint32_t)B, tsum);
    hipLaunchKernelGGL(log_scan_tile_sums, dim3(1), dim3(kScanThreads), 0, s, tsum, tiles, offsets, (uint32_t)B);
    hipLaunchKernelGGL(log_scan_apply, dim3(tiles), dim3(kScanThreads), 0, s, n_log, (uint32_t)B, (const unsigned long long *)tsum, offsets);
    return hipGetLastError();
}

hipError_t ivp_log_gather(const double *pool, const uint32_t *log_cur, const uint32_t *n_log, const unsigned long long *offsets, size_t B,
                          int n, uint32_t shift, unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(log_gather_kernel, dim3((uint32_t)B), dim3(IVP_WAVE), 0, s, pool, log_cur, n_log, offsets, (uint32_t)B, (uint32_t)n, shift,
                       capacity, dst_base, t_log, y_log);
    return hipGetLastError();
}
