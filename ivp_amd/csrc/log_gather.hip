// log_gather.hip -- second half of the one-pass accepted-step log (gfx950): wave pages -> CSR.
//
// The stepping kernels leave every trajectory's records in a chain of segments (a segment = one column of a wave page: up to
// 32 record slots, a bit per slot that holds a record; layout in ivp_kargs.h, writers so_log_open / so_push_log in rk_core.h)
// and its record count in n_log.  Here:
//   ivp_log_scan    offsets[b] = sum of n_log[0 .. b), offsets[B] = total          (three small launches)
//   ivp_log_gather  a group of lanes per trajectory walks its chain from the last segment back to the first; a segment's
//                   k-th record (k = rank of its slot among the set bits) goes to t_log[offsets[b] + k0 + k],
//                   y_log[(offsets[b] + k0 + k) * n + c]  (time-major like the reference's Solution.t / Solution.y,
//                   src/solve/solout.rs:387-428, src/solve/solve_ivp.rs:288-312).  Lane <-> double of the segment: the
//                   n + 1 doubles of a record are read by adjacent lanes, the y part of consecutive records is written to
//                   consecutive addresses.
// Both are pure data movement: HBM-bound, ~2 x 8 (n + 1) bytes per record.
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

// G lanes per trajectory (64 / G trajectories per wavefront), NP1 = n + 1 as a compile-time constant where it is small
// (0 = run-time).  `capacity` = records the destination holds: a log that does not fit is left alone (the host reports it;
// nothing is written out of bounds).
template <int G, int NP1>
__global__ __launch_bounds__(IVP_WAVE) void log_gather_kernel(const double *pool, const unsigned long long *log_cur, const uint32_t *n_log,
                                                              const unsigned long long *offsets, uint32_t B, uint32_t n_rt,
                                                              unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log)
{
    const uint32_t np1 = NP1 ? (uint32_t)NP1 : n_rt + 1u, n = np1 - 1u;
    const uint32_t gl = threadIdx.x & (G - 1);
    const uint32_t j = blockIdx.x * (IVP_WAVE / G) + threadIdx.x / G;
    if (j >= B || offsets[B] > capacity) return;
    if (n_log[j] == 0u) return;
    const unsigned long long off = offsets[j] + dst_base;
    unsigned long long seg = log_cur[j];
    const uint32_t elems = IVP_LOG_SLOTS * np1;
    while (seg != IVP_NO_SEG) {
        const size_t base = (size_t)(seg >> 16), cols = (size_t)((seg >> 8) & 0xFFu), col = (size_t)(seg & 0xFFu);
        const double *hdr = pool + base + 2u * col;
        const unsigned long long prev = *(const unsigned long long *)hdr;
        const uint32_t k0 = ((const uint32_t *)hdr)[2], bits = ((const uint32_t *)hdr)[3];
        const double *body = pool + base + 2u * cols + col * np1;
        const size_t slot_stride = cols * np1;
        const unsigned long long q0 = off + k0;
        if (bits != 0u) {
            for (uint32_t e = gl; e < elems; e += G) {
                const uint32_t s = e / np1, c = e - s * np1;
                if ((bits >> s) & 1u) {
                    const unsigned long long q = q0 + (unsigned long long)__popc(bits & ((1u << s) - 1u));
                    const double v = body[(size_t)s * slot_stride + c];
                    if (c == 0u) t_log[q] = v;
                    else y_log[q * n + (c - 1u)] = v;
                }
            }
        }
        seg = prev;
    }
}

template <int G>
hipError_t launch_gather(const double *pool, const unsigned long long *log_cur, const uint32_t *n_log, const unsigned long long *offsets, size_t B, int n,
                         unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log, hipStream_t s)
{
    const dim3 grid((uint32_t)((B + (IVP_WAVE / G) - 1) / (IVP_WAVE / G))), block(IVP_WAVE);
#define IVP_GATHER_CASE(NP1) case NP1: hipLaunchKernelGGL((log_gather_kernel<G, NP1>), grid, block, 0, s, pool, log_cur, n_log, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log); break;
    switch (n + 1) {
        IVP_GATHER_CASE(2) IVP_GATHER_CASE(3) IVP_GATHER_CASE(4) IVP_GATHER_CASE(5) IVP_GATHER_CASE(6) IVP_GATHER_CASE(7) IVP_GATHER_CASE(8) IVP_GATHER_CASE(9)
    default: hipLaunchKernelGGL((log_gather_kernel<G, 0>), grid, block, 0, s, pool, log_cur, n_log, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log); break;
    }
#undef IVP_GATHER_CASE
    return hipGetLastError();
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

hipError_t ivp_log_gather(const double *pool, const unsigned long long *log_cur, const uint32_t *n_log, const unsigned long long *offsets, size_t B,
                          int n, unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    (void)hipGetLastError();
    // a segment is 32 x (n + 1) doubles: as many lanes per trajectory as keep ~3 doubles per lane and segment in flight, and as
    // many trajectories per wavefront as that leaves room for (the chain walk is a dependent load per segment: more
    // independent chains per wave hide it)
    const int elems = (int)IVP_LOG_SLOTS * (n + 1);
    if (elems >= 192) return launch_gather<64>(pool, log_cur, n_log, offsets, B, n, capacity, dst_base, t_log, y_log, s);
    if (elems >= 96) return launch_gather<32>(pool, log_cur, n_log, offsets, B, n, capacity, dst_base, t_log, y_log, s);
    return launch_gather<16>(pool, log_cur, n_log, offsets, B, n, capacity, dst_base, t_log, y_log, s);
}
