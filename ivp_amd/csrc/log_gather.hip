// log_gather.hip -- second half of the one-pass accepted-step log (gfx950): wave pages -> CSR.
//
// The stepping kernels leave the records in self-describing wave pages (layout in ivp_kargs.h, writers so_log_open /
// so_push_log in rk_core.h) and every trajectory's record count in n_log.  Here:
//   ivp_log_scan    offsets[b] = sum of n_log[0 .. b), offsets[B] = total          (three small launches)
//   ivp_log_gather  one workgroup per page: the column headers (trajectory j, first record index k0, slot bits) go to LDS,
//                   then the page is read front to back -- consecutive lanes read consecutive doubles -- and the record in
//                   slot s of column c goes to t_log[q], y_log[q * n + ..] with q = offsets[j] + k0 + (rank of s among the
//                   column's set bits): time-major like the reference's Solution.t / Solution.y (src/solve/solout.rs:387-428,
//                   src/solve/solve_ivp.rs:288-312).  A column's records land on consecutive addresses slot after slot, so
//                   the scattered 8 (n + 1)-byte writes complete their cache lines while those are still in L2.
// Both are pure data movement: HBM-bound, ~2 x 8 (n + 1) bytes per record (+ the holes rejected attempts leave in a page).
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

// One workgroup (4 wavefronts) per page: blockIdx.y = sub-pool, blockIdx.x = 4 * (arena's directory index) + page of the
// arena.  NP1 = n + 1 as a compile-time constant where it is small (0 = run-time).  `capacity` = records the destination
// holds: a log that does not fit is left alone (the host reports it; nothing is written out of bounds).
//
// A page is slot-major (the records of one attempt side by side: what makes the stepping kernels' stores coalesced), the log
// is trajectory-major: the transposition goes through LDS.  A tile of slots is read front to back -- consecutive lanes,
// consecutive doubles -- into LDS; then every column's records of that tile leave as TWO contiguous runs (t, y) at their place
// in the log: whole cache lines, written once.  (Writing straight from the slot-major order scatters 8 (n + 1)-byte pieces
// over as many destinations as the page has columns, with ~2000 pages in flight: the partly written lines fall out of L2
// and every piece becomes a read-modify-write in HBM -- measured 1.2 ms instead of 0.4 on BASELINE C2.)
constexpr int kGatherThreads = 256;
constexpr int kTileDoubles = 7680;    // 60 KB of LDS per workgroup: two workgroups per CU
template <int NP1>
__global__ __launch_bounds__(kGatherThreads) void log_gather_kernel(const double *pool, unsigned long long region, const unsigned long long *alloc,
                                                                    const unsigned long long *offsets, uint32_t B, uint32_t n_rt,
                                                                    unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log)
{
    const uint32_t np1 = NP1 ? (uint32_t)NP1 : n_rt + 1u, n = np1 - 1u;
    const uint32_t sub = blockIdx.y, e = blockIdx.x >> 2, p = blockIdx.x & 3u;
    if ((unsigned long long)e >= (alloc[(size_t)sub * IVP_LOG_ALLOC_STRIDE] >> 40) || offsets[B] > capacity) return;
    const unsigned long long entry = ((const unsigned long long *)pool)[(size_t)(sub + 1u) * region - 1u - e];
    if (p > (uint32_t)(entry & 3u)) return;
    const size_t acols = (size_t)((entry >> 2) & 0x3Fu) + 1u;
    const size_t page = (size_t)(entry >> 8) + (size_t)p * (1u + acols * (2u + (size_t)IVP_LOG_SLOTS * np1));
    const uint32_t cols = *(const uint32_t *)(pool + page);
    if (cols == 0u) return;   // a page of the arena its wave never opened
    __shared__ double tile[kTileDoubles];
    __shared__ uint32_t s_bits[IVP_WAVE];
    __shared__ unsigned long long s_q0[IVP_WAVE];
    __shared__ unsigned char s_slot[IVP_WAVE][IVP_LOG_SLOTS];   // s_slot[col][r] = the slot of the column's r-th record
    __shared__ uint32_t s_used;
    if (threadIdx.x == 0) s_used = 0u;
    __syncthreads();
    if (threadIdx.x < cols) {
        const uint32_t *hdr = (const uint32_t *)(pool + page + 1u + 2u * (size_t)threadIdx.x);
        const uint32_t bits = hdr[2];
        s_bits[threadIdx.x] = bits;
        s_q0[threadIdx.x] = offsets[hdr[0]] + hdr[1] + dst_base;
        uint32_t r = 0;
        for (uint32_t sl = 0; sl < IVP_LOG_SLOTS; ++sl)
            if ((bits >> sl) & 1u) s_slot[threadIdx.x][r++] = (unsigned char)sl;
        if (bits) atomicOr(&s_used, bits);
    }
    __syncthreads();
    const uint32_t used = s_used;
    if (used == 0u) return;
    const uint32_t last = 32u - (uint32_t)__clz(used);   // slots [0, last) are in use
    const double *body = pool + page + 1u + 2u * (size_t)cols;
    const uint32_t row = cols * np1;
    const uint32_t ts = min(IVP_LOG_SLOTS, (uint32_t)kTileDoubles / row);   // slots per tile (row <= 2052 doubles: ts >= 3)
    const uint32_t tx = threadIdx.x & (IVP_WAVE - 1), ty = threadIdx.x / IVP_WAVE;
    for (uint32_t s0 = 0; s0 < last; s0 += ts) {
        const uint32_t s1 = min(s0 + ts, last);
        __syncthreads();   // the previous tile has left LDS
        for (uint32_t x = threadIdx.x; x < (s1 - s0) * row; x += kGatherThreads) tile[x] = body[(size_t)s0 * row + x];
        __syncthreads();
        const uint32_t tmask = (s1 >= 32u ? 0xFFFFFFFFu : ((1u << s1) - 1u)) & ~((1u << s0) - 1u);
        for (uint32_t col = ty; col < cols; col += kGatherThreads / IVP_WAVE) {
            const uint32_t bits = s_bits[col];
            const uint32_t r0 = (uint32_t)__popc(bits & ((1u << s0) - 1u)), cnt = (uint32_t)__popc(bits & tmask);
            if (cnt == 0u) continue;
            const unsigned long long q = s_q0[col] + r0;
            const double *src = tile + (size_t)col * np1;
            if (tx < cnt) t_log[q + tx] = src[((uint32_t)s_slot[col][r0 + tx] - s0) * row];
            double *dy = y_log + q * n;
            for (uint32_t x = tx; x < cnt * n; x += IVP_WAVE) {
                const uint32_t r = x / n, c = x - r * n;
                dy[x] = src[((uint32_t)s_slot[col][r0 + r] - s0) * row + 1u + c];
            }
        }
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

hipError_t ivp_log_gather(const double *pool, unsigned long long region, const unsigned long long *alloc, uint32_t subs, uint32_t max_arenas,
                          const unsigned long long *offsets, size_t B, int n, unsigned long long capacity, unsigned long long dst_base,
                          double *t_log, double *y_log, hipStream_t s)
{
    if (B == 0 || max_arenas == 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 grid(4u * max_arenas, subs), block(kGatherThreads);
#define IVP_GATHER_CASE(NP1) case NP1: hipLaunchKernelGGL((log_gather_kernel<NP1>), grid, block, 0, s, pool, region, alloc, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log); break;
    switch (n + 1) {
        IVP_GATHER_CASE(2) IVP_GATHER_CASE(3) IVP_GATHER_CASE(4) IVP_GATHER_CASE(5) IVP_GATHER_CASE(6) IVP_GATHER_CASE(7) IVP_GATHER_CASE(8) IVP_GATHER_CASE(9)
    default: hipLaunchKernelGGL((log_gather_kernel<0>), grid, block, 0, s, pool, region, alloc, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log); break;
    }
#undef IVP_GATHER_CASE
    return hipGetLastError();
}
