// log_gather.hip -- second half of the one-pass accepted-step log (gfx950): wave pages -> CSR.
//
// The stepping kernels leave the records in self-describing wave pages (layout in ivp_kargs.h, writers so_log_open /
// so_push_log in rk_core.h) and every trajectory's record count in n_log.  Here:
//   ivp_log_scan    offsets[b] = sum of n_log[0 .. b), offsets[B] = total          (three small launches)
//   ivp_log_gather  one wavefront per COLUMN GROUP of a page (8 trajectories x 32 slots x (n + 1) doubles, one dense block):
//                   the block is read front to back -- consecutive lanes, consecutive doubles -- into LDS, then every
//                   column's records leave as two contiguous runs (t, y) at their place in the log: record (j, k0 + rank of
//                   its slot among the column's set bits) -> t_log[offsets[j] + ..], y_log[(offsets[j] + ..) * n + c],
//                   time-major like the reference's Solution.t / Solution.y (src/solve/solout.rs:387-428,
//                   src/solve/solve_ivp.rs:288-312).  Whole cache lines on both sides, each touched once.
// Both are pure data movement: HBM-bound, ~2 x 8 (n + 1) bytes per record (+ the holes rejected attempts leave in a page).
#include <hip/hip_runtime.h>

#include "ivp_kargs.h"
#include "log_gather.h"

#include <algorithm>

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

// One workgroup (4 wavefronts) per ARENA (blockIdx.x = its directory index, blockIdx.y = sub-pool), one wavefront per column
// group at a time: the arena's (page, group) items are dealt round-robin to the four waves.  (One tiny workgroup per group
// was limited by the rate at which workgroups can be launched: 266k of them for BASELINE C2, 4M for C3.)
// NP1 = n + 1 as a compile-time constant where it is small (0 = run-time).  `capacity` = records the destination holds: a
// log that does not fit is left alone (the host reports it; nothing is written out of bounds).
// LDS (dynamic), per wavefront: the whole group where it is small (n <= 8: at most 32 x 8 x 9 doubles), a tile of its slots
// otherwise.
constexpr int kGatherThreads = 256, kGatherWaves = kGatherThreads / IVP_WAVE;
constexpr int kLoadUnroll = 8;        // loads a lane has in flight per round of the tiled path
template <int NP1>
__global__ __launch_bounds__(kGatherThreads) void log_gather_kernel(const double *pool, unsigned long long region, const unsigned long long *alloc,
                                                                    const unsigned long long *offsets, uint32_t B, uint32_t n_rt,
                                                                    unsigned long long capacity, unsigned long long dst_base, double *t_log, double *y_log,
                                                                    uint32_t tile_doubles)
{
    const uint32_t np1 = NP1 ? (uint32_t)NP1 : n_rt + 1u, n = np1 - 1u, W = IVP_LOG_GROUP(np1);
    const uint32_t sub = blockIdx.y, e = blockIdx.x;
    if ((unsigned long long)e >= (alloc[(size_t)sub * IVP_LOG_ALLOC_STRIDE] >> 40) || offsets[B] > capacity) return;
    const unsigned long long entry = ((const unsigned long long *)pool)[(size_t)(sub + 1u) * region - 1u - e];
    const uint32_t pages = (uint32_t)(entry & 3u) + 1u;
    const size_t acols = (size_t)((entry >> 2) & 0x3Fu) + 1u;
    const uint32_t agroups = (uint32_t)((acols + W - 1u) / W);
    // (arenas of more than one page are made of IVP_LOG_SLOTS-slot pages: so_log_attempt in rk_core.h)
    const size_t page_stride = IVP_LOG_HDR(acols) + (((size_t)agroups * (size_t)IVP_LOG_SLOTS * W * np1 + 15u) & ~(size_t)15u);   // = ivp_log_page_doubles (rk_core.h)
    const uint32_t lane = threadIdx.x & (IVP_WAVE - 1), wv = threadIdx.x / IVP_WAVE, row = W * np1;
    extern __shared__ double tiles[];
    double *tile = tiles + (size_t)wv * tile_doubles;
    __shared__ unsigned char s_slots[kGatherWaves][8][IVP_LOG_SLOTS];   // [wave][c][r] = the slot of column c's r-th record
    unsigned char (*s_slot)[IVP_LOG_SLOTS] = s_slots[wv];
    constexpr bool whole = NP1 != 0 && NP1 <= 9;
    constexpr int kGroupLoads = whole ? (int)(IVP_LOG_SLOTS * 8u * (unsigned)NP1 / IVP_WAVE) : 1;   // loads per lane: the whole group
    const uint32_t waves = blockDim.x / IVP_WAVE;
    for (uint32_t item0 = 0; item0 < pages * agroups; item0 += waves) {   // (the same trip count in every wave: barriers inside)
        const uint32_t item = item0 + wv;
        const uint32_t p = item / agroups, g = item - p * agroups;
        const size_t page = (size_t)(entry >> 8) + (size_t)p * page_stride;
        uint32_t cols = 0, slots = 0;
        if (p < pages) { cols = ((const uint32_t *)(pool + page))[0]; slots = ((const uint32_t *)(pool + page))[1]; }
        const bool have = cols != 0u && g * W < cols;   // not: a page its wave never opened / a group the page does not have
        const uint32_t c0 = g * W, gc = have ? min(W, cols - c0) : 0u;
        const double *src = pool + page + IVP_LOG_HDR((size_t)cols) + (size_t)g * slots * row;
        // The group in one piece (n <= 8): its loads are issued FIRST -- they depend on the page header only -- and travel
        // while the column headers and the trajectories' offsets are fetched.
        double v[kGroupLoads];
        if (whole) {
            const uint32_t count = have ? slots * row : 0u;
#pragma unroll
            for (int u = 0; u < kGroupLoads; ++u) { const uint32_t x = lane + (uint32_t)u * IVP_WAVE; v[u] = x < count ? src[x] : 0.0; }
        }
        uint32_t my_bits = 0;
        unsigned long long my_q0 = 0;
        if (lane < gc) {
            const uint32_t *hdr = (const uint32_t *)(pool + page + 1u + 2u * (size_t)(c0 + lane));
            my_bits = hdr[2];
            if (my_bits) my_q0 = offsets[hdr[0]] + hdr[1] + dst_base;
        }
        uint32_t used = 0;
        __syncthreads();   // the previous item's stores have read s_slot / the tile
        for (uint32_t c = 0; c < gc; ++c) {
            const uint32_t b = (uint32_t)__shfl((int)my_bits, (int)c);
            used |= b;
            if (lane < IVP_LOG_SLOTS && ((b >> lane) & 1u)) s_slot[c][__popc(b & ((1u << lane) - 1u))] = (unsigned char)lane;
        }
        const uint32_t last = used ? 32u - (uint32_t)__clz(used) : 0u;   // slots [0, last) are in use
        const uint32_t ts = whole ? max(slots, 1u) : max(1u, min(slots, tile_doubles / row));
        // (the tile loop runs once for n <= 8; its trip count is not uniform across the waves for larger systems, whose
        // workgroups therefore run ONE wave: see the launch)
        for (uint32_t s0 = 0; s0 < max(last, 1u); s0 += ts) {
            const uint32_t s1 = min(s0 + ts, last);
            if (whole) {
                const uint32_t count = have ? slots * row : 0u;
#pragma unroll
                for (int u = 0; u < kGroupLoads; ++u) { const uint32_t x = lane + (uint32_t)u * IVP_WAVE; if (x < count) tile[x] = v[u]; }
            } else {
                if (s0 > 0u) __syncthreads();
                const uint32_t count = s1 > s0 ? (s1 - s0) * row : 0u;
                const double *in = src + (size_t)s0 * row;
                for (uint32_t x0 = lane; x0 < count; x0 += IVP_WAVE * kLoadUnroll) {
                    double w[kLoadUnroll];
#pragma unroll
                    for (int u = 0; u < kLoadUnroll; ++u) { const uint32_t x = x0 + (uint32_t)u * IVP_WAVE; w[u] = x < count ? in[x] : 0.0; }
#pragma unroll
                    for (int u = 0; u < kLoadUnroll; ++u) { const uint32_t x = x0 + (uint32_t)u * IVP_WAVE; if (x < count) tile[x] = w[u]; }
                }
            }
            __syncthreads();
            const uint32_t tmask = s1 > s0 ? ((s1 >= 32u ? 0xFFFFFFFFu : ((1u << s1) - 1u)) & ~((1u << s0) - 1u)) : 0u;
            for (uint32_t c = 0; c < gc; ++c) {
                const uint32_t bits = (uint32_t)__shfl((int)my_bits, (int)c);
                const unsigned long long q00 = ((unsigned long long)(uint32_t)__shfl((int)(my_q0 >> 32), (int)c) << 32) | (uint32_t)__shfl((int)(uint32_t)my_q0, (int)c);
                const uint32_t r0 = (uint32_t)__popc(bits & ((1u << s0) - 1u)), cnt = (uint32_t)__popc(bits & tmask);
                if (cnt == 0u) continue;
                const unsigned long long q = q00 + r0;
                const double *col = tile + (size_t)c * np1;
                if (lane < cnt) t_log[q + lane] = col[((uint32_t)s_slot[c][r0 + lane] - s0) * row];
                double *dy = y_log + q * n;
                for (uint32_t x = lane; x < cnt * n; x += IVP_WAVE) {
                    const uint32_t r = x / n, cc = x - r * n;
                    dy[x] = col[((uint32_t)s_slot[c][r0 + r] - s0) * row + 1u + cc];
                }
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
    // n <= 8: four wavefronts per workgroup, each with LDS for a whole column group (32 slots x 8 columns x (n + 1) doubles);
    // larger systems (one column per group, the group walked in tiles of at most 32 KB of its slots): one wavefront
    const uint32_t np1 = (uint32_t)n + 1u, row = IVP_LOG_GROUP(np1) * np1;
    const bool small = np1 <= 9u;
    const uint32_t tile_doubles = small ? IVP_LOG_SLOTS * row : std::max<uint32_t>(row, std::min<uint32_t>(IVP_LOG_SLOTS * row, 4096u));
    const dim3 grid(max_arenas, subs), block(small ? kGatherThreads : IVP_WAVE);
    const size_t lds = (size_t)tile_doubles * sizeof(double) * (small ? kGatherWaves : 1);
#define IVP_GATHER_CASE(NP1) case NP1: hipLaunchKernelGGL((log_gather_kernel<NP1>), grid, block, lds, s, pool, region, alloc, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log, tile_doubles); break;
    switch (n + 1) {
        IVP_GATHER_CASE(2) IVP_GATHER_CASE(3) IVP_GATHER_CASE(4) IVP_GATHER_CASE(5) IVP_GATHER_CASE(6) IVP_GATHER_CASE(7) IVP_GATHER_CASE(8) IVP_GATHER_CASE(9)
    default: hipLaunchKernelGGL((log_gather_kernel<0>), grid, block, lds, s, pool, region, alloc, offsets, (uint32_t)B, (uint32_t)n, capacity, dst_base, t_log, y_log, tile_doubles); break;
    }
#undef IVP_GATHER_CASE
    return hipGetLastError();
}
