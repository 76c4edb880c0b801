// rk_global.h -- __global__ wrappers around the per-lane bodies of rk_core.h (device only).
// Shared by the ahead-of-time kernels (rk_kernels.hip) and the hiprtc path (ivp_jit.cpp embeds this
// text together with ivp_kargs.h and rk_core.h).
#pragma once
#ifndef IVP_MIN_WAVES
#define IVP_MIN_WAVES 2
#endif
#ifndef IVP_BDF_MIN_WAVES
#define IVP_BDF_MIN_WAVES 1
#endif

namespace IVP_NS {

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Append the ids of still-running trajectories to perm_out: ballot -> popcount -> one atomicAdd per
// wave -> each lane writes at base + its rank among the set bits.
__device__ __forceinline__ void compact_append(const IvpKArgs &a, uint32_t j, bool still)
{
    const unsigned long long m = __ballot(still);
    if (m == 0ull) return;
    const uint32_t lane = lane_id();
    const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(a.count_out, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (still) a.perm_out[base + rank] = j;
}

template <int M, class R, int FULL>
__global__ __launch_bounds__(IVP_WAVE) void init_kernel_t(const IvpKArgs a)
{
    const uint32_t i = blockIdx.x * IVP_WAVE + threadIdx.x;
    const bool valid = i < a.B;
    int32_t st = 0;
    if (valid) st = any_init_body<M, R, FULL>(a, i);
    if (a.perm_out) compact_append(a, i, valid && st == IVP_RUNNING);
}

// Deferred t_eval sampling (flavour 3): one lane per noted step.  blockIdx.x tiles the trajectories (consecutive lanes,
// consecutive trajectories: the record fields are read coalesced), blockIdx.y strides over a trajectory's noted steps.
template <class R>
__global__ __launch_bounds__(IVP_WAVE) void sample_kernel_t(const IvpKArgs a)
{
    const uint32_t j = blockIdx.x * IVP_WAVE + threadIdx.x;
    if (j >= a.B) return;
    const uint32_t nd = min(a.n_seg[j], a.def_cap);
    for (uint32_t k = blockIdx.y; k < nd; k += gridDim.y) dop853_sample_body<R>(a, j, k);
}

// Deferred event refinement: one lane per noted step, same tiling as the sample kernel (so_events_deferred_body, rk_core.h).
template <int M, class R>
__device__ __forceinline__ void event_kernel_body(const IvpKArgs &a)
{
    const uint32_t j = blockIdx.x * IVP_WAVE + threadIdx.x;
    if (j >= a.B) return;
    if constexpr (R::NE > 0 && M != M_BDF) {
        const uint32_t nd = min(a.evd_cnt[j], a.evd_cap);
        for (uint32_t q = blockIdx.y; q < nd; q += gridDim.y) so_events_deferred_body<M, R>(a, j, q);
    }
}
template <int M, class R>
__global__ __launch_bounds__(IVP_WAVE) void event_kernel_t(const IvpKArgs a) { event_kernel_body<M, R>(a); }

template <int M, class R, int FULL, bool CTL = false>
__device__ __forceinline__ void chunk_kernel_body(const IvpKArgs &a)
{
    const uint32_t lpw = a.lpw ? a.lpw : (uint32_t)IVP_WAVE;   // trajectories per wave (thin waves: ivp_kargs.h)
    const uint32_t i = blockIdx.x * lpw + threadIdx.x;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (a.count_next && blockIdx.x == 0 && threadIdx.x == 0) *a.count_next = 0u;   // before any early exit: both halves of a pair do it
    if (a.spec_min && count <= a.spec_min) return;   // paired launch: the set is small enough for the cooperative kernel (uniform)
    if (blockIdx.x * lpw >= count) return;  // whole wave beyond the active set (stale grid bound)
    if (a.ran_out && blockIdx.x == 0 && threadIdx.x == 0) *a.ran_out = 1u;
    const bool valid = threadIdx.x < lpw && i < count;
    uint32_t j = 0;
    bool active = false;
    if (valid) {
        j = a.perm_in ? a.perm_in[i] : i;
        active = a.status[j] == IVP_RUNNING;
    }
    if (a.window && blockIdx.x * lpw >= a.window) {   // behind the window (wave-uniform): wait for a later launch
        compact_append(a, j, active);
        return;
    }
    uint32_t it = 0;
    int32_t st = 0;
    if (active) it = any_chunk_body<M, R, FULL, CTL>(a, j, st);
    const bool still = active && st == IVP_RUNNING;
    compact_append(a, j, still);
    if (a.slot_counter) {
        // divergence accounting: lanes launched x attempts this wave actually ran (= its slowest lane)
        uint32_t mx = it;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        const unsigned long long act = __ballot(active);
        if (lane_id() == 0) {
            atomicAdd(a.slot_counter, (unsigned long long)mx * IVP_WAVE);
            atomicAdd(a.slot_counter + 1, (unsigned long long)__popcll(act));  // trajectories that moved state this launch
        }
    }
}

// Occupancy hint: two waves per SIMD (256 VGPRs) for the lean build, except where the live set cannot fit them without
// scratch: BDF, and DOP853 with the device DefaultSolOut on systems of 5+ components (10 stage vectors + the SolOut state).
// CTL = true: controller fields from IvpKArgs.ctl_* (a direct method call with non-default struct fields);
// CTL = false keeps them compile-time constants, which is the path solve_ivp() takes.
template <int M, class R, int FULL, bool CTL = false>
// (tried for the recording flavours 2 / 3 of BASELINE C3's system: forcing the end-state kernel's 5 waves per SIMD -- 96 VGPRs
// -- spills 20 registers to scratch and is slower than 4 waves at 118: t_eval 20.1 -> 22.0 ms, step log 21.7 -> 22.6)
__global__ __launch_bounds__(IVP_WAVE, (M == M_BDF) ? IVP_BDF_MIN_WAVES : ((M == M_DOP853 && FULL == 1 && R::N >= 5) ? 1 : IVP_MIN_WAVES)) void chunk_kernel_t(const IvpKArgs a)
{
    chunk_kernel_body<M, R, FULL, CTL>(a);
}

}  // namespace IVP_NS
