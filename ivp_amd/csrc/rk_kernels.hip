// rk_kernels.hip -- gfx950 kernels of the batched explicit RK integrator and their launch table.
//
// Compiled twice (see Makefile):
//   -DIVP_FAST=0 -ffp-contract=off   -> ivp_launch_strict   (reference operation order, no FMA fusion)
//   -DIVP_FAST=1 -ffp-contract=fast  -> ivp_launch_fast     (FMA contraction + reciprocal sharing)
// and each of those once more with -DIVP_HOIST=1 (coefficients resident in registers, see KC() in rk_core.h)
//                                    -> ivp_launch_strict_hoist / ivp_launch_fast_hoist
//
// Launch geometry: one 64-lane wavefront per workgroup, one lane per trajectory.  There is no LDS
// and no barrier (trajectories are independent), so a one-wave workgroup frees its SIMD slot the
// moment its slowest lane retires and the dispatcher back-fills it; grids are >> 256 CUs x 4 SIMDs
// for every BASELINE config (100k trajectories = 1563 waves, 1M = 15625).  SoA state makes each
// per-component access one contiguous 512-byte wave transaction.  Lanes whose trajectory has retired
// are predicated off inside a chunk; between chunks the still-running ids are compacted with a
// wave ballot + one atomic per wave so the next launch runs dense wavefronts again.
#include <hip/hip_runtime.h>

#define IVP_HD __host__ __device__ __forceinline__
#ifndef IVP_HOIST
#define IVP_HOIST 0
#endif
#if IVP_FAST && IVP_HOIST
#define IVP_NS ivp_fast_h
#define IVP_LAUNCH_NAME ivp_launch_fast_hoist
#elif IVP_FAST
#define IVP_NS ivp_fast
#define IVP_LAUNCH_NAME ivp_launch_fast
#elif IVP_HOIST
#define IVP_NS ivp_strict_h
#define IVP_LAUNCH_NAME ivp_launch_strict_hoist
#else
#define IVP_NS ivp_strict
#define IVP_LAUNCH_NAME ivp_launch_strict
#endif
#include "rk_core.h"
#include "bdf_core.h"
#include "rk_global.h"
#include "rk_launch.h"

namespace {

using namespace IVP_NS;

template <int M, class R, int FULL>
hipError_t launch_one(int what, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    const dim3 grid((lanes + IVP_WAVE - 1) / IVP_WAVE), block(IVP_WAVE);
    if (grid.x == 0) return hipSuccess;
    (void)hipGetLastError();   // drop a stale error of some earlier runtime call: the value returned below is this launch's
    if (what == IVP_LAUNCH_INIT) {
        hipLaunchKernelGGL((init_kernel_t<M, R, FULL>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    // a direct method call with its own controller fields (IvpKArgs.has_ctl) runs the CTL = true instantiation;
    // only the lean builds carry it (the host never selects the resident-coefficient variant for such a call)
    constexpr bool kHasCtl = !IVP_HOIST && (M == M_RK23 || M == M_DOPRI5 || M == M_DOP853);
    if constexpr (kHasCtl) {
        if (a.has_ctl) {
            hipLaunchKernelGGL((chunk_kernel_t<M, R, FULL, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
    } else if (a.has_ctl) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL((chunk_kernel_t<M, R, FULL>), grid, block, 0, s, a);
    return hipGetLastError();
}

// flavour 2 (log-only) exists for the adaptive methods of problems without event functions; everything else runs it as 1
template <int M, class R>
hipError_t launch_flavour(int what, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    if constexpr (R::NE == 0 && M != M_RK4) {
        if (full == 2) return launch_one<M, R, 2>(what, a, lanes, s);
    }
    if constexpr (R::NE == 0 && M == M_DOP853) {
        if (full == 3) {
#if !IVP_HOIST
            if (what == IVP_LAUNCH_SAMPLE) {   // one lane per noted step: grid.y strides over a trajectory's noted steps
                const dim3 grid((lanes + IVP_WAVE - 1) / IVP_WAVE, 8), block(IVP_WAVE);
                (void)hipGetLastError();
                hipLaunchKernelGGL((sample_kernel_t<R>), grid, block, 0, s, a);
                return hipGetLastError();
            }
#endif
            return launch_one<M, R, 3>(what, a, lanes, s);
        }
    }
    if (what == IVP_LAUNCH_SAMPLE) return hipErrorInvalidValue;
    if (what == IVP_LAUNCH_EVENTS) {   // one lane per noted step: grid.y strides over a trajectory's noted steps
        if constexpr (R::NE > 0 && !IVP_HOIST) {
            const dim3 grid((lanes + IVP_WAVE - 1) / IVP_WAVE, 4), block(IVP_WAVE);
            (void)hipGetLastError();
            hipLaunchKernelGGL((event_kernel_t<M, R>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    return full ? launch_one<M, R, 1>(what, a, lanes, s) : launch_one<M, R, 0>(what, a, lanes, s);
}

template <class R>
hipError_t launch_rhs(int what, int method, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    switch (method) {
    case M_RK23: return launch_flavour<M_RK23, R>(what, full, a, lanes, s);
    case M_DOPRI5: return launch_flavour<M_DOPRI5, R>(what, full, a, lanes, s);
    case M_DOP853: return launch_flavour<M_DOP853, R>(what, full, a, lanes, s);
    case M_RK4: return launch_flavour<M_RK4, R>(what, full, a, lanes, s);
    case M_BDF:
        return hipErrorInvalidValue;   // BDF lives in rk_bdf.hip (pinned-coefficient build only)
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t IVP_LAUNCH_NAME(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    switch (rhs_id) {
    case 0: return launch_rhs<IVP_NS::RhsDecay>(what, method, full, a, lanes, s);
    case 1: return launch_rhs<IVP_NS::RhsSho>(what, method, full, a, lanes, s);
    case 2: return launch_rhs<IVP_NS::RhsVdp>(what, method, full, a, lanes, s);
    case 3: return launch_rhs<IVP_NS::RhsCr3bp>(what, method, full, a, lanes, s);
    case 4: return launch_rhs<IVP_NS::RhsLorenz>(what, method, full, a, lanes, s);
    case 5: return launch_rhs<IVP_NS::RhsZero>(what, method, full, a, lanes, s);
    case 6: return launch_rhs<IVP_NS::RhsRational>(what, method, full, a, lanes, s);
    case 7: return launch_rhs<IVP_NS::RhsExp2>(what, method, full, a, lanes, s);
    case 8: return launch_rhs<IVP_NS::RhsLinear>(what, method, full, a, lanes, s);
    case 9: return launch_rhs<IVP_NS::RhsRobertson>(what, method, full, a, lanes, s);
    case 10: return launch_rhs<IVP_NS::RhsVdpEps>(what, method, full, a, lanes, s);
    case 11: return launch_rhs<IVP_NS::RhsShoEv>(what, method, full, a, lanes, s);
    case 12: return launch_rhs<IVP_NS::RhsBall>(what, method, full, a, lanes, s);
    case 13: return launch_rhs<IVP_NS::RhsCannon>(what, method, full, a, lanes, s);
    case 14: return launch_rhs<IVP_NS::RhsRationalEv>(what, method, full, a, lanes, s);
    case 15: return launch_rhs<IVP_NS::RhsRobertsonJac>(what, method, full, a, lanes, s);
    }
    return hipErrorInvalidValue;
}
