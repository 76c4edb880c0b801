// rk_group.hip -- the kernels in which several lanes share one trajectory, and their launch tables:
//   * rk_group.h: wave-per-trajectory RK23 / DOPRI5 / DOP853 / RK4 for large state dimensions (8 < n <= 512);
//   * rk_coop.h:  eight lanes per trajectory (n <= 8) for the latency-bound tail of a batch.  Compiled twice like rk_kernels.hip: strict (-ffp-contract=off, index-order error-norm sum) and
// fast (-ffp-contract=fast, __shfl_xor butterfly).  Coefficients are pinned in vector registers (IVP_HOIST = 2, see
// KC() in rk_core.h): a lone wave per SIMD pays an issue slot for every re-materialised constant and has VGPRs to spare.
#include <hip/hip_runtime.h>

#define IVP_HD __host__ __device__ __forceinline__
#define IVP_HOIST 2
#if IVP_FAST
#define IVP_NS ivp_group_fast
#define IVP_LAUNCH_NAME ivp_launch_group_fast
#define IVP_COOP_LAUNCH_NAME ivp_launch_coop_fast
#else
#define IVP_COOP_LAUNCH_NAME ivp_launch_coop_strict
#define IVP_NS ivp_group_strict
#define IVP_LAUNCH_NAME ivp_launch_group_strict
#endif
#include "rk_core.h"
#include "bdf_core.h"
#include "rk_global.h"
#include "rk_group.h"
#include "bdf_group.h"
#include "rk_coop.h"
#include "rk_launch.h"

namespace {

template <int M, class R, int FULL>
hipError_t launch_group_one(int what, const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    const dim3 grid(trajectories), block(IVP_WAVE);   // one wavefront per trajectory
    if (grid.x == 0) return hipSuccess;
    (void)hipGetLastError();   // drop a stale error of some earlier runtime call: the value returned below is this launch's
    if (what == IVP_LAUNCH_INIT) {
        hipLaunchKernelGGL((IVP_NS::group_init_kernel<M, R, FULL>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if constexpr (M == IVP_NS::M_BDF && R::N <= IVP_LDS_LU_MAX_N) {
        if (a.lds_lu) {   // factors of (I - cJ) resident in LDS (bdf_group.h)
            hipLaunchKernelGGL((IVP_NS::group_chunk_kernel<M, R, FULL, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((IVP_NS::group_chunk_kernel<M, R, FULL>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <class R>
hipError_t launch_group(int what, int method, int full, const IvpKArgs &a, uint32_t n, hipStream_t s)
{
    using namespace IVP_NS;
    switch (method) {
    case M_RK23: return full ? launch_group_one<M_RK23, R, true>(what, a, n, s) : launch_group_one<M_RK23, R, false>(what, a, n, s);
    case M_DOPRI5: return full ? launch_group_one<M_DOPRI5, R, true>(what, a, n, s) : launch_group_one<M_DOPRI5, R, false>(what, a, n, s);
    case M_DOP853: return full ? launch_group_one<M_DOP853, R, true>(what, a, n, s) : launch_group_one<M_DOP853, R, false>(what, a, n, s);
    case M_RK4: return full ? launch_group_one<M_RK4, R, true>(what, a, n, s) : launch_group_one<M_RK4, R, false>(what, a, n, s);
    case M_BDF: return full ? launch_group_one<M_BDF, R, true>(what, a, n, s) : launch_group_one<M_BDF, R, false>(what, a, n, s);
    }
    return hipErrorInvalidValue;
}

template <int M, class R, int FULL>
hipError_t launch_coop_one(const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    const dim3 grid((trajectories + 7) / 8), block(IVP_WAVE);   // eight lanes per trajectory
    if (grid.x == 0) return hipSuccess;
    (void)hipGetLastError();   // drop a stale error of some earlier runtime call: the value returned below is this launch's
    hipLaunchKernelGGL((IVP_NS::coop_chunk_kernel<M, R, FULL>), grid, block, 0, s, a);
    return hipGetLastError();
}
template <int M, class R>
hipError_t launch_coop_flavour(int full, const IvpKArgs &a, uint32_t n, hipStream_t s)
{
    if constexpr (R::NE > 0) {   // a problem with event functions always runs its FULL kernels
        if (!full) return hipErrorInvalidValue;
        return launch_coop_one<M, R, 1>(a, n, s);
    } else {
        if (full == 2) return launch_coop_one<M, R, 2>(a, n, s);   // log-only flavour (rk_core.h so_log_accepted)
        if constexpr (M == IVP_NS::M_DOP853) {
            if (full == 3) return launch_coop_one<M, R, 3>(a, n, s);   // deferred t_eval sampling (rk_core.h so_defer_samples)
        }
        return full ? launch_coop_one<M, R, 1>(a, n, s) : launch_coop_one<M, R, 0>(a, n, s);
    }
}
template <class R>
hipError_t launch_coop(int method, int full, const IvpKArgs &a, uint32_t n, hipStream_t s)
{
    using namespace IVP_NS;
    if (method == M_DOPRI5) return launch_coop_flavour<M_DOPRI5, R>(full, a, n, s);
    if (method == M_DOP853) return launch_coop_flavour<M_DOP853, R>(full, a, n, s);
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t IVP_COOP_LAUNCH_NAME(int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    switch (rhs_id) {
    case 0: return launch_coop<IVP_NS::RhsDecay>(method, full, a, trajectories, s);
    case 1: return launch_coop<IVP_NS::RhsSho>(method, full, a, trajectories, s);
    case 2: return launch_coop<IVP_NS::RhsVdp>(method, full, a, trajectories, s);
    case 3: return launch_coop<IVP_NS::RhsCr3bp>(method, full, a, trajectories, s);
    case 4: return launch_coop<IVP_NS::RhsLorenz>(method, full, a, trajectories, s);
    case 5: return launch_coop<IVP_NS::RhsZero>(method, full, a, trajectories, s);
    case 6: return launch_coop<IVP_NS::RhsRational>(method, full, a, trajectories, s);
    case 7: return launch_coop<IVP_NS::RhsExp2>(method, full, a, trajectories, s);
    case 8: return launch_coop<IVP_NS::RhsLinear>(method, full, a, trajectories, s);
    case 9: return launch_coop<IVP_NS::RhsRobertson>(method, full, a, trajectories, s);
    case 10: return launch_coop<IVP_NS::RhsVdpEps>(method, full, a, trajectories, s);
    case 11: return launch_coop<IVP_NS::RhsShoEv>(method, full, a, trajectories, s);
    case 12: return launch_coop<IVP_NS::RhsBall>(method, full, a, trajectories, s);
    case 13: return launch_coop<IVP_NS::RhsCannon>(method, full, a, trajectories, s);
    case 14: return launch_coop<IVP_NS::RhsRationalEv>(method, full, a, trajectories, s);
    case 15: return launch_coop<IVP_NS::RhsRobertsonJac>(method, full, a, trajectories, s);
    }
    return hipErrorInvalidValue;
}

hipError_t IVP_LAUNCH_NAME(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    switch (rhs_id) {
    case 100: return launch_group<IVP_NS::RhsLinearDecay100>(what, method, full, a, trajectories, s);
    case 101: return launch_group<IVP_NS::RhsHeat1D256>(what, method, full, a, trajectories, s);
    case 102: return launch_group<IVP_NS::RhsDense64>(what, method, full, a, trajectories, s);
    }
    return hipErrorInvalidValue;
}
