// rk_group.hip -- wave-per-trajectory DOPRI5 kernels for large state dimensions (see rk_group.h) and their
// launch table.  Compiled twice like rk_kernels.hip: strict (-ffp-contract=off, index-order error-norm sum) and
// fast (-ffp-contract=fast, __shfl_xor butterfly).  Coefficients stay resident in registers (IVP_HOIST): a lone
// wave per trajectory is latency-bound and has VGPRs to spare.
#include <hip/hip_runtime.h>

#define IVP_HD __host__ __device__ __forceinline__
#define IVP_HOIST 1
#if IVP_FAST
#define IVP_NS ivp_group_fast
#define IVP_LAUNCH_NAME ivp_launch_group_fast
#else
#define IVP_NS ivp_group_strict
#define IVP_LAUNCH_NAME ivp_launch_group_strict
#endif
#include "rk_core.h"
#include "rk_group.h"
#include "rk_launch.h"

namespace {

template <class R>
hipError_t launch_group(int what, const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    const dim3 grid(trajectories), block(IVP_WAVE);   // one wavefront per trajectory
    if (grid.x == 0) return hipSuccess;
    if (what == IVP_LAUNCH_INIT) hipLaunchKernelGGL((IVP_NS::group_init_kernel<R>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((IVP_NS::group_chunk_kernel<R>), grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t IVP_LAUNCH_NAME(int what, int rhs_id, const IvpKArgs &a, uint32_t trajectories, hipStream_t s)
{
    switch (rhs_id) {
    case 100: return launch_group<IVP_NS::RhsLinearDecay100>(what, a, trajectories, s);
    case 101: return launch_group<IVP_NS::RhsHeat1D256>(what, a, trajectories, s);
    }
    return hipErrorInvalidValue;
}
