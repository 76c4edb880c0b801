// rk_bdf.hip -- thread-per-trajectory BDF(1..5) kernels (bdf_core.h, n <= 8) and their launch table.
//
// BDF batches are small and long (BASELINE C5: 10 000 trajectories x ~1700 sequential attempts): the waves own their
// SIMDs, so wall time is instructions per attempt x ~4.6 cycles.  These kernels are therefore built with every
// coefficient PINNED in a vector register (IVP_HOIST = 2, see KC() in rk_core.h) -- a lone wave pays a full issue slot
// for every s_mov / v_mov that re-creates a constant, SGPR pressure from scalar constants turns into v_readlane /
// v_writelane spill traffic, and with __launch_bounds__(64, 1) there are 512 registers to spend.  One build per
// floating-point mode; the variant does not depend on the batch, so neither do fast-mode results.
#include <hip/hip_runtime.h>

#define IVP_HD __host__ __device__ __forceinline__
#ifndef IVP_HOIST
#define IVP_HOIST 2
#endif
// Second build (-DIVP_BDF_MIN_WAVES=2 -> *_occ2): the same source under __launch_bounds__(64, 2).  It spills a few
// registers to scratch (84 B per lane at n = 2) and is ~10 % slower per attempt for a lone wave, but two resident waves
// per SIMD hide each other's latencies once a batch over-subscribes the chip (262 144 stiff Van der Pol trajectories:
// 24.7 -> 21.2 ms); the launch loop picks it for batches of more than one full wave per SIMD.  Same arithmetic, same bits.
#ifndef IVP_BDF_MIN_WAVES
#define IVP_BDF_MIN_WAVES 1
#endif
#if IVP_FAST && IVP_BDF_MIN_WAVES > 1
#define IVP_NS ivp_bdf_fast_occ2
#define IVP_LAUNCH_NAME ivp_launch_bdf_fast_occ2
#elif IVP_FAST
#define IVP_NS ivp_bdf_fast
#define IVP_LAUNCH_NAME ivp_launch_bdf_fast
#elif IVP_BDF_MIN_WAVES > 1
#define IVP_NS ivp_bdf_strict_occ2
#define IVP_LAUNCH_NAME ivp_launch_bdf_strict_occ2
#else
#define IVP_NS ivp_bdf_strict
#define IVP_LAUNCH_NAME ivp_launch_bdf_strict
#endif
#include "rk_core.h"
#ifdef IVP_PHASE_PROF
__device__ unsigned long long ivp_phase_ticks[16];
#endif
#include "bdf_core.h"
#include "rk_global.h"
#include "rk_launch.h"

namespace {

using namespace IVP_NS;

template <class R, int FULL>
hipError_t launch_one(int what, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    const uint32_t per_wave = (what == IVP_LAUNCH_CHUNK && a.lpw) ? a.lpw : (uint32_t)IVP_WAVE;   // thin waves (ivp_kargs.h)
    const dim3 grid((lanes + per_wave - 1) / per_wave), block(IVP_WAVE);
    if (grid.x == 0) return hipSuccess;
    (void)hipGetLastError();   // drop a stale error of some earlier runtime call: the value returned below is this launch's
    if (a.has_ctl) return hipErrorInvalidValue;   // BDF has no per-method controller struct on this path
    if (what == IVP_LAUNCH_INIT) hipLaunchKernelGGL((init_kernel_t<M_BDF, R, FULL>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((chunk_kernel_t<M_BDF, R, FULL>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <class R>
hipError_t launch_rhs(int what, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    return full ? launch_one<R, true>(what, a, lanes, s) : launch_one<R, false>(what, a, lanes, s);
}

}  // namespace

hipError_t IVP_LAUNCH_NAME(int what, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s)
{
    switch (rhs_id) {
    case 0: return launch_rhs<IVP_NS::RhsDecay>(what, full, a, lanes, s);
    case 1: return launch_rhs<IVP_NS::RhsSho>(what, full, a, lanes, s);
    case 2: return launch_rhs<IVP_NS::RhsVdp>(what, full, a, lanes, s);
    case 3: return launch_rhs<IVP_NS::RhsCr3bp>(what, full, a, lanes, s);
    case 4: return launch_rhs<IVP_NS::RhsLorenz>(what, full, a, lanes, s);
    case 5: return launch_rhs<IVP_NS::RhsZero>(what, full, a, lanes, s);
    case 6: return launch_rhs<IVP_NS::RhsRational>(what, full, a, lanes, s);
    case 7: return launch_rhs<IVP_NS::RhsExp2>(what, full, a, lanes, s);
    case 8: return launch_rhs<IVP_NS::RhsLinear>(what, full, a, lanes, s);
    case 9: return launch_rhs<IVP_NS::RhsRobertson>(what, full, a, lanes, s);
    case 10: return launch_rhs<IVP_NS::RhsVdpEps>(what, full, a, lanes, s);
    case 11: return launch_rhs<IVP_NS::RhsShoEv>(what, true, a, lanes, s);      // problems with events always run FULL
    case 12: return launch_rhs<IVP_NS::RhsBall>(what, true, a, lanes, s);
    case 13: return launch_rhs<IVP_NS::RhsCannon>(what, true, a, lanes, s);
    case 14: return launch_rhs<IVP_NS::RhsRationalEv>(what, true, a, lanes, s);
    case 15: return launch_rhs<IVP_NS::RhsRobertsonJac>(what, full, a, lanes, s);
    }
    return hipErrorInvalidValue;
}

#ifdef IVP_PHASE_PROF
// tools/bdf_phase_profile.sh: ticks per phase marker summed over all waves since the last reset
extern "C" int ivp_debug_phase_ticks(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ivp_phase_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ivp_phase_ticks), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
