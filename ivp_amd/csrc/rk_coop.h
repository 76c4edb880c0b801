// rk_coop.h -- lane-cooperative DOPRI5 / DOP853 kernels for the latency-bound tail of a batch (device only).
//
// Once the still-running set no longer fills the chip, wall time is the sequential attempt latency of the slowest
// trajectory: a lone wave issues one f64 instruction per ~4 cycles whatever its lane count, so the only way to make
// an attempt faster is to give it fewer instructions.  Here EIGHT lanes own one trajectory (n <= 8): lane c holds
// component c of y and of every k-stage, so the stage combinations, the error estimate and the state update are one
// component's worth of work per lane instead of n.  The right-hand side needs the whole stage vector: it is gathered
// with cross-lane shuffles (n x ds_bpermute pairs) and every lane evaluates the full functor R::ode redundantly,
// keeping its own component -- redundant work on lanes that would otherwise idle.  The weighted error norm gathers
// the n squared terms and adds them in index order in every lane (the reference's left-to-right sum), so the
// controller scalars are identical in the eight lanes and accept/reject is group-uniform.
//
// Like the wave-per-trajectory kernels (rk_group.h) this is not a second integrator: the attempt bodies and the device
// DefaultSolOut of rk_core.h are instantiated with the pseudo right-hand side CoopRhs<R> (one component per lane) that
// overrides ode(), the norm sums, the tolerance lookup and the component map.  Every per-component expression is
// therefore the one the thread-per-trajectory kernels evaluate, so in the strict build the results are bit-identical
// and the launch loop may hand a trajectory from one kernel kind to the other at any launch boundary (tested).  State
// layout in HBM is unchanged.  Scalar per-trajectory state is held and written redundantly by the group's lanes.
#pragma once

namespace IVP_NS {

template <int N>
__device__ __forceinline__ void coop_gather(double v, uint32_t base, double (&out)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __shfl(v, (int)(base + i));
}
template <int N>
__device__ __forceinline__ double coop_select(const double (&v)[N], uint32_t c)
{
    // the opaque copies keep LLVM from folding the select chain into a dynamically indexed load of a scratch array
    double s = v[0];
    IVP_OPAQUE_V(s);
#pragma unroll
    for (int i = 1; i < N; ++i) {
        double t = v[i];
        IVP_OPAQUE_V(t);
        s = (c == (uint32_t)i) ? t : s;
    }
    return s;
}
template <class R, class = void>
struct HasCoopOde { enum { v = 0 }; };
template <class R>
struct HasCoopOde<R, decltype((void)&R::ode_coop)> { enum { v = 1 }; };
// sum_{i<N} term_i in index order; term lives in lane base+i
template <int N>
__device__ __forceinline__ double coop_sum(double term, uint32_t base)
{
    double t[N];
    coop_gather<N>(term, base, t);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += t[i];
    return s;
}
template <class R>
__device__ __forceinline__ double coop_ode(double t, double ystage, uint32_t base, uint32_t c, const double *p)
{
    if constexpr (HasCoopOde<R>::v) {
        return R::ode_coop(t, ystage, base, c, p);   // the functor's own lane-cooperative form
    } else {
        double yf[R::N], df[R::N];
        coop_gather<R::N>(ystage, base, yf);
        R::ode(t, yf, df, p);
        return coop_select<R::N>(df, c);
    }
}

template <class R>
struct CoopRhs {
    enum { NT = R::N, N = 1, P = R::P, NE = R::NE };
    static_assert(R::N <= 8, "eight lanes per trajectory");
    static __device__ __forceinline__ void ode(double t, const double *ys, double *k, const double *p)
    {
        const uint32_t lane = threadIdx.x;
        k[0] = coop_ode<R>(t, ys[0], lane & ~7u, lane & 7u, p);
    }
    // event functions see the whole state: gathered by shuffles, evaluated by every lane of the group (same values in
    // all of them, so the root finder of so_events stays group-uniform)
    static __device__ __forceinline__ void events(double x, const double *ys, double *g, const double *p)
    {
        if constexpr (NE > 0) {
            double yf[R::N];
            coop_gather<R::N>(ys[0], threadIdx.x & ~7u, yf);
            R::events(x, yf, g, p);
        }
    }
};
template <class R>
struct OutMap<CoopRhs<R>, void> {
    struct type {
        enum { NT = R::N };
        static __device__ __forceinline__ int gi(int) { return (int)(threadIdx.x & 7u); }
        static __device__ __forceinline__ bool own(int) { return (threadIdx.x & 7u) < (uint32_t)NT; }
    };
};
template <class R>
struct NormOps<CoopRhs<R>, void> {
    enum { NT = R::N };
    // this lane's rtol / atol (Tolerance, mod.rs:104-214): select chain instead of a dynamically indexed kernarg array
    static __device__ __forceinline__ double pick(const double (&arr)[IVP_MAX_N])
    {
        const uint32_t c = threadIdx.x & 7u;
        double v = arr[0];
#pragma unroll
        for (int q = 1; q < NT; ++q) v = (c == (uint32_t)q) ? arr[q] : v;
        return v;
    }
    static __device__ __forceinline__ double rtol(const IvpKArgs &a, int) { return pick(a.rtol); }
    static __device__ __forceinline__ double atol(const IvpKArgs &a, int) { return pick(a.atol); }
    static __device__ __forceinline__ double sum(const double (&term)[1]) { return coop_sum<NT>(term[0], threadIdx.x & ~7u); }
};

// up to a.chunk step attempts for the 8 trajectories of this wave; controller fields from IvpKArgs (CTL = true)
template <int M, class R, bool FULL>
__device__ __forceinline__ void coop_chunk_body(const IvpKArgs &a)
{
    const uint32_t lane = threadIdx.x, c = lane & 7u;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (a.spec_cap && count > a.spec_cap) return;   // speculative launch declined: the set is still too large (uniform)
    if (blockIdx.x * 8u >= count) return;   // whole wave beyond the active set (stale grid bound)
    const uint32_t i = blockIdx.x * 8u + (lane >> 3);
    const bool valid = i < count;
    uint32_t j = 0;
    bool active = false;
    if (valid) {
        j = a.perm_in ? a.perm_in[i] : i;
        active = a.status[j] == IVP_RUNNING;
    }
    uint32_t it = 0;
    int32_t st = 0;
    if (active) it = chunk_body<M, CoopRhs<R>, FULL, true>(a, j, st);   // the group's 8 lanes stay in lock-step
    compact_append(a, j, active && c == 0 && st == IVP_RUNNING);
    if (a.slot_counter) {
        uint32_t mx = it;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        const unsigned long long act = __ballot(active && c == 0);
        if (lane == 0) {
            atomicAdd(a.slot_counter, (unsigned long long)mx * IVP_WAVE);
            atomicAdd(a.slot_counter + 1, (unsigned long long)__popcll(act));
        }
    }
}

template <int M, class R, bool FULL>
__global__ __launch_bounds__(IVP_WAVE) void coop_chunk_kernel(const IvpKArgs a) { coop_chunk_body<M, R, FULL>(a); }

}  // namespace IVP_NS
