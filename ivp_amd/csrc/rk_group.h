// rk_group.h -- wave-per-trajectory ("thread-group owns one IVP") kernels for state dimensions that do not fit one
// lane's registers (8 < n <= 512), e.g. the reference benchmark's "Large Linear System (N=100)"
// (benches/benchmark.py:139-148) or method-of-lines PDE systems.  Device only.
//
// Mapping.  One 64-lane wavefront integrates ONE trajectory; lane l owns components l, l+64, l+128, ...
// (C = ceil(n/64) per lane), so y, the k-stages and the dense-output coefficients are C-element VGPR arrays per lane.
// The integrators themselves are the very same bodies as in the thread-per-trajectory kernels (rk_core.h:
// init_body / dopri5_attempt / dop853_attempt / rk23_attempt / rk4_attempt, hinit, the device DefaultSolOut): they
// are instantiated with the pseudo right-hand side GroupRhs<R> whose "state" is this lane's C components and which
// supplies the three things that differ:
//   * ode():   component i of f may read any y_j, so the stage state is published through LDS (n doubles per
//              wave; the barrier is wave-local) and each lane evaluates its components with the component-form
//              right-hand side R::ode_comp(i, t, y_lds, p);
//   * NormOps: the weighted RMS norms (error estimate, hinit, stiffness quotient) are the one cross-lane operation.
//              STRICT build: the n terms go to LDS and every lane adds them in index order -- the reference's
//              left-to-right sum (dopri5.rs:343-347) bit for bit; FAST build: lane partials + a 6-step __shfl_xor
//              butterfly over the wavefront.  Either way every lane gets the same value, so the step-size
//              controller runs redundantly in all lanes and accept/reject is a wave-uniform branch;
//   * OutMap:  local component c lives at global index lane + 64 c of every SoA array.
// Scalar per-trajectory state (x, h, counters, ...) is held and written redundantly by all 64 lanes (same value,
// same address).  Per-component tolerance vectors live in device memory (IvpKArgs.rtol_dev / atol_dev).
// Event functions (hiprtc user code) are evaluated on the LDS copy of the state by every lane.
//
// Group width.  G = 64 (one wavefront per trajectory) for the built-in problems; hiprtc modules of smaller systems use
// G = 16 (n <= 16) or G = 32 (n <= 32), i.e. 4 or 2 trajectories per wavefront, each with its own LDS region.  The
// groups of a wavefront then diverge like the lanes of the thread-per-trajectory kernels do (predication); a barrier
// inside a divergent branch is harmless here because the workgroup is a single wavefront.
#pragma once

namespace IVP_NS {

// Large-n right-hand sides (component form).
struct RhsLinearDecay100 {   // benches/benchmark.py:40-42,139-148: y' = -y, N = 100
    enum { N = 100, P = 0 };
    static __device__ __forceinline__ double ode_comp(int i, double, const double *y, const double *) { return -y[i]; }
};
struct RhsHeat1D256 {        // method-of-lines heat equation, Dirichlet ends: y_i' = kappa (y_{i-1} - 2 y_i + y_{i+1})
    enum { N = 256, P = 1 };
    static __device__ __forceinline__ double ode_comp(int i, double, const double *y, const double *p)
    {
        const double left = i > 0 ? y[i - 1] : 0.0;
        const double right = i < N - 1 ? y[i + 1] : 0.0;
        return p[0] * (IVP_MS(left, 2.0, y[i]) + right);
    }
};

struct RhsDense64 {          // y' = A y with a dense, diagonally dominant 64 x 64 matrix (a full Jacobian for BDF's LU)
    enum { N = 64, P = 1 };
    static __device__ __forceinline__ double ode_comp(int i, double, const double *y, const double *p)
    {
        double s = -p[0] * (4.0 + (double)(i % 5)) * y[i];
#pragma unroll 8
        for (int j = 0; j < N; ++j) {
            const double aij = (double)(((i * 5 + j * 3) & 15) - 8) * 0.00390625;   // in [-8, 7] / 256, exact
            if (j != i) s = IVP_MA(s, aij, y[j]);
        }
        return s;
    }
};

// number of event functions of a component-form functor (hiprtc user code defines NE; the built-ins have none)
template <class R, class = void>
struct GroupNE { enum { v = 0 }; };
template <class R>
struct GroupNE<R, decltype((void)R::NE)> { enum { v = R::NE }; };

template <class R, int G = IVP_WAVE>
struct GroupRhs {
    enum { NT = R::N, N = (R::N + G - 1) / G, P = R::P, NE = GroupNE<R>::v, GW = G, NGROUP = IVP_WAVE / G };
    static_assert(G == 16 || G == 32 || G == 64, "group width");
    static __device__ __forceinline__ int gl() { return (int)threadIdx.x & (G - 1); }   // lane within the group
    static __device__ __forceinline__ int gb() { return ((int)threadIdx.x / G) * NT; }   // this group's LDS region
    // ONE n-vector of LDS per group, shared by the three places that publish a vector to the whole group -- the stage
    // state in ode(), the norm terms in NormOps::sum() (strict build), the right-hand side in BdfG::lin_solve().  Each
    // user starts with a barrier, writes, and has read everything back before it returns, so the uses never overlap;
    // one buffer instead of three is what lets two workgroups with an LDS-resident 100 x 100 matrix share a CU.
    static __device__ __forceinline__ double *scratch()
    {
        __shared__ double ivp_group_vec[NGROUP * NT];
        return ivp_group_vec + gb();
    }
    static __device__ __forceinline__ void ode(double t, const double *ys, double *k, const double *p)
    {
        double *st = scratch();
        __syncthreads();   // earlier readers of `stage` are done
#pragma unroll
        for (int c = 0; c < N; ++c) { const int i = gl() + G * c; if (i < NT) st[i] = ys[c]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < N; ++c) { const int i = gl() + G * c; k[c] = i < NT ? R::ode_comp(i, t, st, p) : 0.0; }
    }
    // event functions see the whole state: publish it through LDS, every lane evaluates them (same values in all lanes
    // of the group, so the root finder of so_events stays group-uniform)
    static __device__ __forceinline__ void events(double x, const double *ys, double *g, const double *p)
    {
        if constexpr (NE > 0) {
            __shared__ double estage[NGROUP * NT];
            double *st = estage + gb();
            __syncthreads();
#pragma unroll
            for (int c = 0; c < N; ++c) { const int i = gl() + G * c; if (i < NT) st[i] = ys[c]; }
            __syncthreads();
            R::events(x, st, g, p);
        }
    }
};

template <class R, int G>
struct OutMap<GroupRhs<R, G>, void> {
    struct type {
        enum { NT = R::N };
        static __device__ __forceinline__ int gi(int c) { return GroupRhs<R, G>::gl() + G * c; }
        static __device__ __forceinline__ bool own(int c) { return gi(c) < NT; }
        static __device__ __forceinline__ bool leader() { return GroupRhs<R, G>::gl() == 0; }
        static __device__ __forceinline__ uint32_t bcast(uint32_t v) { return (uint32_t)__shfl((int)v, (int)(threadIdx.x & ~(uint32_t)(G - 1))); }
    };
};

template <class R, int G>
struct NormOps<GroupRhs<R, G>, void> {
    enum { NT = R::N };
    // scalar tolerances in the kernel arguments, or per-component vectors [n] in device memory (Tolerance::Vector)
    static __device__ __forceinline__ double rtol(const IvpKArgs &a, int c)
    {
        const int i = GroupRhs<R, G>::gl() + G * c;
        return (a.rtol_dev && i < NT) ? a.rtol_dev[i] : a.rtol[0];
    }
    static __device__ __forceinline__ double atol(const IvpKArgs &a, int c)
    {
        const int i = GroupRhs<R, G>::gl() + G * c;
        return (a.atol_dev && i < NT) ? a.atol_dev[i] : a.atol[0];
    }
    static __device__ __forceinline__ void pow2(double x1, double e1, double x2, double e2, double &r1, double &r2, uint64_t kz)
    {
        r1 = ivp_pow(x1, e1, kz);
        r2 = ivp_pow(x2, e2, kz);
    }
    template <int C>
    static __device__ __forceinline__ double sum(const double (&term)[C])
    {
        const int gl = GroupRhs<R, G>::gl();
#if IVP_FAST
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) if (gl + G * c < NT) part += term[c];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);   // stays inside the group: o < G
        return part;
#else
        double *rd = GroupRhs<R, G>::scratch();
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) { const int i = gl + G * c; if (i < NT) rd[i] = term[c]; }
        __syncthreads();
        double s = 0.0;
#pragma unroll 8
        for (int i = 0; i < NT; ++i) s += rd[i];
        return s;
#endif
    }
};

// BDF for large n (bdf_group.h, included after this header)
template <class R, int FULL, int G>
__device__ __forceinline__ int32_t bdf_group_init_body(const IvpKArgs &a, uint32_t j);
template <class R, int FULL, int G, bool LDSLU>
__device__ __forceinline__ uint32_t bdf_group_chunk_body(const IvpKArgs &a, uint32_t j, int32_t &status_out);

// init: one group of G lanes per trajectory, 64 / G trajectories per wavefront
template <int M, class R, int FULL, int G = IVP_WAVE>
__device__ __forceinline__ void group_init_body(const IvpKArgs &a)
{
    const uint32_t j = blockIdx.x * (IVP_WAVE / G) + threadIdx.x / G;
    if (j >= a.B) return;
    if constexpr (M == M_BDF) (void)bdf_group_init_body<R, FULL, G>(a, j);
    else (void)init_body<M, GroupRhs<R, G>, FULL>(a, j);
}

// up to a.chunk step attempts for the trajectories of this wave; controller fields from IvpKArgs (CTL = true).
// LDSLU (BDF only): the factors of (I - cJ) live in LDS for the whole launch (bdf_group.h).
template <int M, class R, int FULL, int G = IVP_WAVE, bool LDSLU = false>
__device__ __forceinline__ void group_chunk_body(const IvpKArgs &a)
{
    constexpr uint32_t NG = IVP_WAVE / G;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (a.count_next && blockIdx.x == 0 && threadIdx.x == 0) *a.count_next = 0u;   // before any early exit
    if (blockIdx.x * NG >= count) return;   // whole wave beyond the active set (stale grid bound)
    const uint32_t i = blockIdx.x * NG + threadIdx.x / G;
    uint32_t j = 0;
    bool active = false;
    if (i < count) {
        j = a.perm_in ? a.perm_in[i] : i;
        active = a.status[j] == IVP_RUNNING;
    }
    int32_t st = 0;
    uint32_t it = 0;
    constexpr bool kCtl = M == M_RK23 || M == M_DOPRI5 || M == M_DOP853;
    if (active) {
        if constexpr (M == M_BDF) it = bdf_group_chunk_body<R, FULL, G, LDSLU>(a, j, st);
        else it = chunk_body<M, GroupRhs<R, G>, FULL, kCtl>(a, j, st);
    }
    const bool lead = (threadIdx.x & (G - 1)) == 0;
    compact_append(a, j, active && lead && st == IVP_RUNNING);   // still running: next launch's list
    if (a.slot_counter) {
        uint32_t mx = it;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        const unsigned long long act = __ballot(active && lead);
        if (threadIdx.x == 0) {
            atomicAdd(a.slot_counter, (unsigned long long)mx * IVP_WAVE);
            atomicAdd(a.slot_counter + 1, (unsigned long long)__popcll(act));
        }
    }
}

template <int M, class R, int FULL>
__global__ __launch_bounds__(IVP_WAVE) void group_init_kernel(const IvpKArgs a) { group_init_body<M, R, FULL>(a); }
template <int M, class R, int FULL, bool LDSLU = false>
__global__ __launch_bounds__(IVP_WAVE) void group_chunk_kernel(const IvpKArgs a) { group_chunk_body<M, R, FULL, IVP_WAVE, LDSLU>(a); }

}  // namespace IVP_NS
