// rk_coop.h -- lane-cooperative DOPRI5 / DOP853 kernels for the latency-bound tail of a batch (device only).
//
// Once the still-running set no longer fills the chip, wall time is the sequential attempt latency of the slowest
// trajectory.  A lone wave issues one instruction per ~5.5 cycles (f64; 4-5 for anything else, s_mov and s_nop
// included -- tools/ubench_issue.hip), whatever its lane count, and a dependent f64 operation returns after ~8.5, so
// an attempt only gets faster with FEWER INSTRUCTIONS in the wave's stream.  Here EIGHT lanes own one trajectory
// (n <= 8): every lane holds one component of y and of every k-stage, so the stage combinations, the error estimate
// and the state update are one component's worth of work per lane instead of n.
//
// Data movement inside a group is DPP (v_mov_b32 with a lane-select modifier: ~12 cycles per hop, two movs per
// double), not ds_bpermute (60-70 cycles per hop through the LDS crossbar):
//   quad_bcast<i>   lane i of every quad (4 lanes) to the whole quad          quad_perm:[i,i,i,i]
//   lo_to_hi / hi_to_lo   lanes 0..3 of a group to lanes 4..7 and back        row_shr:4 / row_shl:4 with a bank mask
// The two quads of a group are used as two "halves" that run the SAME instruction stream on different operands
// wherever the reference has two independent evaluations of one expression shape:
//   * the step controller's two powers err^expo1 and facold^beta (NormOps::pow2): one ivp_pow per lane;
//   * a right-hand side may provide its own cooperative form (RhsCr3bp::ode_coop: the distance to the first primary
//     in the lower half, to the second in the upper half -- one sqrt and one division per lane instead of two).
// A functor may also choose which lane holds which component (coop_lane_of) so that those exchanges are single hops.
// Generic functors (incl. hiprtc user code) gather the whole stage vector and evaluate R::ode redundantly.
//
// The weighted error norm adds the n squared terms in index order (the reference's left-to-right sum) walking
// through the quads, and ends up identical in the eight lanes, so the controller scalars agree across the group and
// accept/reject stays group-uniform.
//
// Like the wave-per-trajectory kernels (rk_group.h) this is not a second integrator: the attempt bodies and the device
// DefaultSolOut of rk_core.h are instantiated with the pseudo right-hand side CoopRhs<R> (one component per lane) that
// overrides ode(), the norm sums, the power pair, the tolerance lookup and the component map.  Every per-component
// expression is therefore the one the thread-per-trajectory kernels evaluate, so in the strict build the results are
// bit-identical and the launch loop may hand a trajectory from one kernel kind to the other at any launch boundary
// (tested).  State layout in HBM is unchanged.  Scalar per-trajectory state is held and written redundantly by the
// group's lanes.
#pragma once

namespace IVP_NS {

// ---- DPP moves of doubles -------------------------------------------------------------------------------------
// CTRL: dpp_ctrl (quad_perm 0x00..0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n); BANK: which quads of a 16-lane row are
// written (bit q = lanes 4q..4q+3); unwritten lanes keep `old`.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    // a full-width quad_perm writes every lane from a valid source: bound_ctrl tells the compiler that `old` is dead
    constexpr bool kAll = BANK == 0xF && CTRL < 0x100;
    const uint64_t o = d2u(old), s = d2u(src);
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)s, CTRL, 0xF, BANK, kAll);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(s >> 32), CTRL, 0xF, BANK, kAll);
    return u2d(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
template <int I>
__device__ __forceinline__ double quad_bcast(double v) { return dpp_f64<I * 0x55, 0xF>(0.0, v); }
// lanes 4..7 of every group take `src` of the lane four below; lanes 0..3 keep `old`
__device__ __forceinline__ double lo_to_hi(double old, double src) { return dpp_f64<0x114, 0xA>(old, src); }
// lanes 0..3 of every group take `src` of the lane four above; lanes 4..7 keep `old`
__device__ __forceinline__ double hi_to_lo(double old, double src) { return dpp_f64<0x104, 0x5>(old, src); }
// lane I (0..7) of every group to its eight lanes
template <int I>
__device__ __forceinline__ double grp_bcast(double v)
{
    const double t = quad_bcast<(I & 3)>(v);
    if constexpr (I < 4) return lo_to_hi(t, t);
    else return hi_to_lo(t, t);
}

// ---- which lane of the group holds which component --------------------------------------------------------------
template <class R, class = void>
struct CoopLayout { static constexpr int lane_of(int c) { return c; } };
template <class R>
struct CoopLayout<R, decltype((void)R::coop_lane_of(0))> { static constexpr int lane_of(int c) { return R::coop_lane_of(c); } };
// component held by this lane; >= R::N for a lane that holds none
template <class R>
__device__ __forceinline__ uint32_t coop_comp()
{
    const uint32_t l = threadIdx.x & 7u;
    uint32_t c = 8u;
#pragma unroll
    for (int i = 0; i < R::N; ++i) c = (l == (uint32_t)CoopLayout<R>::lane_of(i)) ? (uint32_t)i : c;
    return c;
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
template <class R, int I>
__device__ __forceinline__ void coop_gather_from(double v, double (&out)[R::N])
{
    if constexpr (I < R::N) {
        out[I] = grp_bcast<CoopLayout<R>::lane_of(I)>(v);
        coop_gather_from<R, I + 1>(v, out);
    }
}
// the whole vector, in component order, in every lane of the group
template <class R>
__device__ __forceinline__ void coop_gather(double v, double (&out)[R::N]) { coop_gather_from<R, 0>(v, out); }

// sum_{i<N} term_i in index order (0.0 + t0 + t1 + ...); term_i lives in lane lane_of(i).  The running sum walks
// through the quads in component order (valid in the quad that holds component i) and is copied to the other quad at
// the end: 2 hops per term + 2 per quad change instead of 4 per term.
template <class R, int I>
__device__ __forceinline__ double coop_sum_from(double s, double t)
{
    using LAY = CoopLayout<R>;
    if constexpr (I == R::N) {
        constexpr int q = LAY::lane_of(R::N - 1) >> 2;
        if constexpr (q == 1) return hi_to_lo(s, s);
        else return lo_to_hi(s, s);
    } else {
        constexpr int li = LAY::lane_of(I), qi = li >> 2;
        if constexpr (I > 0) {
            constexpr int qp = LAY::lane_of(I - 1) >> 2;
            if constexpr (qp != qi) s = (qi == 1) ? lo_to_hi(s, s) : hi_to_lo(s, s);
        }
        s += quad_bcast<(li & 3)>(t);
        return coop_sum_from<R, I + 1>(s, t);
    }
}
template <class R>
__device__ __forceinline__ double coop_sum(double term) { return coop_sum_from<R, 0>(0.0, term); }

// RhsCr3bp::ode_coop (declared and explained in rk_core.h next to the reference form)
__device__ __forceinline__ double RhsCr3bp::ode_coop(double, double ys, const double *p)
{
    const uint32_t l = threadIdx.x & 7u;
    const bool hi = (l & 4u) != 0;
    const uint32_t i = l & 3u;                  // numerator role inside the quad: 0 -> a|b, 1 -> y, 2 -> z, 3 idle
    const double mu = p[0];
    const double pos = lo_to_hi(ys, ys);        // position component i, in both quads
    const double X = quad_bcast<0>(pos), Y = quad_bcast<1>(pos), Z = quad_bcast<2>(pos);
    const double sw = dpp_f64<0xE1, 0xF>(0.0, ys);   // quad_perm:[1,0,2,3]: vy for the vx' lane, vx for the vy' lane
    const double w = (X + (hi ? -1.0 : -0.0)) + mu; // lower quad: a = x + mu; upper quad: b = x - 1.0 + mu
#if IVP_FAST
    // FMA form (RhsCr3bp::ode under IVP_FAST, expression for expression): g = coefficient / r^3 per primary, the upper quad
    // fetches the first primary's g from the lane four below and fuses both attractions into its linear part
    const double d = fma(Z, Z, fma(Y, Y, w * w));
    const double r = sqrt(d);
    const double g = (hi ? mu : 1.0 - mu) / (d * r);
    const double g1 = lo_to_hi(g, g);
    const double q2 = i == 0u ? w : pos;
    const double q1 = i == 0u ? X + mu : pos;
    double lin = fma(i == 0u ? 2.0 : -2.0, sw, pos);
    lin = i == 2u ? -0.0 : lin;
    const double acc = fma(-g, q2, fma(-g1, q1, lin));
#else
    const double r = sqrt(w * w + Y * Y + Z * Z);
    const double r3 = r * r * r;
    const double q = i == 0u ? w : pos;
    const double T = (hi ? mu : 1.0 - mu) * q / r3;
    const double T1 = lo_to_hi(T, T);           // upper quad: the first primary's term from the lane four below
    double lin = pos + (i == 0u ? 2.0 : -2.0) * sw;
    lin = i == 2u ? -0.0 : lin;
    const double acc = (lin - T1) - T;
#endif
    return hi_to_lo(acc, ys);                   // position lanes: d(pos)/dt = the velocity four lanes above
}

template <class R, class = void>
struct HasCoopOde { enum { v = 0 }; };
template <class R>
struct HasCoopOde<R, decltype((void)&R::ode_coop)> { enum { v = 1 }; };
template <class R>
__device__ __forceinline__ double coop_ode(double t, double ystage, const double *p)
{
    if constexpr (HasCoopOde<R>::v) {
        return R::ode_coop(t, ystage, p);   // the functor's own lane-cooperative form
    } else {
        double yf[R::N], df[R::N];
        coop_gather<R>(ystage, yf);
        R::ode(t, yf, df, p);
        return coop_select<R::N>(df, coop_comp<R>());
    }
}

template <class R>
struct CoopRhs {
    enum { NT = R::N, N = 1, P = R::P, NE = R::NE };
    static_assert(R::N <= 8, "eight lanes per trajectory");
    static __device__ __forceinline__ void ode(double t, const double *ys, double *k, const double *p) { k[0] = coop_ode<R>(t, ys[0], p); }
    // event functions see the whole state: gathered, evaluated by every lane of the group (same values in all of
    // them, so the root finder of so_events stays group-uniform)
    static __device__ __forceinline__ void events(double x, const double *ys, double *g, const double *p)
    {
        if constexpr (NE > 0) {
            double yf[R::N];
            coop_gather<R>(ys[0], yf);
            R::events(x, yf, g, p);
        }
    }
};
template <class R>
struct OutMap<CoopRhs<R>, void> {
    struct type {
        enum { NT = R::N };
        static __device__ __forceinline__ int gi(int) { return (int)coop_comp<R>(); }
        static __device__ __forceinline__ bool own(int) { return coop_comp<R>() < (uint32_t)NT; }
        // the group's first lane acts for the trajectory (one-pass step log: page allocation), its seven partners read its result
        static __device__ __forceinline__ bool leader() { return (threadIdx.x & 7u) == 0u; }
        static __device__ __forceinline__ uint32_t bcast(uint32_t v) { return (uint32_t)__shfl((int)v, (int)(threadIdx.x & ~7u)); }
    };
};
template <class R>
struct NormOps<CoopRhs<R>, void> {
    enum { NT = R::N };
    // this lane's rtol / atol (Tolerance, mod.rs:104-214): select chain instead of a dynamically indexed kernarg array
    static __device__ __forceinline__ double pick(const double (&arr)[IVP_MAX_N])
    {
        const uint32_t c = coop_comp<R>();
        double v = arr[0];
#pragma unroll
        for (int q = 1; q < NT; ++q) v = (c == (uint32_t)q) ? arr[q] : v;
        return v;
    }
    static __device__ __forceinline__ double rtol(const IvpKArgs &a, int) { return pick(a.rtol); }
    static __device__ __forceinline__ double atol(const IvpKArgs &a, int) { return pick(a.atol); }
    static __device__ __forceinline__ double sum(const double (&term)[1]) { return coop_sum<R>(term[0]); }
    // the controller's two powers in one instruction stream: x1^e1 in the lower half of the group, x2^e2 in the upper
    static __device__ __forceinline__ void pow2(double x1, double e1, double x2, double e2, double &r1, double &r2, uint64_t kz)
    {
        const bool hi = (threadIdx.x & 4u) != 0;
        const double r = ivp_pow(hi ? x2 : x1, hi ? e2 : e1, kz);
        r1 = lo_to_hi(r, r);
        r2 = hi_to_lo(r, r);
    }
};

// up to a.chunk step attempts for the 8 trajectories of this wave; controller fields from IvpKArgs (CTL = true)
template <int M, class R, int FULL>
__device__ __forceinline__ void coop_chunk_body(const IvpKArgs &a)
{
    const uint32_t lane = threadIdx.x, c = lane & 7u;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (a.count_next && blockIdx.x == 0 && threadIdx.x == 0) *a.count_next = 0u;   // before any early exit: both halves of a pair do it
    if (a.spec_cap && count > a.spec_cap) return;   // speculative launch declined: the set is still too large (uniform)
    if (blockIdx.x * 8u >= count) return;   // whole wave beyond the active set (stale grid bound)
    if (a.ran_out && blockIdx.x == 0 && threadIdx.x == 0) *a.ran_out = 1u;
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

// The FULL kernels (device DefaultSolOut on top of the integrator) may use the whole register file -- at most one wave per
// SIMD -- instead of leaving a spill area in scratch; the end-state kernels keep the compiler's choice (DOPRI5 / CR3BP: 206
// VGPRs, two waves per SIMD, which is what the 1588 waves arriving at C2's hand-over need).
template <int M, class R, int FULL>
__global__ __launch_bounds__(IVP_WAVE) __attribute__((amdgpu_waves_per_eu(1, FULL == 1 ? 1 : 8)))
void coop_chunk_kernel(const IvpKArgs a) { coop_chunk_body<M, R, FULL>(a); }

}  // namespace IVP_NS
