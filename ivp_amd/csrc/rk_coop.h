// rk_coop.h -- lane-cooperative DOPRI5 kernel for the latency-bound tail of a batch (device only).
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
// Every per-component expression is the one in dopri5_attempt (rk_core.h), so in the strict build the results are
// bit-identical to the thread-per-trajectory kernels and the launch loop may switch between the two at any launch
// boundary (tested).  State layout in HBM is unchanged.  End-state (FULL = false) runs only.
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

// up to a.chunk DOPRI5 attempts (dopri5.rs:266-461) for the 8 trajectories of this wave
template <class R>
__device__ __forceinline__ void coop_chunk_body(const IvpKArgs &a)
{
    constexpr int N = R::N, P = R::P;
    static_assert(N <= 8, "eight lanes per trajectory");
    constexpr double C2 = 0.2, C3 = 0.3, C4 = 0.8, C5 = 8.0 / 9.0;
    constexpr double A21 = 0.2, A31 = 3.0 / 40.0, A32 = 9.0 / 40.0;
    constexpr double A41 = 44.0 / 45.0, A42 = -56.0 / 15.0, A43 = 32.0 / 9.0;
    constexpr double A51 = 19372.0 / 6561.0, A52 = -25360.0 / 2187.0, A53 = 64448.0 / 6561.0, A54 = -212.0 / 729.0;
    constexpr double A61 = 9017.0 / 3168.0, A62 = -355.0 / 33.0, A63 = 46732.0 / 5247.0, A64 = 49.0 / 176.0, A65 = -5103.0 / 18656.0;
    constexpr double A71 = 35.0 / 384.0, A73 = 500.0 / 1113.0, A74 = 125.0 / 192.0, A75 = -2187.0 / 6784.0, A76 = 11.0 / 84.0;
    constexpr double E1 = 71.0 / 57600.0, E3 = -71.0 / 16695.0, E4 = 71.0 / 1920.0, E5 = -17253.0 / 339200.0, E6 = 22.0 / 525.0, E7 = -1.0 / 40.0;
    const double uround = a.ctl_uround, safety = a.ctl_safety, beta = a.ctl_beta;   // dopri5.rs:34-72 struct fields
    const double facc1 = a.ctl_facc1, facc2 = a.ctl_facc2, expo1 = a.ctl_expo1;

    const uint32_t lane = threadIdx.x, c = lane & 7u, base = lane & ~7u;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (blockIdx.x * 8u >= count) return;   // whole wave beyond the active set (stale grid bound)
    const uint32_t i = blockIdx.x * 8u + (lane >> 3);
    const bool valid = i < count;
    uint32_t j = 0;
    bool active = false;
    if (valid) {
        j = a.perm_in ? a.perm_in[i] : i;
        active = a.status[j] == IVP_RUNNING;
    }
    const bool own = c < (uint32_t)N;
    const size_t B = a.B;
    double y = 0.0, k1 = 0.0, p[P > 0 ? P : 1];
    double x = 0.0, h = 0.0, facold = 1.0, hlamb = 0.0, x0 = 0.0, xend = 0.0;
    uint32_t flags = 0;
    uint64_t nstep0 = 0, nacc0 = 0;
#pragma unroll
    for (int q = 0; q < (P > 0 ? P : 1); ++q) p[q] = 0.0;
    if (active) {
        if (own) { y = a.y[(size_t)c * B + j]; k1 = a.k1[(size_t)c * B + j]; }
#pragma unroll
        for (int q = 0; q < P; ++q) p[q] = a.params[(size_t)q * B + j];
        x = a.x[j]; h = a.h[j]; facold = a.facold[j]; hlamb = a.hlamb[j];
        x0 = a.t0[(size_t)j * a.t0_stride]; xend = a.t1[(size_t)j * a.t1_stride];
        flags = a.flags[j];
        nstep0 = a.nstep[j]; nacc0 = a.naccpt[j];
    }
    // this lane's tolerance pair (Tolerance, mod.rs:104-214): select chain instead of a dynamically indexed kernarg array
    double rtol_c = a.rtol[0], atol_c = a.atol[0];
#pragma unroll
    for (int q = 1; q < N; ++q) { rtol_c = (c == (uint32_t)q) ? a.rtol[q] : rtol_c; atol_c = (c == (uint32_t)q) ? a.atol[q] : atol_c; }
    const double posneg = rs_signum(xend - x0);
    const double hmax = a.has_max_step ? a.max_step : fabs(xend - x0);   // dopri5.rs:180
    const bool over = nstep0 > a.nmax;
    const uint64_t left = over ? 0 : a.nmax - nstep0;
    const uint32_t budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    uint32_t acc_small = nacc0 > 2 ? 2u : (uint32_t)nacc0;
    uint32_t d_nfev = 0, d_nstep = 0, d_naccpt = 0, d_nrejct = 0;
    int32_t status = active ? IVP_RUNNING : 0;
    uint32_t it = 0;

    for (; it < a.chunk; ++it) {
        if (__ballot(status == IVP_RUNNING) == 0ull) break;   // every group of this wave has retired
        if (status != IVP_RUNNING) continue;                    // this group is done: its 8 lanes sit out together
        if (over || d_nstep > budget) { status = 2; continue; }
        if (0.1 * fabs(h) <= fabs(x) * uround) { status = 3; continue; }
        bool last = (flags & IVP_F_LAST) != 0;
        if ((x + 1.01 * h - xend) * posneg > 0.0) { h = xend - x; last = true; }
        d_nstep += 1;
        double y1 = y + h * A21 * k1;
        double k2 = coop_ode<R>(x + C2 * h, y1, base, c, p);
        y1 = y + h * (A31 * k1 + A32 * k2);
        const double k3 = coop_ode<R>(x + C3 * h, y1, base, c, p);
        y1 = y + h * (A41 * k1 + A42 * k2 + A43 * k3);
        double k4 = coop_ode<R>(x + C4 * h, y1, base, c, p);
        y1 = y + h * (A51 * k1 + A52 * k2 + A53 * k3 + A54 * k4);
        const double k5 = coop_ode<R>(x + C5 * h, y1, base, c, p);
        y1 = y + h * (A61 * k1 + A62 * k2 + A63 * k3 + A64 * k4 + A65 * k5);
        const double xph = x + h;
        const double k6 = coop_ode<R>(xph, y1, base, c, p);
        y1 = y + h * (A71 * k1 + A73 * k3 + A74 * k4 + A75 * k5 + A76 * k6);
        k2 = coop_ode<R>(xph, y1, base, c, p);   // k7 -> k2 (FSAL)
        d_nfev += 6;
        k4 = (E1 * k1 + E3 * k3 + E4 * k4 + E5 * k5 + E6 * k6 + E7 * k2) * h;
        const double sk = atol_c + rtol_c * fmax(fabs(y), fabs(y1));
        double err = coop_sum<N>((k4 / sk) * (k4 / sk), base);
        err = sqrt(err / (double)N);
        const double fac11 = ivp_pow(err, expo1);
        double fac = fac11 / ivp_pow(facold, beta);
        fac = fmax(facc2, fmin(facc1, fac / safety));
        double hnew = h / fac;
        if (err <= 1.0) {
            facold = fmax(err, 1.0e-4);
            d_naccpt += 1;
            if (acc_small < 2) acc_small += 1;
            if (stiff_tick<true>(a, j, flags, d_naccpt)) {   // dopri5.rs:364-391
                const double d1 = k2 - k6;
                const double ysti = y + h * (A61 * k1 + A62 * k2 + A63 * k3 + A64 * k4 + A65 * k5);
                const double d2 = y1 - ysti;
                const double stnum = coop_sum<N>(d1 * d1, base), stden = coop_sum<N>(d2 * d2, base);
                uint32_t iasti = (flags >> IVP_F_IASTI_SHIFT) & 0xFu, nonstiff = (flags >> IVP_F_NONSTIFF_SHIFT) & 0xFu;
                bool stiff_break = false;
                if (stden > 0.0) hlamb = fabs(h) * sqrt(stnum / stden);
                if (hlamb > 3.25) { nonstiff = 0; iasti += 1; if (iasti == 15) stiff_break = true; }
                else { nonstiff += 1; if (nonstiff == 6) iasti = 0; }
                flags = (flags & ~((0xFu << IVP_F_IASTI_SHIFT) | (0xFu << IVP_F_NONSTIFF_SHIFT))) |
                        ((iasti & 0xFu) << IVP_F_IASTI_SHIFT) | ((nonstiff & 0xFu) << IVP_F_NONSTIFF_SHIFT);
                if (stiff_break) { status = 4; continue; }
            }
            k1 = k2;
            y = y1;
            x = xph;
            if (last) { h = hnew; status = 0; continue; }
            if (fabs(hnew) > fabs(hmax)) hnew = posneg * fabs(hmax);
            if (flags & IVP_F_REJECT) { hnew = posneg * fmin(fabs(hnew), fabs(h)); flags &= ~IVP_F_REJECT; }
        } else {
            hnew = h / fmin(facc1, fac11 / safety);
            flags |= IVP_F_REJECT;
            if (acc_small > 1) d_nrejct += 1;
            last = false;
        }
        flags = last ? (flags | IVP_F_LAST) : (flags & ~IVP_F_LAST);
        h = hnew;
    }
    if (active) {
        if (own) { a.y[(size_t)c * B + j] = y; a.k1[(size_t)c * B + j] = k1; }
        if (c == 0) {
            a.x[j] = x; a.h[j] = h; a.facold[j] = facold; a.hlamb[j] = hlamb; a.flags[j] = flags; a.status[j] = status;
            a.nfev[j] += d_nfev; a.nstep[j] += d_nstep; a.naccpt[j] += d_naccpt; a.nrejct[j] += d_nrejct;
        }
    }
    compact_append(a, j, active && c == 0 && status == IVP_RUNNING);
    if (a.slot_counter) {
        const unsigned long long act = __ballot(active && c == 0);
        if (lane == 0) {
            atomicAdd(a.slot_counter, (unsigned long long)it * IVP_WAVE);
            atomicAdd(a.slot_counter + 1, (unsigned long long)__popcll(act));
        }
    }
}

template <class R>
__global__ __launch_bounds__(IVP_WAVE) void coop_chunk_kernel(const IvpKArgs a) { coop_chunk_body<R>(a); }

}  // namespace IVP_NS
