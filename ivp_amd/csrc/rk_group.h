// rk_group.h -- wave-per-trajectory ("thread-group owns one IVP") DOPRI5 kernels for state dimensions that do not
// fit one lane's registers (8 < n <= 512), e.g. the reference benchmark's "Large Linear System (N=100)"
// (benches/benchmark.py:139-148) or method-of-lines PDE systems.  Device only.
//
// Mapping.  One 64-lane wavefront integrates ONE trajectory; lane l owns components l, l+64, l+128, ...
// (C = ceil(n/64) per lane), so y, k1..k7 and the stage state are C-element VGPR arrays per lane.  A stage vector
// has to be visible to every lane before the right-hand side can be evaluated (component i of f may read any
// y_j), so each stage state is staged through LDS (n doubles per wave) and `R::ode_comp(i, t, y_lds, p)` computes
// one component from the shared copy.  Step-size control is wave-uniform: every lane computes the same controller
// scalars, so accept/reject is a uniform branch -- no predication, no divergence.
//
// The weighted RMS error norm is the one cross-lane operation:
//   * STRICT build: the n squared terms are written to LDS and every lane adds them in index order, which is the
//     reference's left-to-right sum (dopri5.rs:343-347) bit for bit (n dependent adds; LDS reads are broadcasts);
//   * FAST build: per-lane partial sums + a 6-step __shfl_xor butterfly over the wavefront.
// The per-component stage arithmetic is the same expression sequence as in rk_core.h.
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
        return p[0] * (left - 2.0 * y[i] + right);
    }
};

template <int N>
struct GroupLds {
    double stage[N];   // stage state visible to all lanes
    double red[N];     // per-component terms of a reduction (strict order)
};

// sum_{i<N} term_i in index order (STRICT) or by lane partials + butterfly (FAST). term[c] belongs to index lane+64c.
template <int N, int C>
__device__ __forceinline__ double group_sum(const double (&term)[C], GroupLds<N> &lds, uint32_t lane)
{
#if IVP_FAST
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) if ((int)lane + 64 * c < N) part += term[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    return part;
#else
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) { const int i = (int)lane + 64 * c; if (i < N) lds.red[i] = term[c]; }
    __syncthreads();
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < N; ++i) s += lds.red[i];
    return s;
#endif
}

// k = f(t, ystage): publish the stage state, then evaluate this lane's components
template <class R, int C>
__device__ __forceinline__ void group_ode(double t, const double (&ys)[C], double (&k)[C], const double *p,
                                          GroupLds<R::N> &lds, uint32_t lane)
{
    __syncthreads();   // previous readers of lds.stage are done
#pragma unroll
    for (int c = 0; c < C; ++c) { const int i = (int)lane + 64 * c; if (i < R::N) lds.stage[i] = ys[c]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) { const int i = (int)lane + 64 * c; k[c] = i < R::N ? R::ode_comp(i, t, lds.stage, p) : 0.0; }
}

template <class R, int C>
__device__ __forceinline__ double group_hinit(const IvpKArgs &a, double x, const double (&y)[C], double posneg,
                                              const double (&f0)[C], const double *p, int iord, double hmax,
                                              GroupLds<R::N> &lds, uint32_t lane)
{   // mod.rs:217-281 (scalar tolerances: rtol[0]/atol[0] apply to every component)
    constexpr int N = R::N;
    double t1[C], t2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double sk = a.atol[0] + a.rtol[0] * fabs(y[c]);
        t1[c] = (f0[c] / sk) * (f0[c] / sk);
        t2[c] = (y[c] / sk) * (y[c] / sk);
    }
    const double dnf = group_sum<N, C>(t1, lds, lane);
    const double dny = group_sum<N, C>(t2, lds, lane);
    double h;
    if (dnf <= 1e-10 || dny <= 1e-10) h = 1.0e-6;
    else h = sqrt(dny / dnf) * 0.01;
    if (h > fabs(hmax)) h = fabs(hmax);
    h = fabs(h) * rs_signum(posneg);
    double y1[C], f1[C];
#pragma unroll
    for (int c = 0; c < C; ++c) y1[c] = y[c] + h * f0[c];
    group_ode<R, C>(x + h, y1, f1, p, lds, lane);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double sk = a.atol[0] + a.rtol[0] * fabs(y[c]);
        const double df = (f1[c] - f0[c]) / sk;
        t1[c] = df * df;
    }
    double der2 = group_sum<N, C>(t1, lds, lane);
    der2 = sqrt(der2) / fabs(h);
    const double der12 = fmax(fabs(der2), sqrt(dnf));
    double h1;
    if (der12 <= 1.0e-15) h1 = fmax(1.0e-6, fabs(h) * 1.0e-3);
    else h1 = ivp_pow(0.01 / der12, 1.0 / (double)iord);
    const double hf = fmin(fmin(fmin(fabs(h), 100.0 * fabs(h)), h1), fabs(hmax));
    return fabs(hf) * rs_signum(posneg);
}

// init: one wave per trajectory (solve_ivp.rs:110-145 short-circuit; dopri5.rs:230-240)
template <class R>
__device__ __forceinline__ void group_init_body(const IvpKArgs &a)
{
    constexpr int N = R::N, P = R::P, C = (N + 63) / 64;
    __shared__ GroupLds<N> lds;
    const uint32_t lane = threadIdx.x;
    const uint32_t j = blockIdx.x;
    if (j >= a.B) return;
    const size_t B = a.B;
    double y[C], k1[C], p[P > 0 ? P : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) { const int i = (int)lane + 64 * c; y[c] = i < N ? a.y0[(size_t)i * B + j] : 0.0; }
#pragma unroll
    for (int c = 0; c < P; ++c) p[c] = a.params[(size_t)c * B + j];
    const double x0 = a.t0[(size_t)j * a.t0_stride], xend = a.t1[(size_t)j * a.t1_stride];
    int32_t status = IVP_RUNNING;
    double h = 0.0;
    uint64_t nfev = 0;
    if (fabs(xend - x0) < 1e-15) status = 0;
    else if (x0 != x0 || xend != xend) status = 3;   // NaN interval: see init_body in rk_core.h
    if (status == IVP_RUNNING) {
        const double posneg = rs_signum(xend - x0);
        const double hmax = a.has_max_step ? a.max_step : fabs(xend - x0);
        group_ode<R, C>(x0, y, k1, p, lds, lane);
        nfev = 1;
        if (a.has_first_step) h = fabs(a.first_step) * posneg;
        else { nfev = 2; h = group_hinit<R, C>(a, x0, y, posneg, k1, p, 5, hmax, lds, lane); }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) k1[c] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = (int)lane + 64 * c;
        if (i < N) { a.y[(size_t)i * B + j] = y[c]; a.k1[(size_t)i * B + j] = k1[c]; }
    }
    if (lane == 0) {
        a.x[j] = x0; a.h[j] = h; a.facold[j] = 1e-4; a.hlamb[j] = 0.0; a.flags[j] = 0; a.status[j] = status;
        a.nfev[j] = nfev; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0;
    }
}

// up to a.chunk DOPRI5 attempts (dopri5.rs:266-461) for the trajectory of this wave
template <class R>
__device__ __forceinline__ void group_chunk_body(const IvpKArgs &a)
{
    constexpr int N = R::N, P = R::P, C = (N + 63) / 64;
    __shared__ GroupLds<N> lds;
    KC_SCOPE
    constexpr double C2 = 0.2, C3 = 0.3, C4 = 0.8, C5 = 8.0 / 9.0;
    constexpr double A21 = 0.2, A31 = 3.0 / 40.0, A32 = 9.0 / 40.0;
    constexpr double A41 = 44.0 / 45.0, A42 = -56.0 / 15.0, A43 = 32.0 / 9.0;
    constexpr double A51 = 19372.0 / 6561.0, A52 = -25360.0 / 2187.0, A53 = 64448.0 / 6561.0, A54 = -212.0 / 729.0;
    constexpr double A61 = 9017.0 / 3168.0, A62 = -355.0 / 33.0, A63 = 46732.0 / 5247.0, A64 = 49.0 / 176.0, A65 = -5103.0 / 18656.0;
    constexpr double A71 = 35.0 / 384.0, A73 = 500.0 / 1113.0, A74 = 125.0 / 192.0, A75 = -2187.0 / 6784.0, A76 = 11.0 / 84.0;
    constexpr double E1 = 71.0 / 57600.0, E3 = -71.0 / 16695.0, E4 = 71.0 / 1920.0, E5 = -17253.0 / 339200.0, E6 = 22.0 / 525.0, E7 = -1.0 / 40.0;
    const double uround = a.ctl_uround, safety = a.ctl_safety, beta = a.ctl_beta;   // dopri5.rs:34-72 struct fields
    const double facc1 = a.ctl_facc1, facc2 = a.ctl_facc2, expo1 = a.ctl_expo1;

    const uint32_t lane = threadIdx.x;
    const uint32_t count = a.perm_in ? *a.count_in : a.B;
    if (blockIdx.x >= count) return;
    const uint32_t j = a.perm_in ? a.perm_in[blockIdx.x] : blockIdx.x;
    if (a.status[j] != IVP_RUNNING) return;
    const size_t B = a.B;
    double y[C], k1[C], p[P > 0 ? P : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = (int)lane + 64 * c;
        y[c] = i < N ? a.y[(size_t)i * B + j] : 0.0;
        k1[c] = i < N ? a.k1[(size_t)i * B + j] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < P; ++c) p[c] = a.params[(size_t)c * B + j];
    double x = a.x[j], h = a.h[j], facold = a.facold[j], hlamb = a.hlamb[j];
    const double x0 = a.t0[(size_t)j * a.t0_stride], xend = a.t1[(size_t)j * a.t1_stride];
    const double posneg = rs_signum(xend - x0);
    const double hmax = a.has_max_step ? a.max_step : fabs(xend - x0);
    uint32_t flags = a.flags[j];
    const uint64_t nstep0 = a.nstep[j], nacc0 = a.naccpt[j];
    const bool over = nstep0 > a.nmax;
    const uint64_t left = over ? 0 : a.nmax - nstep0;
    const uint32_t budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    uint32_t acc_small = nacc0 > 2 ? 2u : (uint32_t)nacc0;
    uint32_t d_nfev = 0, d_nstep = 0, d_naccpt = 0, d_nrejct = 0;
    int32_t status = IVP_RUNNING;

    for (uint32_t it = 0; it < a.chunk && status == IVP_RUNNING; ++it) {
        if (over || d_nstep > budget) { status = 2; break; }
        if (KC(0.1) * fabs(h) <= fabs(x) * uround) { status = 3; break; }
        bool last = (flags & IVP_F_LAST) != 0;
        if ((x + KC(1.01) * h - xend) * posneg > 0.0) { h = xend - x; last = true; }
        d_nstep += 1;
        double k2[C], k3[C], k4[C], k5[C], k6[C], y1[C];
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * A21 * k1[c];
        group_ode<R, C>(x + KC(C2) * h, y1, k2, p, lds, lane);
        { const double c1 = KC(A31), c2 = KC(A32);
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * (c1 * k1[c] + c2 * k2[c]); }
        group_ode<R, C>(x + KC(C3) * h, y1, k3, p, lds, lane);
        { const double c1 = KC(A41), c2 = KC(A42), c3 = KC(A43);
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * (c1 * k1[c] + c2 * k2[c] + c3 * k3[c]); }
        group_ode<R, C>(x + KC(C4) * h, y1, k4, p, lds, lane);
        { const double c1 = KC(A51), c2 = KC(A52), c3 = KC(A53), c4 = KC(A54);
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * (c1 * k1[c] + c2 * k2[c] + c3 * k3[c] + c4 * k4[c]); }
        group_ode<R, C>(x + KC(C5) * h, y1, k5, p, lds, lane);
        { const double c1 = KC(A61), c2 = KC(A62), c3 = KC(A63), c4 = KC(A64), c5 = KC(A65);
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * (c1 * k1[c] + c2 * k2[c] + c3 * k3[c] + c4 * k4[c] + c5 * k5[c]); }
        const double xph = x + h;
        group_ode<R, C>(xph, y1, k6, p, lds, lane);
        { const double c1 = KC(A71), c3 = KC(A73), c4 = KC(A74), c5 = KC(A75), c6 = KC(A76);
#pragma unroll
        for (int c = 0; c < C; ++c) y1[c] = y[c] + h * (c1 * k1[c] + c3 * k3[c] + c4 * k4[c] + c5 * k5[c] + c6 * k6[c]); }
        group_ode<R, C>(xph, y1, k2, p, lds, lane);   // k7 -> k2 (FSAL)
        d_nfev += 6;
        double term[C];
        { const double e1 = KC(E1), e3 = KC(E3), e4 = KC(E4), e5 = KC(E5), e6 = KC(E6), e7 = KC(E7);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            k4[c] = (e1 * k1[c] + e3 * k3[c] + e4 * k4[c] + e5 * k5[c] + e6 * k6[c] + e7 * k2[c]) * h;
            const double sk = a.atol[0] + a.rtol[0] * fmax(fabs(y[c]), fabs(y1[c]));
            term[c] = (k4[c] / sk) * (k4[c] / sk);
        } }
        double err = group_sum<N, C>(term, lds, lane);
        err = sqrt(err / (double)N);
        const double fac11 = ivp_pow(err, expo1);
        double fac = fac11 / ivp_pow(facold, beta);
        fac = fmax(facc2, fmin(facc1, fac / safety));
        double hnew = h / fac;
        if (err <= 1.0) {
            facold = fmax(err, KC(1.0e-4));
            d_naccpt += 1;
            if (acc_small < 2) acc_small += 1;
            if (stiff_tick<true>(a, j, flags, d_naccpt)) {   // dopri5.rs:364-391
                double t1[C], t2[C];
                { const double c1 = KC(A61), c2 = KC(A62), c3 = KC(A63), c4 = KC(A64), c5 = KC(A65);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double d1 = k2[c] - k6[c];
                    const double ysti = y[c] + h * (c1 * k1[c] + c2 * k2[c] + c3 * k3[c] + c4 * k4[c] + c5 * k5[c]);
                    const double d2 = y1[c] - ysti;
                    t1[c] = d1 * d1;
                    t2[c] = d2 * d2;
                } }
                const double stnum = group_sum<N, C>(t1, lds, lane), stden = group_sum<N, C>(t2, lds, lane);
                uint32_t iasti = (flags >> IVP_F_IASTI_SHIFT) & 0xFu, nonstiff = (flags >> IVP_F_NONSTIFF_SHIFT) & 0xFu;
                bool stiff_break = false;
                if (stden > 0.0) hlamb = fabs(h) * sqrt(stnum / stden);
                if (hlamb > 3.25) { nonstiff = 0; iasti += 1; if (iasti == 15) stiff_break = true; }
                else { nonstiff += 1; if (nonstiff == 6) iasti = 0; }
                flags = (flags & ~((0xFu << IVP_F_IASTI_SHIFT) | (0xFu << IVP_F_NONSTIFF_SHIFT))) |
                        ((iasti & 0xFu) << IVP_F_IASTI_SHIFT) | ((nonstiff & 0xFu) << IVP_F_NONSTIFF_SHIFT);
                if (stiff_break) { status = 4; break; }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) { k1[c] = k2[c]; y[c] = y1[c]; }
            x = xph;
            if (last) { h = hnew; status = 0; break; }
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
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = (int)lane + 64 * c;
        if (i < N) { a.y[(size_t)i * B + j] = y[c]; a.k1[(size_t)i * B + j] = k1[c]; }
    }
    if (lane == 0) {
        a.x[j] = x; a.h[j] = h; a.facold[j] = facold; a.hlamb[j] = hlamb; a.flags[j] = flags; a.status[j] = status;
        a.nfev[j] += d_nfev; a.nstep[j] += d_nstep; a.naccpt[j] += d_naccpt; a.nrejct[j] += d_nrejct;
        if (status == IVP_RUNNING) a.perm_out[atomicAdd(a.count_out, 1u)] = j;   // still running: next launch's list
        if (a.slot_counter) {
            atomicAdd(a.slot_counter, (unsigned long long)d_nstep * IVP_WAVE);
            atomicAdd(a.slot_counter + 1, 1ull);
        }
    }
}

template <class R>
__global__ __launch_bounds__(IVP_WAVE) void group_init_kernel(const IvpKArgs a) { group_init_body<R>(a); }
template <class R>
__global__ __launch_bounds__(IVP_WAVE) void group_chunk_kernel(const IvpKArgs a) { group_chunk_body<R>(a); }

}  // namespace IVP_NS
