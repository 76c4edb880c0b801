// bdf_core.h -- per-lane body of the variable-order BDF(1..5) integrator (BASELINE C5).
//
// Restates src/methods/bdf.rs:86-732 with its helpers: the SciPy-style differences array D, simplified Newton
// with LU reuse (`|c - c_old| / max(|c|,1) > 0.1`), `change_d`, the dense LU of src/matrix/lu.rs:37-125 /
// src/matrix/linear.rs:55-96 and the default forward-difference Jacobian of src/ivp.rs:67-107.
//
// One lane owns one trajectory, like the explicit RK kernels.  The per-trajectory (I - cJ) matrices are n x n with
// n <= 8, so "batched LU in LDS" (BASELINE C5's wording) degenerates to LU in registers: every matrix index is a
// compile-time constant, the run-time pivot row and the run-time order are handled with select chains / guarded
// unrolled loops so nothing is dynamically indexed (dynamic indexing would push the arrays to scratch).
// All call sites of the expensive helpers are single: D-rescalings requested by a rejected attempt are deferred to
// the top of the next attempt (nothing reads D in between), so `change_d` is instantiated once inside a 4-pass loop.
#pragma once
#ifndef __HIPCC_RTC__
#include "rk_core.h"
#endif

namespace IVP_NS {

constexpr int BDF_MAXO = 5;
#define IVP_BDF_ORDER_SHIFT 4     // flags bits 4..6   order 1..5
#define IVP_BDF_NEQ_SHIFT 8       // flags bits 8..10  n_equal_steps 0..6
#define IVP_BDF_LU_CURRENT 0x1000u
#define IVP_BDF_PENDING 0x2000u   // a change_d(pending_factor) is owed before D is used again

// Phase clock (tools/bdf_phase_profile.sh builds rk_bdf.hip with -DIVP_PHASE_PROF): at marker k the wave adds the shader
// clock ticks since its previous marker to slot k of an LDS table (two LDS round trips, ~200 ticks per marker); the table
// goes to ivp_phase_ticks with atomics when the launch ends.  One wave per block; all of it a no-op in the product build.
#if defined(IVP_PHASE_PROF) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned long long *ivp_phase_lds()
{
    __shared__ unsigned long long t[18];   // [0..15] ticks / counts, [16] clock at the previous marker
    return t;
}
__device__ __forceinline__ bool ivp_phase_leader()
{
    return __builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0u;
}
__device__ __forceinline__ void ivp_phase_mark(int k)
{
    unsigned long long *t = ivp_phase_lds();
    const unsigned long long now = __builtin_readcyclecounter();
    if (ivp_phase_leader()) {
        t[k] += now - t[16];
        t[16] = now;
        if (k == 0) t[15] += 1ull;
    }
}
__device__ __forceinline__ void ivp_phase_begin()
{
    unsigned long long *t = ivp_phase_lds();
    if (ivp_phase_leader()) {
        for (int k = 0; k < 16; ++k) t[k] = 0ull;
        t[16] = __builtin_readcyclecounter();
    }
}
__device__ __forceinline__ void ivp_phase_end()
{
    unsigned long long *t = ivp_phase_lds();
    if (ivp_phase_leader())
        for (int k = 0; k < 16; ++k) atomicAdd(&::ivp_phase_ticks[k], t[k]);
}
#define IVP_PHASE(k) ivp_phase_mark(k)
#define IVP_PHASE_BEGIN() ivp_phase_begin()
#define IVP_PHASE_END() ivp_phase_end()
#else
#define IVP_PHASE(k) ((void)0)
#define IVP_PHASE_BEGIN() ((void)0)
#define IVP_PHASE_END() ((void)0)
#endif

struct BdfTables {
    double gamma[6], alpha[6], error_const[6];
    constexpr BdfTables() : gamma{}, alpha{}, error_const{}
    {   // bdf.rs:161-172, KAPPA bdf.rs:22
        constexpr double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
        gamma[0] = 0.0;
        for (int k = 1; k <= BDF_MAXO; ++k) gamma[k] = gamma[k - 1] + 1.0 / (double)k;
        for (int k = 0; k <= BDF_MAXO; ++k) alpha[k] = (1.0 - KAPPA[k]) * gamma[k];
        for (int k = 0; k <= BDF_MAXO; ++k) error_const[k] = KAPPA[k] * gamma[k] + 1.0 / ((double)k + 1.0);
    }
};

// t[i] for a run-time i in 0..5 as a select chain over register values.  The opaque copies keep LLVM from folding the
// chain into "select the ADDRESS, then load": that would turn a constant table into a global-memory lookup and a lane's
// difference array into a scratch object -- several hundred cycles of latency on a lone wave's critical path each time.
IVP_HD double bdf_sel6(const double *t, int i)
{
    double v = t[0];
    IVP_OPAQUE_V(v);
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        double tk = t[k];
        IVP_OPAQUE_V(tk);
        v = (i == k) ? tk : v;
    }
    return v;
}

template <int N>
IVP_HD double bdf_wrms(const double *v, const double *scale)
{   // weighted_rms_scaled, bdf.rs:659-667
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double denom = scale[i] == 0.0 ? 2.220446049250313e-16 : scale[i];
        const double ratio = v[i] / denom;
        sum += ratio * ratio;
    }
    return sqrt(sum / (double)N);
}

// change_d (bdf.rs:669-732): D[0..order] <- (R(order, factor) . R(order, 1))^T applied to D.
// The literal form: every product and every zero test of compute_r / matmul / the application loop.  Only taken for a
// factor that is not finite or astronomically large (see bdf_change_d below).
template <int N>
IVP_HD void bdf_change_d_generic(double (&d)[8][N], int order, double factor)
{
    const int size = order + 1;
    double r[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) r[0][j] = 1.0;
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        r[i][0] = r[i - 1][0] * 0.0;   // m[i][0] = 0 (compute_r only fills j >= 1)
#pragma unroll
        for (int j = 1; j < 6; ++j) r[i][j] = r[i - 1][j] * (((double)i - 1.0 - factor * (double)j) / (double)i);
    }
    double scratch[6][N];
#pragma unroll
    for (int row = 0; row < 6; ++row) {
        // column `row` of U = compute_r(order, 1.0)
        double u[6];
        u[0] = 1.0;
#pragma unroll
        for (int i = 1; i < 6; ++i) u[i] = row == 0 ? u[i - 1] * 0.0 : u[i - 1] * (((double)i - 1.0 - 1.0 * (double)row) / (double)i);
#pragma unroll
        for (int c = 0; c < N; ++c) scratch[row][c] = 0.0;
        if (row <= order) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k < size) {
                    // ru[k][row] = sum_m r[k][m] * u[m][row], m < size, zero coefficients skipped (bdf.rs:721-724)
                    double ru = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        if (m < size && r[k][m] != 0.0) ru = IVP_MA(ru, r[k][m], u[m]);
                    if (ru != 0.0) {
#pragma unroll
                        for (int c = 0; c < N; ++c) scratch[row][c] = IVP_MA(scratch[row][c], ru, d[k][c]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i <= order) {
#pragma unroll
            for (int c = 0; c < N; ++c) d[i][c] = scratch[i][c];
        }
}

// U = compute_r(order, 1.0) (bdf.rs:694-712) does not depend on the order (only its size does):
// U[m][j] = prod_{i=1..m} (i - 1 - j) / i, evaluated here with the reference's operations in the reference's order.
struct BdfU {
    double v[6][6];
    constexpr BdfU() : v{}
    {
        for (int j = 0; j < 6; ++j) v[0][j] = 1.0;
        for (int i = 1; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
                v[i][j] = j == 0 ? v[i - 1][j] * 0.0 : v[i - 1][j] * (((double)i - 1.0 - 1.0 * (double)j) / (double)i);
    }
};

// change_d, structured.  With R = compute_r(order, factor) and U as above the reference forms RU = R.U with matmul()
// (terms with a zero R entry skipped) and D[row] <- sum_k RU[k][row] D[k] (zero coefficients skipped).  For a finite
// factor the following holds exactly, rounding included:
//   * U[m][row] is +-0 for m > row, R[k][0] is 0 for k >= 1: those products are +-0, and adding +-0 to an accumulator
//     that started at +0.0 never changes it (it cannot become -0) -- the same reason why matmul's zero-skip is a no-op;
//     so RU[k][row] = sum_{m=1..row} R[k][m] U[m][row] for k, row >= 1, RU[k][0] = 0 for k >= 1;
//   * RU[0][0] = 1 and RU[0][row] = sum_m U[m][row] = 0 exactly (alternating binomials) for row >= 1;
//   * entries with m > order only reach rows > order, which are not stored; a row k > order of D is excluded by
//     multiplying its coefficients with zero, and a zero coefficient times a FINITE difference is a zero that leaves
//     the sum alone, which is also what the reference's "skip zero coefficients" amounts to.
// This is a third of the literal form's instructions (no 6 x 6 x 6 guarded products, no per-term selects), and change_d is
// the fattest phase of an attempt: a wave pays for it whenever any of its lanes rescales.  A factor that is NaN /
// infinite / > 1e50 (R could overflow) or a non-finite entry of D (0 x inf would matter) takes the literal form.
// x / C for a small integer constant C, correctly rounded: q = x * RN(1/C), the exact remainder r = x - C q (one fma),
// q' = RN(q + r * RN(1/C)) -- Markstein's correction step, which yields the IEEE quotient whenever RN(1/C) is the
// correctly rounded reciprocal and nothing under- or overflows (callers keep |x| in [2^-52, 1e51] or +0; a -0 would come
// back as +0, and k - 1 - factor * m with k - 1 >= 2 cannot be one).  Three
// independent-latency instructions instead of the 11-instruction v_div_scale / v_rcp / ... / v_div_fixup chain; checked
// against the hardware division on 2e8 random operands per constant (tests: test_div_by_small_constant_is_ieee_division).
template <int C>
IVP_HD double ivp_div_small_const(double x)
{
    constexpr double c = (double)C, y = 1.0 / (double)C;
    const double q = x * y;
    const double r = fma(-c, q, x);
    return fma(r, y, q);
}

template <int N>
IVP_HD void bdf_change_d(double (&d)[8][N], int order, double factor)
{
    if (factor == 1.0) return;
    if (order > BDF_MAXO) order = BDF_MAXO;
    // the structured form needs finite arithmetic throughout: a finite, not astronomically large factor (so that R stays
    // finite) and finite differences in the rows it reads (so that a zero coefficient times D is a zero)
    bool plain = fabs(factor) < 1e50;
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < N; ++c) plain = plain && fabs(d[k][c]) < u2d(0x7FF0000000000000ull);
    if (!plain) { bdf_change_d_generic<N>(d, order, factor); return; }
    constexpr BdfU U{};
    // Loop nests below run with the SUMMATION index outermost: every accumulator still receives its terms in the
    // reference's order, but consecutive instructions belong to different accumulators (a lone wave waits 8.5 cycles for
    // a dependent f64 result and can issue an independent one after 5.5).
    double r[6][6], fm[6];
#pragma unroll
    for (int m = 1; m < 6; ++m) { fm[m] = factor * (double)m; r[0][m] = 1.0; }
#pragma unroll
    for (int k = 1; k < 6; ++k) {
#pragma unroll
        for (int m = 1; m < 6; ++m) {
            const double num = (double)k - 1.0 - fm[m];
            const double quot = k == 3 ? ivp_div_small_const<3>(num) : (k == 5 ? ivp_div_small_const<5>(num) : num / (double)k);
            r[k][m] = r[k - 1][m] * quot;
        }
    }
    // a coefficient of a row above the order must not contribute: times zero it becomes a zero, which leaves the sum alone
    double keep[6];
#pragma unroll
    for (int k = 1; k < 6; ++k) keep[k] = k <= order ? 1.0 : 0.0;
    double ru[6][6];
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
        for (int row = 1; row < 6; ++row) ru[k][row] = r[k][1] * U.v[1][row];
#pragma unroll
    for (int m = 2; m < 6; ++m)
#pragma unroll
        for (int k = 1; k < 6; ++k)
#pragma unroll
            for (int row = 1; row < 6; ++row)
                if (m <= row) ru[k][row] = IVP_MA(ru[k][row], r[k][m], U.v[m][row]);
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
        for (int row = 1; row < 6; ++row) ru[k][row] = ru[k][row] * keep[k];
    double scratch[6][N];
#pragma unroll
    for (int c = 0; c < N; ++c) scratch[0][c] = 0.0 + 1.0 * d[0][c];
#pragma unroll
    for (int row = 1; row < 6; ++row)
#pragma unroll
        for (int c = 0; c < N; ++c) scratch[row][c] = 0.0;
#pragma unroll
    for (int k = 1; k < 6; ++k)
#pragma unroll
        for (int row = 1; row < 6; ++row)
#pragma unroll
            for (int c = 0; c < N; ++c) scratch[row][c] = IVP_MA(scratch[row][c], ru[k][row], d[k][c]);
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i <= order) {
#pragma unroll
            for (int c = 0; c < N; ++c) d[i][c] = scratch[i][c];
        }
}

// lu_decomp (src/matrix/lu.rs:37-125), row-major a[r][c]; pivots packed 4 bits each. Returns false if singular.
template <int N>
IVP_HD bool bdf_lu_decomp(double (&a)[N][N], uint32_t &piv)
{
    piv = 0;
    if (N == 1) return a[0][0] != 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        int m = k;
        double max_val = fabs(a[k][k]);
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double v = fabs(a[i][k]);
            if (v > max_val) { max_val = v; m = i; }
        }
        piv |= (uint32_t)m << (4 * k);
        double pivot = a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) pivot = (m == i) ? a[i][k] : pivot;
        if (pivot == 0.0) ok = false;
        if (ok) {
            // swap a[m][k] <-> a[k][k]
#pragma unroll
            for (int i = k + 1; i < N; ++i)
                if (m == i) { const double t = a[i][k]; a[i][k] = a[k][k]; a[k][k] = t; }
            const double t = 1.0 / pivot;
#pragma unroll
            for (int i = k + 1; i < N; ++i) a[i][k] = -a[i][k] * t;
#pragma unroll
            for (int j = k + 1; j < N; ++j) {
                double tj = a[k][j];
#pragma unroll
                for (int i = k + 1; i < N; ++i) tj = (m == i) ? a[i][j] : tj;
#pragma unroll
                for (int i = k + 1; i < N; ++i)
                    if (m == i) { const double tmp = a[i][j]; a[i][j] = a[k][j]; a[k][j] = tmp; }
                if (tj != 0.0) {
#pragma unroll
                    for (int i = k + 1; i < N; ++i) a[i][j] = IVP_MA(a[i][j], a[i][k], tj);
                }
            }
        }
    }
    if (ok && a[N - 1][N - 1] == 0.0) ok = false;
    return ok;
}

// lin_solve (src/matrix/linear.rs:55-96)
template <int N>
IVP_HD void bdf_lin_solve(const double (&a)[N][N], double (&b)[N], uint32_t piv)
{
    if (N == 1) { b[0] /= a[0][0]; return; }
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        const int m = (int)((piv >> (4 * k)) & 0xFu);
#pragma unroll
        for (int i = k + 1; i < N; ++i)
            if (m == i) { const double t = b[i]; b[i] = b[k]; b[k] = t; }
#pragma unroll
        for (int i = k + 1; i < N; ++i) b[i] = IVP_MA(b[i], a[i][k], b[k]);
    }
#pragma unroll
    for (int kb = 1; kb < N; ++kb) {
        const int k = N - kb;
        b[k] /= a[k][k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] = IVP_MA(b[i], a[i][k], -b[k]);
    }
    b[0] /= a[0][0];
}

// default IVP::jac: forward differences (src/ivp.rs:67-107)
template <class R>
IVP_HD void bdf_fd_jac(double x, const double *y, const double *p, double (&jac)[R::N][R::N])
{
    constexpr int N = R::N;
    double fo[N], fp[N], yp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) yp[i] = y[i];
    R::ode(x, y, fo, p);
    const double eps = 1.4901161193847656e-08;   // f64::EPSILON.sqrt() = 2^-26
#pragma unroll
    for (int col = 0; col < N; ++col) {
        const double yo = y[col];
        const double pert = eps * fmax(fabs(yo), 1.0);
        yp[col] = yo + pert;
        R::ode(x, yp, fp, p);
        yp[col] = yo;
#pragma unroll
        for (int row = 0; row < N; ++row) jac[row][col] = (fp[row] - fo[row]) / pert;
    }
}

// f.jac(x, y, &mut j): the problem's own `jac` (an `impl IVP` override: the functor defines
//     static void jac(double x, const double* y, double (&j)[N][N], const double* p)
// or, for hiprtc problems, the snippet defines  __device__ void jac(double x, const double* y, double* j /* row-major */, const double* p))
// when it has one, else the trait's default forward-difference implementation above.
template <class R, class = void>
struct HasJac { enum { v = 0 }; };
template <class R>
struct HasJac<R, decltype((void)&R::jac)> { enum { v = 1 }; };
template <class R>
IVP_HD void bdf_eval_jac(double x, const double *y, const double *p, double (&jac)[R::N][R::N])
{
    if constexpr (HasJac<R>::v) R::jac(x, y, jac, p);
    else bdf_fd_jac<R>(x, y, p, jac);
}

template <int N>
struct BdfLane {
    double y[N], d[8][N], jac[N][N], lu[N][N];
    double x, current_h, current_c, pending_factor, xend, x0, direction, hmax, hmin;
    uint32_t piv, flags;
    int32_t status;
    uint32_t d_nfev, d_njev, d_nlu, d_nstep, d_naccpt, d_nrejct, budget;
    bool over;
};

template <class R, int FULL>
IVP_HD int32_t bdf_init_body(const IvpKArgs &a, uint32_t j)
{
    constexpr int N = R::N, P = R::P;
    const size_t B = a.B;
    Lane<N, P> L;   // SolOut registers + params
    double y[N], f0[N];
#pragma unroll
    for (int c = 0; c < N; ++c) y[c] = a.y0[c * B + j];
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    L.x0 = a.t0[(size_t)j * a.t0_stride];
    L.xend = a.t1[(size_t)j * a.t1_stride];
    L.flags = 0;
    L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0; L.n_log = 0;
    if (FULL && a.log_pool != nullptr) so_log_open<IdMap<N>>(a, j, L, 2u, 0u, 1u);   // the initial callback records at most twice
    auto store_so = [&]() {
        if (FULL) {
            a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_log[j] = L.n_log;
            a.n_seg[j] = L.n_seg; a.t_last[j] = L.t_last;
            if (a.log_pool != nullptr) so_log_flush<IdMap<N>>(a, L);
        }
    };
    a.nfev[j] = 0; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; a.njev[j] = 0; a.nlu[j] = 0;
    a.facold[j] = 0.0; a.hlamb[j] = 1.0; a.bdf_piv[j] = 0;
#pragma unroll
    for (int c = 0; c < N; ++c) { a.y[c * B + j] = y[c]; a.k1[c * B + j] = 0.0; }

    if (fabs(L.xend - L.x0) < 1e-15) {  // solve_ivp.rs:110-145
        if (FULL) {
            if (a.n_eval >= 0) {
                const EvalGrid grid = so_grid(a, j);
                for (int32_t i = 0; i < grid.n; ++i)
                    if (fabs(grid.t[i] - L.x0) < 1e-12) so_emit_eval<M_BDF, N, P>(a, j, L, i, y);
            } else if (a.t_log != nullptr) {
                so_push_log<M_BDF, N, P>(a, j, L, L.x0, y);
            }
            if (a.collect_dense && a.max_log > 0) {  // ContinuousOutput::constant, BDF layout (cont.rs:44-51)
#pragma unroll
                for (int c = 0; c < 7 * N; ++c)
                    a.seg_cont[(size_t)c * B + j] = (c % 7 == 0) ? y[c / 7] : ((c % 7 == 6) ? 1.0 : 0.0);
                a.seg_xold[j] = L.x0;
                a.seg_h[j] = 1e-15;
                L.n_seg = 1;
            }
        }
        store_so();
        a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 0;
        return 0;
    }
    if (L.x0 != L.x0 || L.xend != L.xend) {   // NaN interval: see init_body in rk_core.h
        store_so();
        a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 3;
        return 3;
    }
    const double direction = rs_signum(L.xend - L.x0);
    const double hmax = fabs(a.has_max_step ? a.max_step : fabs(L.xend - L.x0));
    R::ode(L.x0, y, f0, L.p);
    double jac[N][N];
    if constexpr (HasJac<R>::v) {
        // the reference hands f.jac() a zero-initialised persistent Matrix (bdf.rs:152): an override that fills only
        // its non-zero entries leaves zeros elsewhere; later calls see the previous Jacobian (S.jac persists)
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c < N; ++c) jac[r][c] = 0.0;
    }
    bdf_eval_jac<R>(L.x0, y, L.p, jac);
    double h_abs;
    if (a.has_first_step) {
        if (a.first_step == 0.0) {   // Err(InvalidStepSize), bdf.rs:192-197
            ivp_flag_error(a, IVP_ERRFLAG_INVALID_STEP);
            store_so();
            a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 0;
            return 0;
        }
        h_abs = fabs(a.first_step);
    } else {
        double guess = hinit<R>(a, L.x0, y, direction, f0, L.p, 1, hmax);
        const double max_h = fabs(L.xend - L.x0);
        if (fabs(guess) > max_h) guess = max_h * direction;
        h_abs = fabs(guess);
    }
    h_abs = fmin(h_abs, fmax(hmax, 2.2250738585072014e-308));
#pragma unroll
    for (int c = 0; c < N; ++c) {
        a.bdf_d[(size_t)(0 * N + c) * B + j] = y[c];
        a.bdf_d[(size_t)(1 * N + c) * B + j] = f0[c] * h_abs * direction;
#pragma unroll
        for (int k = 2; k < 8; ++k) a.bdf_d[(size_t)(k * N + c) * B + j] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) { a.bdf_jac[(size_t)(r * N + c) * B + j] = jac[r][c]; a.bdf_lu[(size_t)(r * N + c) * B + j] = 0.0; }
    L.x = L.x0;
    if (FULL) (void)solout_full<M_BDF, R>(a, j, L, L.x0, L.x0, y, (const double *)y, (const double *)nullptr, 0.0, L.x0);
    store_so();
    a.nfev[j] = 1; a.njev[j] = 1;
    a.x[j] = L.x0; a.h[j] = h_abs;
    a.flags[j] = (1u << IVP_BDF_ORDER_SHIFT) | (L.flags & IVP_F_FIRSTOUT);
    a.status[j] = IVP_RUNNING;
    return IVP_RUNNING;
}

// One pass of the main loop (bdf.rs:276-607). Returns false when the trajectory retired.
template <class R, int FULL>
IVP_HD bool bdf_attempt(const IvpKArgs &a, uint32_t j, BdfLane<R::N> &S, Lane<R::N, R::P> &L)
{
    KC_SCOPE_KZ(L.kz)
    constexpr int N = R::N;
    constexpr BdfTables T{};
    constexpr double EPS = 2.220446049250313e-16, MIN_POSITIVE = 2.2250738585072014e-308;
    constexpr int newton_maxiter = 4;
    int order = (int)((S.flags >> IVP_BDF_ORDER_SHIFT) & 7u);
    int n_equal = (int)((S.flags >> IVP_BDF_NEQ_SHIFT) & 7u);
    bool lu_current = (S.flags & IVP_BDF_LU_CURRENT) != 0;
    auto pack = [&]() {
        S.flags = (S.flags & ~((7u << IVP_BDF_ORDER_SHIFT) | (7u << IVP_BDF_NEQ_SHIFT) | IVP_BDF_LU_CURRENT)) |
                  ((uint32_t)order << IVP_BDF_ORDER_SHIFT) | ((uint32_t)n_equal << IVP_BDF_NEQ_SHIFT) |
                  (lu_current ? IVP_BDF_LU_CURRENT : 0u);
    };

    IVP_PHASE(0);   // everything since the previous attempt's last marker (its tail, the loop, the exits)
    if (S.over || S.d_nstep >= S.budget) { S.status = 2; return false; }                 // steps.total >= nmax
    if (S.current_h < MIN_POSITIVE) { S.status = 3; return false; }
    double h_try = S.current_h;
    double h_signed = 0.0, x_new = 0.0;
    bool finished = false;
    // The four D-rescalings that may precede the predictor (a pending one from the previous attempt, the h_max clamp,
    // the h_min clamp, the last-step clamp; bdf.rs:276-340).  Their conditions only involve scalars, so the scalar
    // bookkeeping runs first, straight-line, and the rescalings follow in the reference's order through ONE instance of
    // change_d (it is the fattest helper; a factor of exactly 1.0 is its own early exit).
    double fpass[4] = {1.0, 1.0, 1.0, 1.0};
    if (S.flags & IVP_BDF_PENDING) { fpass[0] = S.pending_factor; S.flags &= ~IVP_BDF_PENDING; }
    if (h_try > S.hmax) {
        fpass[1] = S.hmax / h_try;
        h_try = S.hmax; S.current_h = h_try; n_equal = 0; lu_current = false;
    }
    if (h_try < S.hmin && S.hmin > 0.0) {
        fpass[2] = fmax(S.hmin / h_try, 1.0);
        h_try = S.hmin; S.current_h = h_try; n_equal = 0; lu_current = false;
    }
    h_signed = S.direction * h_try;
    x_new = S.x + h_signed;
    if (S.direction * (x_new - S.xend) > 0.0) {
        const double step_to_end = fabs(S.xend - S.x);
        if (step_to_end == 0.0) { finished = true; }
        else {
            fpass[3] = step_to_end / h_try;
            S.current_h *= fpass[3];
            h_try = S.current_h;
            h_signed = S.direction * h_try;
            x_new = S.x + h_signed;
            n_equal = 0; lu_current = false;
        }
    }
    IVP_PHASE(1);   // step-size clamps
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const double factor = pass == 0 ? fpass[0] : (pass == 1 ? fpass[1] : (pass == 2 ? fpass[2] : fpass[3]));
        if (factor != 1.0) bdf_change_d<N>(S.d, order, factor);
    }
    IVP_PHASE(2);   // change_d
    if (finished) { pack(); S.status = 0; return false; }
    if ((S.x + KC(0.1) * fabs(h_signed)) == S.x) { pack(); S.status = 3; return false; }
    const double x_start = S.x;
    S.d_nstep += 1;

    double y_predict[N], scale[N], psi[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k <= order) sum += S.d[k][i];
        y_predict[i] = sum;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        scale[i] = IVP_MA(a.atol[i], a.rtol[i], fabs(y_predict[i]));
        if (scale[i] == 0.0) scale[i] = EPS;
    }
    const double alpha_o = bdf_sel6(T.alpha, order);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int jj = 1; jj < 6; ++jj) if (jj <= order) sacc = IVP_MA(sacc, T.gamma[jj], S.d[jj][i]);
        psi[i] = sacc / alpha_o;
    }
    const double c = h_signed / alpha_o;
    bool lu_failed = false;
    IVP_PHASE(3);   // predictor, scale, psi
    if (!lu_current || fabs(c - S.current_c) / fmax(fabs(c), 1.0) > KC(0.1)) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int ci = 0; ci < N; ++ci) S.lu[r][ci] = -c * S.jac[r][ci];
            S.lu[r][r] = IVP_MA(1.0, -c, S.jac[r][r]);   // (I - cJ)'s diagonal: -c j + 1
        }
        S.d_nlu += 1;
        if (bdf_lu_decomp<N>(S.lu, S.piv)) { lu_current = true; S.current_c = c; }
        else lu_failed = true;
    }
    if (lu_failed) {   // bdf.rs:373-381
        S.pending_factor = 0.5; S.flags |= IVP_BDF_PENDING;
        S.current_h *= 0.5; n_equal = 0; lu_current = false; S.d_nrejct += 1;
        pack();
        return true;
    }

    IVP_PHASE(4);   // LU refresh
    double y_new[N], delta[N], rhs[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { y_new[i] = y_predict[i]; delta[i] = 0.0; }
    bool converged = false, has_prev = false;
    double dy_norm_prev = 0.0;
    int iters = 0;
    double rtol_min = u2d(0x7FF0000000000000ull);
#pragma unroll
    for (int i = 0; i < N; ++i) rtol_min = fmin(rtol_min, a.rtol[i]);
    rtol_min = fmax(rtol_min, EPS);
    double newton_tol = fmax(10.0 * EPS / rtol_min, fmin(sqrt(rtol_min), 0.03));   // bdf.rs:174-185
    if (newton_tol <= 0.0) newton_tol = 1e-9;
    // bdf.rs:385-447.  One exit test per iteration: the contraction rate and both convergence estimates are evaluated
    // unconditionally (a division by zero or a NaN simply fails every comparison that would have been skipped) and the
    // reference's early exits become flags, in the reference's order of precedence.  The original nest of branches cost a
    // lone wave more in exec-mask bookkeeping and taken branches (~40 cycles each) than in arithmetic, and computed
    // dy_norm / dy_norm_prev twice.
    bool running = true;
#pragma unroll 1
    while (running) {
        R::ode(x_new, y_new, rhs, L.p);
        S.d_nfev += 1;
#pragma unroll
        for (int i = 0; i < N; ++i) rhs[i] = IVP_MB(c, rhs[i], psi[i]) - delta[i];
        bdf_lin_solve<N>(S.lu, rhs, S.piv);
        const double dy_norm = bdf_wrms<N>(rhs, scale);
        const bool have = has_prev && dy_norm_prev > 0.0;
        const double rate = dy_norm / dy_norm_prev;
        const double one_minus = 1.0 - rate;
        // rate.powf(remaining), remaining = 3, 2, 1 iterations left: the product (oracle: orc_pow_small_int)
        static_assert(newton_maxiter == 4, "the contraction-rate power is unrolled for 1..3 iterations left");
        const int remaining = newton_maxiter - iters;
        double rate_pow = rate;
        rate_pow = remaining >= 2 ? rate_pow * rate : rate_pow;
        rate_pow = remaining >= 3 ? rate_pow * rate : rate_pow;
        const double estimate_left = rate_pow / one_minus * dy_norm;   // error left after the remaining iterations
        const double estimate_now = rate / one_minus * dy_norm;
        const bool rate_condition = have && (rate >= 1.0 || estimate_left > newton_tol);
#pragma unroll
        for (int i = 0; i < N; ++i) { y_new[i] += rhs[i]; delta[i] += rhs[i]; }
        converged = dy_norm == 0.0 || (have && rate < 1.0 && estimate_now < newton_tol);
        const bool stop = converged || rate_condition;
        dy_norm_prev = dy_norm; has_prev = true;
        iters += stop ? 0 : 1;
        running = !stop && iters < newton_maxiter;
    }
    IVP_PHASE(5);   // Newton iterations
    if (!converged) {   // bdf.rs:448-459: refresh the Jacobian at the predictor, halve the step
        bdf_eval_jac<R>(x_new, y_predict, L.p, S.jac);
        S.d_njev += 1;
        lu_current = false;
        S.pending_factor = 0.5; S.flags |= IVP_BDF_PENDING;
        S.current_h *= 0.5; n_equal = 0; S.d_nrejct += 1;
        pack();
        return true;
    }
    const double safety = 0.9 * (2.0 * (double)newton_maxiter + 1.0) / (2.0 * (double)newton_maxiter + (double)(iters + 1));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        scale[i] = IVP_MA(a.atol[i], a.rtol[i], fabs(y_new[i]));
        if (scale[i] == 0.0) scale[i] = EPS;
    }
    const double ec_o = bdf_sel6(T.error_const, order);
#pragma unroll
    for (int i = 0; i < N; ++i) rhs[i] = ec_o * delta[i];
    const double error_norm = bdf_wrms<N>(rhs, scale);
    // One power site per attempt.  A rejected step needs error_norm^(-1/(order+1)) (bdf.rs:481-489), the order / step
    // adaptation of an accepted one needs that very value and its two neighbours err_m^(-1/order), err_p^(-1/(order+2))
    // (bdf.rs:551-606).  A wave executes whatever any of its lanes needs, and with 40 trajectories some lane rejects and
    // some lane adapts in nearly every attempt: both kinds of lane meet at ivp_pow3 below instead of running one power
    // for the rejection and three more, one after the other, for the adaptation.
    IVP_PHASE(6);   // error estimate
    const bool reject = error_norm > 1.0;
    bool adapt = false;
    double err_m = u2d(0x7FF0000000000000ull), err_p = u2d(0x7FF0000000000000ull);
    if (!reject) {
    S.d_naccpt += 1;
    n_equal += 1;
    S.x = x_new;
    double yold[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { yold[i] = S.y[i]; S.y[i] = y_new[i]; }
    // d[order+2] = delta - d[order+1]; d[order+1] = delta; d[k] += d[k+1] for k = order..0
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int k = 2; k < 8; ++k) {
            if (k == order + 2) S.d[k][i] = delta[i] - S.d[k - 1][i];
        }
#pragma unroll
        for (int k = 1; k < 7; ++k) {
            if (k == order + 1) S.d[k][i] = delta[i];
        }
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            if (k <= order) S.d[k][i] += S.d[k + 1][i];
        }
    }
    if (FULL) {
        double cont[7 * N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cont[i * 7] = S.d[0][i];
#pragma unroll
            for (int k = 0; k < 5; ++k) cont[i * 7 + 1 + k] = (k + 1 <= order) ? S.d[k + 1][i] : 0.0;
            cont[i * 7 + 6] = (double)order;
        }
        L.x0 = S.x0;
        // bdf.rs:518-519: the interpolant is anchored at x_start, the callback's xold argument is x - h_signed
        if (solout_full<M_BDF, R>(a, j, L, S.x - h_signed, S.x, S.y, yold, cont, h_signed, x_start)) { pack(); S.status = 1; return false; }   // bdf.rs:521-524
    }
    if (S.direction * (S.x - S.xend) >= 0.0) { pack(); S.status = 0; return false; }

    IVP_PHASE(7);   // accepted step: D update, output
    if (n_equal >= order + 1) {   // order / step adaptation, bdf.rs:551-606
        adapt = true;
        if (order > 1) {
            const double ecm = bdf_sel6(T.error_const, order - 1);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double dv = S.d[1][i];
                IVP_OPAQUE_V(dv);
#pragma unroll
                for (int k = 2; k < 6; ++k) { double dk = S.d[k][i]; IVP_OPAQUE_V(dk); dv = (k == order) ? dk : dv; }
                rhs[i] = ecm * dv;
            }
            err_m = bdf_wrms<N>(rhs, scale);
        }
        if (order < BDF_MAXO) {
            const double ecp = bdf_sel6(T.error_const, order + 1);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double dv = S.d[3][i];
                IVP_OPAQUE_V(dv);
#pragma unroll
                for (int k = 4; k < 8; ++k) { double dk = S.d[k][i]; IVP_OPAQUE_V(dk); dv = (k == order + 2) ? dk : dv; }
                rhs[i] = ecp * dv;
            }
            err_p = bdf_wrms<N>(rhs, scale);
        }
    }
    }   // !reject
    IVP_PHASE(8);   // neighbouring-order error norms
    if (reject || adapt) {
        double factors[3];
        const double errors[3] = {err_m, error_norm, err_p};
        const double expo[3] = {-1.0 / ((double)order + 0.0), -1.0 / ((double)order + 1.0), -1.0 / ((double)order + 2.0)};
        ivp_pow3(errors, expo, factors, IVP_KZ_ARG);
        IVP_PHASE(9);   // the three powers
        if (reject) {   // bdf.rs:481-489
            double factor = safety * factors[1];
            factor = fmax(factor, 0.2);
            S.pending_factor = factor; S.flags |= IVP_BDF_PENDING;
            S.current_h *= factor; n_equal = 0; S.d_nrejct += 1;
            pack();
            return true;
        }
        int best = 0;   // Iterator::max_by keeps a later element unless the current maximum is strictly greater
        double bestv = factors[0];
        if (!(bestv > factors[1])) { best = 1; bestv = factors[1]; }
        if (!(bestv > factors[2])) { best = 2; bestv = factors[2]; }
        int new_order = order;
        if (best == 0 && order > 1) new_order -= 1;
        else if (best == 2 && order < BDF_MAXO) new_order += 1;
        double max_factor = fmax(fmax(fmax(0.0, factors[0]), factors[1]), factors[2]);
        const double step_factor = fmin(safety * max_factor, 10.0);
        const int old_order = order;
        S.pending_factor = step_factor; S.flags |= IVP_BDF_PENDING;   // change_d(d, new_order, step_factor)
        S.current_h *= step_factor;
        order = new_order;
        n_equal = 0;
        lu_current = false;
        if (new_order != old_order) { bdf_eval_jac<R>(S.x, S.y, L.p, S.jac); S.d_njev += 1; }
    }
    pack();
    return true;
}

template <class R, int FULL>
IVP_HD uint32_t bdf_chunk_body(const IvpKArgs &a, uint32_t j, int32_t &status_out)
{
    constexpr int N = R::N, P = R::P;
    const size_t B = a.B;
    BdfLane<N> S;
    Lane<N, P> L;
#pragma unroll
    for (int c = 0; c < N; ++c) S.y[c] = a.y[c * B + j];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < N; ++c) S.d[k][c] = a.bdf_d[(size_t)(k * N + c) * B + j];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) { S.jac[r][c] = a.bdf_jac[(size_t)(r * N + c) * B + j]; S.lu[r][c] = a.bdf_lu[(size_t)(r * N + c) * B + j]; }
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    S.x = a.x[j];
    S.current_h = a.h[j];
    S.current_c = a.facold[j];
    S.pending_factor = a.hlamb[j];
    S.piv = a.bdf_piv[j];
    S.flags = a.flags[j];
    S.x0 = a.t0[(size_t)j * a.t0_stride];
    S.xend = a.t1[(size_t)j * a.t1_stride];
    S.direction = rs_signum(S.xend - S.x0);
    S.hmax = fabs(a.has_max_step ? a.max_step : fabs(S.xend - S.x0));
    S.hmin = fabs(a.has_min_step ? a.min_step : 0.0);
    S.status = IVP_RUNNING;
    S.d_nfev = S.d_njev = S.d_nlu = S.d_nstep = S.d_naccpt = S.d_nrejct = 0;
    const uint64_t nstep0 = a.nstep[j];
    S.over = nstep0 >= a.nmax;
    const uint64_t left = S.over ? 0 : a.nmax - nstep0;
    S.budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    L.flags = S.flags & IVP_F_FIRSTOUT;
    L.x0 = S.x0;
#if defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 2
    L.kz = ivp_opaque_zero_v();   // pinned-coefficient build: see KC() in rk_core.h
#else
    L.kz = 0;
#endif
    if (FULL) {
        L.next_idx = a.next_idx[j]; L.n_filled = a.n_filled[j]; L.n_log = a.n_log[j];
        L.n_seg = a.n_seg[j]; L.t_last = a.t_last[j];
    } else {
        L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    }
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0;
    uint32_t it = 0;
    bool run = true;
    IVP_PHASE_BEGIN();
    while (run && it < a.chunk) {
        if (FULL && a.log_pool != nullptr) so_log_attempt<IdMap<N>, 1>(a, j, L, it);
        run = bdf_attempt<R, FULL>(a, j, S, L);
        ++it;
    }
    IVP_PHASE_END();
    if (FULL && a.log_pool != nullptr) so_log_flush<IdMap<N>>(a, L);
    uint32_t js = j;
    IVP_OPAQUE_V(js);
#pragma unroll
    for (int c = 0; c < N; ++c) a.y[c * B + js] = S.y[c];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < N; ++c) a.bdf_d[(size_t)(k * N + c) * B + js] = S.d[k][c];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) { a.bdf_jac[(size_t)(r * N + c) * B + js] = S.jac[r][c]; a.bdf_lu[(size_t)(r * N + c) * B + js] = S.lu[r][c]; }
    a.x[js] = S.x;
    // IntegrationResult.h = direction * current_h (bdf.rs:609-614); kept unsigned while the trajectory runs
    a.h[js] = S.status == IVP_RUNNING ? S.current_h : S.direction * S.current_h;
    a.facold[js] = S.current_c;
    a.hlamb[js] = S.pending_factor;
    a.bdf_piv[js] = S.piv;
    a.flags[js] = (S.flags & ~IVP_F_FIRSTOUT) | (L.flags & IVP_F_FIRSTOUT);
    a.status[js] = S.status;
    a.nfev[js] += S.d_nfev; a.njev[js] += S.d_njev; a.nlu[js] += S.d_nlu;
    a.nstep[js] += S.d_nstep; a.naccpt[js] += S.d_naccpt; a.nrejct[js] += S.d_nrejct;
    if (FULL) {
        a.next_idx[js] = L.next_idx; a.n_filled[js] = L.n_filled; a.n_log[js] = L.n_log;
        a.n_seg[js] = L.n_seg; a.t_last[js] = L.t_last;
    }
    status_out = S.status;
    return it;
}

// method dispatch used by the kernels (rk_global.h) and the CPU emulation harness
template <int M, class R, int FULL>
IVP_HD int32_t any_init_body(const IvpKArgs &a, uint32_t j)
{
    if constexpr (M == M_BDF) return bdf_init_body<R, FULL>(a, j);
    else return init_body<M, R, FULL>(a, j);
}
template <int M, class R, int FULL, bool CTL = false>
IVP_HD uint32_t any_chunk_body(const IvpKArgs &a, uint32_t j, int32_t &status_out)
{
    if constexpr (M == M_BDF) return bdf_chunk_body<R, FULL>(a, j, status_out);
    else return chunk_body<M, R, FULL, CTL>(a, j, status_out);
}

}  // namespace IVP_NS
