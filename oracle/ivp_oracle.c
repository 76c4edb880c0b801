/*
 * ivp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See ivp_oracle.h.
 *
 * Scalar, one-trajectory-per-call restatement of the reference's explicit RK path.
 * Build with -ffp-contract=off: the reference (Rust) never contracts a*b+c into an FMA, and
 * every expression below keeps the reference's left-to-right association so that the
 * arithmetic is the same IEEE-754 operation sequence.
 *
 * Third build, -DORC_FMA -DORC_DETPOW (liboracle_fma.so): the checker of the kernels' FMA arithmetic mode
 * (ivp_options_t.fp_mode = IVP_FP_FMA).  That mode is a defined arithmetic: the multiply-add sites marked MA / MS / MB /
 * LCn below -- stage combinations, error estimates, dense coefficients, interpolants, tolerance scales, the built-in
 * right-hand sides (which there also share one reciprocal per primary) -- are single fused operations, everything else
 * is unchanged, and norms of systems with n > 8 are summed in the wave-per-trajectory kernels' order (orc_sum).  The
 * same sites are fused in ivp_amd/csrc/rk_core.h (IVP_MA / IVP_MS / IVP_MB / IVP_LC), still without compiler contraction.
 *
 * Citations are file:line in the reference tree.
 */
#include "ivp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Step-controller power function
 * ---------------------------------------------------------------------------------------- */

/* the multiply-add sites: two IEEE operations in the reference's association, or one fused operation (ORC_FMA) */
#ifdef ORC_FMA
#define MA(acc, a, b) fma((a), (b), (acc))     /* acc + a * b */
#define MS(acc, a, b) fma(-(a), (b), (acc))    /* acc - a * b */
#define MB(a, b, c) fma((a), (b), -(c))        /* a * b - c   */
#else
#define MA(acc, a, b) ((acc) + (a) * (b))
#define MS(acc, a, b) ((acc) - (a) * (b))
#define MB(a, b, c) ((a) * (b) - (c))
#endif
/* ((c1 k1 + c2 k2) + c3 k3) + ... : the reference's left-to-right sums of products */
#define LC2(c1, k1, c2, k2) MA((c1) * (k1), c2, k2)
#define LC3(c1, k1, c2, k2, c3, k3) MA(LC2(c1, k1, c2, k2), c3, k3)
#define LC4(c1, k1, c2, k2, c3, k3, c4, k4) MA(LC3(c1, k1, c2, k2, c3, k3), c4, k4)
#define LC5(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5) MA(LC4(c1, k1, c2, k2, c3, k3, c4, k4), c5, k5)
#define LC6(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6) MA(LC5(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5), c6, k6)
#define LC7(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6, c7, k7) MA(LC6(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6), c7, k7)
#define LC8(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6, c7, k7, c8, k8) \
    MA(LC7(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6, c7, k7), c8, k8)
#define LC9(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6, c7, k7, c8, k8, c9, k9) \
    MA(LC8(c1, k1, c2, k2, c3, k3, c4, k4, c5, k5, c6, k6, c7, k7, c8, k8), c9, k9)

/* Sum of the n terms of a weighted norm.  Reference: left to right (dopri5.rs:343-347).  ORC_FMA build, n > 8: the
 * order of the wave-per-trajectory kernels' FMA build (ivp_amd/csrc/rk_group.h, NormOps<GroupRhs>::sum): lane l of a
 * group of G lanes adds its terms l, l + G, ... starting from 0.0, then log2 G butterfly steps
 * part += shfl_xor(part, o), o = G/2 .. 1; G = 16 / 32 / 64 for n <= 16 / <= 32 / larger (ivp_group_width). */
static double orc_sum(const double *t, int n)
{
#ifdef ORC_FMA
    if (n > 8) {
        const int G = n <= 16 ? 16 : (n <= 32 ? 32 : 64);
        double part[64], nxt[64];
        for (int l = 0; l < G; l++) {
            part[l] = 0.0;
            for (int i = l; i < n; i += G) part[l] += t[i];
        }
        for (int o = G / 2; o > 0; o >>= 1) {
            for (int l = 0; l < G; l++) nxt[l] = part[l] + part[l ^ o];
            memcpy(part, nxt, sizeof(double) * (size_t)G);
        }
        return part[0];
    }
#endif
    double s = 0.0;
    for (int i = 0; i < n; i++) s += t[i];
    return s;
}

static inline uint64_t d2bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double bits2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* Portable exp2(e*log2(x)) for x >= 0: only IEEE +,-,*,/, fma, rint and bit moves, so a device
 * restatement of the same operation sequence gives the same bits.  A few ulp accurate, which a
 * step-size factor does not notice. */
double orc_detpow(double x, double e)
{
    if (e == 0.0) return 1.0;
    if (x != x || e != e) return x + e;
    if (x < 0.0) return NAN;
    if (x == 0.0) return e > 0.0 ? 0.0 : INFINITY;
    if (x == INFINITY) return e > 0.0 ? INFINITY : 0.0;

    int k = 0;
    uint64_t u = d2bits(x);
    if ((u >> 52) == 0) { /* subnormal */
        x *= 0x1p54;
        u = d2bits(x);
        k = -54;
    }
    int ex = (int)(u >> 52) - 1023;
    uint64_t mant = u & 0x000FFFFFFFFFFFFFull;
    double m;
    if (mant > 0x6A09E667F3BCDull) { /* m > sqrt(2): use m/2 in [sqrt(1/2), 1) */
        m = bits2d(mant | 0x3FE0000000000000ull);
        ex += 1;
    } else {
        m = bits2d(mant | 0x3FF0000000000000ull);
    }
    k += ex;

    /* ln(m) = 2 atanh(t), t = (m-1)/(m+1), |t| <= 0.1716 */
    double t = (m - 1.0) / (m + 1.0);
    double z = t * t;
    double p = 1.0 / 25.0;
    p = fma(p, z, 1.0 / 23.0);
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    double lnm = (2.0 * t) * p;
    double l2 = fma(lnm, 0x1.71547652b82fep+0 /* 1/ln2 */, (double)k);
    double w = e * l2;

    if (w >= 1024.0) return INFINITY;
    if (w <= -1022.0) return 0.0;
    double kd = rint(w);
    double r = w - kd;
    double v = r * 0x1.62e42fefa39efp-1; /* ln2 */
    double q = 1.0 / 87178291200.0; /* 1/14! */
    q = fma(q, v, 1.0 / 6227020800.0);
    q = fma(q, v, 1.0 / 479001600.0);
    q = fma(q, v, 1.0 / 39916800.0);
    q = fma(q, v, 1.0 / 3628800.0);
    q = fma(q, v, 1.0 / 362880.0);
    q = fma(q, v, 1.0 / 40320.0);
    q = fma(q, v, 1.0 / 5040.0);
    q = fma(q, v, 1.0 / 720.0);
    q = fma(q, v, 1.0 / 120.0);
    q = fma(q, v, 1.0 / 24.0);
    q = fma(q, v, 1.0 / 6.0);
    q = fma(q, v, 0.5);
    q = fma(q, v, 1.0);
    q = fma(q, v, 1.0);
    int ki = (int)kd;
    double scale = bits2d((uint64_t)(ki + 1023) << 52);
    return q * scale;
}

#ifdef ORC_DETPOW
#define ORC_POW(x, e) orc_detpow((x), (e))
int orc_uses_detpow(void) { return 1; }
/* powf with a small positive INTEGER exponent (the contraction-rate estimate of BDF's Newton loop, bdf.rs:408: the
 * exponent is the number of iterations left, 1..3): the portable stand-in is the product, which is what a correctly
 * rounded pow returns for n = 1, 2 and within an ulp of it for n = 3 -- closer to libm than the 2-ulp exp2/log2 form. */
static inline double orc_pow_small_int(double x, int n)
{
    if (n == 1) return x;
    if (n == 2) return x * x;
    if (n == 3) return x * x * x;
    return orc_detpow(x, (double)n);
}
#else
#define ORC_POW(x, e) pow((x), (e))
int orc_uses_detpow(void) { return 0; }
static inline double orc_pow_small_int(double x, int n) { return pow(x, (double)n); }   /* the reference's own call */
#endif

#ifdef ORC_FMA
int orc_uses_fma(void) { return 1; }
#else
int orc_uses_fma(void) { return 0; }
#endif

/* Rust f64::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN. */
static inline double rs_signum(double v) { return v != v ? v : copysign(1.0, v); }

/* ------------------------------------------------------------------------------------------
 * Built-in right-hand sides
 * ---------------------------------------------------------------------------------------- */

static void rhs_decay(double x, const double *y, double *d, const double *p)
{   /* examples/exponential_decay.rs:11 */
    (void)x;
    d[0] = -p[0] * y[0];
}
static void rhs_sho(double x, const double *y, double *d, const double *p)
{   /* tests/common.rs:5-8 */
    (void)x; (void)p;
    d[0] = y[1];
    d[1] = -y[0];
}
static void rhs_vdp(double x, const double *y, double *d, const double *p)
{   /* benches/benchmark.py:22-27 */
    (void)x;
    double mu = p[0];
    d[0] = y[1];
    d[1] = MB(mu * MS(1.0, y[0], y[0]), y[1], y[0]);   /* mu * (1 - y0 * y0) * y1 - y0 */
}
static void rhs_cr3bp(double t, const double *s, double *d, const double *p)
{   /* examples/cr3bp.rs:24-35 */
    (void)t;
    double mu = p[0];
    double x = s[0], y = s[1], z = s[2], vx = s[3], vy = s[4], vz = s[5];
    double a = x + mu;
    double b = x - 1.0 + mu;
#ifdef ORC_FMA
    /* FMA form (RhsCr3bp::ode under IVP_FAST in rk_core.h): one division per primary, every a * b + c fused */
    double d1 = fma(z, z, fma(y, y, a * a));
    double d2 = fma(z, z, fma(y, y, b * b));
    double r1 = sqrt(d1), r2 = sqrt(d2);
    double g1 = (1.0 - mu) / (d1 * r1);
    double g2 = mu / (d2 * r2);
    d[0] = vx;
    d[1] = vy;
    d[2] = vz;
    d[3] = fma(-g2, b, fma(-g1, a, fma(2.0, vy, x)));
    d[4] = fma(-g2, y, fma(-g1, y, fma(-2.0, vx, y)));
    d[5] = fma(-g2, z, fma(-g1, z, -0.0));
#else
    double r1 = sqrt(a * a + y * y + z * z);
    double r2 = sqrt(b * b + y * y + z * z);
    double r13 = r1 * r1 * r1; /* powi(3) */
    double r23 = r2 * r2 * r2;
    d[0] = vx;
    d[1] = vy;
    d[2] = vz;
    d[3] = x + 2.0 * vy - (1.0 - mu) * (x + mu) / r13 - mu * (x - 1.0 + mu) / r23;
    d[4] = y - 2.0 * vx - (1.0 - mu) * y / r13 - mu * y / r23;
    d[5] = -(1.0 - mu) * z / r13 - mu * z / r23;
#endif
}
static void rhs_lorenz(double t, const double *s, double *d, const double *p)
{   /* benches/benchmark.py:30-37 */
    (void)t;
    double sigma = p[0], rho = p[1], beta = p[2];
    double x = s[0], y = s[1], z = s[2];
    d[0] = sigma * (y - x);
    d[1] = MB(x, rho - z, y);        /* x * (rho - z) - y */
    d[2] = MS(x * y, beta, z);       /* x * y - beta * z */
}
static void rhs_zero(double t, const double *s, double *d, const double *p)
{   /* tests/ivp.rs:14-18 */
    (void)t; (void)s; (void)p;
    d[0] = 0.0; d[1] = 0.0; d[2] = 0.0;
}
static void rhs_rational(double t, const double *y, double *d, const double *p)
{   /* tests/test_helpers.py:23-25 */
    (void)p;
    d[0] = y[1] / t;
    d[1] = y[1] * (MA(y[0], 2.0, y[1]) - 1.0) / (t * (y[0] - 1.0));
}
static void rhs_exp2(double t, const double *y, double *d, const double *p)
{   /* tests/ivp.rs:293-297 */
    (void)t; (void)p;
    d[0] = y[0];
    d[1] = y[1];
}

static void rhs_linear(double t, const double *y, double *d, const double *p)
{   /* tests/test_helpers.py:11-12 */
    (void)t; (void)p;
    d[0] = MS(-y[0], 5.0, y[1]);
    d[1] = y[0] + y[1];
}
static void rhs_robertson(double t, const double *s, double *d, const double *p)
{   /* tests/test_ivp.py:327-333 */
    (void)t; (void)p;
    double x = s[0], y = s[1], z = s[2];
    d[0] = MA(-0.04 * x, 1e4 * y, z);                  /* -0.04 x + 1e4 y z */
    d[1] = MS(MS(0.04 * x, 1e4 * y, z), 3e7 * y, y);   /* 0.04 x - 1e4 y z - 3e7 y y */
    d[2] = 3e7 * y * y;
}
static void rhs_vdp_eps(double t, const double *y, double *d, const double *p)
{   /* examples/van_der_pol.rs:9-14 */
    (void)t;
    d[0] = y[1];
    d[1] = MB(MS(1.0, y[0], y[0]), y[1], y[0]) / p[0];   /* ((1 - y0 y0) y1 - y0) / eps */
}

static void rhs_ball(double t, const double *s, double *d, const double *p)
{   /* examples/bouncing_ball.rs:10-15  p = {gravity, drag} */
    (void)t;
    double vy = s[1];
    d[0] = vy;
    d[1] = MS(-p[0], p[1] * vy, fabs(vy));   /* -g - drag vy |vy| */
}
static void rhs_cannon(double t, const double *y, double *d, const double *p)
{   /* tests/test_ivp.py:153-154 */
    (void)t; (void)p;
    d[0] = y[1];
    d[1] = -9.80665;
}
static void rhs_linear_decay100(double t, const double *y, double *d, const double *p)
{   /* benches/benchmark.py:40-42,139-148: y' = -y, N = 100 */
    (void)t; (void)p;
    for (int i = 0; i < 100; i++) d[i] = -y[i];
}
static void rhs_heat1d256(double t, const double *y, double *d, const double *p)
{   /* method-of-lines heat equation, Dirichlet ends (no reference counterpart: a coupled large-n case) */
    (void)t;
    for (int i = 0; i < 256; i++) {
        double left = i > 0 ? y[i - 1] : 0.0;
        double right = i < 255 ? y[i + 1] : 0.0;
        d[i] = p[0] * (MS(left, 2.0, y[i]) + right);
    }
}
static void rhs_dense64(double t, const double *y, double *d, const double *p)
{   /* y' = A y: a_ii = -k (4 + i mod 5), a_ij = (((5 i + 3 j) & 15) - 8) / 256 (kernel: RhsDense64 in rk_group.h) */
    (void)t;
    for (int i = 0; i < 64; i++) {
        double s = -p[0] * (4.0 + (double)(i % 5)) * y[i];
        for (int j = 0; j < 64; j++) {
            double aij = (double)(((i * 5 + j * 3) & 15) - 8) * 0.00390625;
            if (j != i) s = MA(s, aij, y[j]);
        }
        d[i] = s;
    }
}
/* event functions: trait IVP::events (src/ivp.rs:31-40) */
static void ev_y0(double x, const double *y, double *g, const double *p)
{   /* tests/ivp.rs:157-159, examples/bouncing_ball.rs:17-19, tests/test_ivp.py:156-157 */
    (void)x; (void)p;
    g[0] = y[0];
}
static void ev_rational3(double t, const double *y, double *g, const double *p)
{   /* tests/test_ivp.py:346-353 */
    (void)p;
    g[0] = y[0] - pow(y[1], 0.7);
    g[1] = pow(y[1], 0.6) - y[0];
    g[2] = t - 7.4;
}
orc_event_fn orc_builtin_events(int rhs_id, int *n_events)
{
    switch (rhs_id) {
    case ORC_RHS_SHO_EV: case ORC_RHS_BALL: case ORC_RHS_CANNON: *n_events = 1; return ev_y0;
    case ORC_RHS_RATIONAL_EV: *n_events = 3; return ev_rational3;
    default: *n_events = 0; return NULL;
    }
}

/* `impl IVP { fn jac }` of the Robertson problem: the analytic Jacobian of rhs_robertson */
static void jac_robertson(double t, const double *s, double *j, const double *p)
{
    (void)t; (void)p;
    const double y = s[1], z = s[2];
    j[0] = -0.04;  j[1] = 1e4 * z;              j[2] = 1e4 * y;
    j[3] = 0.04;   j[4] = MS(-1e4 * z, 6e7, y);   j[5] = -1e4 * y;
    j[6] = 0.0;    j[7] = 6e7 * y;              j[8] = 0.0;
}
orc_jac_fn orc_builtin_jac(int rhs_id) { return rhs_id == ORC_RHS_ROBERTSON_JAC ? jac_robertson : NULL; }

orc_ode_fn orc_builtin_rhs(int rhs_id, int *n_out, int *np_out)
{
    static const struct { orc_ode_fn f; int n, np; } tab[ORC_RHS_COUNT] = {
        {rhs_decay, 1, 1}, {rhs_sho, 2, 0}, {rhs_vdp, 2, 1}, {rhs_cr3bp, 6, 1},
        {rhs_lorenz, 3, 3}, {rhs_zero, 3, 0}, {rhs_rational, 2, 0}, {rhs_exp2, 2, 0},
        {rhs_linear, 2, 0}, {rhs_robertson, 3, 0}, {rhs_vdp_eps, 2, 1},
        {rhs_sho, 2, 0}, {rhs_ball, 2, 2}, {rhs_cannon, 2, 0}, {rhs_rational, 2, 0}, {rhs_robertson, 3, 0},
    };
    if (rhs_id == ORC_RHS_LINEAR_DECAY_100) { if (n_out) *n_out = 100; if (np_out) *np_out = 0; return rhs_linear_decay100; }
    if (rhs_id == ORC_RHS_HEAT1D_256) { if (n_out) *n_out = 256; if (np_out) *np_out = 1; return rhs_heat1d256; }
    if (rhs_id == ORC_RHS_DENSE_64) { if (n_out) *n_out = 64; if (np_out) *np_out = 1; return rhs_dense64; }
    if (rhs_id < 0 || rhs_id >= ORC_RHS_COUNT) return NULL;
    if (n_out) *n_out = tab[rhs_id].n;
    if (np_out) *np_out = tab[rhs_id].np;
    return tab[rhs_id].f;
}

/* ------------------------------------------------------------------------------------------
 * Tolerance (src/methods/mod.rs:104-214): Index returns the scalar for any i.
 * ---------------------------------------------------------------------------------------- */
typedef struct { const double *v; int len; } tol_t;
static inline double tol_at(const tol_t *t, int i) { return t->len == 1 ? t->v[0] : t->v[i]; }

/* ------------------------------------------------------------------------------------------
 * Dense interpolants
 * ---------------------------------------------------------------------------------------- */
static void interp_dopri5(double xi, double *yi, const double *cont, int n, double xold, double h)
{   /* dopri5.rs:467-478 */
    double theta = (xi - xold) / h;
    double theta1 = 1.0 - theta;
    for (int i = 0; i < n; i++) {
        yi[i] = MA(cont[i], theta, MA(cont[n + i], theta1, MA(cont[2 * n + i], theta, MA(cont[3 * n + i], theta1, cont[4 * n + i]))));
    }
}
static void interp_dop853(double xi, double *yi, const double *cont, int n, double xold, double h)
{   /* dop853.rs:659-670 */
    double s = (xi - xold) / h;
    double s1 = 1.0 - s;
    for (int i = 0; i < n; i++) {
        double conpar = MA(cont[4 * n + i], s, MA(cont[5 * n + i], s1, MA(cont[6 * n + i], s, cont[7 * n + i])));
        yi[i] = MA(cont[i], s, MA(cont[n + i], s1, MA(cont[2 * n + i], s, MA(cont[3 * n + i], s1, conpar))));
    }
}
static void interp_rk23(double xi, double *yi, const double *cont, int n, double xold, double h)
{   /* rk23.rs:313-321 */
    double xc = (xi - xold) / h;
    double x2 = xc * xc;
    double x3 = x2 * xc;
    for (int i = 0; i < n; i++) {
        yi[i] = MA(cont[i], h, LC3(cont[n + i], xc, cont[2 * n + i], x2, cont[3 * n + i], x3));
    }
}
static void interp_rk4(double xi, double *yi, const double *cont, int n, double xold, double h)
{   /* rk4.rs:229-244: cubic Hermite on (y_old, k4, k1_new, y_new) as stored at rk4.rs:186-190 */
    double t = (xi - xold) / h;
    double t2 = t * t;
    double t3 = t2 * t;
    double h00 = 2.0 * t3 - 3.0 * t2 + 1.0;
    double h10 = t3 - 2.0 * t2 + t;
    double h01 = -2.0 * t3 + 3.0 * t2;
    double h11 = t3 - t2;
    for (int i = 0; i < n; i++)
        yi[i] = LC4(h00, cont[i], h10 * h, cont[n + i], h01, cont[3 * n + i], h11 * h, cont[2 * n + i]);
}
static void interp_bdf(double xi, double *yi, const double *cont, int n, double xold, double h)
{   /* bdf.rs:618-656; cont is per-state blocks [D0, D1..D5, order] */
    if (h == 0.0 || n == 0) return;
    double ordf = round(cont[6]);
    if (ordf < 1.0) ordf = 1.0;
    if (ordf > 5.0) ordf = 5.0;
    int order = (int)ordf;
    double x_new = xold + h;
    double p[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < order; k++) {
        double denom = h * ((double)k + 1.0);
        double t_shift = x_new - h * (double)k;
        double xf = (xi - t_shift) / denom;
        p[k] = k == 0 ? xf : p[k - 1] * xf;
    }
    for (int i = 0; i < n; i++) {
        const double *b = cont + (size_t)i * 7;
        double sum = b[0];
        for (int k = 0; k < order; k++) sum += b[1 + k] * p[k];
        yi[i] = sum;
    }
}
static int ncoef_of(int method)
{   /* options.rs:34-43 */
    return method == ORC_DOPRI5 ? 5 : method == ORC_DOP853 ? 8 : method == ORC_BDF ? 7 : 4;
}
static void interp_any(int method, double xi, double *yi, const double *cont, int n, double xold, double h)
{
    if (method == ORC_DOPRI5) interp_dopri5(xi, yi, cont, n, xold, h);
    else if (method == ORC_DOP853) interp_dop853(xi, yi, cont, n, xold, h);
    else if (method == ORC_RK4) interp_rk4(xi, yi, cont, n, xold, h);
    else if (method == ORC_BDF) interp_bdf(xi, yi, cont, n, xold, h);
    else interp_rk23(xi, yi, cont, n, xold, h);
}

/* ------------------------------------------------------------------------------------------
 * DefaultSolOut (src/solve/solout.rs) without events: dense collection, t_eval sampling,
 * accepted-step recording with first_step enforcement.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int method, n;
    const double *t_eval; int n_eval; /* n_eval < 0: None */
    size_t next_idx;
    double tol;
    size_t len, cap; double *t; double *y;
    int collect_dense;
    size_t nseg, segcap; double *seg_cont, *seg_xold, *seg_h;
    int has_first_step; double first_step;
    double x0;
    int first_output_done;
    /* events (solout.rs:158-331) */
    int n_events;
    orc_event_fn ev;
    const double *ev_params;
    int ev_direction[ORC_MAX_EVENTS];       /* 0 All, >0 Positive, <0 Negative (event.rs:59-77) */
    uint64_t ev_terminal[ORC_MAX_EVENTS];   /* terminal_count, 0 = None */
    double prev_event[ORC_MAX_EVENTS];
    uint64_t event_hits[ORC_MAX_EVENTS];
    int has_yold;
    double yold[ORC_MAX_N];
    size_t ev_len[ORC_MAX_EVENTS], ev_cap[ORC_MAX_EVENTS];
    double *t_events[ORC_MAX_EVENTS];
    double *y_events[ORC_MAX_EVENTS];
} solout_t;

static void so_push(solout_t *s, double t, const double *y)
{
    if (s->len == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 64;
        s->t = (double *)realloc(s->t, s->cap * sizeof(double));
        s->y = (double *)realloc(s->y, s->cap * (size_t)s->n * sizeof(double));
    }
    s->t[s->len] = t;
    memcpy(s->y + s->len * (size_t)s->n, y, (size_t)s->n * sizeof(double));
    s->len++;
}

/* Returns 0 = Continue (the only flag DefaultSolOut produces without events). */
static int so_call2(solout_t *s, double xold, double x, const double *y,
                    const double *cont /* NULL = no interpolant */, double ixold, double h);
static int so_call(solout_t *s, double xold, double x, const double *y, const double *cont, double h)
{
    return so_call2(s, xold, x, y, cont, xold, h);
}
/* `xold` is what the integrator passes as the callback's first argument; `ixold`/`h` are the interpolant's own
 * anchor (StepInterpolant.xold, .h): identical for the RK methods, different for BDF (bdf.rs:518-519). */
static int so_call2(solout_t *s, double xold, double x, const double *y,
                    const double *cont /* NULL = no interpolant */, double ixold, double h)
{
    int n = s->n;
    /* solout.rs:141-146 */
    if (s->collect_dense && x != xold && cont) {
        if (h != 0.0) {
            int nc = ncoef_of(s->method) * n;
            if (s->nseg == s->segcap) {
                s->segcap = s->segcap ? 2 * s->segcap : 64;
                s->seg_cont = (double *)realloc(s->seg_cont, s->segcap * (size_t)nc * sizeof(double));
                s->seg_xold = (double *)realloc(s->seg_xold, s->segcap * sizeof(double));
                s->seg_h = (double *)realloc(s->seg_h, s->segcap * sizeof(double));
            }
            memcpy(s->seg_cont + s->nseg * (size_t)nc, cont, (size_t)nc * sizeof(double));
            s->seg_xold[s->nseg] = ixold;
            s->seg_h[s->nseg] = h;
            s->nseg++;
        }
    }

    /* Event detection, solout.rs:158-331 */
    if (s->n_events > 0) {
        double g_curr[ORC_MAX_EVENTS];
        s->ev(x, y, g_curr, s->ev_params);
        if (!s->has_yold) {
            memcpy(s->prev_event, g_curr, sizeof(double) * (size_t)s->n_events);
        } else {
            double det_t[ORC_MAX_EVENTS], det_y[ORC_MAX_EVENTS][ORC_MAX_N];
            int det_i[ORC_MAX_EVENTS], ndet = 0;
            for (int i = 0; i < s->n_events; i++) {
                double g_prev = s->prev_event[i], g_cur = g_curr[i];
                int dir = s->ev_direction[i], crossed;
                if (dir == 0) crossed = (g_prev <= 0.0 && g_cur >= 0.0) || (g_prev >= 0.0 && g_cur <= 0.0);
                else if (dir > 0) crossed = g_prev < 0.0 && g_cur >= 0.0;
                else crossed = g_prev > 0.0 && g_cur <= 0.0;
                if (!crossed) continue;
                const double XTOL = 2e-12, RTOL = 2.220446049250313e-16;
                double a = xold, b = x, fa = g_prev, fb = g_cur;
                double ymid[ORC_MAX_N], gmid[ORC_MAX_EVENTS];
                if (fabs(fa) <= XTOL) { det_t[ndet] = a; memcpy(det_y[ndet], s->yold, sizeof(double) * (size_t)n); }
                else if (fabs(fb) <= XTOL) { det_t[ndet] = b; memcpy(det_y[ndet], y, sizeof(double) * (size_t)n); }
                else {   /* Brent's method (matches SciPy's brentq), solout.rs:204-291 */
                    double c = a, fc = fa, d = b - a, e = d;
                    for (int it = 0; it < 100; it++) {
                        if (fb * fc > 0.0) { c = a; fc = fa; d = b - a; e = d; }
                        if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
                        double tol1 = 2.0 * RTOL * fabs(b) + 0.5 * XTOL;
                        double xm = 0.5 * (c - b);
                        if (fabs(xm) <= tol1 || fb == 0.0) break;
                        if (fabs(e) >= tol1 && fabs(fa) > fabs(fb)) {
                            double sq, pp, qq;
                            if (a == c) {
                                sq = fb / fa;
                                pp = 2.0 * xm * sq;
                                qq = 1.0 - sq;
                            } else {
                                double q_val = fa / fc, r = fb / fc;
                                sq = fb / fa;
                                pp = sq * (2.0 * xm * q_val * (q_val - r) - (b - a) * (r - 1.0));
                                qq = (q_val - 1.0) * (r - 1.0) * (sq - 1.0);
                            }
                            if (qq > 0.0) pp = -pp; else qq = -qq;
                            if (2.0 * pp < fmin(3.0 * xm * qq - fabs(tol1 * qq), fabs(e * qq))) { e = d; d = pp / qq; }
                            else { d = xm; e = d; }
                        } else { d = xm; e = d; }
                        a = b; fa = fb;
                        if (fabs(d) > tol1) b += d;
                        else b += xm > 0.0 ? tol1 : -tol1;
                        interp_any(s->method, b, ymid, cont, n, ixold, h);
                        s->ev(b, ymid, gmid, s->ev_params);
                        fb = gmid[i];
                    }
                    interp_any(s->method, b, ymid, cont, n, ixold, h);
                    det_t[ndet] = b;
                    memcpy(det_y[ndet], ymid, sizeof(double) * (size_t)n);
                }
                det_i[ndet] = i;
                ndet++;
            }
            /* stable sort by time: ascending when integrating forward, descending backward (solout.rs:297-303) */
            int forward = x > xold;
            for (int u = 1; u < ndet; u++)
                for (int v = u; v > 0; v--) {
                    int swap = forward ? (det_t[v] < det_t[v - 1]) : (det_t[v] > det_t[v - 1]);
                    if (!swap) break;
                    double tt = det_t[v]; det_t[v] = det_t[v - 1]; det_t[v - 1] = tt;
                    int ti = det_i[v]; det_i[v] = det_i[v - 1]; det_i[v - 1] = ti;
                    double ty[ORC_MAX_N]; memcpy(ty, det_y[v], sizeof ty); memcpy(det_y[v], det_y[v - 1], sizeof ty); memcpy(det_y[v - 1], ty, sizeof ty);
                }
            for (int u = 0; u < ndet; u++) {
                int i = det_i[u];
                if (s->ev_len[i] == s->ev_cap[i]) {
                    s->ev_cap[i] = s->ev_cap[i] ? 2 * s->ev_cap[i] : 8;
                    s->t_events[i] = (double *)realloc(s->t_events[i], s->ev_cap[i] * sizeof(double));
                    s->y_events[i] = (double *)realloc(s->y_events[i], s->ev_cap[i] * (size_t)n * sizeof(double));
                }
                s->t_events[i][s->ev_len[i]] = det_t[u];
                memcpy(s->y_events[i] + s->ev_len[i] * (size_t)n, det_y[u], sizeof(double) * (size_t)n);
                s->ev_len[i]++;
                s->event_hits[i]++;
                if (s->ev_terminal[i] && s->event_hits[i] >= s->ev_terminal[i]) {
                    so_push(s, det_t[u], det_y[u]);
                    memcpy(s->prev_event, g_curr, sizeof(double) * (size_t)s->n_events);
                    return 1;   /* ControlFlag::Interrupt */
                }
            }
            memcpy(s->prev_event, g_curr, sizeof(double) * (size_t)s->n_events);
        }
    }
    memcpy(s->yold, y, sizeof(double) * (size_t)n);
    s->has_yold = 1;

    double yi[ORC_MAX_N];
    if (s->n_eval >= 0) {
        /* Mode 1, solout.rs:344-386 */
        size_t i = s->next_idx;
        size_t ne = (size_t)s->n_eval;
        if (fabs(xold - x) <= s->tol) {
            while (i < ne && fabs(s->t_eval[i] - x) <= s->tol) {
                so_push(s, s->t_eval[i], y);
                i++;
            }
        } else {
            int forward = x > xold;
            if (forward) {
                while (i < ne && s->t_eval[i] <= x + s->tol) {
                    if (s->t_eval[i] >= xold - s->tol) {
                        interp_any(s->method, s->t_eval[i], yi, cont, n, ixold, h);
                        so_push(s, s->t_eval[i], yi);
                    }
                    i++;
                }
            } else {
                while (i < ne && s->t_eval[i] >= x - s->tol) {
                    if (s->t_eval[i] <= xold + s->tol) {
                        interp_any(s->method, s->t_eval[i], yi, cont, n, ixold, h);
                        so_push(s, s->t_eval[i], yi);
                    }
                    i++;
                }
            }
        }
        s->next_idx = i;
    } else {
        /* Mode 2, solout.rs:387-428 */
        if (s->has_first_step) {
            if (!s->first_output_done && fabs(xold - x) > s->tol) {
                double direction = rs_signum(x - xold);
                double target = s->x0 + direction * s->first_step;
                if (direction * (x - target) >= -s->tol) {
                    if (cont) {
                        interp_any(s->method, target, yi, cont, n, ixold, h);
                        so_push(s, target, yi);
                        s->first_output_done = 1;
                    }
                    if (fabs(x - target) > s->tol) so_push(s, x, y);
                    return 0;
                } else {
                    return 0;
                }
            }
        }
        if (s->len == 0 || fabs(s->t[s->len - 1] - x) > s->tol) so_push(s, x, y);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * hinit (src/methods/mod.rs:217-281)
 * ---------------------------------------------------------------------------------------- */
static double hinit(orc_ode_fn f, const double *p, int n, double x, const double *y, double posneg,
                    const double *f0, double *f1, double *y1, int iord, double hmax,
                    const tol_t *atol, const tol_t *rtol)
{
    double t_a[ORC_MAX_N], t_b[ORC_MAX_N];
    for (int i = 0; i < n; i++) {
        double sk = MA(tol_at(atol, i), tol_at(rtol, i), fabs(y[i]));
        t_a[i] = (f0[i] / sk) * (f0[i] / sk);
        t_b[i] = (y[i] / sk) * (y[i] / sk);
    }
    double dnf = orc_sum(t_a, n), dny = orc_sum(t_b, n);
    double h;
    if (dnf <= 1e-10 || dny <= 1e-10) h = 1.0e-6;
    else h = sqrt(dny / dnf) * 0.01;
    if (h > fabs(hmax)) h = fabs(hmax);
    h = fabs(h) * rs_signum(posneg);

    for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, f0[i]);
    f(x + h, y1, f1, p);

    for (int i = 0; i < n; i++) {
        double sk = MA(tol_at(atol, i), tol_at(rtol, i), fabs(y[i]));
        double df = (f1[i] - f0[i]) / sk;
        t_a[i] = df * df;
    }
    double der2 = orc_sum(t_a, n);
    der2 = sqrt(der2) / fabs(h);
    double der12 = fmax(fabs(der2), sqrt(dnf));
    double h1;
    if (der12 <= 1.0e-15) h1 = fmax(1.0e-6, fabs(h) * 1.0e-3);
    else h1 = ORC_POW(0.01 / der12, 1.0 / (double)iord);
    /* mod.rs:279: min(|h|, 100|h|, h1, hmax) -- the 100|h| term is dead but kept as written */
    double hf = fmin(fmin(fmin(fabs(h), 100.0 * fabs(h)), h1), fabs(hmax));
    return fabs(hf) * rs_signum(posneg);
}

typedef struct {
    double h; int status;
    uint64_t nfev, nstep, naccpt, nrejct;
} int_result;

/* ------------------------------------------------------------------------------------------
 * DOPRI5 (src/methods/dopri5.rs)
 * ---------------------------------------------------------------------------------------- */
static int dopri5_solve(orc_ode_fn f, const double *p, int n, double x0, const double *y0, double xend,
                        const tol_t *rtol, const tol_t *atol, const orc_options *opt, solout_t *so,
                        int_result *res, double *y_final, double *x_final)
{
    /* tableau, dopri5.rs:482-520 */
    const double C2 = 0.2, C3 = 0.3, C4 = 0.8, C5 = 8.0 / 9.0;
    const double A21 = 0.2;
    const double A31 = 3.0 / 40.0, A32 = 9.0 / 40.0;
    const double A41 = 44.0 / 45.0, A42 = -56.0 / 15.0, A43 = 32.0 / 9.0;
    const double A51 = 19372.0 / 6561.0, A52 = -25360.0 / 2187.0, A53 = 64448.0 / 6561.0, A54 = -212.0 / 729.0;
    const double A61 = 9017.0 / 3168.0, A62 = -355.0 / 33.0, A63 = 46732.0 / 5247.0, A64 = 49.0 / 176.0,
                 A65 = -5103.0 / 18656.0;
    const double A71 = 35.0 / 384.0, A73 = 500.0 / 1113.0, A74 = 125.0 / 192.0, A75 = -2187.0 / 6784.0,
                 A76 = 11.0 / 84.0;
    const double E1 = 71.0 / 57600.0, E3 = -71.0 / 16695.0, E4 = 71.0 / 1920.0, E5 = -17253.0 / 339200.0,
                 E6 = 22.0 / 525.0, E7 = -1.0 / 40.0;
    const double D1 = -12715105075.0 / 11282082432.0, D3 = 87487479700.0 / 32700410799.0,
                 D4 = -10690763975.0 / 1880347072.0, D5 = 701980252875.0 / 199316789632.0,
                 D6 = -1453857185.0 / 822651844.0, D7 = 69997945.0 / 29380423.0;

    /* struct defaults, dopri5.rs:34-72; solve_ivp only overrides max_step/first_step/max_steps */
    const int hs = opt->has_settings;
    const double uround = hs ? opt->uround : 2.3e-16, safety = hs ? opt->safety_factor : 0.9;
    const double scale_min = hs ? opt->scale_min : 0.2, scale_max = hs ? opt->scale_max : 10.0, beta = hs ? opt->beta : 0.04;
    const uint64_t nstiff = hs ? opt->stiff_test : 1000;
    const uint64_t nmax = opt->has_max_steps ? opt->max_steps : UINT64_MAX;
    /* validation in the reference's order, dopri5.rs:143-198 */
    if (uround <= 1e-35 || uround >= 1.0) return ORC_ERR_OUT_OF_RANGE;
    if (safety >= 1.0 || safety <= 1e-4) return ORC_ERR_OUT_OF_RANGE;
    if (beta > 0.2) return ORC_ERR_OUT_OF_RANGE;
    if (nmax == 0) return ORC_ERR_MUST_BE_POSITIVE; /* dopri5.rs:184-189 */
    if (nstiff == 0) return ORC_ERR_MUST_BE_POSITIVE;

    double x = x0;
    double *w = (double *)malloc((size_t)n * (8 + 5 + 2) * sizeof(double));
    double *y = w, *k1 = w + n, *k2 = w + 2 * n, *k3 = w + 3 * n, *k4 = w + 4 * n, *k5 = w + 5 * n,
           *k6 = w + 6 * n, *y1 = w + 7 * n, *cont = w + 8 * n, *t_a = w + 13 * n, *t_b = w + 14 * n;
    memcpy(y, y0, (size_t)n * sizeof(double));
    memset(k1, 0, (size_t)n * 12 * sizeof(double));

    const double facc1 = 1.0 / scale_min, facc2 = 1.0 / scale_max;
    const double h_max = opt->has_max_step ? opt->max_step : fabs(xend - x); /* dopri5.rs:180 */
    double facold = 1e-4;
    int last = 0, reject = 0;
    int nonstiff = 0, iasti = 0;
    double hlamb = 0.0;
    uint64_t nfev = 0, nstep = 0, naccpt = 0, nrejct = 0, attempts = 0;
    double xold = x;
    int status;
    const double expo1 = 0.2 - beta * 0.75;
    const double posneg = rs_signum(xend - x);

    f(x, y, k1, p);
    nfev += 1;
    double h;
    if (opt->has_first_step) h = fabs(opt->first_step) * posneg;
    else { nfev += 1; h = hinit(f, p, n, x, y, posneg, k1, k2, k3, 5, h_max, atol, rtol); }

    so_call(so, xold, x, y, NULL, 0.0);

    for (;;) {
        if (nstep > nmax) { status = ORC_NEED_LARGER_NMAX; break; }
        if (0.1 * fabs(h) <= fabs(x) * uround) { status = ORC_STEP_SIZE_TOO_SMALL; break; }
        if (opt->attempt_guard && attempts >= opt->attempt_guard) { status = ORC_NEED_LARGER_NMAX; break; }
        if ((x + 1.01 * h - xend) * posneg > 0.0) { h = xend - x; last = 1; }
        nstep += 1;
        attempts += 1;

        /* dopri5.rs:287-325; stage 2 is y + (h * A21) * k1, later stages y + h * (sum in source order) */
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h * A21, k1[i]);
        f(x + C2 * h, y1, k2, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC2(A31, k1[i], A32, k2[i]));
        f(x + C3 * h, y1, k3, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC3(A41, k1[i], A42, k2[i], A43, k3[i]));
        f(x + C4 * h, y1, k4, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC4(A51, k1[i], A52, k2[i], A53, k3[i], A54, k4[i]));
        f(x + C5 * h, y1, k5, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC5(A61, k1[i], A62, k2[i], A63, k3[i], A64, k4[i], A65, k5[i]));
        double xph = x + h;
        f(xph, y1, k6, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC5(A71, k1[i], A73, k3[i], A74, k4[i], A75, k5[i], A76, k6[i]));
        f(xph, y1, k2, p);
        nfev += 6;

        /* dense block 4 (always: struct default dense_output = true, dopri5.rs:329-334) */
        for (int i = 0; i < n; i++)
            cont[4 * n + i] = h * LC6(D1, k1[i], D3, k3[i], D4, k4[i], D5, k5[i], D6, k6[i], D7, k2[i]);

        for (int i = 0; i < n; i++)
            k4[i] = LC6(E1, k1[i], E3, k3[i], E4, k4[i], E5, k5[i], E6, k6[i], E7, k2[i]) * h;

        for (int i = 0; i < n; i++) {
            double sk = MA(tol_at(atol, i), tol_at(rtol, i), fmax(fabs(y[i]), fabs(y1[i])));
            t_a[i] = (k4[i] / sk) * (k4[i] / sk);
        }
        double err = orc_sum(t_a, n);
        err = sqrt(err / (double)n);

        double fac11 = ORC_POW(err, expo1);
        double fac = fac11 / ORC_POW(facold, beta);
        fac = fmax(facc2, fmin(facc1, fac / safety));
        double hnew = h / fac;

        if (err <= 1.0) {
            facold = fmax(err, 1.0e-4);
            naccpt += 1;

            if ((naccpt % nstiff == 0) || (iasti > 0)) { /* dopri5.rs:364-391 */
                for (int i = 0; i < n; i++) {
                    double d1 = k2[i] - k6[i];
                    double ysti = MA(y[i], h, LC5(A61, k1[i], A62, k2[i], A63, k3[i], A64, k4[i], A65, k5[i]));
                    double d2 = y1[i] - ysti;
                    t_a[i] = d1 * d1;
                    t_b[i] = d2 * d2;
                }
                double stnum = orc_sum(t_a, n), stden = orc_sum(t_b, n);
                if (stden > 0.0) hlamb = fabs(h) * sqrt(stnum / stden);
                if (hlamb > 3.25) {
                    nonstiff = 0;
                    iasti += 1;
                    if (iasti == 15) { status = ORC_PROBABLY_STIFF; break; }
                } else {
                    nonstiff += 1;
                    if (nonstiff == 6) iasti = 0;
                }
            }

            for (int i = 0; i < n; i++) { /* dopri5.rs:394-403 */
                double ydiff = y1[i] - y[i];
                double bspl = MB(h, k1[i], ydiff);
                cont[i] = y[i];
                cont[n + i] = ydiff;
                cont[2 * n + i] = bspl;
                cont[3 * n + i] = MA(ydiff, -h, k2[i]) - bspl;   /* -h k7 + ydiff - bspl */
            }

            memcpy(k1, k2, (size_t)n * sizeof(double));
            memcpy(y, y1, (size_t)n * sizeof(double));
            xold = x;
            x = xph;

            if (so_call(so, xold, x, y, cont, h)) { status = ORC_USER_INTERRUPT; break; }   /* dopri5.rs:418-421 */

            if (last) { h = hnew; status = ORC_SUCCESS; break; }
            if (fabs(hnew) > fabs(h_max)) hnew = posneg * fabs(h_max);
            if (reject) { hnew = posneg * fmin(fabs(hnew), fabs(h)); reject = 0; }
        } else {
            hnew = h / fmin(facc1, fac11 / safety);
            reject = 1;
            if (naccpt > 1) nrejct += 1;
            last = 0;
        }
        h = hnew;
    }

    res->h = h; res->status = status;
    res->nfev = nfev; res->nstep = nstep; res->naccpt = naccpt; res->nrejct = nrejct;
    memcpy(y_final, y, (size_t)n * sizeof(double));
    *x_final = x;
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * DOP853 (src/methods/dop853.rs)
 * ---------------------------------------------------------------------------------------- */
static int dop853_solve(orc_ode_fn f, const double *p, int n, double x0, const double *y0, double xend,
                        const tol_t *rtol, const tol_t *atol, const orc_options *opt, solout_t *so,
                        int_result *res, double *y_final, double *x_final)
{
    /* tableau, dop853.rs:674-848 (Hairer's DOP853 coefficients) */
    const double C2 = 0.526001519587677318785587544488e-01, C3 = 0.789002279381515978178381316732e-01,
                 C4 = 0.118350341907227396726757197510e+00, C5 = 0.281649658092772603273242802490e+00,
                 C6 = 0.333333333333333333333333333333e+00, C7 = 0.25e+00,
                 C8 = 0.307692307692307692307692307692e+00, C9 = 0.651282051282051282051282051282e+00,
                 C10 = 0.6e+00, C11 = 0.857142857142857142857142857142e+00, C14 = 0.1e+00, C15 = 0.2e+00,
                 C16 = 7.777777777777778e-1;
    const double A21 = 5.26001519587677318785587544488e-2;
    const double A31 = 1.97250569845378994544595329183e-2, A32 = 5.91751709536136983633785987549e-2;
    const double A41 = 2.95875854768068491816892993775e-2, A43 = 8.87627564304205475450678981324e-2;
    const double A51 = 2.41365134159266685502369798665e-1, A53 = -8.84549479328286085344864962717e-1,
                 A54 = 9.24834003261792003115737966543e-1;
    const double A61 = 3.7037037037037037037037037037e-2, A64 = 1.70828608729473871279604482173e-1,
                 A65 = 1.25467687566822425016691814123e-1;
    const double A71 = 3.7109375e-2, A74 = 1.70252211019544039314978060272e-1,
                 A75 = 6.02165389804559606850219397283e-2, A76 = -1.7578125e-2;
    const double A81 = 3.70920001185047927108779319836e-2, A84 = 1.70383925712239993810214054705e-1,
                 A85 = 1.07262030446373284651809199168e-1, A86 = -1.53194377486244017527936158236e-2,
                 A87 = 8.27378916381402288758473766002e-3;
    const double A91 = 6.24110958716075717114429577812e-1, A94 = -3.36089262944694129406857109825e0,
                 A95 = -8.68219346841726006818189891453e-1, A96 = 2.75920996994467083049415600797e1,
                 A97 = 2.01540675504778934086186788979e1, A98 = -4.34898841810699588477366255144e1;
    const double A101 = 4.77662536438264365890433908527e-1, A104 = -2.48811461997166764192642586468e0,
                 A105 = -5.90290826836842996371446475743e-1, A106 = 2.12300514481811942347288949897e1,
                 A107 = 1.52792336328824235832596922938e1, A108 = -3.32882109689848629194453265587e1,
                 A109 = -2.03312017085086261358222928593e-2;
    const double A111 = -9.3714243008598732571704021658e-1, A114 = 5.18637242884406370830023853209e0,
                 A115 = 1.09143734899672957818500254654e0, A116 = -8.14978701074692612513997267357e0,
                 A117 = -1.85200656599969598641566180701e1, A118 = 2.27394870993505042818970056734e1,
                 A119 = 2.49360555267965238987089396762e0, A1110 = -3.0467644718982195003823669022e0;
    const double A121 = 2.27331014751653820792359768449e0, A124 = -1.05344954667372501984066689879e1,
                 A125 = -2.00087205822486249909675718444e0, A126 = -1.79589318631187989172765950534e1,
                 A127 = 2.79488845294199600508499808837e1, A128 = -2.85899827713502369474065508674e0,
                 A129 = -8.87285693353062954433549289258e0, A1210 = 1.23605671757943030647266201528e1,
                 A1211 = 6.43392746015763530355970484046e-1;
    const double B1 = 5.42937341165687622380535766363e-2, B6 = 4.45031289275240888144113950566e0,
                 B7 = 1.89151789931450038304281599044e0, B8 = -5.8012039600105847814672114227e0,
                 B9 = 3.1116436695781989440891606237e-1, B10 = -1.52160949662516078556178806805e-1,
                 B11 = 2.01365400804030348374776537501e-1, B12 = 4.47106157277725905176885569043e-2;
    const double BH1 = 0.244094488188976377952755905512e+00, BH2 = 0.733846688281611857341361741547e+00,
                 BH3 = 0.220588235294117647058823529412e-01;
    const double ER1 = 0.1312004499419488073250102996e-01, ER6 = -0.1225156446376204440720569753e+01,
                 ER7 = -0.4957589496572501915214079952e+00, ER8 = 0.1664377182454986536961530415e+01,
                 ER9 = -0.3503288487499736816886487290e+00, ER10 = 0.3341791187130174790297318841e+00,
                 ER11 = 0.8192320648511571246570742613e-01, ER12 = -0.2235530786388629525884427845e-01;
    const double A141 = 5.61675022830479523392909219681e-2, A147 = 2.53500210216624811088794765333e-1,
                 A148 = -2.46239037470802489917441475441e-1, A149 = -1.24191423263816360469010140626e-1,
                 A1410 = 1.5329179827876569731206322685e-1, A1411 = 8.20105229563468988491666602057e-3,
                 A1412 = 7.56789766054569976138603589584e-3, A1413 = -8.298e-3;
    const double A151 = 3.18346481635021405060768473261e-2, A156 = 2.83009096723667755288322961402e-2,
                 A157 = 5.35419883074385676223797384372e-2, A158 = -5.49237485713909884646569340306e-2,
                 A1511 = -1.08347328697249322858509316994e-4, A1512 = 3.82571090835658412954920192323e-4,
                 A1513 = -3.40465008687404560802977114492e-4, A1514 = 1.41312443674632500278074618366e-1;
    const double A161 = -4.28896301583791923408573538692e-1, A166 = -4.69762141536116384314449447206e0,
                 A167 = 7.68342119606259904184240953878e0, A168 = 4.06898981839711007970213554331e0,
                 A169 = 3.56727187455281109270669543021e-1, A1613 = -1.39902416515901462129418009734e-3,
                 A1614 = 2.9475147891527723389556272149e0, A1615 = -9.15095847217987001081870187138e0;
    const double D41 = -0.84289382761090128651353491142e+01, D46 = 0.56671495351937776962531783590e+00,
                 D47 = -0.30689499459498916912797304727e+01, D48 = 0.23846676565120698287728149680e+01,
                 D49 = 0.21170345824450282767155149946e+01, D410 = -0.87139158377797299206789907490e+00,
                 D411 = 0.22404374302607882758541771650e+01, D412 = 0.63157877876946881815570249290e+00,
                 D413 = -0.88990336451333310820698117400e-01, D414 = 0.18148505520854727256656404962e+02,
                 D415 = -0.91946323924783554000451984436e+01, D416 = -0.44360363875948939664310572000e+01;
    const double D51 = 0.10427508642579134603413151009e+02, D56 = 0.24228349177525818288430175319e+03,
                 D57 = 0.16520045171727028198505394887e+03, D58 = -0.37454675472269020279518312152e+03,
                 D59 = -0.22113666853125306036270938578e+02, D510 = 0.77334326684722638389603898808e+01,
                 D511 = -0.30674084731089398182061213626e+02, D512 = -0.93321305264302278729567221706e+01,
                 D513 = 0.15697238121770843886131091075e+02, D514 = -0.31139403219565177677282850411e+02,
                 D515 = -0.93529243588444783865713862664e+01, D516 = 0.35816841486394083752465898540e+02;
    const double D61 = 0.19985053242002433820987653617e+02, D66 = -0.38703730874935176555105901742e+03,
                 D67 = -0.18917813819516756882830838328e+03, D68 = 0.52780815920542364900561016686e+03,
                 D69 = -0.11573902539959630126141871134e+02, D610 = 0.68812326946963000169666922661e+01,
                 D611 = -0.10006050966910838403183860980e+01, D612 = 0.77771377980534432092869265740e+00,
                 D613 = -0.27782057523535084065932004339e+01, D614 = -0.60196695231264120758267380846e+02,
                 D615 = 0.84320405506677161018159903784e+02, D616 = 0.11992291136182789328035130030e+02;
    const double D71 = -0.25693933462703749003312586129e+02, D76 = -0.15418974869023643374053993627e+03,
                 D77 = -0.23152937917604549567536039109e+03, D78 = 0.35763911791061412378285349910e+03,
                 D79 = 0.93405324183624310003907691704e+02, D710 = -0.37458323136451633156875139351e+02,
                 D711 = 0.10409964950896230045147246184e+03, D712 = 0.29840293426660503123344363579e+02,
                 D713 = -0.43533456590011143754432175058e+02, D714 = 0.96324553959188282948394950600e+02,
                 D715 = -0.39177261675615439165231486172e+02, D716 = -0.14972683625798562581422125276e+03;

    /* struct defaults, dop853.rs:34-63 */
    const int hs = opt->has_settings;
    const double uround = hs ? opt->uround : 2.3e-16, safety = hs ? opt->safety_factor : 0.9;
    const double scale_min = hs ? opt->scale_min : 0.333, scale_max = hs ? opt->scale_max : 6.0, beta = hs ? opt->beta : 0.0;
    const uint64_t nstiff = hs ? opt->stiff_test : 1000;
    const uint64_t nmax = opt->has_max_steps ? opt->max_steps : UINT64_MAX;
    /* validation in the reference's order, dop853.rs:135-193 */
    if (uround <= 1e-35 || uround >= 1.0) return ORC_ERR_OUT_OF_RANGE;
    if (safety >= 1.0 || safety <= 1e-4) return ORC_ERR_OUT_OF_RANGE;
    if (beta > 0.2) return ORC_ERR_OUT_OF_RANGE;
    if (nmax == 0) return ORC_ERR_MUST_BE_POSITIVE;
    if (nstiff == 0) return ORC_ERR_MUST_BE_POSITIVE;

    double x = x0;
    double *w = (double *)malloc((size_t)n * (12 + 8 + 2) * sizeof(double));
    double *y = w, *y1 = w + n, *k1 = w + 2 * n, *k2 = w + 3 * n, *k3 = w + 4 * n, *k4 = w + 5 * n,
           *k5 = w + 6 * n, *k6 = w + 7 * n, *k7 = w + 8 * n, *k8 = w + 9 * n, *k9 = w + 10 * n,
           *k10 = w + 11 * n, *cont = w + 12 * n, *t_a = w + 20 * n, *t_b = w + 21 * n;
    memcpy(y, y0, (size_t)n * sizeof(double));
    memset(y1, 0, (size_t)n * 19 * sizeof(double));

    const double facc1 = 1.0 / scale_min, facc2 = 1.0 / scale_max;
    const double h_max = opt->has_max_step ? fabs(opt->max_step) : fabs(xend - x); /* dop853.rs:172-175 */
    int nonstiff = 0, iasti = 0;
    double facold = 1e-4, hlamb = 0.0;
    int last = 0, reject = 0;
    uint64_t nfev = 0, nstep = 0, naccpt = 0, nrejct = 0, attempts = 0;
    double xold = x;
    const double expo1 = 1.0 / 8.0 - beta * 0.2;
    int status;
    const double posneg = rs_signum(xend - x);

    f(x, y, k1, p);
    nfev += 1;
    double h;
    if (opt->has_first_step) h = fabs(opt->first_step) * posneg;
    else { nfev += 1; h = hinit(f, p, n, x, y, posneg, k1, k2, y1, 8, h_max, atol, rtol); }

    so_call(so, xold, x, y, NULL, 0.0);

    for (;;) {
        if (nstep > nmax) { status = ORC_NEED_LARGER_NMAX; break; }
        if (0.1 * fabs(h) <= fabs(x) * uround) { status = ORC_STEP_SIZE_TOO_SMALL; break; }
        if (opt->attempt_guard && attempts >= opt->attempt_guard) { status = ORC_NEED_LARGER_NMAX; break; }
        if ((x + 1.01 * h - xend) * posneg > 0.0) { h = xend - x; last = 1; }
        nstep += 1;
        attempts += 1;

        /* dop853.rs:293-390 */
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h * A21, k1[i]);
        f(x + C2 * h, y1, k2, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC2(A31, k1[i], A32, k2[i]));
        f(x + C3 * h, y1, k3, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC2(A41, k1[i], A43, k3[i]));
        f(x + C4 * h, y1, k4, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC3(A51, k1[i], A53, k3[i], A54, k4[i]));
        f(x + C5 * h, y1, k5, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC3(A61, k1[i], A64, k4[i], A65, k5[i]));
        f(x + C6 * h, y1, k6, p);
        for (int i = 0; i < n; i++) y1[i] = MA(y[i], h, LC4(A71, k1[i], A74, k4[i], A75, k5[i], A76, k6[i]));
        f(x + C7 * h, y1, k7, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC5(A81, k1[i], A84, k4[i], A85, k5[i], A86, k6[i], A87, k7[i]));
        f(x + C8 * h, y1, k8, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC6(A91, k1[i], A94, k4[i], A95, k5[i], A96, k6[i], A97, k7[i], A98, k8[i]));
        f(x + C9 * h, y1, k9, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC7(A101, k1[i], A104, k4[i], A105, k5[i], A106, k6[i], A107, k7[i],
                                    A108, k8[i], A109, k9[i]));
        f(x + C10 * h, y1, k10, p);
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC8(A111, k1[i], A114, k4[i], A115, k5[i], A116, k6[i], A117, k7[i],
                                    A118, k8[i], A119, k9[i], A1110, k10[i]));
        f(x + C11 * h, y1, k2, p);
        double xph = x + h;
        for (int i = 0; i < n; i++)
            y1[i] = MA(y[i], h, LC9(A121, k1[i], A124, k4[i], A125, k5[i], A126, k6[i], A127, k7[i],
                                    A128, k8[i], A129, k9[i], A1210, k10[i], A1211, k2[i]));
        f(xph, y1, k3, p);
        nfev += 11;

        for (int i = 0; i < n; i++) {
            k4[i] = LC8(B1, k1[i], B6, k6[i], B7, k7[i], B8, k8[i], B9, k9[i], B10, k10[i], B11, k2[i], B12, k3[i]);
            k5[i] = MA(y[i], h, k4[i]);
        }

        for (int i = 0; i < n; i++) {
            double sk = MA(tol_at(atol, i), tol_at(rtol, i), fmax(fabs(y[i]), fabs(k5[i])));
            double erri = MS(MS(MS(k4[i], BH1, k1[i]), BH2, k9[i]), BH3, k3[i]);
            double q = erri / sk;
            t_b[i] = q * q; /* powi(2) */
            erri = LC8(ER1, k1[i], ER6, k6[i], ER7, k7[i], ER8, k8[i], ER9, k9[i], ER10, k10[i], ER11, k2[i], ER12, k3[i]);
            q = erri / sk;
            t_a[i] = q * q;
        }
        double err = orc_sum(t_a, n), err2 = orc_sum(t_b, n);
        double deno = MA(err, 0.01, err2);
        if (deno <= 0.0) deno = 1.0;
        err = fabs(h) * err * sqrt(1.0 / ((double)n * deno));

        double fac11 = ORC_POW(err, expo1);
        double fac = fac11 / ORC_POW(facold, beta);
        fac = fmax(facc2, fmin(facc1, fac / safety));
        double hnew = h / fac;

        if (err <= 1.0) {
            facold = fmax(err, 1.0e-4);
            naccpt += 1;
            f(xph, k5, k4, p);
            nfev += 1;

            if ((naccpt % nstiff == 0) || (iasti > 0)) { /* dop853.rs:447-472 */
                for (int i = 0; i < n; i++) {
                    double d1 = k4[i] - k3[i];
                    double d2 = k5[i] - y1[i];
                    t_a[i] = d1 * d1;
                    t_b[i] = d2 * d2;
                }
                double stnum = orc_sum(t_a, n), stden = orc_sum(t_b, n);
                if (stden > 0.0) hlamb = fabs(h) * sqrt(stnum / stden);
                if (hlamb > 6.1) {
                    nonstiff = 0;
                    iasti += 1;
                    if (iasti == 15) { status = ORC_PROBABLY_STIFF; break; }
                } else {
                    nonstiff += 1;
                    if (nonstiff == 6) iasti = 0;
                }
            }

            /* dense output, always on (struct default), dop853.rs:476-592 */
            for (int i = 0; i < n; i++) {
                cont[i] = y[i];
                double ydiff = k5[i] - y[i];
                cont[n + i] = ydiff;
                double bspl = MB(h, k1[i], ydiff);
                cont[2 * n + i] = bspl;
                cont[3 * n + i] = MS(ydiff, h, k4[i]) - bspl;
                cont[4 * n + i] = LC8(D41, k1[i], D46, k6[i], D47, k7[i], D48, k8[i], D49, k9[i],
                                      D410, k10[i], D411, k2[i], D412, k3[i]);
                cont[5 * n + i] = LC8(D51, k1[i], D56, k6[i], D57, k7[i], D58, k8[i], D59, k9[i],
                                      D510, k10[i], D511, k2[i], D512, k3[i]);
                cont[6 * n + i] = LC8(D61, k1[i], D66, k6[i], D67, k7[i], D68, k8[i], D69, k9[i],
                                      D610, k10[i], D611, k2[i], D612, k3[i]);
                cont[7 * n + i] = LC8(D71, k1[i], D76, k6[i], D77, k7[i], D78, k8[i], D79, k9[i],
                                      D710, k10[i], D711, k2[i], D712, k3[i]);
            }
            for (int i = 0; i < n; i++)
                y1[i] = MA(y[i], h, LC8(A141, k1[i], A147, k7[i], A148, k8[i], A149, k9[i], A1410, k10[i],
                                        A1411, k2[i], A1412, k3[i], A1413, k4[i]));
            f(x + C14 * h, y1, k10, p);
            for (int i = 0; i < n; i++)
                y1[i] = MA(y[i], h, LC8(A151, k1[i], A156, k6[i], A157, k7[i], A158, k8[i], A1511, k2[i],
                                        A1512, k3[i], A1513, k4[i], A1514, k10[i]));
            f(x + C15 * h, y1, k2, p);
            for (int i = 0; i < n; i++)
                y1[i] = MA(y[i], h, LC8(A161, k1[i], A166, k6[i], A167, k7[i], A168, k8[i], A169, k9[i],
                                        A1613, k4[i], A1614, k10[i], A1615, k2[i]));
            f(x + C16 * h, y1, k3, p);
            nfev += 3;
            for (int i = 0; i < n; i++) {
                cont[4 * n + i] = h * MA(MA(MA(MA(cont[4 * n + i], D413, k4[i]), D414, k10[i]), D415, k2[i]), D416, k3[i]);
                cont[5 * n + i] = h * MA(MA(MA(MA(cont[5 * n + i], D513, k4[i]), D514, k10[i]), D515, k2[i]), D516, k3[i]);
                cont[6 * n + i] = h * MA(MA(MA(MA(cont[6 * n + i], D613, k4[i]), D614, k10[i]), D615, k2[i]), D616, k3[i]);
                cont[7 * n + i] = h * MA(MA(MA(MA(cont[7 * n + i], D713, k4[i]), D714, k10[i]), D715, k2[i]), D716, k3[i]);
            }

            memcpy(k1, k4, (size_t)n * sizeof(double));
            memcpy(y, k5, (size_t)n * sizeof(double));
            xold = x;
            x = xph;

            if (so_call(so, xold, x, y, cont, h)) { status = ORC_USER_INTERRUPT; break; }   /* dop853.rs:609-612 */

            if (last) { h = hnew; status = ORC_SUCCESS; break; }
            if (fabs(hnew) > fabs(h_max)) hnew = posneg * fabs(h_max);
            if (reject) { hnew = posneg * fmin(fabs(hnew), fabs(h)); reject = 0; }
        } else {
            hnew = h / fmin(facc1, fac11 / safety);
            reject = 1;
            if (naccpt > 1) nrejct += 1;
            last = 0;
        }
        h = hnew;
    }

    res->h = h; res->status = status;
    res->nfev = nfev; res->nstep = nstep; res->naccpt = naccpt; res->nrejct = nrejct;
    memcpy(y_final, y, (size_t)n * sizeof(double));
    *x_final = x;
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * RK23 (src/methods/rk23.rs)
 * ---------------------------------------------------------------------------------------- */
static int rk23_solve(orc_ode_fn f, const double *p, int n, double x0, const double *y0, double xend,
                      const tol_t *rtol, const tol_t *atol, const orc_options *opt, solout_t *so,
                      int_result *res, double *y_final, double *x_final)
{
    /* tableau, rk23.rs:325-347 */
    const double C2 = 0.5, C3 = 0.75, A21 = 0.5, A32 = 0.75;
    const double B1 = 2.0 / 9.0, B2 = 1.0 / 3.0, B3 = 4.0 / 9.0;
    const double E1 = 5.0 / 72.0, E2 = -1.0 / 12.0, E3 = -1.0 / 9.0, E4 = 1.0 / 8.0;
    const double D21 = -4.0 / 3.0, D22 = 1.0, D23 = 4.0 / 3.0, D24 = -1.0;
    const double D31 = 5.0 / 9.0, D32 = -2.0 / 3.0, D33 = -8.0 / 9.0, D34 = 1.0;

    /* struct defaults, rk23.rs:16-36 */
    const int hs = opt->has_settings;
    const double safety = hs ? opt->safety_factor : 0.9, scale_min = hs ? opt->scale_min : 0.2, scale_max = hs ? opt->scale_max : 10.0;
    const uint64_t nmax = opt->has_max_steps ? opt->max_steps : UINT64_MAX;
    /* validation in the reference's order, rk23.rs:102-129 */
    if (nmax == 0) return ORC_ERR_MUST_BE_POSITIVE;
    if (safety >= 1.0 || safety <= 1e-4) return ORC_ERR_OUT_OF_RANGE;
    if (scale_min <= 0.0 || scale_max <= scale_min) return ORC_ERR_INVALID_SCALE_FACTORS;
    const double error_exponent = -1.0 / 3.0;

    double x = x0;
    const double hmax = opt->has_max_step ? fabs(opt->max_step) : fabs(xend - x); /* rk23.rs:135 */
    double *w = (double *)malloc((size_t)n * (7 + 4 + 1) * sizeof(double));
    double *y = w, *k1 = w + n, *k2 = w + 2 * n, *k3 = w + 3 * n, *k4 = w + 4 * n, *yt = w + 5 * n,
           *ye = w + 6 * n, *cont = w + 7 * n, *t_a = w + 11 * n;
    memcpy(y, y0, (size_t)n * sizeof(double));
    memset(k1, 0, (size_t)n * 10 * sizeof(double));
    uint64_t nfev = 0, nstep = 0, naccpt = 0, nrejct = 0, attempts = 0;
    int status = ORC_SUCCESS;
    double xold = x;
    const double posneg = rs_signum(xend - x);

    f(x, y, k1, p);
    nfev += 1;
    double h;
    if (opt->has_first_step) h = fabs(opt->first_step) * posneg;
    else { nfev += 1; h = hinit(f, p, n, x, y, posneg, k1, k2, k3, 3, hmax, atol, rtol); }

    so_call(so, xold, x, y, NULL, 0.0);

    for (;;) {
        if (nstep >= nmax) { status = ORC_NEED_LARGER_NMAX; break; }
        if (opt->attempt_guard && attempts >= opt->attempt_guard) { status = ORC_NEED_LARGER_NMAX; break; }
        attempts += 1;
        if ((x + h - xend) * posneg > 0.0) h = xend - x;

        /* rk23.rs:201-234 */
        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h * A21, k1[i]);
        f(x + C2 * h, yt, k2, p);
        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h * A32, k2[i]);
        f(x + C3 * h, yt, k3, p);
        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h, LC3(B1, k1[i], B2, k2[i], B3, k3[i]));
        f(x + h, yt, k4, p);
        nfev += 3;

        for (int i = 0; i < n; i++) ye[i] = h * LC4(E1, k1[i], E2, k2[i], E3, k3[i], E4, k4[i]);
        for (int i = 0; i < n; i++) {
            double tol = MA(tol_at(atol, i), tol_at(rtol, i), fmax(fabs(yt[i]), fabs(y[i])));
            double q = ye[i] / tol;
            t_a[i] = q * q;
        }
        double err = orc_sum(t_a, n);
        err = sqrt(err / (double)n);

        if (err <= 1.0) {
            nstep += 1;
            naccpt += 1;
            memcpy(ye, y, (size_t)n * sizeof(double));
            memcpy(y, yt, (size_t)n * sizeof(double));
            xold = x;
            x += h;

            /* dense (struct default true and solout is Some), rk23.rs:248-255 */
            memcpy(cont, ye, (size_t)n * sizeof(double));
            for (int i = 0; i < n; i++) {
                cont[n + i] = k1[i];
                cont[2 * n + i] = LC4(D21, k1[i], D22, k2[i], D23, k3[i], D24, k4[i]);
                cont[3 * n + i] = LC4(D31, k1[i], D32, k2[i], D33, k3[i], D34, k4[i]);
            }
            if (so_call(so, xold, x, y, cont, h)) { status = ORC_USER_INTERRUPT; break; }   /* rk23.rs:266-269 */
            memcpy(k1, k4, (size_t)n * sizeof(double)); /* ControlFlag::Continue arm, rk23.rs:281-284 */

            h *= fmax(fmin(safety * ORC_POW(err, error_exponent), scale_max), scale_min);
            if (fabs(h) > hmax) h = hmax * posneg;
            if (x == xend) break;
        } else {
            nrejct += 1;
            h *= fmax(fmin(safety * ORC_POW(err, error_exponent), 1.0), scale_min);
        }
    }

    res->h = h; res->status = status;
    res->nfev = nfev; res->nstep = nstep; res->naccpt = naccpt; res->nrejct = nrejct;
    memcpy(y_final, y, (size_t)n * sizeof(double));
    *x_final = x;
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * RK4 (src/methods/rk4.rs:64-226): fixed step.  Quirks kept: the initial f(x0,y0) is not counted
 * in nfev (rk4.rs:119), steps.accepted is never incremented, and the last step is NOT shortened
 * (rk4.rs:148-152 only sets `last`), so the final x is x0 + k*h, wherever that lands.
 * ---------------------------------------------------------------------------------------- */
static int rk4_solve(orc_ode_fn f, const double *p, int n, double x0, const double *y0, double xend, double h,
                     const orc_options *opt, solout_t *so, int_result *res, double *y_final, double *x_final)
{
    const double C2 = 0.5, C3 = 0.5, C4 = 1.0, A21 = 0.5, A32 = 0.5, A43 = 1.0;
    const double B1 = 1.0 / 6.0, B2 = 1.0 / 3.0, B3 = 1.0 / 3.0, B4 = 1.0 / 6.0;
    double x = x0;
    const double posneg = rs_signum(xend - x);
    if (h == 0.0 || rs_signum(h) != posneg) return ORC_ERR_INVALID_STEP_SIZE; /* rk4.rs:81-87 */
    const uint64_t nmax = opt->has_max_steps ? opt->max_steps : UINT64_MAX;
    if (nmax == 0) return ORC_ERR_MUST_BE_POSITIVE;

    double *w = (double *)malloc((size_t)n * (6 + 4) * sizeof(double));
    double *y = w, *k1 = w + n, *k2 = w + 2 * n, *k3 = w + 3 * n, *k4 = w + 4 * n, *yt = w + 5 * n, *cont = w + 6 * n;
    memcpy(y, y0, (size_t)n * sizeof(double));
    uint64_t nfev = 0, nstep = 0;
    int status = ORC_SUCCESS;
    double xold = x;

    f(x, y, k1, p); /* not counted */
    so_call(so, xold, x, y, NULL, 0.0);

    for (;;) {
        if (nstep >= nmax) { status = ORC_NEED_LARGER_NMAX; break; }
        int last = 0;
        if ((x + 1.01 * h - xend) * rs_signum(h) > 0.0) last = 1;

        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h * A21, k1[i]);
        f(x + C2 * h, yt, k2, p);
        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h * A32, k2[i]);
        f(x + C3 * h, yt, k3, p);
        for (int i = 0; i < n; i++) yt[i] = MA(y[i], h * A43, k3[i]);
        f(x + C4 * h, yt, k4, p);

        xold = x;
        memcpy(yt, y, (size_t)n * sizeof(double));
        x += h;
        for (int i = 0; i < n; i++) y[i] = MA(y[i], h, LC4(B1, k1[i], B2, k2[i], B3, k3[i], B4, k4[i]));
        f(x, y, k1, p);
        nfev += 4;
        nstep += 1;

        memcpy(cont, yt, (size_t)n * sizeof(double));
        for (int i = 0; i < n; i++) { cont[n + i] = k4[i]; cont[2 * n + i] = k1[i]; }
        memcpy(cont + 3 * n, y, (size_t)n * sizeof(double));
        if (so_call(so, xold, x, y, cont, h)) { status = ORC_USER_INTERRUPT; break; }   /* rk4.rs:201-204 */
        if (last) break;
    }
    res->h = h; res->status = status;
    res->nfev = nfev; res->nstep = nstep; res->naccpt = 0; res->nrejct = 0;
    memcpy(y_final, y, (size_t)n * sizeof(double));
    *x_final = x;
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Dense LU with partial pivoting and the matching solve (Hairer's DEC/SOL):
 * src/matrix/lu.rs:37-125, src/matrix/linear.rs:55-96.  Row-major a[r*n + c].
 * ---------------------------------------------------------------------------------------- */
static int lu_decomp(double *a, int *ip, int n)
{
    if (n == 1) {
        if (a[0] == 0.0) return -1;
        ip[0] = 0;
        return 0;
    }
    for (int k = 0; k < n - 1; k++) {
        int m = k;
        double max_val = fabs(a[k * n + k]);
        for (int i = k + 1; i < n; i++) {
            double v = fabs(a[i * n + k]);
            if (v > max_val) { max_val = v; m = i; }
        }
        ip[k] = m;
        double pivot = a[m * n + k];
        if (pivot == 0.0) return -1;
        if (m != k) { double t = a[m * n + k]; a[m * n + k] = a[k * n + k]; a[k * n + k] = t; }
        double t = 1.0 / pivot;
        for (int i = k + 1; i < n; i++) a[i * n + k] = -a[i * n + k] * t;
        for (int j = k + 1; j < n; j++) {
            double tj = a[m * n + j];
            if (m != k) { double tmp = a[m * n + j]; a[m * n + j] = a[k * n + j]; a[k * n + j] = tmp; }
            if (tj != 0.0)
                for (int i = k + 1; i < n; i++) a[i * n + j] = MA(a[i * n + j], a[i * n + k], tj);
        }
    }
    if (a[(n - 1) * n + (n - 1)] == 0.0) return -1;
    return 0;
}
static void lin_solve(const double *a, double *b, const int *ip, int n)
{
    if (n == 1) { b[0] /= a[0]; return; }
    for (int k = 0; k < n - 1; k++) {
        int m = ip[k];
        double t = b[m]; b[m] = b[k]; b[k] = t;
        for (int i = k + 1; i < n; i++) b[i] = MA(b[i], a[i * n + k], b[k]);
    }
    for (int kb = 1; kb < n; kb++) {
        int k = n - kb;
        b[k] /= a[k * n + k];
        for (int i = 0; i < k; i++) b[i] = MA(b[i], a[i * n + k], -b[k]);
    }
    b[0] /= a[0];
}

/* Default finite-difference Jacobian, trait IVP::jac (src/ivp.rs:67-107). */
static void fd_jac(orc_ode_fn f, const double *p, int n, double x, const double *y, double *jac)
{
    double yp[ORC_MAX_N], fp[ORC_MAX_N], fo[ORC_MAX_N];
    memcpy(yp, y, (size_t)n * sizeof(double));
    f(x, y, fo, p);
    const double eps = sqrt(2.220446049250313e-16);
    for (int col = 0; col < n; col++) {
        double yo = y[col];
        double pert = eps * fmax(fabs(yo), 1.0);
        yp[col] = yo + pert;
        f(x, yp, fp, p);
        yp[col] = yo;
        for (int row = 0; row < n; row++) jac[row * n + col] = (fp[row] - fo[row]) / pert;
    }
}

/* f.jac(x, y, &mut j): the user's override when the problem has one, else the trait default above */
static void eval_jac(const orc_options *opt, orc_ode_fn f, const double *p, int n, double x, const double *y, double *jac)
{
    if (opt->jac) opt->jac(x, y, jac, p);
    else fd_jac(f, p, n, x, y, jac);
}

/* ------------------------------------------------------------------------------------------
 * BDF 1..5 (src/methods/bdf.rs:86-732)
 * ---------------------------------------------------------------------------------------- */
#define BDF_MAXO 5
static double wrms_scaled(const double *v, const double *scale, int n)
{   /* bdf.rs:659-667 */
    double t[ORC_MAX_N];
    for (int i = 0; i < n; i++) {
        double denom = scale[i] == 0.0 ? 2.220446049250313e-16 : scale[i];
        double ratio = v[i] / denom;
        t[i] = ratio * ratio;
    }
    return sqrt(orc_sum(t, n) / (double)n);
}
static void compute_r(int order, double factor, double r[6][6])
{   /* bdf.rs:694-713 */
    int size = order + 1;
    double m[6][6];
    memset(m, 0, sizeof m);
    memset(r, 0, sizeof(double) * 36);
    for (int j = 0; j < size; j++) m[0][j] = 1.0;
    for (int i = 1; i < size; i++)
        for (int j = 1; j < size; j++) m[i][j] = ((double)i - 1.0 - factor * (double)j) / (double)i;
    for (int j = 0; j < size; j++) r[0][j] = m[0][j];
    for (int i = 1; i < size; i++)
        for (int j = 0; j < size; j++) r[i][j] = r[i - 1][j] * m[i][j];
}
static void change_d(double *d /* [8][n] */, int n, int order, double factor)
{   /* bdf.rs:669-692 */
    if (factor == 1.0) return;
    if (order > BDF_MAXO) order = BDF_MAXO;
    int size = order + 1;
    double r[6][6], u[6][6], ru[6][6];
    compute_r(order, factor, r);
    compute_r(order, 1.0, u);
    memset(ru, 0, sizeof ru);
    for (int i = 0; i < size; i++)          /* matmul, bdf.rs:715-732 */
        for (int k = 0; k < size; k++) {
            double coeff = r[i][k];
            if (coeff == 0.0) continue;
            for (int j = 0; j < size; j++) ru[i][j] = MA(ru[i][j], coeff, u[k][j]);
        }
    double scratch[6][ORC_MAX_N];
    for (int row = 0; row <= order; row++) {
        for (int i = 0; i < n; i++) scratch[row][i] = 0.0;
        for (int k = 0; k <= order; k++) {
            double coeff = ru[k][row];
            if (coeff == 0.0) continue;
            for (int i = 0; i < n; i++) scratch[row][i] = MA(scratch[row][i], coeff, d[k * n + i]);
        }
    }
    for (int i = 0; i <= order; i++) memcpy(d + (size_t)i * n, scratch[i], (size_t)n * sizeof(double));
}

static int bdf_solve(orc_ode_fn f, const double *p, int n, double x0, const double *y0, double xend,
                     const tol_t *rtol, const tol_t *atol, const orc_options *opt, solout_t *so,
                     int_result *res, double *y_final, double *x_final, uint64_t *njev_out, uint64_t *nlu_out)
{
    static const double KAPPA[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    const double MIN_FACTOR = 0.2, MAX_FACTOR = 10.0, SAFETY_DEFAULT = 0.9;
    const double EPS = 2.220446049250313e-16, MIN_POSITIVE = 2.2250738585072014e-308;
    double x = x0;
    for (int i = 0; i < n; i++) {   /* bdf.rs:112-128 */
        if (tol_at(rtol, i) < 0.0 || tol_at(atol, i) < 0.0) return ORC_ERR_NEGATIVE_TOLERANCE;
    }
    const uint64_t nmax = opt->has_max_steps ? opt->max_steps : UINT64_MAX;
    if (nmax == 0) return ORC_ERR_MUST_BE_POSITIVE;
    const double direction = rs_signum(xend - x);
    const double hmax = fabs(opt->has_max_step ? opt->max_step : fabs(xend - x));
    const double hmin = fabs(opt->has_min_step ? opt->min_step : 0.0);
    uint64_t nfev = 0, njev = 0, nlu = 0, nstep = 0, naccpt = 0, nrejct = 0;

    double *w = (double *)calloc((size_t)n * (8 + 8 + 7) + (size_t)n * n * 2 + 8, sizeof(double));
    double *y = w, *f0 = w + n, *psi = w + 2 * n, *scale = w + 3 * n, *y_predict = w + 4 * n, *y_new = w + 5 * n,
           *delta = w + 6 * n, *rhs = w + 7 * n, *d = w + 8 * n, *cont = w + 16 * n, *jac = w + 23 * n,
           *lu = w + 23 * n + (size_t)n * n;
    int pivot[ORC_MAX_N];
    memcpy(y, y0, (size_t)n * sizeof(double));
    f(x, y, f0, p);
    nfev += 1;
    eval_jac(opt, f, p, n, x, y, jac);
    njev += 1;
    int lu_is_current = 0;
    double current_c = 0.0;

    double gamma[6], alpha[6], error_const[6];
    gamma[0] = 0.0;
    for (int k = 1; k <= BDF_MAXO; k++) gamma[k] = gamma[k - 1] + 1.0 / (double)k;
    for (int k = 0; k <= BDF_MAXO; k++) alpha[k] = (1.0 - KAPPA[k]) * gamma[k];
    for (int k = 0; k <= BDF_MAXO; k++) error_const[k] = KAPPA[k] * gamma[k] + 1.0 / ((double)k + 1.0);

    double rtol_min = INFINITY;
    for (int i = 0; i < n; i++) rtol_min = fmin(rtol_min, tol_at(rtol, i));
    rtol_min = fmax(rtol_min, EPS);
    double newton_tol = fmax(10.0 * EPS / rtol_min, fmin(sqrt(rtol_min), 0.03));
    if (newton_tol <= 0.0) newton_tol = 1e-9;
    const int newton_maxiter = 4;

    double h_abs;
    if (opt->has_first_step) {
        if (opt->first_step == 0.0) { free(w); return ORC_ERR_INVALID_STEP_SIZE; }
        h_abs = fabs(opt->first_step);
    } else {
        double f1[ORC_MAX_N], y1[ORC_MAX_N];
        double guess = hinit(f, p, n, x, y, direction, f0, f1, y1, 1, hmax, atol, rtol);
        double max_h = fabs(xend - x);
        if (fabs(guess) > max_h) guess = max_h * direction;
        h_abs = fabs(guess);
    }
    h_abs = fmin(h_abs, fmax(hmax, MIN_POSITIVE));
    double current_h = h_abs;

    memcpy(d, y, (size_t)n * sizeof(double));
    for (int i = 0; i < n; i++) d[n + i] = f0[i] * current_h * direction;
    int order = 1, n_equal_steps = 0, status;

    so_call(so, x, x, y, NULL, 0.0);

    for (;;) {
        if (nstep >= nmax) { status = ORC_NEED_LARGER_NMAX; break; }
        if (current_h < MIN_POSITIVE) { status = ORC_STEP_SIZE_TOO_SMALL; break; }
        double h_try = current_h;
        if (h_try > hmax) {
            change_d(d, n, order, hmax / h_try);
            h_try = hmax; current_h = h_try; n_equal_steps = 0; lu_is_current = 0;
        }
        if (h_try < hmin && hmin > 0.0) {
            change_d(d, n, order, fmax(hmin / h_try, 1.0));
            h_try = hmin; current_h = h_try; n_equal_steps = 0; lu_is_current = 0;
        }
        double h_signed = direction * h_try;
        const double x_start = x;
        double x_new = x + h_signed;
        if (direction * (x_new - xend) > 0.0) {
            double step_to_end = fabs(xend - x);
            if (step_to_end == 0.0) { status = ORC_SUCCESS; break; }
            double factor = step_to_end / h_try;
            change_d(d, n, order, factor);
            current_h *= factor;
            h_try = current_h;
            h_signed = direction * h_try;
            x_new = x + h_signed;
            n_equal_steps = 0; lu_is_current = 0;
        }
        if ((x + 0.1 * fabs(h_signed)) == x) { status = ORC_STEP_SIZE_TOO_SMALL; break; }
        nstep += 1;

        for (int i = 0; i < n; i++) {
            double sum = 0.0;
            for (int k = 0; k <= order; k++) sum += d[k * n + i];
            y_predict[i] = sum;
        }
        for (int i = 0; i < n; i++) {
            scale[i] = MA(tol_at(atol, i), tol_at(rtol, i), fabs(y_predict[i]));
            if (scale[i] == 0.0) scale[i] = EPS;
        }
        for (int i = 0; i < n; i++) {
            double sacc = 0.0;
            for (int j = 1; j <= order; j++) sacc = MA(sacc, gamma[j], d[j * n + i]);
            psi[i] = sacc / alpha[order];
        }
        const double c = h_signed / alpha[order];
        if (!lu_is_current || fabs(c - current_c) / fmax(fabs(c), 1.0) > 0.1) {
            for (int r = 0; r < n; r++) {
                for (int ci = 0; ci < n; ci++) lu[r * n + ci] = -c * jac[r * n + ci];
                lu[r * n + r] = MA(1.0, -c, jac[r * n + r]);   /* -c j + 1 */
            }
            nlu += 1;
            if (lu_decomp(lu, pivot, n) == 0) { lu_is_current = 1; current_c = c; }
            else {
                change_d(d, n, order, 0.5);
                current_h *= 0.5; n_equal_steps = 0; lu_is_current = 0; nrejct += 1;
                continue;
            }
        }

        memcpy(y_new, y_predict, (size_t)n * sizeof(double));
        for (int i = 0; i < n; i++) delta[i] = 0.0;
        int converged = 0, has_prev = 0, iters = 0;
        double dy_norm_prev = 0.0;
        while (iters < newton_maxiter) {
            f(x_new, y_new, rhs, p);
            nfev += 1;
            for (int i = 0; i < n; i++) rhs[i] = MB(c, rhs[i], psi[i]) - delta[i];
            lin_solve(lu, rhs, pivot, n);
            double dy_norm = wrms_scaled(rhs, scale, n);
            int rate_condition = 0;
            if (has_prev && dy_norm_prev > 0.0) {
                double rate = dy_norm / dy_norm_prev;
                if (rate >= 1.0) rate_condition = 1;
                else {
                    double estimate = orc_pow_small_int(rate, newton_maxiter - iters) / (1.0 - rate) * dy_norm;
                    if (estimate > newton_tol) rate_condition = 1;
                }
            }
            for (int i = 0; i < n; i++) { y_new[i] += rhs[i]; delta[i] += rhs[i]; }
            if (dy_norm == 0.0) { converged = 1; break; }
            if (has_prev && dy_norm_prev > 0.0) {
                double rate = dy_norm / dy_norm_prev;
                if (rate < 1.0) {
                    double estimate = rate / (1.0 - rate) * dy_norm;
                    if (estimate < newton_tol) { converged = 1; break; }
                }
            }
            if (rate_condition) break;
            dy_norm_prev = dy_norm; has_prev = 1;
            iters += 1;
        }
        if (!converged) {
            eval_jac(opt, f, p, n, x_new, y_predict, jac);
            njev += 1;
            lu_is_current = 0;
            change_d(d, n, order, 0.5);
            current_h *= 0.5; n_equal_steps = 0; nrejct += 1;
            continue;
        }
        const double safety = SAFETY_DEFAULT * (2.0 * (double)newton_maxiter + 1.0)
                            / (2.0 * (double)newton_maxiter + (double)(iters + 1));
        for (int i = 0; i < n; i++) {
            scale[i] = MA(tol_at(atol, i), tol_at(rtol, i), fabs(y_new[i]));
            if (scale[i] == 0.0) scale[i] = EPS;
        }
        for (int i = 0; i < n; i++) rhs[i] = error_const[order] * delta[i];
        const double error_norm = wrms_scaled(rhs, scale, n);
        if (error_norm > 1.0) {
            double factor = safety * ORC_POW(error_norm, -1.0 / ((double)order + 1.0));
            factor = fmax(factor, MIN_FACTOR);
            change_d(d, n, order, factor);
            current_h *= factor; n_equal_steps = 0; nrejct += 1;
            continue;
        }
        naccpt += 1;
        n_equal_steps += 1;
        x = x_new;
        memcpy(y, y_new, (size_t)n * sizeof(double));
        for (int i = 0; i < n; i++) {
            d[(order + 2) * n + i] = delta[i] - d[(order + 1) * n + i];
            d[(order + 1) * n + i] = delta[i];
        }
        for (int k = order; k >= 0; k--)
            for (int i = 0; i < n; i++) d[k * n + i] += d[(k + 1) * n + i];
        for (int i = 0; i < n; i++) {
            double *b = cont + (size_t)i * 7;
            b[0] = d[i];
            for (int k = 0; k < BDF_MAXO; k++) b[1 + k] = (k + 1 <= order) ? d[(k + 1) * n + i] : 0.0;
            b[6] = (double)order;
        }
        /* bdf.rs:518-519: interpolant anchored at x_start, callback xold = x - h_signed */
        if (so_call2(so, x - h_signed, x, y, cont, x_start, h_signed)) { status = ORC_USER_INTERRUPT; break; }   /* bdf.rs:521-524 */
        if (direction * (x - xend) >= 0.0) { status = ORC_SUCCESS; break; }

        if (n_equal_steps >= order + 1) {
            double err_m = INFINITY, err_p = INFINITY;
            if (order > 1) {
                for (int i = 0; i < n; i++) rhs[i] = error_const[order - 1] * d[order * n + i];
                err_m = wrms_scaled(rhs, scale, n);
            }
            if (order < BDF_MAXO) {
                for (int i = 0; i < n; i++) rhs[i] = error_const[order + 1] * d[(order + 2) * n + i];
                err_p = wrms_scaled(rhs, scale, n);
            }
            double errors[3] = {err_m, error_norm, err_p}, factors[3];
            for (int idx = 0; idx < 3; idx++) factors[idx] = ORC_POW(errors[idx], -1.0 / ((double)order + (double)idx));
            int best = 0;   /* Iterator::max_by: a later element replaces unless the current max is strictly greater */
            for (int idx = 1; idx < 3; idx++) if (!(factors[best] > factors[idx])) best = idx;
            int new_order = order;
            if (best == 0 && order > 1) new_order -= 1;
            else if (best == 2 && order < BDF_MAXO) new_order += 1;
            double max_factor = 0.0;
            for (int idx = 0; idx < 3; idx++) max_factor = fmax(max_factor, factors[idx]);
            double step_factor = fmin(safety * max_factor, MAX_FACTOR);
            int old_order = order;
            change_d(d, n, new_order, step_factor);
            current_h *= step_factor;
            order = new_order;
            n_equal_steps = 0;
            lu_is_current = 0;
            if (new_order != old_order) { eval_jac(opt, f, p, n, x, y, jac); njev += 1; }
        }
    }
    res->h = direction * current_h; res->status = status;
    res->nfev = nfev; res->nstep = nstep; res->naccpt = naccpt; res->nrejct = nrejct;
    *njev_out = njev; *nlu_out = nlu;
    memcpy(y_final, y, (size_t)n * sizeof(double));
    *x_final = x;
    free(w);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * solve_ivp (src/solve/solve_ivp.rs:99-313)
 * ---------------------------------------------------------------------------------------- */
static void constant_solution(orc_solution *sol, int method, int n, double x0, const double *y0)
{   /* ContinuousOutput::constant, cont.rs:32-64 */
    int nc = ncoef_of(method);
    sol->has_dense = 1;
    sol->ncoef = nc;
    sol->nseg = 1;
    sol->seg_cont = (double *)calloc((size_t)(nc * n > 0 ? nc * n : 1), sizeof(double));
    if (method == ORC_BDF) {   /* cont.rs:44-51: per-state blocks, order marker 1 */
        for (int i = 0; i < n; i++) { sol->seg_cont[i * nc] = y0[i]; sol->seg_cont[i * nc + nc - 1] = 1.0; }
    } else {
        for (int i = 0; i < n; i++) sol->seg_cont[i] = y0[i];
    }
    sol->seg_xold = (double *)malloc(sizeof(double));
    sol->seg_h = (double *)malloc(sizeof(double));
    sol->seg_xold[0] = x0;
    sol->seg_h[0] = 1e-15;
}

static int solve_core(orc_ode_fn f, const double *params, int n, double x0, double xend, const double *y0,
                      const orc_options *opt, orc_solution *sol, double *y_final, double *x_final)
{
    memset(sol, 0, sizeof(*sol));
    sol->n = n;
    sol->ncoef = ncoef_of(opt->method);
    if (n > ORC_MAX_N) return ORC_ERR_BAD_ARGUMENT;
    if (opt->method < ORC_RK23 || opt->method > ORC_BDF || opt->method == ORC_RADAU) return ORC_ERR_BAD_ARGUMENT;

    if (fabs(xend - x0) < 1e-15) { /* solve_ivp.rs:110-145 */
        if (opt->n_eval >= 0) {
            size_t m = 0;
            for (int i = 0; i < opt->n_eval; i++) if (fabs(opt->t_eval[i] - x0) < 1e-12) m++;
            sol->t = (double *)malloc((m ? m : 1) * sizeof(double));
            sol->y = (double *)malloc((m ? m : 1) * (size_t)(n ? n : 1) * sizeof(double));
            size_t k = 0;
            for (int i = 0; i < opt->n_eval; i++) if (fabs(opt->t_eval[i] - x0) < 1e-12) {
                sol->t[k] = opt->t_eval[i];
                memcpy(sol->y + k * (size_t)n, y0, (size_t)n * sizeof(double));
                k++;
            }
            sol->len = m;
        } else {
            sol->t = (double *)malloc(sizeof(double));
            sol->y = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
            sol->t[0] = x0;
            memcpy(sol->y, y0, (size_t)n * sizeof(double));
            sol->len = 1;
        }
        if (opt->dense_output) constant_solution(sol, opt->method, n, x0, y0);
        sol->status = ORC_SUCCESS;
        if (y_final) memcpy(y_final, y0, (size_t)n * sizeof(double));
        if (x_final) *x_final = x0;
        return ORC_OK;
    }
    if (n == 0) { /* solve_ivp.rs:148-176 */
        size_t m = opt->n_eval >= 0 ? (size_t)opt->n_eval : 2;
        sol->t = (double *)malloc((m ? m : 1) * sizeof(double));
        sol->y = (double *)malloc(sizeof(double));
        if (opt->n_eval >= 0) memcpy(sol->t, opt->t_eval, m * sizeof(double));
        else { sol->t[0] = x0; sol->t[1] = xend; }
        sol->len = m;
        if (opt->dense_output) constant_solution(sol, opt->method, 0, x0, y0);
        sol->status = ORC_SUCCESS;
        if (x_final) *x_final = x0;
        return ORC_OK;
    }

    tol_t rtol = {opt->rtol, opt->rtol_len}, atol = {opt->atol, opt->atol_len};
    /* Tolerance::Vector with the wrong length panics in the reference (mod.rs:156-161 / index OOB) */
    if (opt->method != ORC_RK4 && ((rtol.len != 1 && rtol.len != n) || (atol.len != 1 && atol.len != n)))
        return ORC_ERR_TOLERANCE_SIZE_MISMATCH;

    solout_t so;
    memset(&so, 0, sizeof(so));
    so.method = opt->method; so.n = n;
    so.t_eval = opt->t_eval; so.n_eval = opt->n_eval;
    so.tol = 1e-12;
    so.collect_dense = opt->dense_output;
    so.has_first_step = opt->has_first_step; so.first_step = opt->first_step;
    so.x0 = x0;
    so.n_events = opt->events ? opt->n_events : 0;
    so.ev = opt->events;
    so.ev_params = params;
    for (int i = 0; i < ORC_MAX_EVENTS; i++) { so.ev_direction[i] = opt->ev_direction[i]; so.ev_terminal[i] = opt->ev_terminal[i]; }

    int_result r;
    double yf[ORC_MAX_N], xf = x0;
    int rc;
    if (opt->method == ORC_DOPRI5) rc = dopri5_solve(f, params, n, x0, y0, xend, &rtol, &atol, opt, &so, &r, yf, &xf);
    else if (opt->method == ORC_DOP853) rc = dop853_solve(f, params, n, x0, y0, xend, &rtol, &atol, opt, &so, &r, yf, &xf);
    else if (opt->method == ORC_BDF) { /* solve_ivp.rs:268-285 */
        uint64_t njev = 0, nlu = 0;
        rc = bdf_solve(f, params, n, x0, y0, xend, &rtol, &atol, opt, &so, &r, yf, &xf, &njev, &nlu);
        sol->njev = njev; sol->nlu = nlu;
    }
    else if (opt->method == ORC_RK4) { /* solve_ivp.rs:184-196: h = first_step or (xend - x0) / 100 */
        double h4 = opt->has_first_step ? opt->first_step : (xend - x0) / 100.0;
        rc = rk4_solve(f, params, n, x0, y0, xend, h4, opt, &so, &r, yf, &xf);
    }
    else rc = rk23_solve(f, params, n, x0, y0, xend, &rtol, &atol, opt, &so, &r, yf, &xf);
    if (rc != ORC_OK) {
        free(so.t); free(so.y); free(so.seg_cont); free(so.seg_xold); free(so.seg_h);
        for (int i = 0; i < ORC_MAX_EVENTS; i++) { free(so.t_events[i]); free(so.y_events[i]); }
        return rc;
    }
    sol->n_events = so.n_events;
    for (int i = 0; i < ORC_MAX_EVENTS; i++) { sol->ev_len[i] = so.ev_len[i]; sol->t_events[i] = so.t_events[i]; sol->y_events[i] = so.y_events[i]; }
    sol->len = so.len; sol->t = so.t; sol->y = so.y;
    sol->nfev = r.nfev;
    sol->nstep = r.nstep; sol->naccpt = r.naccpt; sol->nrejct = r.nrejct;
    sol->status = r.status;
    sol->h_next = r.h;
    if (opt->dense_output) {
        sol->has_dense = 1;
        sol->nseg = so.nseg; sol->seg_cont = so.seg_cont; sol->seg_xold = so.seg_xold; sol->seg_h = so.seg_h;
    } else {
        free(so.seg_cont); free(so.seg_xold); free(so.seg_h);
    }
    if (y_final) memcpy(y_final, yf, (size_t)n * sizeof(double));
    if (x_final) *x_final = xf;
    return ORC_OK;
}

int orc_solve_ivp(orc_ode_fn f, const double *params, int n, double x0, double xend,
                  const double *y0, const orc_options *opt, orc_solution *sol)
{
    return solve_core(f, params, n, x0, xend, y0, opt, sol, NULL, NULL);
}

void orc_solution_free(orc_solution *sol)
{
    free(sol->t); free(sol->y); free(sol->seg_cont); free(sol->seg_xold); free(sol->seg_h);
    for (int i = 0; i < ORC_MAX_EVENTS; i++) { free(sol->t_events[i]); free(sol->y_events[i]); }
    memset(sol, 0, sizeof(*sol));
}

/* ContinuousOutput::find_segment, cont.rs:100-117 */
static long find_segment(const orc_solution *sol, double t)
{
    const double tol = 1e-12;
    for (size_t s = 0; s < sol->nseg; s++) {
        double a = sol->seg_xold[s], b = sol->seg_xold[s] + sol->seg_h[s];
        double left = fmin(a, b), right = fmax(a, b);
        if (t >= left - tol && t <= right + tol) return (long)s;
    }
    return -1;
}

int orc_solution_eval(const orc_solution *sol, int method, double t, double *out)
{   /* Solution::sol, solution.rs:25-47 */
    if (!sol->has_dense || sol->nseg == 0) return -1;
    double start = sol->seg_xold[0];
    double end = sol->seg_xold[sol->nseg - 1] + sol->seg_h[sol->nseg - 1];
    double lo = fmin(start, end), hi = fmax(start, end);
    if (t < lo || t > hi) return -2;
    long s = find_segment(sol, t);
    if (s < 0) return -2;
    interp_any(method, t, out, sol->seg_cont + (size_t)s * sol->ncoef * sol->n, sol->n, sol->seg_xold[s], sol->seg_h[s]);
    return 0;
}

int orc_solution_eval_extrapolate(const orc_solution *sol, int method, double t, double *out)
{   /* cont.rs:119-153 */
    if (!sol->has_dense || sol->nseg == 0) return -1;
    long s = find_segment(sol, t);
    if (s < 0) {
        size_t L = sol->nseg - 1;
        double fl = fmin(sol->seg_xold[0], sol->seg_xold[0] + sol->seg_h[0]);
        double lr = fmax(sol->seg_xold[L], sol->seg_xold[L] + sol->seg_h[L]);
        if (t < fl) s = 0;
        else if (t > lr) s = (long)L;
        else return -2;
    }
    interp_any(method, t, out, sol->seg_cont + (size_t)s * sol->ncoef * sol->n, sol->n, sol->seg_xold[s], sol->seg_h[s]);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Batch driver
 * ---------------------------------------------------------------------------------------- */
int64_t orc_batch_solve(int rhs_id, size_t B, const double *y0, const double *params,
                        const double *t0, int t0_len, const double *t1, int t1_len,
                        const orc_options *opt, int threads,
                        double *y_end, double *t_end, int32_t *status,
                        uint64_t *nfev, uint64_t *nstep, uint64_t *naccpt, uint64_t *nrejct,
                        double *h_next, double *y_eval, int32_t *n_filled, uint64_t *njev, uint64_t *nlu)
{
    int n = 0, np = 0;
    orc_ode_fn f = orc_builtin_rhs(rhs_id, &n, &np);
    if (!f) return ORC_ERR_BAD_ARGUMENT;
    orc_options opt_j = *opt;           /* a built-in problem with an analytic jac override brings it along */
    if (!opt_j.jac) opt_j.jac = orc_builtin_jac(rhs_id);
    opt = &opt_j;
    int64_t total = 0;
    int err = 0;
    (void)threads;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : total)
#endif
    for (long long b = 0; b < (long long)B; b++) {
        double y0b[ORC_MAX_N], pb[4], yf[ORC_MAX_N], xf;
        for (int i = 0; i < n; i++) y0b[i] = y0[(size_t)i * B + (size_t)b];
        for (int i = 0; i < np; i++) pb[i] = params[(size_t)i * B + (size_t)b];
        double a = t0_len == 1 ? t0[0] : t0[b];
        double e = t1_len == 1 ? t1[0] : t1[b];
        orc_solution s;
        int rc = solve_core(f, pb, n, a, e, y0b, opt, &s, yf, &xf);
        if (rc != ORC_OK) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = rc;
            continue;
        }
        for (int i = 0; i < n; i++) y_end[(size_t)i * B + (size_t)b] = yf[i];
        t_end[b] = xf;
        status[b] = s.status;
        if (nfev) nfev[b] = s.nfev;
        if (nstep) nstep[b] = s.nstep;
        if (naccpt) naccpt[b] = s.naccpt;
        if (nrejct) nrejct[b] = s.nrejct;
        if (h_next) h_next[b] = s.h_next;
        if (njev) njev[b] = s.njev;
        if (nlu) nlu[b] = s.nlu;
        if (y_eval && opt->n_eval > 0) {
            size_t m = s.len < (size_t)opt->n_eval ? s.len : (size_t)opt->n_eval;
            for (size_t k = 0; k < m; k++)
                for (int i = 0; i < n; i++)
                    y_eval[(k * (size_t)n + (size_t)i) * B + (size_t)b] = s.y[k * (size_t)n + (size_t)i];
            if (n_filled) n_filled[b] = (int32_t)m;
        }
        total += (int64_t)s.naccpt;
        orc_solution_free(&s);
    }
    if (err) return err;
    return total;
}
