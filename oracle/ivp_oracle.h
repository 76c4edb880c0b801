/*
 * ivp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the explicit Runge-Kutta path of the reference crate
 * Ryan-D-Gast/ivp 0.5.1 (Rust), one trajectory per call exactly like the
 * reference.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.  The product (ivp_amd/, libivp_hip.so) never
 * links, imports or calls anything in oracle/.
 *
 * Pinning status: the reference cannot be compiled here (no Rust toolchain) and
 * holds NO golden step-sequence vectors for this path, so
 * the oracle is pinned by (i) every known-answer / tolerance assertion the
 * reference's own tests make for the explicit-RK path, restated in
 * tests/test_oracle_reference_pins.py, (ii) analytic solutions and (iii)
 * independent SciPy fixtures under tests/golden/.
 *
 * Reference files followed (relative to the reference tree):
 *   src/methods/mod.rs:217-281      hinit
 *   src/methods/dopri5.rs:122-520   DOPRI5::solve, interpolate, tableau
 *   src/methods/dop853.rs:114-848   DOP853::solve, interpolate, tableau
 *   src/methods/rk23.rs:81-347      RK23::solve, interpolate, tableau
 *   src/methods/rk4.rs:64-257       RK4::solve, interpolate, tableau
 *   src/methods/bdf.rs:86-732       BDF::solve, interpolate, change_d ("next" row)
 *   src/matrix/lu.rs:37-125, src/matrix/linear.rs:55-96   lu_decomp / lin_solve
 *   src/ivp.rs:67-107               default finite-difference Jacobian
 *   src/solve/solve_ivp.rs:99-313   solve_ivp front end
 *   src/solve/solout.rs:127-431     DefaultSolOut (dense collection, t_eval, step record)
 *   src/solve/cont.rs:16-153        ContinuousOutput
 *   src/solve/solution.rs:25-79     Solution::sol
 *   src/status.rs:4-19              Status
 */
#ifndef IVP_ORACLE_H
#define IVP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Method enum order follows src/solve/options.rs:14-27. */
enum { ORC_RK23 = 0, ORC_DOPRI5 = 1, ORC_DOP853 = 2, ORC_RK4 = 3, ORC_RADAU = 4 /* not restated */, ORC_BDF = 5 };

/* Status order follows src/status.rs:4-19. */
enum {
    ORC_SUCCESS = 0,
    ORC_USER_INTERRUPT = 1,
    ORC_NEED_LARGER_NMAX = 2,
    ORC_STEP_SIZE_TOO_SMALL = 3,
    ORC_PROBABLY_STIFF = 4,
    ORC_SINGULAR_MATRIX = 5,
    ORC_POOR_CONVERGENCE = 6
};

/* Built-in right-hand sides (the `impl IVP` blocks of the reference's examples/tests). */
enum {
    ORC_RHS_DECAY = 0,    /* examples/exponential_decay.rs:9-13   p={k}          n=1 */
    ORC_RHS_SHO = 1,      /* tests/common.rs:3-9                                 n=2 */
    ORC_RHS_VDP = 2,      /* benches/benchmark.py:22-27           p={mu}         n=2 */
    ORC_RHS_CR3BP = 3,    /* examples/cr3bp.rs:23-36              p={mu}         n=6 */
    ORC_RHS_LORENZ = 4,   /* benches/benchmark.py:30-37           p={s,r,b}      n=3 */
    ORC_RHS_ZERO = 5,     /* tests/ivp.rs:11-19                                  n=3 */
    ORC_RHS_RATIONAL = 6, /* tests/test_helpers.py:23-25                         n=2 */
    ORC_RHS_EXP2 = 7,     /* tests/ivp.rs:291-298                                n=2 */
    ORC_RHS_LINEAR = 8,   /* tests/test_helpers.py:11-12                         n=2 */
    ORC_RHS_ROBERTSON = 9,/* tests/test_ivp.py:327-333                           n=3 */
    ORC_RHS_VDP_EPS = 10, /* examples/van_der_pol.rs:9-14         p={eps}        n=2 */
    ORC_RHS_SHO_EV = 11,  /* tests/ivp.rs:151-221: SHO + event g = y0           n=2 */
    ORC_RHS_BALL = 12,    /* examples/bouncing_ball.rs            p={g,drag}     n=2, event g = y0 */
    ORC_RHS_CANNON = 13,  /* tests/test_ivp.py:152-170                           n=2, event g = y0 */
    ORC_RHS_RATIONAL_EV = 14, /* tests/test_ivp.py:345-353: rational + 3 events  n=2 */
    ORC_RHS_ROBERTSON_JAC = 15, /* Robertson with the analytic IVP::jac override (src/ivp.rs:67-107)  n=3 */
    ORC_RHS_COUNT = 16,
    /* large-n problems (wave-per-trajectory kernels on the GPU side) */
    ORC_RHS_LINEAR_DECAY_100 = 100, /* benches/benchmark.py:40-42,139-148             n=100 */
    ORC_RHS_HEAT1D_256 = 101,       /* y_i' = kappa (y_{i-1} - 2 y_i + y_{i+1}), p={kappa}  n=256 */
    ORC_RHS_DENSE_64 = 102          /* y' = A y, dense 64 x 64 (no reference counterpart: a full Jacobian for BDF's LU), p={k} */
};

typedef void (*orc_ode_fn)(double x, const double *y, double *dydx, const double *p);
typedef void (*orc_event_fn)(double x, const double *y, double *g, const double *p);   /* IVP::events, src/ivp.rs:31-40 */
typedef void (*orc_jac_fn)(double x, const double *y, double *jac, const double *p);   /* IVP::jac override, jac[row*n+col] */
#define ORC_MAX_EVENTS 16
#define ORC_MAX_N 512   /* largest state dimension the fixed-size work arrays accept */

typedef struct {
    int method;           /* ORC_RK23 / ORC_DOPRI5 / ORC_DOP853 */
    const double *rtol;   /* scalar (len 1) or per-component (len n): Tolerance, mod.rs:104-214 */
    int rtol_len;
    const double *atol;
    int atol_len;
    int has_max_steps;    /* Options.max_steps: None => usize::MAX (solve_ivp.rs:218) */
    uint64_t max_steps;
    const double *t_eval; /* Options.t_eval */
    int n_eval;           /* < 0 => None */
    int has_first_step;
    double first_step;
    int has_max_step;
    double max_step;
    int dense_output;     /* Options.dense_output: collect per-step interpolants */
    int has_min_step;     /* Options.min_step (BDF only) */
    double min_step;
    /* events (trait IVP::events / n_events / event_config, src/ivp.rs:31-52, src/solve/event.rs) */
    orc_event_fn events;  /* NULL = none */
    int n_events;
    int ev_direction[ORC_MAX_EVENTS];      /* 0 All, >0 Positive, <0 Negative */
    uint64_t ev_terminal[ORC_MAX_EVENTS];  /* EventConfig.terminal_count, 0 = None */
    /* Oracle-only guard (not in the reference): stop after this many step attempts
     * with ORC_NEED_LARGER_NMAX.  0 => no guard.  Needed because RK23 in the reference
     * never terminates when the error estimate is NaN (rk23.rs:300-306 leaves h unchanged). */
    uint64_t attempt_guard;
    /* Per-method struct fields for a direct `DOPRI5{..}.solve()` / `DOP853{..}.solve()` / `RK23{..}.solve()` call
     * (dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37).  has_settings == 0: the struct defaults, which is what
     * solve_ivp() uses.  RK23 reads safety_factor / scale_min / scale_max only. */
    int has_settings;
    double uround, safety_factor, scale_min, scale_max, beta;
    uint64_t stiff_test;
    /* trait IVP::jac (src/ivp.rs:67-107): NULL = the default forward-difference implementation; else the user's
     * override, called wherever BDF calls f.jac() (bdf.rs: start, Newton failure, order change) */
    orc_jac_fn jac;
} orc_options;

/* Solution (src/solve/solution.rs:7-20) minus events. Owned by the library; free with orc_solution_free. */
typedef struct {
    size_t len;        /* number of samples */
    double *t;         /* [len] */
    double *y;         /* [len][n] time-major */
    uint64_t nfev, njev, nlu, nstep, naccpt, nrejct;
    int status;
    double h_next;     /* IntegrationResult.h */
    /* continuous_sol (src/solve/cont.rs): segments of (cont[ncoef*n], xold, h) */
    int has_dense;
    int ncoef;         /* Method::coeffs_per_state */
    int n;
    size_t nseg;
    double *seg_cont;  /* [nseg][ncoef*n] */
    double *seg_xold;  /* [nseg] */
    double *seg_h;     /* [nseg] */
    /* Solution.t_events / y_events */
    int n_events;
    size_t ev_len[ORC_MAX_EVENTS];
    double *t_events[ORC_MAX_EVENTS];   /* [ev_len[i]] */
    double *y_events[ORC_MAX_EVENTS];   /* [ev_len[i]][n] */
} orc_solution;

/* Error codes for whole-call validation failures (Error::Config, src/error.rs:18-60). */
enum {
    ORC_OK = 0,
    ORC_ERR_MUST_BE_POSITIVE = -1,
    ORC_ERR_OUT_OF_RANGE = -2,
    ORC_ERR_NEGATIVE_TOLERANCE = -3,
    ORC_ERR_TOLERANCE_SIZE_MISMATCH = -4,
    ORC_ERR_INVALID_STEP_SIZE = -5,
    ORC_ERR_INVALID_SCALE_FACTORS = -6,
    ORC_ERR_BAD_ARGUMENT = -100
};

orc_ode_fn orc_builtin_rhs(int rhs_id, int *n_out, int *n_params_out);
orc_event_fn orc_builtin_events(int rhs_id, int *n_events);
orc_jac_fn orc_builtin_jac(int rhs_id);   /* analytic Jacobian override of a built-in problem, or NULL */

/* solve_ivp (src/solve/solve_ivp.rs:99-313) for one trajectory. Returns ORC_OK or a negative error. */
int orc_solve_ivp(orc_ode_fn f, const double *params, int n, double x0, double xend,
                  const double *y0, const orc_options *opt, orc_solution *sol);

void orc_solution_free(orc_solution *sol);

/* Solution::sol (solution.rs:25-47): returns 0 and fills out[n]; -1 not enabled; -2 out of range. */
int orc_solution_eval(const orc_solution *sol, int method, double t, double *out);
/* ContinuousOutput::evaluate_extrapolate (cont.rs:90-98). */
int orc_solution_eval_extrapolate(const orc_solution *sol, int method, double t, double *out);

/*
 * Batch driver: B back-to-back solve_ivp calls (what "a batch" means for the reference,
 * SURVEY.md section 3.5), optionally spread over OpenMP threads.  SoA in/out like the GPU ABI:
 *   y0[n][B], params[p][B], t0[B or 1], t1[B or 1]  ->  y_end[n][B], t_end[B], status[B],
 *   nfev/nstep/naccpt/nrejct[B], h_next[B]; optional y_eval[n_eval][n][B] + n_filled[B].
 * Returns total accepted steps, or a negative error code.
 */
int64_t orc_batch_solve(int rhs_id, size_t B, const double *y0, const double *params,
                        const double *t0, int t0_len, const double *t1, int t1_len,
                        const orc_options *opt, int threads,
                        double *y_end, double *t_end, int32_t *status,
                        uint64_t *nfev, uint64_t *nstep, uint64_t *naccpt, uint64_t *nrejct,
                        double *h_next, double *y_eval, int32_t *n_filled, uint64_t *njev, uint64_t *nlu);

/* Step-controller power function. Default build: libm pow (what Rust's f64::powf calls on
 * Linux). Built with -DORC_DETPOW the oracle uses orc_detpow instead: a portable, branch-light
 * exp2(e*log2(x)) that the GPU kernels restate operation by operation, which makes the
 * strict-FP GPU path comparable BIT FOR BIT (see tests/test_parity_bitexact.py). */
double orc_detpow(double x, double e);
int orc_uses_detpow(void);
int orc_uses_fma(void);      /* 1 in liboracle_fma.so: the multiply-add sites are fused (the kernels' FMA arithmetic mode) */

#ifdef __cplusplus
}
#endif
#endif
