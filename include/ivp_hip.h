/*
 * ivp_hip.h -- C ABI of libivp_hip.so: the MI355X (gfx950) batched explicit Runge-Kutta integrator.
 *
 * This is the drop-in boundary for ONE path of the Rust crate Ryan-D-Gast/ivp 0.5.1: the explicit
 * RK stepping core (DOPRI5 / DOP853 / RK23 stage evaluation, weighted error norm, step-size
 * controller, initial-step heuristic) behind solve_ivp().  The reference has no C ABI of its own
 * (its only exported symbol is the PyO3 module init, src/python/mod.rs:26); the entry points below
 * are what a `#[link(name = "ivp_hip")] extern "C"` block in the crate would bind -- see
 * INTEGRATION.md for that binding.  Each declaration cites the reference interface it replaces
 * (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types;
 *   - all state is struct-of-arrays: component c of trajectory b lives at a[c*B + b];
 *   - a "trajectory" is one reference solve_ivp() call; a batch is B independent calls;
 *   - whole-call validation failures are returned as negative codes (the reference's
 *     Err(Error::Config(..)) values, src/error.rs:18-60); per-trajectory integration failures are
 *     reported only in status[b] (the reference's Ok(..) with a non-success Status,
 *     src/methods/dopri5.rs:268-277); nothing throws or aborts across this boundary;
 *   - the library never falls back to a CPU path: without a usable HIP device every compute entry
 *     point returns IVP_ERR_NO_DEVICE.
 */
#ifndef IVP_HIP_H
#define IVP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVP_HIP_ABI_VERSION 5

/* Method: same order as `enum Method`, src/solve/options.rs:14-27. The explicit RK methods (RK23, DOPRI5,
 * DOP853, the fixed-step RK4) and the variable-order implicit BDF are on the accelerated path; RADAU returns
 * IVP_ERR_UNSUPPORTED_METHOD. */
typedef enum {
    IVP_RK23 = 0,
    IVP_DOPRI5 = 1, /* "RK45" */
    IVP_DOP853 = 2,
    IVP_RK4 = 3,
    IVP_RADAU = 4,
    IVP_BDF = 5
} ivp_method_t;

/* Status: same order as `enum Status`, src/status.rs:4-19. */
typedef enum {
    IVP_STATUS_SUCCESS = 0,
    IVP_STATUS_USER_INTERRUPT = 1,
    IVP_STATUS_NEED_LARGER_NMAX = 2,
    IVP_STATUS_STEP_SIZE_TOO_SMALL = 3,
    IVP_STATUS_PROBABLY_STIFF = 4,
    IVP_STATUS_SINGULAR_MATRIX = 5,
    IVP_STATUS_POOR_CONVERGENCE = 6
} ivp_status_t;

/* Return codes. The negative config codes map 1:1 onto `enum ConfigError`, src/error.rs:18-60. */
typedef enum {
    IVP_OK = 0,
    IVP_ERR_MUST_BE_POSITIVE = -1,        /* ConfigError::MustBePositive       */
    IVP_ERR_OUT_OF_RANGE = -2,            /* ConfigError::OutOfRange           */
    IVP_ERR_NEGATIVE_TOLERANCE = -3,      /* ConfigError::NegativeTolerance    */
    IVP_ERR_TOLERANCE_SIZE_MISMATCH = -4, /* ConfigError::ToleranceSizeMismatch (a panic in the reference, src/methods/mod.rs:156-161) */
    IVP_ERR_INVALID_STEP_SIZE = -5,       /* ConfigError::InvalidStepSize (RK4: first_step zero / wrong sign, rk4.rs:81-87) */
    IVP_ERR_INVALID_SCALE_FACTORS = -6,   /* ConfigError::InvalidScaleFactors  */
    IVP_ERR_BAD_ARGUMENT = -100,          /* NULL pointer, unknown rhs id, n mismatch ...            */
    IVP_ERR_UNSUPPORTED_METHOD = -101,    /* RADAU: not on the accelerated path (every other method is, for every n) */
    IVP_ERR_NO_DEVICE = -102,             /* no HIP device: there is deliberately no CPU fallback    */
    IVP_ERR_HIP = -103,                   /* a HIP runtime call failed; see ivp_last_error_string()  */
    IVP_ERR_JIT = -104,                   /* hiprtc compilation of a user RHS failed                 */
    IVP_ERR_LOG_CAPACITY = -105           /* ivp_step_log_t: the caller's t / y hold fewer than `total` records (total is set) */
} ivp_error_t;

/* Right-hand sides that ship as device functors: the `impl IVP for ..` blocks of the reference's
 * examples, tests and benchmark (trait IVP::ode, src/ivp.rs:27-29). User-defined systems are
 * compiled at run time from a HIP source snippet with ivp_rhs_compile(). */
typedef enum {
    IVP_RHS_DECAY = 0,    /* y' = -k y                 p={k}           n=1  examples/exponential_decay.rs:9-13 */
    IVP_RHS_SHO = 1,      /* y0'=y1, y1'=-y0                           n=2  tests/common.rs:3-9                */
    IVP_RHS_VDP = 2,      /* Van der Pol               p={mu}          n=2  benches/benchmark.py:22-27         */
    IVP_RHS_CR3BP = 3,    /* restricted 3-body         p={mu}          n=6  examples/cr3bp.rs:23-36            */
    IVP_RHS_LORENZ = 4,   /* Lorenz                    p={sigma,rho,beta} n=3  benches/benchmark.py:30-37      */
    IVP_RHS_ZERO = 5,     /* y' = 0                                    n=3  tests/ivp.rs:11-19                 */
    IVP_RHS_RATIONAL = 6, /* SciPy "rational" problem                  n=2  tests/test_helpers.py:23-25        */
    IVP_RHS_EXP2 = 7,     /* y' = y                                    n=2  tests/ivp.rs:291-298               */
    IVP_RHS_LINEAR = 8,   /* y' = A y, A = [[-1,-5],[1,1]]            n=2  tests/test_helpers.py:11-12        */
    IVP_RHS_ROBERTSON = 9,/* Robertson kinetics                        n=3  tests/test_ivp.py:327-333          */
    IVP_RHS_VDP_EPS = 10, /* stiff Van der Pol        p={eps}          n=2  examples/van_der_pol.rs:9-14       */
    /* problems that carry event functions (trait IVP::events / n_events, src/ivp.rs:31-46) */
    IVP_RHS_SHO_EV = 11,  /* SHO, event g = y0                          n=2  tests/ivp.rs:151-221               */
    IVP_RHS_BALL = 12,    /* bouncing ball  p={gravity,drag}, g = height n=2  examples/bouncing_ball.rs:5-31     */
    IVP_RHS_CANNON = 13,  /* y'' = -9.80665, g = y0                     n=2  tests/test_ivp.py:152-160          */
    IVP_RHS_RATIONAL_EV = 14, /* rational problem + 3 events            n=2  tests/test_ivp.py:345-353          */
    IVP_RHS_ROBERTSON_JAC = 15, /* Robertson + analytic `jac` override (trait IVP::jac, src/ivp.rs:67-107)  n=3 */
    IVP_RHS_BUILTIN_COUNT = 16,
    /* Large state dimensions (8 < n <= 512): one 64-lane wavefront integrates one trajectory, the state is
     * distributed over its lanes and the error norm is a wavefront reduction.  RK23 / DOPRI5 / DOP853 / RK4 with
     * every output mode (t_eval, step log, dense output), events (hiprtc problems) and scalar or vector tolerances;
     * BDF with the n x n matrices J and LU = (I - cJ) per trajectory (in LDS for n <= 128, see ivp_options_t.variant). */
    IVP_RHS_LINEAR_DECAY_100 = 100, /* y' = -y                            n=100 benches/benchmark.py:40-42,139-148 */
    IVP_RHS_HEAT1D_256 = 101,       /* y_i' = k (y_{i-1} - 2 y_i + y_{i+1}), p={k}  n=256 (method of lines)  */
    IVP_RHS_DENSE_64 = 102,         /* y' = A y, A dense 64 x 64: a_ii = -k (4 + i mod 5), a_ij = (((5 i + 3 j) & 15) - 8) / 256,
                                       p={k}: a FULL Jacobian for the per-trajectory LU of BDF (n=64) */
    IVP_RHS_JIT = 1000    /* problem.jit holds a handle from ivp_rhs_compile() */
} ivp_rhs_id_t;

/* The device-side analogue of `impl IVP for T` (src/ivp.rs:27-121). */
typedef struct {
    int32_t rhs_id;   /* ivp_rhs_id_t */
    int32_t n;        /* state dimension (must match the functor's) */
    int32_t n_params; /* per-trajectory parameters (the fields of the user's struct, e.g. CR3BP{mu}) */
    void *jit;        /* ivp_rhs_compile() handle when rhs_id == IVP_RHS_JIT, else NULL */
} ivp_problem_t;

/* Floating-point mode of the kernels. */
typedef enum {
    IVP_FP_STRICT = 0, /* no FMA contraction, reference expression order: bit-comparable with a CPU
                          restatement of the reference (Rust never contracts a*b+c) */
    IVP_FP_FMA = 1,    /* the FMA arithmetic mode: the same operation sequence with the multiply-add sites of the stage
                          combinations, error estimates, dense coefficients, interpolants and tolerance scales as single
                          fused operations, and the built-in right-hand sides in their FMA form (one reciprocal per
                          denominator).  A DEFINED arithmetic (written out in the kernel source, compiled without
                          compiler contraction): results are identical in every kernel variant and for every batch
                          size, and bit-comparable with the oracle's FMA build (oracle/liboracle_fma.so).  Differs from
                          STRICT at the 1e-16-per-operation level; a hiprtc right-hand side is evaluated as written
                          (call fma() yourself where you want it) */
    IVP_FP_FAST = IVP_FP_FMA   /* ABI v3 name */
} ivp_fp_mode_t;

/* Mirrors `struct Options` (src/solve/options.rs:75-123) field by field for the fields the explicit
 * RK path reads; the remaining fields (jac_storage, mass_storage, nind1-3, min_step) belong to
 * RADAU/BDF only. Fill with ivp_options_default() first. */
typedef struct {
    int32_t method;         /* ivp_method_t; Options.method, default DOPRI5 */
    double rtol;            /* Options.rtol scalar form, default 1e-3 */
    double atol;            /* Options.atol scalar form, default 1e-6 */
    const double *rtol_vec; /* optional per-component rtol[n] (Tolerance::Vector), host pointer */
    const double *atol_vec; /* optional per-component atol[n], host pointer */
    int32_t rtol_vec_len;   /* must equal n when rtol_vec != NULL */
    int32_t atol_vec_len;
    uint64_t max_steps;     /* Options.max_steps; 0 = None = unlimited (src/solve/solve_ivp.rs:218) */
    const double *t_eval;   /* Options.t_eval: output grid shared by the batch (or B concatenated grids, see
                               t_eval_offsets below), host pointer, or NULL = None */
    int64_t n_eval;         /* number of t_eval points (ignored when t_eval == NULL) */
    int32_t has_first_step; /* Options.first_step is Some(..) */
    double first_step;
    int32_t has_max_step;   /* Options.max_step is Some(..) */
    double max_step;
    int32_t dense_output;   /* Options.dense_output: record per-step interpolants (needs max_log > 0) */
    /* event_config(i) of the problem (src/solve/event.rs:5-77), for problems whose functor defines events */
    int32_t ev_direction[4];  /* Direction: 0 All, > 0 Positive, < 0 Negative  (up to 4 events; more: *_vec below) */
    uint32_t ev_terminal[4];  /* terminal_count: 0 = None, k = interrupt at the k-th occurrence   */
    uint32_t max_events;      /* capacity of t_events / y_events per event and trajectory          */
    int32_t has_min_step;   /* Options.min_step is Some(..) (read by BDF only, src/solve/solve_ivp.rs:271) */
    double min_step;
    /* ---- knobs that exist only on the GPU path ---- */
    int32_t fp_mode;        /* ivp_fp_mode_t, default IVP_FP_STRICT */
    int32_t chunk_attempts; /* step attempts per kernel launch between compactions; 0 = auto */
    uint32_t max_log;       /* capacity (per trajectory) of the accepted-step log t_log/y_log and of the
                               dense-segment log; 0 = do not record (end state only) */
    int32_t variant;        /* stepping-kernel variant: 0 = auto, 1 = lean registers, 2 = coefficients resident,
                               3 = lane-cooperative (eight lanes per trajectory: DOPRI5 / DOP853, every output mode,
                               problems with events, built-in and hiprtc systems with n <= 8; otherwise as 0).
                               Large-n BDF (8 < n <= 128, built-in problems): 0 / 2 = factors of (I - cJ) resident in
                               LDS, 1 = in global memory.  Results do not depend on the variant in either arithmetic mode. */
    int32_t profile;        /* 1: time every kernel launch with HIP events (see ivp_run_stats_t);
                               2: additionally sum naccpt / attempts over the batch on the host */
    /* ---- direct per-method call: `DOPRI5 {..}.solve()`, `DOP853 {..}.solve()`, `RK23 {..}.solve()` ----
     * (src/methods/dopri5.rs:34-72,122-198, dop853.rs:34-63,114-193, rk23.rs:17-37,81-129).
     * has_settings == 0 (default): the struct defaults, i.e. exactly what solve_ivp() runs.
     * has_settings != 0: the fields below replace the struct's controller fields and are validated like
     * XXX::solve() validates them (OutOfRange / MustBePositive / InvalidScaleFactors); max_steps is then the struct's
     * `usize` field taken literally (0 => IVP_ERR_MUST_BE_POSITIVE).  Fill with ivp_options_method_defaults() first.
     * RK23 reads safety_factor / scale_min / scale_max only; RK4 / BDF reject has_settings. */
    int32_t has_settings;
    double uround;          /* default 2.3e-16                                   */
    double safety_factor;   /* default 0.9                                       */
    double scale_min;       /* default 0.2 (DOPRI5, RK23), 0.333 (DOP853)        */
    double scale_max;       /* default 10  (DOPRI5, RK23), 6 (DOP853)            */
    double beta;            /* default 0.04 (DOPRI5), 0 (DOP853)                 */
    uint64_t stiff_test;    /* default 1000                                      */
    int32_t count_log;      /* 1: run DefaultSolOut's accepted-step recording without storing anything: out->n_log
                               receives the number of records per trajectory (first pass of the CSR log) */
    /* ---- ABI v4 ---- */
    /* Per-trajectory output grids.  Every reference solve_ivp() call has its own Options.t_eval
     * (src/solve/options.rs:75-123); t_eval_offsets != NULL (host pointer, [B + 1], offsets[0] = 0, non-decreasing,
     * offsets[B] = n_eval) makes t_eval the concatenation of B grids: trajectory b samples
     * t_eval[offsets[b] .. offsets[b+1]).  The outputs are then time-major CSR records like Solution.y
     * (Vec<Vec<f64>>): with e = 1 for problems with event functions (a terminal event appends its own sample,
     * src/solve/solout.rs:316-319) and 0 otherwise, the k-th sample of trajectory b is record
     * q = offsets[b] + b * e + k:  y_eval[q * n + c], eval_idx[q] (index into b's OWN grid, -1 for the terminal sample);
     * y_eval holds (n_eval + B * e) * n doubles, eval_idx n_eval + B * e entries, n_filled[b] counts b's samples.
     * Every entry point takes them: ivp_batch_solve / ivp_batch_solve_multi_host copy the records back to the host arrays,
     * ivp_batch_solve_multi hands each shard its slice of the grids (offsets re-based to 0: a shard's y_eval / eval_idx
     * hold its own records) and places the shards' runs in the gathered arrays at the batch-wide offsets. */
    const uint64_t *t_eval_offsets;
    /* event_config(i) for problems with MORE than 4 event functions (trait IVP::n_events is unbounded,
     * src/ivp.rs:31-52): host arrays of n_event_cfg entries that replace ev_direction / ev_terminal above */
    const int32_t *ev_direction_vec;
    const uint32_t *ev_terminal_vec;
    int32_t n_event_cfg;
} ivp_options_t;

/* Per-trajectory results: `struct Solution` (src/solve/solution.rs:7-20) + IntegrationResult.h
 * (src/methods/mod.rs:30-39), struct-of-arrays over the batch. Any pointer may be NULL = not wanted.
 * For ivp_batch_solve() these are host pointers, for ivp_batch_solve_device() device pointers. */
typedef struct {
    double *y_end;      /* [n][B]  state at t_end (Solution.y.last())                           */
    double *t_end;      /* [B]     final x (xend on Success)                                    */
    int32_t *status;    /* [B]     ivp_status_t                                                 */
    uint64_t *nfev;     /* [B]     Solution.nfev  (njev = nlu = 0 on this path)                 */
    uint64_t *nstep;    /* [B]     Solution.nstep                                               */
    uint64_t *naccpt;   /* [B]     Solution.naccpt                                              */
    uint64_t *nrejct;   /* [B]     Solution.nrejct                                              */
    double *h_next;     /* [B]     IntegrationResult.h: the step the controller would try next  */
    /* t_eval mode (Options.t_eval = Some): Solution.t/y hold the sampled points, in order */
    double *y_eval;     /* [n_eval (+1 if the problem has events)][n][B]  k-th emitted sample   */
    int32_t *eval_idx;  /* [same rows][B]  index into t_eval of the k-th emitted sample; -1 = the terminal-event point */
    int32_t *n_filled;  /* [B]             number of emitted samples                            */
    /* accepted-step mode (Options.t_eval = None, max_log > 0): Solution.t/y, capped at max_log */
    double *t_log;      /* [max_log][B]                                                         */
    double *y_log;      /* [max_log][n][B]                                                      */
    uint32_t *n_log;    /* [B]  number of records the reference would hold (may exceed max_log) */
    /* dense_output: ContinuousOutput segments (src/solve/cont.rs:9-28), capped at max_log      */
    double *seg_cont;   /* [max_log][ncoef*n][B]  ncoef = Method::coeffs_per_state (options.rs:34-43) */
    double *seg_xold;   /* [max_log][B]                                                         */
    double *seg_h;      /* [max_log][B]                                                         */
    uint32_t *n_seg;    /* [B]                                                                  */
    /* events: Solution.t_events / Solution.y_events (src/solve/solution.rs:10-11), capped at max_events */
    double *t_events;   /* [n_events][max_events][B]                                            */
    double *y_events;   /* [n_events][max_events][n][B]                                         */
    uint32_t *n_event_hits; /* [n_events][B] occurrences detected (may exceed max_events)       */
    double *t_term;     /* [B] t_eval mode: time of the appended terminal-event sample (eval_idx = -1) */
    /* implicit methods (BDF): Solution.njev / Solution.nlu; zero for the explicit RK methods  */
    uint64_t *njev;     /* [B]                                                                  */
    uint64_t *nlu;      /* [B]                                                                  */
    /* Unbounded accepted-step log in CSR form (the reference's Solution.t / Solution.y are Vecs that grow with every
     * accepted step, src/solve/solout.rs:387-428): an INPUT, like the other members a host array for ivp_batch_solve() /
     * ivp_batch_solve_multi_host() and a device array for the device-pointer entry points.  When non-NULL it holds B+1 record
     * offsets; trajectory b's k-th record then lives at t_log[log_offsets[b] + k] and
     * y_log[(log_offsets[b] + k) * n + c] (time-major like Vec<Vec<f64>>), so the log takes sum(n_log) records
     * instead of max_log x B.  Callers that want the whole log should use ivp_batch_solve_logged*() below: ONE integration
     * that records as it goes and delivers offsets, t_log and y_log in this very layout (ABI v5).  The two-pass form stays
     * for callers that manage the offsets themselves: a counting solve (options.count_log = 1, only n_log is written), an
     * exclusive scan of n_log, then the filling solve with these offsets. */
    const uint64_t *log_offsets;
} ivp_batch_result_t;

/* What the last solve on a context did (filled when options.profile == 1). */
typedef struct {
    uint32_t launches;          /* stepping-kernel launches (chunks)                     */
    uint32_t init_launches;
    double step_kernel_ms;      /* sum of HIP-event durations of the stepping kernels    */
    double init_kernel_ms;
    double total_ms;            /* first launch -> last kernel done, on the stream       */
    uint64_t total_accepted;    /* sum over the batch of naccpt                          */
    uint64_t total_attempts;    /* sum over the batch of step attempts                   */
    uint64_t lane_attempt_slots;/* sum over launches of (lanes launched x attempts the wave ran): divergence accounting */
    uint64_t lane_launches;     /* sum over launches of trajectories that loaded + stored their state */
    uint32_t coop_launches;     /* of `launches`: lane-cooperative tail launches (variant 3 / auto policy) */
    double coop_kernel_ms;      /* of `step_kernel_ms`: time in those launches                            */
    /* ABI v4: (bulk, cooperative) launch pairs -- exactly one launch of a pair works, the other returns at once */
    uint32_t declined_launches;      /* pair halves that declined (not counted in `launches`)            */
    uint32_t declined_coop_launches; /* of those: cooperative-kernel halves                              */
    double declined_ms;              /* their HIP-event durations (included in step_kernel_ms)           */
    double declined_coop_ms;         /* of that: cooperative-kernel halves (included in coop_kernel_ms)  */
} ivp_run_stats_t;

typedef struct ivp_ctx ivp_ctx_t;

/* Library / device queries. */
int ivp_abi_version(void);
int ivp_device_count(void);

/* A context owns the device scratch of one (device, stream) pair; it is not thread-safe, distinct
 * contexts may be used concurrently (the reference is re-entrant and single-threaded per call,
 * src/ivp.rs:27). */
int ivp_ctx_create(ivp_ctx_t **ctx, int device);
void ivp_ctx_destroy(ivp_ctx_t *ctx);
const char *ivp_last_error_string(const ivp_ctx_t *ctx);
int ivp_ctx_get_stats(const ivp_ctx_t *ctx, ivp_run_stats_t *stats);

/* Options::builder().build() defaults (src/solve/options.rs:75-123). */
void ivp_options_default(ivp_options_t *opt);

/* `DOPRI5::default()` / `DOP853::default()` / `RK23::default()` (src/methods/dopri5.rs:34-72, dop853.rs:34-81,
 * rk23.rs:17-50): sets method and fills uround .. stiff_test with that struct's defaults (has_settings stays as it is). */
int ivp_options_method_defaults(ivp_options_t *opt, int32_t method);

/* Dimension lookup for built-in right-hand sides; ivp_rhs_n_events = IVP::n_events() (src/ivp.rs:42-46). */
int ivp_rhs_dims(int32_t rhs_id, int32_t *n, int32_t *n_params);
int ivp_rhs_n_events(int32_t rhs_id);

/*
 * B independent solve_ivp() calls (src/solve/solve_ivp.rs:99-108:
 *   solve_ivp(f, x0, xend, y0, options) -> Result<Solution, Error>), host buffers.
 *   y0[n][B], params[n_params][B] (may be NULL when n_params == 0),
 *   t0/t1: [B] when *_len == B, or a single shared value when *_len == 1.
 * Returns IVP_OK or a negative ivp_error_t.
 */
int ivp_batch_solve(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                    const double *params, const double *t0, size_t t0_len, const double *t1,
                    size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out);

/*
 * Same, with every array argument (y0, params, t0, t1 and all `out` members) already resident in
 * device memory; work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and the
 * call returns after the stream has drained the integration (it has to poll the active count).
 * opt->t_eval / rtol_vec / atol_vec stay host pointers.  y_end may alias y0.
 */
int ivp_batch_solve_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                           const double *params, const double *t0, size_t t0_len, const double *t1,
                           size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out,
                           void *hip_stream);

/*
 * The same solve (src/solve/solve_ivp.rs:99-108, B calls) as a resumable operation, so that one host thread can keep several contexts (= several batches,
 * each on its own stream) in flight: a solve is a sequence of rounds (a few kernel launches, then the count of
 * still-running trajectories travels to the host), and only the hand-over between rounds needs the host.
 *   ivp_batch_submit_device  validates, enqueues the init kernel and the first round, returns without waiting;
 *                            arguments as for ivp_batch_solve_device; one solve in flight per context
 *   ivp_batch_poll           non-blocking: if the current round has finished, starts the next one or completes the
 *                            solve; *done = 1 once the results are final (also when nothing is in flight)
 *   ivp_batch_wait           blocks until the solve is complete
 * ivp_batch_solve_device() is submit + wait.  A failed poll / wait abandons the solve.
 */
int ivp_batch_submit_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                            const double *params, const double *t0, size_t t0_len, const double *t1,
                            size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out,
                            void *hip_stream);
int ivp_batch_poll(ivp_ctx_t *ctx, int *done);
int ivp_batch_wait(ivp_ctx_t *ctx);

/*
 * Solution.t / Solution.y of B solve_ivp() calls in ONE call and ONE integration (ABI v5).
 *
 * Without Options.t_eval the reference records every accepted step while it integrates (DefaultSolOut mode 2,
 * src/solve/solout.rs:387-428) and hands the two growing Vecs back as Solution.t / Solution.y
 * (src/solve/solve_ivp.rs:288-312).  The entry points below do the same for a batch: the stepping kernels append each
 * accepted step to pages drawn from a device pool (one page per wavefront and 32 record slots, the records of one attempt
 * side by side, chained per trajectory), and once every trajectory has finished a gather kernel lays the records out as a
 * CSR log in trajectory order:
 *     trajectory b's k-th record:  t[offsets[b] + k],  y[(offsets[b] + k) * n + c]      (time-major like Vec<Vec<f64>>)
 * with offsets[B] = total.  `out` takes the end-state members / statistics as in ivp_batch_solve (its t_log / y_log /
 * log_offsets are ignored; n_log, if given, receives the counts); opt->t_eval must be NULL.  Everything else about the
 * solve (events, first_step enforcement, dense_output segments, every method and state dimension) is unchanged, and the
 * records are bit-identical to those of the dense [max_log] log and of the counted two-pass CSR log above.
 *
 * ivp_step_log_t -- who owns what:
 *   offsets   [B + 1], ALWAYS the caller's (device memory for the *_device / *_multi forms, host memory for the host form)
 *   t, y      the caller's buffers of `capacity` records, or both NULL: the library allocates exactly `total` records
 *             (hipMalloc on the context's / gather device, malloc for the host form), sets owned = 1, and the caller
 *             releases them with ivp_step_log_free() -- the C rendering of a Vec the callee returns
 *   reserve   in, optional: expected total number of records; sizes the page pool.  0 = automatic (the previous logged
 *             solve of this batch size on the context, else 1024 records per trajectory and at least 256 MB, never more
 *             than half of the free device memory).  A pool that runs dry is not an error: the solve is repeated as the counted fill pass
 *             (passes = 2), so a log is never truncated
 *   defer     in: 1 = integrate and count only (total, offsets and `out` are final on return); the records stay in the
 *             context's pool until its next solve and are fetched with ivp_step_log_fetch_device() into buffers the
 *             caller sizes from `total`
 * Returns IVP_ERR_LOG_CAPACITY (with total and offsets set) when the caller's t / y are too small; the records are then
 * still in the pool: enlarge the buffers and call ivp_step_log_fetch_device().
 */
typedef struct {
    uint64_t *offsets;
    double *t;
    double *y;
    uint64_t capacity;
    uint64_t reserve;
    int32_t defer;
    /* ---- out ---- */
    int32_t owned;        /* 1: t / y were allocated by the library                                           */
    int32_t device;       /* HIP device of owned device memory, -1 for host memory                             */
    uint32_t passes;      /* integrations it took: 1 (page pool) or 2 (the pool ran dry: counted fill pass)    */
    uint64_t total;       /* number of records = offsets[B]                                                    */
    uint64_t pool_bytes;       /* size of the page pool during the solve / bytes of it the pages took (records + the slots  */
    uint64_t pool_used_bytes;  /* of rejected attempts and retired lanes + column headers)                                  */
    uint32_t page_slots;       /* record slots per trajectory and page                                                      */
} ivp_step_log_t;

int ivp_batch_solve_logged_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                                  const double *params, const double *t0, size_t t0_len, const double *t1,
                                  size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out,
                                  ivp_step_log_t *log, void *hip_stream);
/* the records of the context's last logged solve (defer = 1, or after IVP_ERR_LOG_CAPACITY) into log->t / log->y
 * (device memory, `capacity` records; both NULL: allocated here); log->offsets is not touched */
int ivp_step_log_fetch_device(ivp_ctx_t *ctx, ivp_step_log_t *log, void *hip_stream);
/* host pointers throughout (arguments as ivp_batch_solve; log->offsets / t / y are host memory) */
int ivp_batch_solve_logged(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                           const double *params, const double *t0, size_t t0_len, const double *t1,
                           size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out, ivp_step_log_t *log);
void ivp_step_log_free(ivp_step_log_t *log);

/*
 * One batch over several devices.  The reference has no parallelism (a batch is B back-to-back solve_ivp() calls,
 * src/solve/solve_ivp.rs:99-313, with no coupling between them), so the batch shards by trajectory range: shard k is
 * integrated by its own context on its own device with no communication, and the only data movement is the final
 * gather of the result arrays.  ONE host thread drives all contexts through ivp_batch_submit_device / ivp_batch_poll.
 *
 * ivp_shard_t: trajectories [first, first + count) of a batch of B; every pointer is device memory on
 * ctx's device with SoA stride `count` (component c of the shard's i-th trajectory at a[c*count + i]); `out` members
 * may be NULL like in ivp_batch_solve_device.  count == 0 is allowed (the shard is skipped).
 *
 * ivp_batch_solve_multi: integrates all shards concurrently; when `gathered` is not NULL every member that is
 * non-NULL both in `gathered` and in a shard's `out` is copied into `gathered` (device memory on `gather_device`,
 * stride B) at column offset `first` -- same-device copies for shards that live on gather_device,
 * hipMemcpyPeerAsync / peer-enabled 2-D copies (xGMI) otherwise -- and the call returns when everything has landed.
 * Two contexts on ONE device are legal (the degenerate case the single-GPU tests run).
 * What travels is the reference's whole `Solution` (src/solve/solution.rs:7-20), not only the end states: the t_eval
 * samples (y_eval / eval_idx / n_filled), event records and dense [max_log] logs are SoA members like y_end; a CSR
 * step log (out.log_offsets / t_log / y_log per shard, offsets starting at 0 in the shard's own buffers: run the
 * counting pass first, options.count_log = 1, with out.n_log gathered) is gathered into gathered->t_log /
 * gathered->y_log in trajectory order, and gathered->log_offsets[B + 1] (device memory, written here) holds the shard
 * offsets re-based by the records of the shards before them.
 *
 * ivp_batch_solve_multi_host: host-pointer convenience form (arguments as ivp_batch_solve): splits the batch into
 * n_ctx contiguous balanced shards (the first B % n_ctx shards get one more), stages each to its context's device,
 * integrates concurrently and writes the results back into the caller's host arrays.
 */
typedef struct {
    ivp_ctx_t *ctx;
    size_t first, count;
    const double *y0;         /* [n][count]        */
    const double *params;     /* [n_params][count] */
    const double *t0;         /* [count] or [1]    */
    size_t t0_len;
    const double *t1;
    size_t t1_len;
    ivp_batch_result_t out;   /* stride count      */
    void *hip_stream;         /* stream on ctx's device, NULL = its default stream */
} ivp_shard_t;

int ivp_batch_solve_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B,
                          const ivp_options_t *opt, int32_t gather_device, ivp_batch_result_t *gathered);
int ivp_batch_solve_multi_host(ivp_ctx_t *const *ctxs, int32_t n_ctx, const ivp_problem_t *prob, size_t B,
                               const double *y0, const double *params, const double *t0, size_t t0_len,
                               const double *t1, size_t t1_len, const ivp_options_t *opt, ivp_batch_result_t *out);
/*
 * ivp_batch_solve_multi + the one-pass step log (BASELINE config C4's "gather of sol.y": Solution.y is every accepted
 * step, src/solve/solution.rs:7-20): every shard integrates ONCE into its own context's page pool; afterwards each
 * shard's chains are laid out as one contiguous run of the batch-wide CSR log on gather_device -- directly by the
 * gather kernel for shards that live there, through a staging buffer and a peer copy (xGMI) otherwise -- in trajectory
 * order, and log->offsets [B + 1] (device memory on gather_device) holds the batch-wide offsets.  `gathered` takes the
 * SoA members as in ivp_batch_solve_multi (may be NULL); log->t / log->y as in ivp_batch_solve_logged_device
 * (gather_device memory; NULL = allocated there).  A shard whose pool ran dry repeats its solve as the counted fill pass.
 */
int ivp_batch_solve_logged_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B,
                                 const ivp_options_t *opt, int32_t gather_device, ivp_batch_result_t *gathered,
                                 ivp_step_log_t *log);
/* log->defer = 1 in the call above: the records of every shard into log->t / log->y now (same shards, problem, options) */
int ivp_step_log_fetch_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B,
                             const ivp_options_t *opt, int32_t gather_device, ivp_step_log_t *log);

/*
 * User-defined right-hand side: the device-side `impl IVP for T { fn ode(..) }` (src/ivp.rs:29).
 * `ode_source` is HIP device code defining
 *     __device__ void ode(double x, const double* y, double* dydx, const double* p);
 * for state dimension n <= 8 and n_params <= 16 parameters (the fields of the user's struct, one value of each per
 * trajectory); it is compiled with hiprtc for the context's
 * device and instantiates the same stepping kernels as the built-ins.  Free with ivp_rhs_free().
 * For 8 < n <= 512 (wave-per-trajectory kernels, see IVP_RHS_LINEAR_DECAY_100) the snippet defines the
 * component form instead -- y points at the whole state, the function returns dy_i/dx:
 *     __device__ double ode_comp(int i, double x, const double* y, const double* p);
 * (its events(), if any, keep the whole-state signature below).
 * Compiled code objects are cached on disk when the environment variable IVP_JIT_CACHE_DIR names an existing
 * directory (key: hash of the generated source, options and hiprtc version).
 * ivp_rhs_compile_events: the snippet additionally defines the trait's event functions
 *     __device__ void events(double x, const double* y, double* g, const double* p);   // g[0..n_events)
 * ivp_rhs_compile_ex: flags & IVP_RHS_HAS_JAC -- the snippet also overrides the trait's Jacobian (src/ivp.rs:67-107;
 * used by BDF instead of the default forward differences).  n <= 8: the whole matrix, row-major j[row*n + col],
 *     __device__ void jac(double x, const double* y, double* j, const double* p);
 * 8 < n <= 512 (one wavefront per trajectory): column form -- write column `col` of dF/dy into column[0..n),
 *     __device__ void jac_col(int col, double x, const double* y, double* column, const double* p);
 * Either way the storage starts zeroed and persists between calls like the reference's Matrix (bdf.rs:152): entries an
 * override never writes are 0.
 */
#define IVP_RHS_HAS_JAC 1u
int ivp_rhs_compile(ivp_ctx_t *ctx, const char *ode_source, int32_t n, int32_t n_params, void **handle);
int ivp_rhs_compile_events(ivp_ctx_t *ctx, const char *source, int32_t n, int32_t n_params, int32_t n_events, void **handle);
int ivp_rhs_compile_ex(ivp_ctx_t *ctx, const char *source, int32_t n, int32_t n_params, int32_t n_events, uint32_t flags, void **handle);
void ivp_rhs_free(void *handle);

#ifdef __cplusplus
}
#endif
#endif /* IVP_HIP_H */
