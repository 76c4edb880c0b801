// ivp_kargs.h -- kernel argument block shared by the host library and the gfx950 kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define IVP_MAX_N 8        // largest state dimension a thread-per-trajectory kernel keeps in VGPRs
#define IVP_MAX_P 16
#define IVP_WAVE 64        // CDNA wavefront
#define IVP_RUNNING (-1)   // status[] sentinel while a trajectory is still being integrated

// flags[] bit layout (per trajectory, persists between chunk launches)
#define IVP_F_LAST      0x1u          // `last`   (dopri5.rs:211)
#define IVP_F_REJECT    0x2u          // `reject` (dopri5.rs:212)
#define IVP_F_FIRSTOUT  0x4u          // DefaultSolOut.first_output_done (solout.rs:53)
#define IVP_F_IASTI_SHIFT 4           // iasti   0..15  (dopri5.rs:215)
#define IVP_F_NONSTIFF_SHIFT 8        // nonstiff 0..6  (dopri5.rs:213)
#define IVP_F_STIFFCTR_SHIFT 12       // bits 12..31: naccpt mod nstiff for nstiff < 2^20, replaces the 64-bit modulo

struct IvpKArgs {
    // ---- batch geometry ----
    uint32_t B;               // trajectories in the batch = stride of every SoA array
    // ---- inputs (init kernel) ----
    const double *y0;         // [N][B]
    const double *params;     // [P][B]
    const double *t0;         // [B] or [1]
    const double *t1;         // [B] or [1]
    uint32_t t0_stride;       // 1 = per trajectory, 0 = shared
    uint32_t t1_stride;
    // ---- options (uniform -> SGPRs) ----
    double rtol[IVP_MAX_N];
    double atol[IVP_MAX_N];
    const double *rtol_dev;   // large-n problems with per-component tolerances: [n] in device memory, else NULL
    const double *atol_dev;
    double first_step;
    double max_step;
    uint64_t nmax;            // Options.max_steps or UINT64_MAX
    int32_t has_first_step;
    int32_t has_max_step;
    // step-size controller settings: the method struct's fields (dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37).
    // Always filled (struct defaults unless ivp_options_t.has_settings); read by the CTL = true kernels and by the
    // wave-per-trajectory kernels, the CTL = false kernels have the defaults compiled in.
    double ctl_uround;
    double ctl_safety;        // safety_factor
    double ctl_facc1;         // 1 / scale_min   (DOPRI5, DOP853)
    double ctl_facc2;         // 1 / scale_max
    double ctl_beta;
    double ctl_expo1;         // 0.2 - 0.75 beta (DOPRI5), 1/8 - 0.2 beta (DOP853)
    double ctl_scale_min;     // RK23 uses the factors themselves
    double ctl_scale_max;
    uint64_t ctl_nstiff;      // stiff_test
    int32_t has_ctl;          // ivp_options_t.has_settings
    // ---- persistent per-trajectory state (doubles as the result arrays) ----
    double *y;                // [N][B]  current state; y_end on exit
    double *k1;               // [N][B]  FSAL derivative at (x, y)
    double *x;                // [B]     t_end on exit
    double *h;                // [B]     h_next on exit
    double *facold;           // [B]
    double *hlamb;            // [B]
    uint32_t *flags;          // [B]
    int32_t *status;          // [B]     IVP_RUNNING until the trajectory retires
    uint64_t *nfev, *nstep, *naccpt, *nrejct;  // [B]
    // ---- active-set compaction ----
    const uint32_t *perm_in;  // [count_in] trajectory ids to process; NULL = identity over B
    const uint32_t *count_in; // device scalar; ignored when perm_in == NULL
    uint32_t *perm_out;       // ids still running after this launch (appended wave by wave)
    uint32_t *count_out;      // device scalar, zeroed by the host before the launch
    uint32_t chunk;           // step attempts per launch
    // ---- DefaultSolOut outputs (FULL kernels only) ----
    const double *t_eval;     // [n_eval] shared grid, device
    int32_t n_eval;           // < 0: Options.t_eval == None
    double *y_eval;           // [n_eval][N][B]
    int32_t *eval_idx;        // [n_eval][B]
    int32_t *n_filled;        // [B] emitted samples so far
    int32_t *next_idx;        // [B] DefaultSolOut.next_idx
    const unsigned long long *teval_off;  // per-trajectory grids (every reference call has its own Options.t_eval): [B+1] offsets into
                              // t_eval; trajectory j samples t_eval[teval_off[j] .. teval_off[j+1]) and its k-th emitted sample is
                              // record q = teval_off[j] + j * teval_extra + k of y_eval [..][N] / eval_idx [..] (time-major CSR);
                              // NULL = one grid of n_eval points shared by the batch, y_eval [n_eval][N][B]
    uint32_t teval_extra;     // spare records per trajectory in the CSR outputs (1 for problems with events: the terminal sample)
    uint32_t max_log;
    double *t_log;            // [max_log][B]
    double *y_log;            // [max_log][N][B]
    uint32_t *n_log;          // [B]
    double *t_last;           // [B] last recorded t (dedupe test, solout.rs:424)
    const unsigned long long *log_off;  // CSR step log: [B+1] record offsets; then t_log is [total] and y_log [total][N]
                                        // (time-major like the reference's Vec<Vec<f64>>); NULL = dense [max_log][..][B]
    int32_t collect_dense;
    double *seg_cont;         // [max_log][ncoef*N][B]
    double *seg_xold;         // [max_log][B]
    double *seg_h;            // [max_log][B]
    uint32_t *n_seg;          // [B]
    // ---- events (trait IVP::events / event_config; FULL kernels of problems with NE > 0) ----
    int32_t ev_direction[4];  // 0 All, > 0 Positive, < 0 Negative (event.rs:59-77)
    uint32_t ev_terminal[4];  // EventConfig.terminal_count, 0 = None
    const int32_t *ev_direction_dev;   // [NE] in device memory for problems with more than 4 event functions, else NULL
    const uint32_t *ev_terminal_dev;
    uint32_t max_events;      // capacity of t_events / y_events per event and trajectory
    double *t_events;         // [NE][max_events][B]
    double *y_events;         // [NE][max_events][N][B]
    uint32_t *n_ev;           // [NE][B]  event_hits
    double *prev_event;       // [NE][B]
    double *t_term;           // [B] time of the terminal-event sample in t_eval mode (eval_idx = -1)
    // ---- BDF (variable-order implicit) state ----
    double min_step;          // Options.min_step
    int32_t has_min_step;
    double *bdf_d;            // [8*N][B]   differences array D (bdf.rs:220)
    double *bdf_jac;          // [N*N][B]   Jacobian
    double *bdf_lu;           // [N*N][B]   LU of (I - c J)
    uint32_t *bdf_piv;        // [B]        pivot rows, 4 bits each
    uint64_t *njev, *nlu;     // [B]
    uint32_t *err_flag;       // device word: IVP_ERRFLAG_* bits raised by the init kernel; the two words behind it collect, for the
                              // first 64 trajectories of a large-n BDF batch, {trailing columns updated, trailing columns looked at} of
                              // their factorisations (bdf_group.h lu_decomp): how dense the elimination is decides LDS vs global factors
    // ---- profiling ----
    unsigned long long *slot_counter;  // optional: += lanes x attempts the wave executed
    // ---- speculative launch of the lane-cooperative kernel (rk_coop.h) ----
    uint32_t spec_cap;        // != 0: do nothing unless *count_in <= spec_cap (the host enqueued this launch before
                              // it knew the active count; it reads count_in afterwards to see which way it went)
    uint32_t spec_min;        // thread-per-trajectory chunk kernels: != 0: do nothing unless the active count is > spec_min.
                              // The launch loop enqueues (bulk launch with spec_min = T, cooperative launch with spec_cap = T)
                              // PAIRS on the same input / output lists: exactly one of the two works, decided on the device
    uint32_t *ran_out;        // profiling: a launch that did work sets *ran_out = 1 (which kernel of a pair ran)
    // ---- thin waves ----
    uint32_t lds_lu;          // large-n BDF (bdf_group.h): != 0 selects the kernels that keep the factors of (I - cJ) in LDS
    uint32_t lpw;             // trajectories per wavefront of a thread-per-trajectory chunk launch (0 = 64).  A wave executes the
                              // UNION of its lanes' control flow; when the active set leaves SIMDs idle anyway, fewer lanes per
                              // wave on more SIMDs cost nothing and shrink that union (BDF: Newton iteration counts, D-rescaling,
                              // order adaptation, refactorisation and rejection differ from lane to lane on every attempt)
    uint32_t *count_next;     // the count_out of the NEXT launch of the chain (nobody reads or writes it during this one): every
                              // stepping kernel resets it, which saves a 4-byte fill kernel (~5 us of an idle GPU) per launch
    // ---- windowed bulk launches ----
    uint32_t window;          // != 0: only the first `window` entries of the input list take steps in this launch; the waves behind
                              // them pass their entries straight to the output list (where, having nothing else to do, they land
                              // FIRST, so the next launch's window starts with them).  A thread-per-trajectory wave saturates the
                              // f64 pipe of its SIMD on its own, so a launch of 1563 waves on 1024 SIMDs lasts as long as one of
                              // 2048: the launch loop cuts such a launch down to whole multiples of one wave per SIMD.
    // ---- deferred t_eval sampling (kernel flavour 3: DOP853, so_defer_samples / dop853_sample_body in rk_core.h) ----
    double *def_rec;          // [def_cap][n + 4][B] noted steps: x, h, {first, end} t_eval index, first output position, y[n];
                              // a trajectory's count of noted steps lives in n_seg (dense-output segments are not collected in this flavour)
    uint32_t def_cap;         // noted steps per trajectory the buffer holds (one per t_eval point at most)
    // ---- one-pass accepted-step log (so_push_log): records go to WAVE PAGES drawn from a device pool ----
    // The reference pushes every accepted step into growing Vecs (solout.rs:387-428); a GPU lane cannot grow a Vec, and a
    // counting solve before the filling solve costs a whole second integration.  With log_pool != NULL the stepping kernels
    // record as they go.  A page belongs to ONE WAVE and covers IVP_LOG_SLOTS record slots of every trajectory the wave is
    // stepping (32 attempts of a log-only kernel, 16 of a full one, whose attempts may record twice):
    //     [page header][cols x column header (2 doubles)][group 0: slots x W records][group 1: slots x W records] ...
    // cols = trajectories the wave is stepping when it opens the page, column = a trajectory's rank among them, slot = attempt
    // index within the page (wave-uniform), record = [t, y_0 .. y_{n-1}].  The columns are cut into GROUPS of W = 8 (n <= 8;
    // W = 1 for the wave-per-trajectory kernels of larger systems): within a group the records of one attempt lie side by
    // side, W x (n + 1) doubles, and a group's slots follow each other.  A wave's stores of one attempt therefore fill 8 runs
    // of 8 x 8 (n + 1) bytes -- a few whole cache lines each, completed by the very next stores (lanes that rejected leave
    // holes) -- and a group is a small dense block (32 x 8 x (n + 1) doubles: 6 KB at n = 2, 14 KB at n = 6) that the gather
    // kernel transposes through LDS in one piece.  (Per-trajectory pages touch one partly written line per trajectory and
    // attempt; 1M trajectories x 2 lines fit no cache and every 8-byte store became a partial HBM write -- measured on
    // BASELINE C3: 28.9 ms against 12.7 for the end state.  Whole-wave rows -- W = 64 -- made the stores perfect and the
    // gather slow: 64 destinations per row.)
    // A column header is {j: the trajectory, k0: its record count when the page was opened, bits: which slots hold a record},
    // so a page describes itself: the gather kernel (log_gather.hip) takes one column group per wavefront, reads it front to
    // back (coalesced) into LDS and writes every column's records -- record (j, k0 + rank of the slot) -- to their place in the
    // CSR log as one contiguous run: no per-trajectory chains, whole cache lines on both sides.
    // Allocation: the pool is cut into up to IVP_LOG_SUBPOOLS equal regions, each with its own 64-bit counter
    // (pages << 40 | doubles): pages grow up from the start of a region, their directory entries (page offset << 8 | cols, one
    // per page, what the gather enumerates) grow down from its end.  A wave draws up to four pages at a time (one atomicAdd
    // on the counter blockIdx picks: 1563 waves that start a launch together do not queue on one address).
    double *log_pool;                    // NULL = the dense / two-pass CSR forms above
    unsigned long long log_region;       // doubles per sub-pool region
    uint32_t log_sub_mask;               // sub-pools in use - 1 (a power of two <= IVP_LOG_SUBPOOLS: few waves, few sub-pools,
                                         // so that no region is left idle while another runs dry)
    unsigned long long *log_alloc;       // [IVP_LOG_SUBPOOLS * IVP_LOG_ALLOC_STRIDE] counters (one per 128-byte line)
    // ---- deferred event refinement (so_events_note / so_events_deferred_body in rk_core.h; no terminal events) ----
    double *evd_rec;          // [evd_cap][F][B] noted steps, F = 4 + 3 NE + n + NCoef n (EvdRec); NULL: roots are found in the stepping kernel
    uint32_t *evd_cnt;        // [B] noted steps of a trajectory
    uint32_t evd_cap;         // noted steps per trajectory the buffer holds (NE * max_events: each noted step fills an output slot)
};
#define IVP_LOG_SLOTS 32u
/* doubles before a page's first column group: page header + column headers, rounded up to a whole 128-byte line so that every
 * slot row of a group -- W (n + 1) doubles = 64 (n + 1) bytes at W = 8 -- starts on a 64-byte boundary: a wave's stores of one
 * attempt then fill whole 64-byte sectors (pages themselves are multiples of 128 bytes from a 128-byte-aligned pool: the
 * active pages of a big batch, ~50 KB per resident wave, fit no cache, and sectors written in two halves went to HBM twice) */
#define IVP_LOG_HDR(cols) ((1u + 2u * (cols) + 15u) & ~15u)
#define IVP_LOG_GROUP(np1) ((np1) <= 9 ? 8u : 1u)   /* columns per group of a page, by record length n + 1 */
#define IVP_LOG_SUBPOOLS 64u
#define IVP_LOG_ALLOC_STRIDE 16u
#define IVP_NO_SEG 0xFFFFFFFFFFFFFFFFull
