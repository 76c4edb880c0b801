//! Raw FFI of `include/ivp_hip.h` (ABI v5): the C boundary of the MI355X batched integrator that stands behind
//! `ivp::solve_ivp` for the explicit Runge-Kutta path (and BDF).
//!
//! UNCOMPILED in this repository -- the authoring image has no Rust toolchain.  What keeps it honest:
//! `tests/test_abi_layout.py` compiles `tests/c_abi/abi_layout.c` against the header, dumps `offsetof` / `sizeof` of every
//! member of every ABI struct and compares the dump with `abi_layout.json` next to this crate; the same test parses THIS file
//! and checks that every `#[repr(C)]` struct lists the same members in the same order, and that every function the header
//! declares is declared here.
//!
//! Field meanings, ownership and error codes: `include/ivp_hip.h`.  Reference interfaces replaced: `INTEGRATION.md`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const IVP_HIP_ABI_VERSION: c_int = 5;

// enum Method, src/solve/options.rs:14-27
pub const IVP_RK23: i32 = 0;
pub const IVP_DOPRI5: i32 = 1;
pub const IVP_DOP853: i32 = 2;
pub const IVP_RK4: i32 = 3;
pub const IVP_RADAU: i32 = 4;
pub const IVP_BDF: i32 = 5;

// return codes; the config codes map 1:1 onto enum ConfigError, src/error.rs:18-60
pub const IVP_OK: c_int = 0;
pub const IVP_ERR_MUST_BE_POSITIVE: c_int = -1;
pub const IVP_ERR_OUT_OF_RANGE: c_int = -2;
pub const IVP_ERR_NEGATIVE_TOLERANCE: c_int = -3;
pub const IVP_ERR_TOLERANCE_SIZE_MISMATCH: c_int = -4;
pub const IVP_ERR_INVALID_STEP_SIZE: c_int = -5;
pub const IVP_ERR_INVALID_SCALE_FACTORS: c_int = -6;
pub const IVP_ERR_BAD_ARGUMENT: c_int = -100;
pub const IVP_ERR_UNSUPPORTED_METHOD: c_int = -101;
pub const IVP_ERR_NO_DEVICE: c_int = -102;
pub const IVP_ERR_HIP: c_int = -103;
pub const IVP_ERR_JIT: c_int = -104;
pub const IVP_ERR_LOG_CAPACITY: c_int = -105;

pub const IVP_RHS_JIT: i32 = 1000;
pub const IVP_RHS_HAS_JAC: u32 = 1;

#[repr(C)]
pub struct ivp_problem_t {
    pub rhs_id: i32,
    pub n: i32,
    pub n_params: i32,
    pub jit: *mut c_void,
}

/// `struct Options` (src/solve/options.rs:75-123) + the per-method controller fields + the GPU-only knobs.
#[repr(C)]
pub struct ivp_options_t {
    pub method: i32,
    pub rtol: f64,
    pub atol: f64,
    pub rtol_vec: *const f64,
    pub atol_vec: *const f64,
    pub rtol_vec_len: i32,
    pub atol_vec_len: i32,
    pub max_steps: u64,
    pub t_eval: *const f64,
    pub n_eval: i64,
    pub has_first_step: i32,
    pub first_step: f64,
    pub has_max_step: i32,
    pub max_step: f64,
    pub dense_output: i32,
    pub ev_direction: [i32; 4],
    pub ev_terminal: [u32; 4],
    pub max_events: u32,
    pub has_min_step: i32,
    pub min_step: f64,
    pub fp_mode: i32,
    pub chunk_attempts: i32,
    pub max_log: u32,
    pub variant: i32,
    pub profile: i32,
    pub has_settings: i32,
    pub uround: f64,
    pub safety_factor: f64,
    pub scale_min: f64,
    pub scale_max: f64,
    pub beta: f64,
    pub stiff_test: u64,
    pub count_log: i32,
    pub t_eval_offsets: *const u64,
    pub ev_direction_vec: *const i32,
    pub ev_terminal_vec: *const u32,
    pub n_event_cfg: i32,
}

/// `struct Solution` (src/solve/solution.rs:7-20) + IntegrationResult.h, struct of arrays over the batch.
#[repr(C)]
pub struct ivp_batch_result_t {
    pub y_end: *mut f64,
    pub t_end: *mut f64,
    pub status: *mut i32,
    pub nfev: *mut u64,
    pub nstep: *mut u64,
    pub naccpt: *mut u64,
    pub nrejct: *mut u64,
    pub h_next: *mut f64,
    pub y_eval: *mut f64,
    pub eval_idx: *mut i32,
    pub n_filled: *mut i32,
    pub t_log: *mut f64,
    pub y_log: *mut f64,
    pub n_log: *mut u32,
    pub seg_cont: *mut f64,
    pub seg_xold: *mut f64,
    pub seg_h: *mut f64,
    pub n_seg: *mut u32,
    pub t_events: *mut f64,
    pub y_events: *mut f64,
    pub n_event_hits: *mut u32,
    pub t_term: *mut f64,
    pub njev: *mut u64,
    pub nlu: *mut u64,
    pub log_offsets: *const u64,
}

#[repr(C)]
pub struct ivp_run_stats_t {
    pub launches: u32,
    pub init_launches: u32,
    pub step_kernel_ms: f64,
    pub init_kernel_ms: f64,
    pub total_ms: f64,
    pub total_accepted: u64,
    pub total_attempts: u64,
    pub lane_attempt_slots: u64,
    pub lane_launches: u64,
    pub coop_launches: u32,
    pub coop_kernel_ms: f64,
    pub declined_launches: u32,
    pub declined_coop_launches: u32,
    pub declined_ms: f64,
    pub declined_coop_ms: f64,
}

/// Solution.t / Solution.y of a batch as a CSR log (src/solve/solve_ivp.rs:288-312): `t` / `y` null on entry = allocated by
/// the library like the Vecs the reference returns (`owned` = 1; release with `ivp_step_log_free`).
#[repr(C)]
pub struct ivp_step_log_t {
    pub offsets: *mut u64,
    pub t: *mut f64,
    pub y: *mut f64,
    pub capacity: u64,
    pub reserve: u64,
    pub defer: i32,
    pub owned: i32,
    pub device: i32,
    pub passes: u32,
    pub total: u64,
    pub pool_bytes: u64,
    pub pool_used_bytes: u64,
    pub page_slots: u32,
}

pub enum ivp_ctx_t {}

/// trajectories [first, first + count) of a batch, resident on ctx's device (SoA stride `count`)
#[repr(C)]
pub struct ivp_shard_t {
    pub ctx: *mut ivp_ctx_t,
    pub first: usize,
    pub count: usize,
    pub y0: *const f64,
    pub params: *const f64,
    pub t0: *const f64,
    pub t0_len: usize,
    pub t1: *const f64,
    pub t1_len: usize,
    pub out: ivp_batch_result_t,
    pub hip_stream: *mut c_void,
}

#[link(name = "ivp_hip")]
extern "C" {
    pub fn ivp_abi_version() -> c_int;
    pub fn ivp_device_count() -> c_int;
    pub fn ivp_ctx_create(ctx: *mut *mut ivp_ctx_t, device: c_int) -> c_int;
    pub fn ivp_ctx_destroy(ctx: *mut ivp_ctx_t);
    pub fn ivp_last_error_string(ctx: *const ivp_ctx_t) -> *const c_char;
    pub fn ivp_ctx_get_stats(ctx: *const ivp_ctx_t, stats: *mut ivp_run_stats_t) -> c_int;
    pub fn ivp_options_default(opt: *mut ivp_options_t);
    pub fn ivp_options_method_defaults(opt: *mut ivp_options_t, method: i32) -> c_int;
    pub fn ivp_rhs_dims(rhs_id: i32, n: *mut i32, n_params: *mut i32) -> c_int;
    pub fn ivp_rhs_n_events(rhs_id: i32) -> c_int;
    // B independent solve_ivp() calls (src/solve/solve_ivp.rs:99-108), host / device buffers
    pub fn ivp_batch_solve(ctx: *mut ivp_ctx_t, prob: *const ivp_problem_t, b: usize, y0: *const f64, params: *const f64,
                           t0: *const f64, t0_len: usize, t1: *const f64, t1_len: usize, opt: *const ivp_options_t,
                           out: *mut ivp_batch_result_t) -> c_int;
    pub fn ivp_batch_solve_device(ctx: *mut ivp_ctx_t, prob: *const ivp_problem_t, b: usize, y0: *const f64, params: *const f64,
                                  t0: *const f64, t0_len: usize, t1: *const f64, t1_len: usize, opt: *const ivp_options_t,
                                  out: *mut ivp_batch_result_t, hip_stream: *mut c_void) -> c_int;
    // resumable form: several contexts (batches) in flight from one thread
    pub fn ivp_batch_submit_device(ctx: *mut ivp_ctx_t, prob: *const ivp_problem_t, b: usize, y0: *const f64, params: *const f64,
                                   t0: *const f64, t0_len: usize, t1: *const f64, t1_len: usize, opt: *const ivp_options_t,
                                   out: *mut ivp_batch_result_t, hip_stream: *mut c_void) -> c_int;
    pub fn ivp_batch_poll(ctx: *mut ivp_ctx_t, done: *mut c_int) -> c_int;
    pub fn ivp_batch_wait(ctx: *mut ivp_ctx_t) -> c_int;
    // Solution.t / Solution.y in ONE call and ONE integration (page pool + gather kernel)
    pub fn ivp_batch_solve_logged_device(ctx: *mut ivp_ctx_t, prob: *const ivp_problem_t, b: usize, y0: *const f64,
                                         params: *const f64, t0: *const f64, t0_len: usize, t1: *const f64, t1_len: usize,
                                         opt: *const ivp_options_t, out: *mut ivp_batch_result_t, log: *mut ivp_step_log_t,
                                         hip_stream: *mut c_void) -> c_int;
    pub fn ivp_step_log_fetch_device(ctx: *mut ivp_ctx_t, log: *mut ivp_step_log_t, hip_stream: *mut c_void) -> c_int;
    pub fn ivp_batch_solve_logged(ctx: *mut ivp_ctx_t, prob: *const ivp_problem_t, b: usize, y0: *const f64, params: *const f64,
                                  t0: *const f64, t0_len: usize, t1: *const f64, t1_len: usize, opt: *const ivp_options_t,
                                  out: *mut ivp_batch_result_t, log: *mut ivp_step_log_t) -> c_int;
    pub fn ivp_step_log_free(log: *mut ivp_step_log_t);
    // one batch over several devices: N contexts driven by this thread, shards gathered by peer copies (xGMI)
    pub fn ivp_batch_solve_multi(shards: *mut ivp_shard_t, n_shards: i32, prob: *const ivp_problem_t, b: usize,
                                 opt: *const ivp_options_t, gather_device: i32, gathered: *mut ivp_batch_result_t) -> c_int;
    pub fn ivp_batch_solve_multi_host(ctxs: *const *mut ivp_ctx_t, n_ctx: i32, prob: *const ivp_problem_t, b: usize,
                                      y0: *const f64, params: *const f64, t0: *const f64, t0_len: usize, t1: *const f64,
                                      t1_len: usize, opt: *const ivp_options_t, out: *mut ivp_batch_result_t) -> c_int;
    pub fn ivp_batch_solve_logged_multi(shards: *mut ivp_shard_t, n_shards: i32, prob: *const ivp_problem_t, b: usize,
                                        opt: *const ivp_options_t, gather_device: i32, gathered: *mut ivp_batch_result_t,
                                        log: *mut ivp_step_log_t) -> c_int;
    pub fn ivp_step_log_fetch_multi(shards: *mut ivp_shard_t, n_shards: i32, prob: *const ivp_problem_t, b: usize,
                                    opt: *const ivp_options_t, gather_device: i32, log: *mut ivp_step_log_t) -> c_int;
    // the device-side `impl IVP for T` (src/ivp.rs:27-121): HIP source compiled at run time
    pub fn ivp_rhs_compile(ctx: *mut ivp_ctx_t, ode_source: *const c_char, n: i32, n_params: i32, handle: *mut *mut c_void) -> c_int;
    pub fn ivp_rhs_compile_events(ctx: *mut ivp_ctx_t, source: *const c_char, n: i32, n_params: i32, n_events: i32,
                                  handle: *mut *mut c_void) -> c_int;
    pub fn ivp_rhs_compile_ex(ctx: *mut ivp_ctx_t, source: *const c_char, n: i32, n_params: i32, n_events: i32, flags: u32,
                              handle: *mut *mut c_void) -> c_int;
    pub fn ivp_rhs_free(handle: *mut c_void);
}
