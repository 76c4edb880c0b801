// ivp_capi.cpp -- host side of libivp_hip.so: the C ABI declared in include/ivp_hip.h.
//
// What lives here: option validation (the reference's Err(Error::Config) values), device scratch
// management, the launch loop (init kernel, then chunks of step attempts with on-device compaction
// of the still-running set) and result plumbing.  No arithmetic of the integration happens on the
// host, and there is no CPU fallback: without a HIP device every compute entry point fails.
#include "../../include/ivp_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ivp_jit.h"
#include "ivp_kargs.h"
#include "rk_launch.h"
#include "ivp_ctx.h"

using namespace ivp_host;

namespace {

struct RhsDim { int n, p; };
const RhsDim kRhsDims[IVP_RHS_BUILTIN_COUNT] = {{1, 1}, {2, 0}, {2, 1}, {6, 1}, {3, 3}, {3, 0}, {2, 0}, {2, 0}, {2, 0}, {3, 0}, {2, 1},
                                                {2, 0}, {2, 2}, {2, 0}, {2, 0}, {3, 0}};
const int kRhsEvents[IVP_RHS_BUILTIN_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 3, 0};

// wave-per-trajectory problems (rk_group.h): ids 100.., n > IVP_MAX_N
#define IVP_MAX_GROUP_N 512
bool group_builtin(int rhs_id, int *n, int *p)
{
    switch (rhs_id) {
    case IVP_RHS_LINEAR_DECAY_100: *n = 100; *p = 0; return true;
    case IVP_RHS_HEAT1D_256: *n = 256; *p = 1; return true;
    case IVP_RHS_DENSE_64: *n = 64; *p = 1; return true;
    }
    return false;
}

// controller fields of the method structs (dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37)
struct MethodSettings { double uround, safety, scale_min, scale_max, beta; uint64_t nstiff; };
MethodSettings method_defaults(int method)
{
    if (method == IVP_DOP853) return {2.3e-16, 0.9, 0.333, 6.0, 0.0, 1000};
    if (method == IVP_DOPRI5) return {2.3e-16, 0.9, 0.2, 10.0, 0.04, 1000};
    return {2.3e-16, 0.9, 0.2, 10.0, 0.0, 1000};   // RK23 (reads safety / scale_min / scale_max only)
}
MethodSettings settings_of(const ivp_options_t *opt)
{
    if (!opt->has_settings) return method_defaults(opt->method);
    return {opt->uround, opt->safety_factor, opt->scale_min, opt->scale_max, opt->beta, opt->stiff_test};
}

int ncoef_of(int method) { return method == IVP_DOPRI5 ? 5 : method == IVP_DOP853 ? 8 : method == IVP_BDF ? 7 : 4; }

}  // namespace

namespace {

// solves in flight in this process on each device, over all its contexts (several batches overlapped on streams, ivp_amd/pipeline.py): the launch
// loop sizes its launches for an otherwise idle chip only when it is alone
std::atomic<int> g_inflight[64];   // per device (zero-initialised: static storage)
std::atomic<int> &inflight_on(int device) { return g_inflight[(unsigned)device & 63u]; }
void set_active(ivp_ctx *ctx, bool on)
{
    if (ctx->pend.active == on) return;
    ctx->pend.active = on;
    inflight_on(ctx->device).fetch_add(on ? 1 : -1, std::memory_order_relaxed);
}

// Options validation: the checks XXX::solve() makes before integrating
// (dopri5.rs:143-198, dop853.rs:135-193, rk23.rs:102-129) for the fields solve_ivp() can set, plus
// the Tolerance length rule (mod.rs:156-161).
}  // namespace
int ivp_host::validate(ivp_ctx *ctx, const ivp_problem_t *prob, size_t B, const ivp_options_t *opt, int *n_out, int *p_out)
{
    if (!prob || !opt) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null problem/options");
    int n, p;
    if (prob->rhs_id == IVP_RHS_JIT) {
        if (!prob->jit) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "IVP_RHS_JIT without a handle");
        ivp_jit_dims(prob->jit, &n, &p);
    } else if (prob->rhs_id >= 0 && prob->rhs_id < IVP_RHS_BUILTIN_COUNT) {
        n = kRhsDims[prob->rhs_id].n;
        p = kRhsDims[prob->rhs_id].p;
    } else if (group_builtin(prob->rhs_id, &n, &p)) {
    } else {
        return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unknown rhs_id %d", prob->rhs_id);
    }
    if (prob->n != n || prob->n_params != p)
        return fail(ctx, IVP_ERR_BAD_ARGUMENT, "problem dims (n=%d,p=%d) do not match rhs (n=%d,p=%d)", prob->n, prob->n_params, n, p);
    if (n < 1 || n > IVP_MAX_GROUP_N || p > IVP_MAX_P) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unsupported dimensions");
    if (B == 0 || B > 0x7FFFFFFFull) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "batch size %zu out of range", B);
    if (opt->method == IVP_RADAU)
        return fail(ctx, IVP_ERR_UNSUPPORTED_METHOD, "method %d (RADAU) is not on the accelerated path", opt->method);
    if (opt->method < IVP_RK23 || opt->method > IVP_BDF) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unknown method %d", opt->method);
    if (opt->method == IVP_BDF) {   // BDF::solve validates tolerances, bdf.rs:112-128
        for (int i = 0; i < n; ++i) {
            const double rt = opt->rtol_vec ? opt->rtol_vec[i < opt->rtol_vec_len ? i : 0] : opt->rtol;
            const double at = opt->atol_vec ? opt->atol_vec[i < opt->atol_vec_len ? i : 0] : opt->atol;
            if (rt < 0.0 || at < 0.0) return fail(ctx, IVP_ERR_NEGATIVE_TOLERANCE, "negative tolerance at component %d", i);
        }
    }
    if (opt->has_settings) {   // XXX::solve() input validation, in the reference's order
        const MethodSettings m = settings_of(opt);
        if (opt->method == IVP_DOPRI5 || opt->method == IVP_DOP853) {   // dopri5.rs:143-198, dop853.rs:135-193
            if (m.uround <= 1e-35 || m.uround >= 1.0) return fail(ctx, IVP_ERR_OUT_OF_RANGE, "uround = %g outside (1e-35, 1)", m.uround);
            if (m.safety >= 1.0 || m.safety <= 1e-4) return fail(ctx, IVP_ERR_OUT_OF_RANGE, "safety_factor = %g outside (1e-4, 1)", m.safety);
            if (m.beta > 0.2) return fail(ctx, IVP_ERR_OUT_OF_RANGE, "beta = %g outside [0, 0.2]", m.beta);
            if (opt->max_steps == 0) return fail(ctx, IVP_ERR_MUST_BE_POSITIVE, "max_steps must be positive");
            if (m.nstiff == 0) return fail(ctx, IVP_ERR_MUST_BE_POSITIVE, "stiff_test must be positive");
        } else if (opt->method == IVP_RK23) {                           // rk23.rs:102-129
            if (opt->max_steps == 0) return fail(ctx, IVP_ERR_MUST_BE_POSITIVE, "max_steps must be positive");
            if (m.safety >= 1.0 || m.safety <= 1e-4) return fail(ctx, IVP_ERR_OUT_OF_RANGE, "safety_factor = %g outside (1e-4, 1)", m.safety);
            if (m.scale_min <= 0.0 || m.scale_max <= m.scale_min)
                return fail(ctx, IVP_ERR_INVALID_SCALE_FACTORS, "scale factors min = %g, max = %g", m.scale_min, m.scale_max);
        } else {
            return fail(ctx, IVP_ERR_BAD_ARGUMENT, "has_settings applies to RK23 / DOPRI5 / DOP853");
        }
    }
    if (opt->rtol_vec && opt->rtol_vec_len != n) return fail(ctx, IVP_ERR_TOLERANCE_SIZE_MISMATCH, "rtol: expected %d, got %d", n, opt->rtol_vec_len);
    if (opt->atol_vec && opt->atol_vec_len != n) return fail(ctx, IVP_ERR_TOLERANCE_SIZE_MISMATCH, "atol: expected %d, got %d", n, opt->atol_vec_len);
    if (opt->t_eval && opt->n_eval < 0) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "negative n_eval");
    // one shared grid: its length is a 32-bit kernel argument; per-trajectory grids: the concatenation may be longer, each
    // trajectory's own grid may not (checked with the offsets)
    if (opt->t_eval && !opt->t_eval_offsets && opt->n_eval > 0x7FFFFFFFll) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "n_eval too large");
    if (opt->fp_mode != IVP_FP_STRICT && opt->fp_mode != IVP_FP_FAST) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unknown fp_mode");
    if (opt->chunk_attempts < 0) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "negative chunk_attempts");
    if (opt->variant < 0 || opt->variant > 3) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unknown kernel variant %d", opt->variant);
    *n_out = n;
    *p_out = p;
    return IVP_OK;
}
namespace {

// Launch-policy knobs.  The defaults are the measured optimum for the BASELINE configs on MI355X; the environment
// overrides exist for tuning runs (tools/tune_policy.py) and are read once per process.
struct Tune {
    size_t coop_cap_lanes = 0;       // lanes (8 per trajectory) up to which the lane-cooperative kernels take over; 0 = two waves per SIMD of the device
    uint32_t bulk_chunk = 64;        // attempts per bulk launch
    int bdf_occ2 = -1;               // BDF occupancy-2 build: -1 = automatic (batches wider than one wave per SIMD), 0 = never, 1 = always
    int window = 1;                  // windowed bulk launches (IvpKArgs.window): 1 = automatic, 0 = never, 2 = also for systems with fewer than four components
    int launches_per_poll = 3;       // bulk launches between two host polls
    int launches_per_poll_set = 0;   // IVP_TUNE_LAUNCHES_PER_POLL given: taken literally (no extra hand-over pair)
    int lds_lu = 1;                  // large-n BDF: 0 = never keep the factors in LDS
    int defer_eval = 1;              // DOP853 t_eval sampling in a second, sample-parallel kernel (flavour 3): 0 = sample in the stepping kernel
    int defer_events = 1;            // event roots (no terminal event) in a second kernel, one lane per step with a crossing: 0 = Brent in the stepping kernel
    int bdf_lpw = 0;                 // trajectories per wave of the BDF chunk launches: 0 = auto (spread the active set over the SIMDs)
    Tune()
    {
        if (const char *e = getenv("IVP_TUNE_COOP_CAP_LANES")) coop_cap_lanes = (size_t)strtoull(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_BULK_CHUNK")) bulk_chunk = (uint32_t)std::max(1l, strtol(e, nullptr, 10));
        if (const char *e = getenv("IVP_TUNE_BDF_OCC2")) bdf_occ2 = (int)strtol(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_WINDOW")) window = (int)strtol(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_LDS_LU")) lds_lu = (int)strtol(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_DEFER_EVAL")) defer_eval = (int)strtol(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_DEFER_EVENTS")) defer_events = (int)strtol(e, nullptr, 10);
        if (const char *e = getenv("IVP_TUNE_BDF_LPW")) bdf_lpw = (int)std::min(64l, std::max(0l, strtol(e, nullptr, 10)));
        if (const char *e = getenv("IVP_TUNE_LAUNCHES_PER_POLL")) { launches_per_poll = (int)std::min(16l, std::max(1l, strtol(e, nullptr, 10))); launches_per_poll_set = 1; }
    }
};
const Tune &tune() { static const Tune t; return t; }
// lanes (8 per trajectory) up to which the lane-cooperative kernels take over: two cooperative waves per SIMD
// IVP_TRACE_LAUNCHES=1 with Options.profile: one stderr line per stepping-kernel launch (kind, ran / declined, duration)
bool trace_launches() { static const bool on = getenv("IVP_TRACE_LAUNCHES") != nullptr; return on; }
size_t coop_cap_lanes(const ivp_ctx *ctx) { return tune().coop_cap_lanes ? tune().coop_cap_lanes : 2u * (size_t)ctx->one_wave_per_simd(); }

constexpr uint32_t kMaxRanSlots = 1u << 16;   // profiled launches whose "did work" flag is recorded

hipEvent_t pend_event(ivp_ctx *ctx)
{
    ivp_ctx::Pending &P = ctx->pend;
    if (P.ev_used == ctx->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->events.push_back(e);
    }
    return ctx->events[P.ev_used++];
}

hipError_t pend_launch(ivp_ctx *ctx, int what, const IvpKArgs &ka, uint32_t lanes, bool use_hoist, bool use_coop)
{
    const ivp_ctx::Pending &P = ctx->pend;
    hipStream_t s = P.stream;
    const bool fast = P.fp_mode == IVP_FP_FAST;
    if (P.jit) return ivp_jit_launch(P.prob.jit, (use_coop && what == IVP_LAUNCH_CHUNK) ? IVP_LAUNCH_COOP : what, P.method, P.fp_mode, P.full, ka, lanes, s);
    if (P.group) return (fast ? ivp_launch_group_fast : ivp_launch_group_strict)(what, P.method, P.prob.rhs_id, P.full, ka, lanes, s);
    if (P.method == IVP_BDF) {
        // more than one full wave per SIMD still running: the two-waves-per-SIMD build (rk_bdf.hip); same bits either way
        const bool occ2 = what == IVP_LAUNCH_CHUNK && (lanes > ctx->one_wave_per_simd() || tune().bdf_occ2 == 1) && tune().bdf_occ2 != 0;
        if (occ2) return (fast ? ivp_launch_bdf_fast_occ2 : ivp_launch_bdf_strict_occ2)(what, P.prob.rhs_id, P.full, ka, lanes, s);
        return (fast ? ivp_launch_bdf_fast : ivp_launch_bdf_strict)(what, P.prob.rhs_id, P.full, ka, lanes, s);
    }
    if (use_coop && what == IVP_LAUNCH_CHUNK) return (fast ? ivp_launch_coop_fast : ivp_launch_coop_strict)(P.method, P.prob.rhs_id, P.full, ka, lanes, s);
    if (use_hoist) return (fast ? ivp_launch_fast_hoist : ivp_launch_strict_hoist)(what, P.method, P.prob.rhs_id, P.full, ka, lanes, s);
    return (fast ? ivp_launch_fast : ivp_launch_strict)(what, P.method, P.prob.rhs_id, P.full, ka, lanes, s);
}

// a kernel launch; when a hiprtc module had to be built for it and the build failed, the context's error string
// carries the build log (IVP_ERR_JIT) instead of a bare HIP error
#define LAUNCH_TRY(ctx, expr)                                                                                   \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if ((ctx)->pend.jit && ivp_jit_last_log((ctx)->pend.prob.jit)[0])                                   \
                return fail((ctx), IVP_ERR_JIT, "building the kernel for this launch failed: %.400s", ivp_jit_last_log((ctx)->pend.prob.jit)); \
            return fail((ctx), IVP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                            \
        }                                                                                                       \
    } while (0)

// One round = the chunk launches between two host polls of the active count; the still-running ids are compacted on
// the device from launch to launch.
// Launch policy.  While the active set still over-subscribes the chip (more than one wave per SIMD:
// 256 CUs x 4 SIMDs x 64 lanes = 65536 trajectories) short chunks + compaction keep wavefronts dense and
// the SIMDs evenly loaded.  Once it fits one wave per SIMD, wall time is the sequential attempt latency of
// the slowest trajectory: compaction cannot help any more and every extra launch only adds a gap, so the
// remainder runs in long chunks with one launch per host poll.
int enqueue_round(ivp_ctx *ctx)
{
    ivp_ctx::Pending &P = ctx->pend;
    hipStream_t s = P.stream;
    const uint32_t kOneWavePerSimd = ctx->one_wave_per_simd();
    uint32_t *counts = (uint32_t *)ctx->counts.p;
    const uint32_t lanes = P.lanes;
    const bool profile = P.profile != 0;
    const bool fits_one_wave = (size_t)lanes * (P.group ? (P.jit ? (uint32_t)ivp_group_width(P.n) : (uint32_t)IVP_WAVE) : 1u) <= kOneWavePerSimd;
    // problems whose stragglers may be handed to the lane-cooperative kernels by a speculative launch (see below)
    const bool spec_ok = P.coop_ok && P.variant == 0 && P.adaptive && P.n >= 4 && !P.jit;
    // kernel variant: 1 = lean registers (coefficients re-materialised per use), 2 = coefficients resident in
    // registers, 3 = lane-cooperative.  Results are bit-identical in all of them, in BOTH arithmetic modes (strict: the
    // reference's operation sequence; FMA: the same sequence with the IVP_MA sites fused, written out in the source and
    // compiled without contraction), so the choice follows the shrinking active set: resident once at most two waves
    // per SIMD are left to run, cooperative for the tail.
    bool use_hoist = !P.has_settings &&   // run-time controller fields exist in the lean builds only
                     (P.variant == 2 || (P.variant == 0 && (size_t)lanes <= 2 * (size_t)kOneWavePerSimd));
    // eight lanes per trajectory pay off once the cooperative waves fit two per SIMD, and only for systems
    // with enough components to share out (measured: break-even at n = 3, a loss at n = 2)
    const size_t coop_cap = coop_cap_lanes(ctx);
    const bool use_coop = P.coop_ok && (P.variant == 3 ||
                                        (P.variant == 0 && P.adaptive && P.n >= 4 && (size_t)lanes * 8u <= coop_cap));
    // Long chunks (one launch per poll) once compaction cannot help any more: in the cooperative kernels, and for
    // problems without a cooperative kernel when the active set fits one wave per SIMD.  A set that will still be
    // handed to the cooperative kernels keeps short chunks + the speculative hand-over whatever its size: an attempt
    // of a lone thread-per-trajectory wave costs 4.2 us against 1.9 us in the cooperative kernel, so every attempt a
    // straggler spends in a 1024-attempt bulk launch is paid twice.
    // three short launches per poll: measured on C2 (attempts per trajectory peak at 160-200) the hand-over to the
    // cooperative kernel then happens after 192 instead of 256 attempts (3.33 -> 3.25 ms); more polls cost ~40 us each
    const bool tail = P.adaptive && (use_coop || (fits_one_wave && !spec_ok));
    int launches_per_sync = tail ? 1 : tune().launches_per_poll;
    uint32_t this_chunk = tail ? 1024u : P.chunk_now;
    // Large-n BDF batches too big for the "two wavefronts per CU" rule: ONE short launch first (every trajectory factorises in
    // its first attempt), so that the density of the eliminations is known before the bulk of the work is enqueued (see lds_lu
    // below; costs one host poll, ~50 us of a solve that takes tens of milliseconds)
    if (P.lds_lu_ok && P.variant == 0 && !P.lu_probed && (size_t)lanes > 2u * (size_t)ctx->cus) {
        P.lu_probed = true;
        launches_per_sync = 1;
        this_chunk = 2u;
    }
    // BDF (thread per trajectory): thin waves.  BASELINE C5's 10 000 trajectories are 157 full waves on a chip with 256 CUs.
    // A wave pays for the union of its lanes' control flow on every attempt, and (measured, MI355X) a wave that has its
    // CU to itself runs this branch-heavy kernel fastest: 10.95 ms with 64 lanes per wave (157 waves), 10.5 ms with 40
    // (250 waves), but 13.7-16.8 ms with 313-625 waves and 11.2 ms with 1000 -- so the active set is spread over at most
    // one wave per CU -- down to a single trajectory per wave (256 trajectories: 4.9 ms with one lane per wave, 6.6 with
    // two, 9.6 with eight: the fewer lanes, the fewer phases a wave runs on behalf of some other lane).
    uint32_t lpw = 0;
    if (P.method == IVP_BDF && !P.group) {   // built-in and hiprtc right-hand sides alike
        const uint32_t cus = ctx->cus;
        const uint32_t want = tune().bdf_lpw > 0 ? (uint32_t)tune().bdf_lpw : std::max(1u, (lanes + cus - 1u) / cus);
        lpw = std::min(64u, want);
    }
    // Windowed bulk launches (IvpKArgs.window).  A thread-per-trajectory wave of a system with four or more components keeps
    // the f64 pipe of its SIMD busy on its own (measured, BASELINE C2: 3.75-4.1 us per attempt alone, 7.5 us for each of two
    // waves that share a SIMD), so a launch lasts ceil(waves / SIMDs) wave-rounds whether its last round is full or not:
    // C2's 1563 waves on 1024 SIMDs take as long as 2048 would.  When the last round would be less than ~80 % full the
    // launch therefore works on whole rounds only -- the first `window` entries of the list -- and the waves behind them
    // pass their entries on; those land first in the output list, so every trajectory is at most one launch behind.
    // Results never depend on how attempts are cut into launches.
    uint32_t window = 0;
    // (not for problems with event functions: the root-finding of a crossing is a long divergent stretch that a second wave
    // on the SIMD hides; measured on C2 with the x-axis crossing event 4.5 ms with full launches, 5.0 with windows.  With
    // deferred refinement there is no such stretch, but the full DefaultSolOut kernel gains nothing from windows either:
    // 5 x 0.38 ms against 3 x 0.58 ms, 3.82 against 3.79 ms per solve -- round 4, profiles/EXPERIMENTS.md A6)
    // (and only while this solve has the device to itself: with other solves in flight the SIMDs a ragged round leaves idle
    // are not idle)
    const bool alone = inflight_on(ctx->device).load(std::memory_order_relaxed) <= 1;
    if (tune().window && alone && P.adaptive && !P.group && !use_coop && !tail && lpw == 0 && P.method != IVP_BDF && !P.has_events &&
        (P.n >= 4 || tune().window == 2)) {
        const uint32_t full = lanes / kOneWavePerSimd;
        if (full >= 1 && full < 4 && (uint64_t)lanes * 5u < (uint64_t)(full + 1u) * kOneWavePerSimd * 4u) window = full * kOneWavePerSimd;
        // a round still covers `launches_per_poll` chunks of the WHOLE list (a host poll idles the GPU for ~50 us)
        if (window) launches_per_sync = (int)(((uint64_t)launches_per_sync * lanes + window - 1u) / window);
        // what runs at once is the window, not the list: at most two waves per SIMD take the resident build
        if (window && !P.has_settings && P.variant == 0 && (size_t)window <= 2 * (size_t)kOneWavePerSimd) use_hoist = true;
    }
    // Paired launches.  For a problem whose stragglers go to the lane-cooperative kernels, every bulk launch of a round
    // is followed by a cooperative launch on the SAME input / output lists: the bulk one works while more than T
    // trajectories are still running, the cooperative one once at most T are (T = two cooperative waves per SIMD); the
    // decision is taken on the device from the active count, so the hand-over happens at the first LAUNCH boundary at
    // which it pays (every `chunk` attempts), not at the next host poll (every 3 x chunk), and costs no poll at all.  A
    // launch that declines returns at once (~3 us).  Results never depend on where the hand-over falls.
    const bool paired = !tail && !use_coop && spec_ok;
    const uint32_t pair_threshold = (uint32_t)(coop_cap / 8u);
    // one more pair per round: a cooperative partner looks at the count its own bulk launch STARTED from, so the pair after
    // the launch that crossed the threshold is the one that hands over -- with it in the same round the hand-over needs no
    // host poll (~40 us of an idle GPU); if the threshold was not crossed the extra launch is one more chunk of useful work
    if (paired && tune().launches_per_poll_set == 0) launches_per_sync += 1;
    auto ran_slot = [&]() -> uint32_t * {   // profiling: which launch of a pair did the work
        if (!profile || !paired || P.step_ev.size() >= kMaxRanSlots) return nullptr;
        return (uint32_t *)ctx->ran.p + P.step_ev.size();
    };
    for (int r = 0; r < launches_per_sync; ++r, ++P.c) {
        const uint64_t c = P.c;
        IvpKArgs ka = P.a;
        ka.chunk = this_chunk;
        ka.lpw = lpw;
        ka.window = window;
        ka.spec_min = (paired && c > 0) ? pair_threshold : 0u;
        ka.ran_out = ran_slot();
        // LDS-resident factors (bdf_group.h) cost occupancy: an 80 KB matrix leaves room for two wavefronts per CU instead of
        // four.  Measured (MI355X, profiles/r03_large_n_bdf_lds_vs_global.jsonl): while the active set fits two wavefronts per
        // CU the LDS form is 2-12 % faster (a pivot step waits for LDS, not for L2); beyond that a dense Jacobian still gains
        // 1.2x but a sparse one (whose trailing updates are mostly skipped) loses 1.5x to the lost occupancy -- so the
        // automatic choice follows the active count, launch by launch (current factors travel through global memory
        // between launches either way); variant 2 forces the LDS form for every launch.  Round 4: the kernels report how
        // dense their eliminations are (the two words behind err_flag); once more than half of the trailing columns a pivot
        // looks at need an update the LDS form wins at any batch size (20 000 dense 64-state systems: 234 ms against 369)
        ka.lds_lu = (P.lds_lu_ok && (P.variant == 2 || (size_t)lanes <= 2u * (size_t)ctx->cus || P.lu_dense)) ? 1u : 0u;
        if (c == 0) {
            ka.perm_in = nullptr;
            ka.count_in = nullptr;
        } else {
            ka.perm_in = (const uint32_t *)ctx->perm[(c - 1) & 1].p;
            ka.count_in = counts + ((c - 1) & 3);
        }
        ka.perm_out = (uint32_t *)ctx->perm[c & 1].p;
        ka.count_out = counts + (c & 3);
        // slot (c+1)&3 is the next launch's count_out; nothing reads it during this launch, which resets it itself
        // (a 4-byte hipMemsetAsync is a ~5 us kernel of its own)
        ka.count_next = counts + ((c + 1) & 3);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile) { e0 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(e0, s)); }
        LAUNCH_TRY(ctx, pend_launch(ctx, IVP_LAUNCH_CHUNK, ka, lanes, use_hoist, use_coop));
        if (profile) { e1 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(e1, s)); P.step_ev.emplace_back(e0, e1); P.step_is_coop.push_back((use_coop ? 1 : 0) | (ka.ran_out ? 2 : 0)); P.step_lanes.push_back(lanes); }
        if (!profile) { ctx->stats.launches += 1; if (use_coop) ctx->stats.coop_launches += 1; }   // with profile: counted from the ran flags
        if (paired && c > 0) {   // the cooperative partner of this launch: same lists, the complementary condition
            IvpKArgs kc = ka;
            kc.chunk = 1024u;
            kc.spec_min = 0u;
            kc.spec_cap = pair_threshold;
            kc.lpw = 0u;
            kc.window = 0u;
            kc.ran_out = ran_slot();
            if (profile) { e0 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(e0, s)); }
            LAUNCH_TRY(ctx, pend_launch(ctx, IVP_LAUNCH_CHUNK, kc, std::min<uint32_t>(lanes, pair_threshold), false, true));
            if (profile) { e1 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(e1, s)); P.step_ev.emplace_back(e0, e1); P.step_is_coop.push_back(1 | (kc.ran_out ? 2 : 0)); P.step_lanes.push_back(lanes); }
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned, counts + ((P.c - 1) & 3), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (!P.err_checked || P.paged || P.lds_lu_ok)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned + 1, counts + 4, sizeof(uint32_t) * (P.lds_lu_ok ? 3u : 1u), hipMemcpyDeviceToHost, s));
    // one-pass step log: the sub-pools' counters travel with every round's active count, so that the round that finishes the
    // solve also tells the host how many pages there are to gather (ivp_log.cpp) -- no extra round trip
    if (P.paged) HIP_TRY(ctx, hipMemcpyAsync(ctx->alloc_host, ctx->log_alloc.p, sizeof(unsigned long long) * ctx->log_state.subs * IVP_LOG_ALLOC_STRIDE, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(P.round_done, s));
    return IVP_OK;
}

// the round's active count has arrived: finish the solve or enqueue the next round
int finish_round(ivp_ctx *ctx, int *done)
{
    ivp_ctx::Pending &P = ctx->pend;
    if (!P.err_checked) {
        P.err_checked = true;
        if (ctx->pinned[1] & 0x1u) {  // IVP_ERRFLAG_INVALID_STEP: RK4::solve's Err(InvalidStepSize), rk4.rs:81-87
            set_active(ctx, false);
            return fail(ctx, IVP_ERR_INVALID_STEP_SIZE, "RK4: step size is zero or its sign does not match xend - x0 for at least one trajectory");
        }
    }
    // Chunk length from the observed decay of the active set: three rounds in a row that retired nobody (< 1 %) say that the
    // trajectories are long compared with the chunk (two were not enough for BASELINE C3, whose retirements start right
    // after 384 attempts: the doubled round then cost 3 %) -- launch boundaries only cost (state round trip, launch gap, the
    // wait for each launch's slowest wave), so the next round's launches run twice as many attempts (up to 256); as
    // soon as trajectories start to retire (> 3 % in a round), compaction matters again and the chunk returns to its base.
    {
        const uint32_t before = P.lanes, after = ctx->pinned[0];
        // (problems with a cooperative tail keep the base chunk: their hand-over happens at launch boundaries, and measured
        // on C2 at rtol 1e-10 coarser boundaries cost 8 % where problems without one gain 3-5 %)
        const bool has_coop_tail = P.coop_ok && P.variant == 0 && P.n >= 4 && !P.jit;
        if (P.adaptive && !has_coop_tail) {
            if ((uint64_t)after * 100u >= (uint64_t)before * 99u) {
                P.quiet_rounds += 1;
                if (P.quiet_rounds >= 3) P.chunk_now = std::min(P.chunk_now * 2u, std::max(P.chunk, 256u));   // three quiet rounds in a row
            } else {
                P.quiet_rounds = 0;
                if ((uint64_t)after * 100u < (uint64_t)before * 97u) P.chunk_now = P.chunk;
            }
        }
    }
    P.lanes = ctx->pinned[0];
    if (P.lds_lu_ok && ctx->pinned[3] >= 64u) P.lu_dense = (uint64_t)ctx->pinned[2] * 2u > (uint64_t)ctx->pinned[3];   // large-n BDF: see enqueue_round
    if (P.lanes != 0) return enqueue_round(ctx);
    if (P.a.evd_rec != nullptr && !P.sampled) {
        // deferred event refinement: the roots of every noted step, one lane each (same hand-over as the sample kernel below)
        P.sampled = true;
        LAUNCH_TRY(ctx, pend_launch(ctx, IVP_LAUNCH_EVENTS, P.a, (uint32_t)P.B, false, false));
        HIP_TRY(ctx, hipEventRecord(P.round_done, P.stream));
        return IVP_OK;
    }
    if (P.full == 3 && !P.sampled) {
        // deferred t_eval sampling: every trajectory has finished stepping; its noted steps are evaluated now, one lane each
        // (the solve is complete when THIS kernel is: one more turn of the round-done event)
        P.sampled = true;
        const bool fast = P.fp_mode == IVP_FP_FAST;
        LAUNCH_TRY(ctx, (fast ? ivp_launch_fast : ivp_launch_strict)(IVP_LAUNCH_SAMPLE, P.method, P.prob.rhs_id, 3, P.a, (uint32_t)P.B, P.stream));
        HIP_TRY(ctx, hipEventRecord(P.round_done, P.stream));
        return IVP_OK;
    }
    set_active(ctx, false);
    *done = 1;
    if (P.paged) {   // one-pass step log: the pool holds every record unless it ran dry on the way (ivp_log.cpp decides what follows)
        ctx->log_state.overflow = (ctx->pinned[1] & 0x2u) != 0;   // IVP_ERRFLAG_LOG_OVERFLOW
        ctx->log_state.valid = !ctx->log_state.overflow;
    }
    if (P.profile) {
        hipStream_t s = P.stream;
        hipEvent_t ev_end = pend_event(ctx);
        HIP_TRY(ctx, hipEventRecord(ev_end, s));
        HIP_TRY(ctx, hipEventSynchronize(ev_end));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, P.ev_t0, P.ev_init1));
        ctx->stats.init_kernel_ms = ms;
        // which launches did work (a launch of a (bulk, cooperative) pair that declined set no flag: its few microseconds
        // count as stepping time of the kernel kind that ran, and it is not counted as a launch)
        std::vector<uint32_t> ran(std::min(P.step_ev.size(), (size_t)kMaxRanSlots), 1u);
        if (!ran.empty()) HIP_TRY(ctx, hipMemcpy(ran.data(), ctx->ran.p, ran.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < P.step_ev.size(); ++q) {   // step_is_coop: bit 0 = cooperative kernel, bit 1 = half of a pair (has a ran flag)
            HIP_TRY(ctx, hipEventElapsedTime(&ms, P.step_ev[q].first, P.step_ev[q].second));
            const bool coop = (P.step_is_coop[q] & 1) != 0, flagged = (P.step_is_coop[q] & 2) != 0;
            const bool did = !flagged || q >= ran.size() || ran[q] != 0;
            if (trace_launches()) fprintf(stderr, "ivp launch %3zu  %-4s %-8s lanes<=%-8u %8.4f ms\n", q, coop ? "coop" : "bulk", did ? "ran" : "declined", P.step_lanes[q], ms);
            ctx->stats.step_kernel_ms += ms;
            if (coop) ctx->stats.coop_kernel_ms += ms;
            if (did) {
                ctx->stats.launches += 1;
                if (coop) ctx->stats.coop_launches += 1;
            } else {
                ctx->stats.declined_launches += 1;
                ctx->stats.declined_ms += ms;
                if (coop) { ctx->stats.declined_coop_launches += 1; ctx->stats.declined_coop_ms += ms; }
            }
        }
        HIP_TRY(ctx, hipEventElapsedTime(&ms, P.ev_t0, ev_end));
        ctx->stats.total_ms = ms;
        unsigned long long slots[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpy(slots, ctx->slot.p, sizeof slots, hipMemcpyDeviceToHost));
        ctx->stats.lane_attempt_slots = slots[0];
        ctx->stats.lane_launches = slots[1];
        if (P.profile >= 2) {
            const size_t B = P.B;
            std::vector<uint64_t> tmp(B);
            HIP_TRY(ctx, hipMemcpy(tmp.data(), P.a.naccpt, sizeof(uint64_t) * B, hipMemcpyDeviceToHost));
            uint64_t acc = 0, att = 0;
            for (uint64_t v : tmp) acc += v;
            ctx->stats.total_accepted = acc;
            if (P.method == IVP_RK23) {  // RK23 counts only accepted steps in nstep (rk23.rs:238)
                HIP_TRY(ctx, hipMemcpy(tmp.data(), P.a.nrejct, sizeof(uint64_t) * B, hipMemcpyDeviceToHost));
                att = acc;
                for (uint64_t v : tmp) att += v;
            } else {
                HIP_TRY(ctx, hipMemcpy(tmp.data(), P.a.nstep, sizeof(uint64_t) * B, hipMemcpyDeviceToHost));
                for (uint64_t v : tmp) att += v;
            }
            ctx->stats.total_attempts = att;
        }
    }
    return IVP_OK;
}

}  // namespace

extern "C" {

int ivp_abi_version(void) { return IVP_HIP_ABI_VERSION; }

int ivp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ivp_ctx_create(ivp_ctx_t **out, int device)
{
    if (!out) return IVP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return IVP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return IVP_ERR_BAD_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return IVP_ERR_HIP;
    ivp_ctx *c = new ivp_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        c->cus = (uint32_t)prop.multiProcessorCount;
        c->simds = c->cus * 4u;   // CDNA: four SIMDs per compute unit
    }
    if (hipHostMalloc((void **)&c->pinned, 64 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        delete c;
        return IVP_ERR_HIP;
    }
    *out = c;
    return IVP_OK;
}

void ivp_ctx_destroy(ivp_ctx_t *c)
{
    if (!c) return;
    set_active(c, false);   // a context destroyed with a solve in flight must not stay in the count
    (void)hipSetDevice(c->device);
    DevBuf *bufs[] = {&c->k1, &c->facold, &c->hlamb, &c->flags, &c->perm[0], &c->perm[1], &c->counts, &c->slot, &c->ran, &c->teval, &c->teval_off, &c->evcfg, &c->tolvec, &c->zero_off,
                      &c->sc_y, &c->sc_x, &c->sc_h, &c->sc_status, &c->sc_nfev, &c->sc_nstep, &c->sc_naccpt, &c->sc_nrejct,
                      &c->sc_next_idx, &c->sc_n_filled, &c->sc_n_log, &c->sc_n_seg, &c->sc_t_last,
                      &c->bdf_d, &c->bdf_jac, &c->bdf_lu, &c->bdf_piv, &c->sc_njev, &c->sc_nlu, &c->prev_event, &c->sc_n_ev,
                      &c->st_y0, &c->st_params, &c->st_t0, &c->st_t1, &c->st_logoff,
                      &c->log_pool, &c->log_alloc, &c->def_rec, &c->evd_rec, &c->evd_cnt, &c->log_off, &c->log_bsum, &c->st_log_t, &c->st_log_y};
    for (DevBuf *b : bufs) b->release();
    for (DevBuf &b : c->st_out) b.release();
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->pend.round_done) (void)hipEventDestroy(c->pend.round_done);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->alloc_host) (void)hipHostFree(c->alloc_host);
    delete c;
}

const char *ivp_last_error_string(const ivp_ctx_t *c) { return c ? c->err.c_str() : "no context"; }

int ivp_ctx_get_stats(const ivp_ctx_t *c, ivp_run_stats_t *s)
{
    if (!c || !s) return IVP_ERR_BAD_ARGUMENT;
    *s = c->stats;
    return IVP_OK;
}

void ivp_options_default(ivp_options_t *o)
{   // Options::builder().build(), src/solve/options.rs:75-123
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->method = IVP_DOPRI5;
    o->rtol = 1e-3;
    o->atol = 1e-6;
    o->fp_mode = IVP_FP_STRICT;
}

int ivp_options_method_defaults(ivp_options_t *o, int32_t method)
{
    if (!o || (method != IVP_RK23 && method != IVP_DOPRI5 && method != IVP_DOP853)) return IVP_ERR_BAD_ARGUMENT;
    const MethodSettings m = method_defaults(method);
    o->method = method;
    o->uround = m.uround;
    o->safety_factor = m.safety;
    o->scale_min = m.scale_min;
    o->scale_max = m.scale_max;
    o->beta = m.beta;
    o->stiff_test = m.nstiff;
    return IVP_OK;
}

int ivp_rhs_dims(int32_t rhs_id, int32_t *n, int32_t *np)
{
    int gn = 0, gp = 0;
    if (group_builtin(rhs_id, &gn, &gp)) {
        if (n) *n = gn;
        if (np) *np = gp;
        return IVP_OK;
    }
    if (rhs_id < 0 || rhs_id >= IVP_RHS_BUILTIN_COUNT) return IVP_ERR_BAD_ARGUMENT;
    if (n) *n = kRhsDims[rhs_id].n;
    if (np) *np = kRhsDims[rhs_id].p;
    return IVP_OK;
}

int ivp_rhs_n_events(int32_t rhs_id)
{
    if (rhs_id < 0 || rhs_id >= IVP_RHS_BUILTIN_COUNT) return 0;
    return kRhsEvents[rhs_id];
}

int ivp_batch_submit_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0, const double *params,
                            const double *t0, size_t t0_len, const double *t1, size_t t1_len, const ivp_options_t *opt,
                            ivp_batch_result_t *out, void *hip_stream)
{
    if (!ctx) return IVP_ERR_BAD_ARGUMENT;
    if (ctx->pend.active) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "a solve is already in flight on this context");
    ctx->err.clear();
    int n = 0, np = 0;
    int rc = validate(ctx, prob, B, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    if (!y0 || !t0 || !t1 || !out) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null y0/t0/t1/out");
    if (np > 0 && !params) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "params required (n_params=%d)", np);
    if ((t0_len != 1 && t0_len != B) || (t1_len != 1 && t1_len != B)) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t0/t1 length must be 1 or B");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    // one-pass step log: ivp_log.cpp set the plan for THIS submit (Solution.t / Solution.y into the page pool)
    const bool paged = ctx->log_plan.want;
    const uint64_t log_reserve = ctx->log_plan.reserve;
    ctx->log_plan = ivp_ctx::LogPlan{};
    if (paged) { ctx->log_state.valid = false; ctx->log_state.overflow = false; }
    if (paged && opt->t_eval) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "the accepted-step log is what solve_ivp records when t_eval is None");

    const bool want_eval = opt->t_eval != nullptr;
    const bool csr_log = !paged && !want_eval && out->log_offsets && out->t_log && out->y_log;    // fill pass of the CSR step log
    const bool count_log = !paged && !want_eval && opt->count_log != 0 && !csr_log;               // counting pass: n_log only
    if (opt->count_log && !out->n_log) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "count_log needs out.n_log");
    const bool want_log = paged || csr_log || count_log || (!want_eval && opt->max_log > 0 && out->t_log && out->y_log);
    const bool want_dense = opt->dense_output && opt->max_log > 0 && out->seg_cont && out->seg_xold && out->seg_h;
    const bool group = n > IVP_MAX_N;
    const int n_events = prob->rhs_id == IVP_RHS_JIT ? ivp_jit_n_events(prob->jit) : group ? 0 : kRhsEvents[prob->rhs_id];
    if (n_events > 4 && !(opt->ev_direction_vec && opt->ev_terminal_vec && opt->n_event_cfg == n_events))
        return fail(ctx, IVP_ERR_BAD_ARGUMENT, "%d event functions need ev_direction_vec / ev_terminal_vec with n_event_cfg = %d entries", n_events, n_events);
    if (opt->t_eval_offsets) {   // per-trajectory grids: B + 1 offsets into the concatenated t_eval
        if (!opt->t_eval) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets without t_eval");
        if (opt->t_eval_offsets[0] != 0 || opt->t_eval_offsets[B] != (uint64_t)opt->n_eval)
            return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must run from 0 to n_eval");
        for (size_t b = 0; b < B; ++b) {
            if (opt->t_eval_offsets[b + 1] < opt->t_eval_offsets[b]) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must be non-decreasing");
            if (opt->t_eval_offsets[b + 1] - opt->t_eval_offsets[b] > 0x7FFFFFFFull) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "a trajectory's t_eval grid is too long");
        }
    }
    const bool full = want_eval || want_log || want_dense || n_events > 0;
    if (want_eval && opt->n_eval > 0 && !out->y_eval) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t_eval given but out.y_eval is NULL");
    if (opt->dense_output && !want_dense) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "dense_output needs max_log > 0 and seg_cont/seg_xold/seg_h");

    IvpKArgs a;
    std::memset(&a, 0, sizeof a);
    a.B = (uint32_t)B;
    a.y0 = y0;
    a.params = params;
    a.t0 = t0;
    a.t1 = t1;
    a.t0_stride = t0_len == 1 ? 0u : 1u;
    a.t1_stride = t1_len == 1 ? 0u : 1u;
    for (int i = 0; i < IVP_MAX_N; ++i) {
        a.rtol[i] = (opt->rtol_vec && i < n && n <= IVP_MAX_N) ? opt->rtol_vec[i] : opt->rtol;
        a.atol[i] = (opt->atol_vec && i < n && n <= IVP_MAX_N) ? opt->atol_vec[i] : opt->atol;
    }
    if (n > IVP_MAX_N && (opt->rtol_vec || opt->atol_vec)) {   // Tolerance::Vector for a large-n problem: [n] on the device
        std::vector<double> tv(2 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            tv[i] = opt->rtol_vec ? opt->rtol_vec[i] : opt->rtol;
            tv[n + i] = opt->atol_vec ? opt->atol_vec[i] : opt->atol;
        }
        HIP_TRY(ctx, ctx->tolvec.reserve(sizeof(double) * 2 * n));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tolvec.p, tv.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
        HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)hip_stream));   // tv is a stack-lifetime host buffer
        a.rtol_dev = (const double *)ctx->tolvec.p;
        a.atol_dev = (const double *)ctx->tolvec.p + n;
    }
    a.first_step = opt->first_step;
    a.max_step = opt->max_step;
    a.has_first_step = opt->has_first_step ? 1 : 0;
    a.has_max_step = opt->has_max_step ? 1 : 0;
    a.nmax = opt->max_steps ? opt->max_steps : UINT64_MAX;  // None => usize::MAX, solve_ivp.rs:218
    a.min_step = opt->min_step;
    a.has_min_step = opt->has_min_step ? 1 : 0;
    {   // controller settings: IEEE operations exactly as XXX::solve() derives them (no contraction on the host)
        const MethodSettings m = settings_of(opt);
        volatile double prod = m.beta * (opt->method == IVP_DOP853 ? 0.2 : 0.75);
        a.ctl_uround = m.uround;
        a.ctl_safety = m.safety;
        a.ctl_facc1 = 1.0 / m.scale_min;
        a.ctl_facc2 = 1.0 / m.scale_max;
        a.ctl_beta = m.beta;
        a.ctl_expo1 = (opt->method == IVP_DOP853 ? 1.0 / 8.0 : 0.2) - prod;   // dop853.rs:229, dopri5.rs:226
        a.ctl_scale_min = m.scale_min;
        a.ctl_scale_max = m.scale_max;
        a.ctl_nstiff = m.nstiff;
        a.has_ctl = opt->has_settings ? 1 : 0;
    }

    // ---- state / result arrays: the caller's buffers where given, context scratch otherwise ----
#define BIND(field, userptr, scratch, bytes)                                   \
    do {                                                                       \
        if (userptr) a.field = userptr;                                        \
        else {                                                                 \
            HIP_TRY(ctx, ctx->scratch.reserve(bytes));                         \
            a.field = (decltype(a.field))ctx->scratch.p;                       \
        }                                                                      \
    } while (0)
    BIND(y, out->y_end, sc_y, sizeof(double) * n * B);
    BIND(x, out->t_end, sc_x, sizeof(double) * B);
    BIND(h, out->h_next, sc_h, sizeof(double) * B);
    BIND(status, out->status, sc_status, sizeof(int32_t) * B);
    BIND(nfev, out->nfev, sc_nfev, sizeof(uint64_t) * B);
    BIND(nstep, out->nstep, sc_nstep, sizeof(uint64_t) * B);
    BIND(naccpt, out->naccpt, sc_naccpt, sizeof(uint64_t) * B);
    BIND(nrejct, out->nrejct, sc_nrejct, sizeof(uint64_t) * B);
    if (opt->method == IVP_BDF) {
        BIND(njev, out->njev, sc_njev, sizeof(uint64_t) * B);
        BIND(nlu, out->nlu, sc_nlu, sizeof(uint64_t) * B);
        HIP_TRY(ctx, ctx->bdf_d.reserve(sizeof(double) * 8 * n * B));
        HIP_TRY(ctx, ctx->bdf_jac.reserve(sizeof(double) * n * n * B));
        HIP_TRY(ctx, ctx->bdf_lu.reserve(sizeof(double) * n * n * B));
        HIP_TRY(ctx, ctx->bdf_piv.reserve(sizeof(uint32_t) * B * (group ? (size_t)n : 1)));   // large n: [B][n] pivot rows
        a.bdf_d = (double *)ctx->bdf_d.p;
        a.bdf_jac = (double *)ctx->bdf_jac.p;
        a.bdf_lu = (double *)ctx->bdf_lu.p;
        a.bdf_piv = (uint32_t *)ctx->bdf_piv.p;
    } else {
        a.njev = (uint64_t *)out->njev;   // njev = nlu = 0 for explicit RK: written by the init kernel (NULL = not wanted)
        a.nlu = (uint64_t *)out->nlu;
    }
    HIP_TRY(ctx, ctx->k1.reserve(sizeof(double) * n * B));
    HIP_TRY(ctx, ctx->facold.reserve(sizeof(double) * B));
    HIP_TRY(ctx, ctx->hlamb.reserve(sizeof(double) * B));
    HIP_TRY(ctx, ctx->flags.reserve(sizeof(uint32_t) * B));
    HIP_TRY(ctx, ctx->perm[0].reserve(sizeof(uint32_t) * B));
    HIP_TRY(ctx, ctx->perm[1].reserve(sizeof(uint32_t) * B));
    HIP_TRY(ctx, ctx->counts.reserve(sizeof(uint32_t) * 8));  // [0..3] active-count ring, [4] error flags
    a.k1 = (double *)ctx->k1.p;
    a.facold = (double *)ctx->facold.p;
    a.hlamb = (double *)ctx->hlamb.p;
    a.flags = (uint32_t *)ctx->flags.p;

    a.n_eval = -1;
    // Deferred t_eval sampling (kernel flavour 3): DOP853 with Options.t_eval and nothing else asked of the device DefaultSolOut,
    // built-in thread-per-trajectory problems.  The stepping kernels note the sampled steps, a second kernel with one lane per
    // noted step evaluates the dense stages and the samples (rk_core.h so_defer_samples / dop853_sample_body): in lock-step a
    // wave pays DOP853's three dense stages whenever ANY of its trajectories has a sample in the step.
    const bool defer_eval = want_eval && opt->method == IVP_DOP853 && !want_dense && n_events == 0 && !group && prob->rhs_id != IVP_RHS_JIT &&
                            opt->n_eval > 0 && tune().defer_eval != 0;
    if (n_events > 0) {   // event state: prev_event, hit counters; outputs where given
        for (int i = 0; i < 4; ++i) { a.ev_direction[i] = opt->ev_direction[i]; a.ev_terminal[i] = opt->ev_terminal[i]; }
        if (opt->ev_direction_vec && opt->ev_terminal_vec && opt->n_event_cfg == n_events) {   // any number of event functions
            HIP_TRY(ctx, ctx->evcfg.reserve(sizeof(int32_t) * 2 * (size_t)n_events));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->evcfg.p, opt->ev_direction_vec, sizeof(int32_t) * n_events, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync((int32_t *)ctx->evcfg.p + n_events, opt->ev_terminal_vec, sizeof(uint32_t) * n_events, hipMemcpyHostToDevice, s));
            a.ev_direction_dev = (const int32_t *)ctx->evcfg.p;
            a.ev_terminal_dev = (const uint32_t *)ctx->evcfg.p + n_events;
            for (int i = 0; i < 4 && i < n_events; ++i) { a.ev_direction[i] = opt->ev_direction_vec[i]; a.ev_terminal[i] = opt->ev_terminal_vec[i]; }
        }
        const bool store = out->t_events && out->y_events && opt->max_events > 0;
        a.max_events = store ? opt->max_events : 0;
        a.t_events = out->t_events;
        a.y_events = out->y_events;
        a.t_term = out->t_term;
        HIP_TRY(ctx, ctx->prev_event.reserve(sizeof(double) * n_events * B));
        a.prev_event = (double *)ctx->prev_event.p;
        if (out->n_event_hits) a.n_ev = out->n_event_hits;
        else { HIP_TRY(ctx, ctx->sc_n_ev.reserve(sizeof(uint32_t) * n_events * B)); a.n_ev = (uint32_t *)ctx->sc_n_ev.p; }
        HIP_TRY(ctx, hipMemsetAsync(a.n_ev, 0, sizeof(uint32_t) * n_events * B, s));
        // Deferred event refinement (rk_core.h so_events_note / so_events_deferred_body): with no terminal event nothing the
        // integration does depends on the roots, so the stepping kernels only note the steps that hold a crossing and a second
        // kernel refines them, one lane per noted step.  Explicit methods, thread-per-trajectory / lane-cooperative kernels.
        bool any_terminal = false;
        for (int i = 0; i < n_events; ++i) any_terminal = any_terminal || (a.ev_terminal_dev ? opt->ev_terminal_vec[i] : opt->ev_terminal[i < 4 ? i : 3]) != 0;
        if (!any_terminal && !group && opt->method != IVP_BDF && tune().defer_events != 0) {
            const uint64_t ncoef = opt->method == IVP_DOPRI5 ? 5 : (opt->method == IVP_DOP853 ? 8 : 4);
            const uint64_t fields = 4 + 3 * (uint64_t)n_events + (uint64_t)n + ncoef * (uint64_t)n;
            const uint64_t cap = std::max<uint64_t>((uint64_t)n_events * a.max_events, 1);   // every noted step fills at least one output slot
            const uint64_t bytes = sizeof(double) * cap * fields * B;
            size_t free_b = 0, total_b = 0;
            const bool fits = cap <= 0xFFFFFFFFull && (bytes <= ctx->evd_rec.cap || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes <= (free_b + ctx->evd_rec.cap) / 4));
            if (fits) {
                HIP_TRY(ctx, ctx->evd_rec.reserve((size_t)bytes));
                HIP_TRY(ctx, ctx->evd_cnt.reserve(sizeof(uint32_t) * B));
                HIP_TRY(ctx, hipMemsetAsync(ctx->evd_cnt.p, 0, sizeof(uint32_t) * B, s));
                a.evd_rec = (double *)ctx->evd_rec.p;
                a.evd_cnt = (uint32_t *)ctx->evd_cnt.p;
                a.evd_cap = (uint32_t)cap;
            }
        }
    }
    if (full) {
        if (want_eval) {
            a.n_eval = (int32_t)std::min<int64_t>(opt->n_eval, 0x7FFFFFFF);   // per-trajectory grids: only its sign is read (lengths come from the offsets)
            HIP_TRY(ctx, ctx->teval.reserve(sizeof(double) * std::max<int64_t>(opt->n_eval, 1)));
            if (opt->n_eval > 0)
                HIP_TRY(ctx, hipMemcpyAsync(ctx->teval.p, opt->t_eval, sizeof(double) * opt->n_eval, hipMemcpyHostToDevice, s));
            a.t_eval = (const double *)ctx->teval.p;
            a.y_eval = out->y_eval;
            a.eval_idx = out->eval_idx;
            if (opt->t_eval_offsets) {
                HIP_TRY(ctx, ctx->teval_off.reserve(sizeof(unsigned long long) * (B + 1)));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->teval_off.p, opt->t_eval_offsets, sizeof(unsigned long long) * (B + 1), hipMemcpyHostToDevice, s));
                a.teval_off = (const unsigned long long *)ctx->teval_off.p;
                a.teval_extra = n_events > 0 ? 1u : 0u;
            }
        }
        if (defer_eval) {   // noted steps: at most one per t_eval point of a trajectory (every noted step holds a sample)
            uint64_t cap = (uint64_t)opt->n_eval;
            if (opt->t_eval_offsets) {
                cap = 0;
                for (size_t b = 0; b < B; ++b) cap = std::max<uint64_t>(cap, opt->t_eval_offsets[b + 1] - opt->t_eval_offsets[b]);
            }
            cap = std::max<uint64_t>(cap, 1);
            HIP_TRY(ctx, ctx->def_rec.reserve(sizeof(double) * (size_t)cap * (size_t)(n + 4) * B));
            a.def_rec = (double *)ctx->def_rec.p;
            a.def_cap = (uint32_t)cap;
        }
        BIND(n_filled, out->n_filled, sc_n_filled, sizeof(int32_t) * B);
        BIND(n_log, out->n_log, sc_n_log, sizeof(uint32_t) * B);
        BIND(n_seg, out->n_seg, sc_n_seg, sizeof(uint32_t) * B);
        HIP_TRY(ctx, ctx->sc_next_idx.reserve(sizeof(int32_t) * B));
        HIP_TRY(ctx, ctx->sc_t_last.reserve(sizeof(double) * B));
        a.next_idx = (int32_t *)ctx->sc_next_idx.p;
        a.t_last = (double *)ctx->sc_t_last.p;
        a.max_log = (want_log || want_dense) ? opt->max_log : 0;
        if (paged) {
            // One-pass step log (ivp_kargs.h): wave pages drawn from a pool sized from the caller's estimate, the last logged
            // solve of this batch size on this context, or 1024 records per trajectory (at least 256 MB) -- a pool that turns
            // out too small costs a second integration (the counted fill pass), never a wrong or truncated log.  A page holds
            // a slot for every ATTEMPT of every trajectory its wave steps, so the pool takes ~1.25x the doubles of the records
            // (rejected attempts, retired lanes, column headers) -- 1.5x is reserved.
            ivp_ctx::LogState &LS = ctx->log_state;
            const bool learnt = LS.total && LS.last_B == B;
            const uint64_t recs = log_reserve ? log_reserve : (learnt ? LS.total : (uint64_t)B * 1024u);
            uint64_t doubles = recs * (uint64_t)(n + 1) / 2 * 3 + (uint64_t)B * (2u + 2u * (uint64_t)(n + 1)) + (1u << 16);
            // sub-pools: one per eight waves of the first launch, at most IVP_LOG_SUBPOOLS (they exist to spread the waves'
            // allocation atomics; a batch of few waves would leave most regions idle)
            uint32_t subs = 1;
            while (subs < IVP_LOG_SUBPOOLS && (uint64_t)subs * 2u * 512u <= (uint64_t)B * (group ? 64u : 1u)) subs *= 2;
            // what the last solve of this size really took: its fullest region, in every region
            if (learnt && !log_reserve && LS.subs == subs) doubles = std::max<uint64_t>(doubles, (LS.region_used_max + LS.region_used_max / 8) * subs);
            if (!log_reserve && !learnt) doubles = std::max<uint64_t>(doubles, ((uint64_t)256 << 20) / 8);
            if (doubles * 8 > ctx->log_pool.cap) {   // growing: never ask for more than half of what the device can give
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    const uint64_t most = (uint64_t)(free_b + ctx->log_pool.cap) / 2 / 8;
                    if (doubles > most) doubles = std::max<uint64_t>(most, ctx->log_pool.cap / 8);
                }
            }
            doubles = std::min<uint64_t>(std::max<uint64_t>(doubles, (uint64_t)1 << 16), (uint64_t)1 << 45);
            HIP_TRY(ctx, ctx->log_pool.reserve((size_t)doubles * 8));
            HIP_TRY(ctx, ctx->log_alloc.reserve(sizeof(unsigned long long) * IVP_LOG_SUBPOOLS * IVP_LOG_ALLOC_STRIDE));
            if (!ctx->alloc_host) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->alloc_host, sizeof(unsigned long long) * IVP_LOG_SUBPOOLS * IVP_LOG_ALLOC_STRIDE, hipHostMallocDefault));
            HIP_TRY(ctx, hipMemsetAsync(ctx->log_alloc.p, 0, sizeof(unsigned long long) * IVP_LOG_SUBPOOLS * IVP_LOG_ALLOC_STRIDE, s));
            a.log_pool = (double *)ctx->log_pool.p;
            a.log_region = ((ctx->log_pool.cap / 8) / subs) & ~(unsigned long long)15u;   // whole 128-byte lines (pages are; IVP_LOG_HDR); a region stays below 2^40 doubles: the counters' low field
            a.log_sub_mask = subs - 1u;
            a.log_alloc = (unsigned long long *)ctx->log_alloc.p;
            a.t_log = a.log_pool;   // "mode 2" marker of the device DefaultSolOut (so_sample); the records go to the pages
            a.y_log = a.log_pool;
            LS.B = B; LS.n = n; LS.pool_doubles = a.log_region * subs; LS.region = a.log_region; LS.subs = subs; LS.n_log = a.n_log;
        } else if (count_log) {
            // counting pass = a CSR log whose offsets are all zero: so_sample runs (t_log != NULL), every record finds
            // capacity 0 and only n_log advances
            HIP_TRY(ctx, ctx->zero_off.reserve(sizeof(unsigned long long) * (B + 1)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->zero_off.p, 0, sizeof(unsigned long long) * (B + 1), s));
            a.t_log = (double *)ctx->counts.p;
            a.y_log = (double *)ctx->counts.p;
            a.log_off = (const unsigned long long *)ctx->zero_off.p;
        } else if (want_log) {
            a.t_log = out->t_log;
            a.y_log = out->y_log;
            a.log_off = csr_log ? (const unsigned long long *)out->log_offsets : nullptr;
        }
        a.collect_dense = want_dense ? 1 : 0;
        a.seg_cont = out->seg_cont;
        a.seg_xold = out->seg_xold;
        a.seg_h = out->seg_h;
    }
#undef BIND

    const bool profile = opt->profile != 0;
    ctx->stats = ivp_run_stats_t{};
    if (profile) {
        HIP_TRY(ctx, ctx->slot.reserve(2 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->slot.p, 0, 2 * sizeof(unsigned long long), s));
        a.slot_counter = (unsigned long long *)ctx->slot.p;
        HIP_TRY(ctx, ctx->ran.reserve(kMaxRanSlots * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->ran.p, 0, kMaxRanSlots * sizeof(uint32_t), s));
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->counts.p, 0, sizeof(uint32_t) * 8, s));
    a.err_flag = (uint32_t *)ctx->counts.p + 4;
    // large-n BDF: the factors of (I - cJ) can live in LDS where the matrix fits (n <= 128, built-in problems: one wavefront
    // per trajectory); enqueue_round decides per launch (a.lds_lu), the results do not depend on it
    const bool lds_lu_ok = group && opt->method == IVP_BDF && prob->rhs_id != IVP_RHS_JIT && n <= 128 && opt->variant != 1 && tune().lds_lu != 0;

    ivp_ctx::Pending &P = ctx->pend;
    P.a = a;
    P.prob = *prob;
    P.method = opt->method;
    P.fp_mode = opt->fp_mode;
    P.profile = opt->profile;
    P.n = n;
    // the log-only flavour: every accepted step recorded and nothing else asked of DefaultSolOut (no t_eval, no first_step
    // enforcement, no dense-output segments, no event functions) -- then no interpolant is ever evaluated, the kernels skip
    // the dense-output coefficients and keep the registers (occupancy) of the end-state kernels
    const bool log_only = want_log && !want_eval && !want_dense && n_events == 0 && !opt->has_first_step;
    P.full = log_only ? 2 : (defer_eval ? 3 : (full ? 1 : 0));
    P.sampled = false;
    P.group = group;
    P.jit = prob->rhs_id == IVP_RHS_JIT;
    P.has_settings = opt->has_settings != 0;
    P.lds_lu_ok = lds_lu_ok;
    P.lu_dense = false;
    P.lu_probed = false;
    P.has_events = n_events > 0;
    P.paged = paged;
    // lane-cooperative DOPRI5 / DOP853 kernels (rk_coop.h: eight lanes per trajectory): available for problems with
    // n <= 8.  Results are bit-identical to the thread-per-trajectory kernels in both arithmetic modes, so the loop
    // switches to them for the latency-bound tail.
    P.coop_ok = !group && (opt->method == IVP_DOPRI5 || opt->method == IVP_DOP853);
    P.variant = (opt->variant == 3 && !P.coop_ok) ? 0 : opt->variant;
    P.chunk = opt->chunk_attempts > 0 ? (uint32_t)opt->chunk_attempts : tune().bulk_chunk;
    P.chunk_now = P.chunk;
    P.quiet_rounds = 0;
    P.adaptive = opt->chunk_attempts == 0;
    P.B = B;
    P.lanes = (uint32_t)B;
    P.c = 0;
    P.err_checked = false;
    P.stream = s;
    P.ev_used = 0;
    P.step_ev.clear();
    P.step_is_coop.clear();
    P.step_lanes.clear();
    if (!P.round_done) HIP_TRY(ctx, hipEventCreateWithFlags(&P.round_done, hipEventDisableTiming));

    // ---- init: f0, hinit / first_step, initial SolOut call ----
    IvpKArgs ka = P.a;
    ka.chunk = 0;
    ka.perm_in = nullptr;
    ka.count_in = nullptr;
    ka.perm_out = nullptr;
    ka.count_out = nullptr;
    if (profile) { P.ev_t0 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(P.ev_t0, s)); }
    LAUNCH_TRY(ctx, pend_launch(ctx, IVP_LAUNCH_INIT, ka, (uint32_t)B, false, false));
    if (profile) { P.ev_init1 = pend_event(ctx); HIP_TRY(ctx, hipEventRecord(P.ev_init1, s)); }
    ctx->stats.init_launches = 1;
    set_active(ctx, true);
    rc = enqueue_round(ctx);
    if (rc != IVP_OK) set_active(ctx, false);
    return rc;
}

int ivp_batch_poll(ivp_ctx_t *ctx, int *done)
{
    if (!ctx || !done) return IVP_ERR_BAD_ARGUMENT;
    *done = 0;
    if (!ctx->pend.active) { *done = 1; return IVP_OK; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const hipError_t q = hipEventQuery(ctx->pend.round_done);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return IVP_OK; }
    if (q != hipSuccess) { set_active(ctx, false); return fail(ctx, IVP_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q)); }
    const int rc = finish_round(ctx, done);
    if (rc != IVP_OK) set_active(ctx, false);
    return rc;
}

int ivp_batch_wait(ivp_ctx_t *ctx)
{
    if (!ctx) return IVP_ERR_BAD_ARGUMENT;
    while (ctx->pend.active) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const hipError_t e = hipEventSynchronize(ctx->pend.round_done);
        if (e != hipSuccess) { set_active(ctx, false); return fail(ctx, IVP_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e)); }
        int done = 0;
        const int rc = finish_round(ctx, &done);
        if (rc != IVP_OK) { set_active(ctx, false); return rc; }
    }
    return IVP_OK;
}

int ivp_batch_solve_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0, const double *params,
                           const double *t0, size_t t0_len, const double *t1, size_t t1_len, const ivp_options_t *opt,
                           ivp_batch_result_t *out, void *hip_stream)
{
    const int rc = ivp_batch_submit_device(ctx, prob, B, y0, params, t0, t0_len, t1, t1_len, opt, out, hip_stream);
    if (rc != IVP_OK) return rc;
    return ivp_batch_wait(ctx);
}

int ivp_batch_solve(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0, const double *params,
                    const double *t0, size_t t0_len, const double *t1, size_t t1_len, const ivp_options_t *opt,
                    ivp_batch_result_t *out)
{
    ivp_ctx_t *one[1] = {ctx};
    return ivp_batch_solve_multi_host(one, 1, prob, B, y0, params, t0, t0_len, t1, t1_len, opt, out);
}

}  // extern "C"

ivp_host::ResultShape ivp_host::result_shape(const ivp_problem_t *prob, const ivp_options_t *opt, int n)
{
    ResultShape r;
    r.n = (size_t)n;
    r.nev = prob->rhs_id == IVP_RHS_JIT ? (size_t)ivp_jit_n_events(prob->jit) : n > IVP_MAX_N ? 0 : (size_t)kRhsEvents[prob->rhs_id];
    const size_t ne = opt->t_eval ? (size_t)opt->n_eval : 0;
    r.ne_rows = opt->t_eval ? ne + (r.nev > 0 ? 1 : 0) : 0;   // a terminal event appends one more sample
    r.ml = opt->max_log;
    r.nc = (size_t)ncoef_of(opt->method) * n;
    r.mev = opt->max_events;
    return r;
}
void ivp_host::member_table(const ResultShape &r, MemberDesc (&m)[kMembers])
{
#define M(field, elem, rows) MemberDesc{offsetof(ivp_batch_result_t, field), (size_t)(elem), (size_t)(rows)}
    const MemberDesc t[kMembers] = {
        M(y_end, 8, r.n), M(t_end, 8, 1), M(status, 4, 1), M(nfev, 8, 1), M(nstep, 8, 1), M(naccpt, 8, 1), M(nrejct, 8, 1), M(h_next, 8, 1),
        M(y_eval, 8, r.ne_rows * r.n), M(eval_idx, 4, r.ne_rows), M(n_filled, 4, 1),
        M(t_log, 8, r.ml), M(y_log, 8, r.ml * r.n), M(n_log, 4, 1),
        M(seg_cont, 8, r.ml * r.nc), M(seg_xold, 8, r.ml), M(seg_h, 8, r.ml), M(n_seg, 4, 1),
        M(njev, 8, 1), M(nlu, 8, 1),
        M(t_events, 8, r.nev * r.mev), M(y_events, 8, r.nev * r.mev * r.n), M(n_event_hits, 4, r.nev), M(t_term, 8, 1),
    };
#undef M
    for (int i = 0; i < kMembers; ++i) m[i] = t[i];
}
// rows x (count elements) between two SoA arrays of different stride
hipError_t ivp_host::copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t elem, size_t count, size_t rows,
                     hipMemcpyKind kind, hipStream_t s)
{
    if (!rows || !count) return hipSuccess;
    if (dst_stride == count && src_stride == count) return hipMemcpyAsync(dst, src, elem * count * rows, kind, s);
    return hipMemcpy2DAsync(dst, dst_stride * elem, src, src_stride * elem, count * elem, rows, kind, s);
}

// device -> device, possibly across devices: peer-enabled 2-D copy (xGMI) when the runtime allows it, else one
// hipMemcpyPeerAsync per row
hipError_t ivp_host::copy_rows_peer(void *dst, int dst_dev, size_t dst_stride, const void *src, int src_dev, size_t src_stride, size_t elem,
                          size_t count, size_t rows, hipStream_t s)
{
    if (!rows || !count) return hipSuccess;
    if (dst_dev == src_dev) return copy_rows(dst, dst_stride, src, src_stride, elem, count, rows, hipMemcpyDeviceToDevice, s);
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, src_dev, dst_dev) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(dst_dev, 0);   // current device = src_dev (the caller set it)
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            return copy_rows(dst, dst_stride, src, src_stride, elem, count, rows, hipMemcpyDeviceToDevice, s);
        }
        (void)hipGetLastError();
    }
    for (size_t r = 0; r < rows; ++r) {
        const hipError_t e = hipMemcpyPeerAsync((char *)dst + r * dst_stride * elem, dst_dev, (const char *)src + r * src_stride * elem, src_dev,
                                                count * elem, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

namespace {

// per-trajectory t_eval grids: the sample records of trajectories [first, first + count) in the batch-wide CSR arrays
struct EvalRun { size_t first, count; };
inline EvalRun eval_run(const ivp_options_t *opt, size_t first, size_t count, bool has_events)
{
    const size_t e = has_events ? 1 : 0;   // a terminal event appends its own sample
    return EvalRun{(size_t)opt->t_eval_offsets[first] + first * e, (size_t)(opt->t_eval_offsets[first + count] - opt->t_eval_offsets[first]) + count * e};
}

// drive every submitted shard to completion from this thread; on an error the other solves are still drained
int drive_all(ivp_ctx_t *const *ctxs, const char *live, int n)
{
    std::vector<char> done(n, 0);
    int rc_first = IVP_OK, pending = 0;
    for (int i = 0; i < n; ++i) { done[i] = live[i] ? 0 : 1; pending += live[i] ? 1 : 0; }
    while (pending) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            if (done[i]) continue;
            int d = 0;
            const uint64_t before = ctxs[i]->pend.c;
            const int rc = ivp_batch_poll(ctxs[i], &d);
            if (rc != IVP_OK) { if (rc_first == IVP_OK) rc_first = rc; d = 1; }
            if (d) { done[i] = 1; --pending; }
            progressed = progressed || d || ctxs[i]->pend.c != before;
        }
        if (!progressed) std::this_thread::yield();   // every shard's round is still in flight: do not spin on the event queries
    }
    return rc_first;
}

}  // namespace

extern "C" {

int ivp_batch_solve_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B, const ivp_options_t *opt,
                          int32_t gather_device, ivp_batch_result_t *gathered)
{
    if (!shards || n_shards <= 0 || n_shards > 64) return IVP_ERR_BAD_ARGUMENT;
    DeviceGuard restore_device;   // hipSetDevice below must not leak into the caller (its allocations / launches follow the current device)
    ivp_ctx_t *c0 = nullptr;
    for (int i = 0; i < n_shards; ++i) {
        if (!shards[i].ctx) return IVP_ERR_BAD_ARGUMENT;
        if (!c0) c0 = shards[i].ctx;
        for (int k = 0; k < i; ++k)
            if (shards[k].ctx == shards[i].ctx) return fail(c0, IVP_ERR_BAD_ARGUMENT, "shards %d and %d share one context", k, i);
        if (shards[i].first > B || shards[i].count > B - shards[i].first)
            return fail(c0, IVP_ERR_BAD_ARGUMENT, "shard %d: [%zu, %zu) is outside the batch of %zu", i, shards[i].first, shards[i].first + shards[i].count, B);
    }
    if (gathered) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || gather_device < 0 || gather_device >= ndev)
            return fail(c0, IVP_ERR_BAD_ARGUMENT, "gather_device %d", gather_device);
    }
    std::vector<ivp_ctx_t *> ctxs(n_shards);
    std::vector<char> live(n_shards, 0);
    // Per-trajectory t_eval grids (ivp_options_t.t_eval_offsets): the offsets index the whole batch, a shard integrates a
    // slice -- every shard gets its own view of the options: its part of the concatenated grid, offsets re-based to 0.
    const bool grids = opt && opt->t_eval_offsets && opt->t_eval;
    std::vector<ivp_options_t> shard_opt(grids ? n_shards : 0);
    std::vector<std::vector<uint64_t>> shard_off(grids ? n_shards : 0);
    if (grids) {
        for (size_t b = 0; b < B; ++b)
            if (opt->t_eval_offsets[b + 1] < opt->t_eval_offsets[b]) return fail(c0, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must be non-decreasing");
        if (opt->t_eval_offsets[0] != 0 || opt->t_eval_offsets[B] != (uint64_t)opt->n_eval)
            return fail(c0, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must run from 0 to n_eval");
        for (int i = 0; i < n_shards; ++i) {
            const ivp_shard_t &sh = shards[i];
            const uint64_t lo = opt->t_eval_offsets[sh.first];
            shard_off[i].resize(sh.count + 1);
            for (size_t k = 0; k <= sh.count; ++k) shard_off[i][k] = opt->t_eval_offsets[sh.first + k] - lo;
            shard_opt[i] = *opt;
            shard_opt[i].t_eval = opt->t_eval + lo;
            shard_opt[i].n_eval = (int64_t)shard_off[i][sh.count];
            shard_opt[i].t_eval_offsets = shard_off[i].data();
        }
    }
    int rc = IVP_OK;
    for (int i = 0; i < n_shards && rc == IVP_OK; ++i) {
        ivp_shard_t &sh = shards[i];
        ctxs[i] = sh.ctx;
        if (sh.count == 0) continue;
        rc = ivp_batch_submit_device(sh.ctx, prob, sh.count, sh.y0, sh.params, sh.t0, sh.t0_len, sh.t1, sh.t1_len, grids ? &shard_opt[i] : opt, &sh.out,
                                     sh.hip_stream);
        if (rc == IVP_OK) live[i] = 1;
        else if (sh.ctx != c0) c0->err = sh.ctx->err;   // the caller reads the first context's message
    }
    const int rc_drive = drive_all(ctxs.data(), live.data(), n_shards);
    if (rc == IVP_OK && rc_drive != IVP_OK) {
        rc = rc_drive;
        for (int i = 0; i < n_shards; ++i) if (live[i] && !shards[i].ctx->err.empty() && shards[i].ctx != c0) { c0->err = shards[i].ctx->err; break; }
    }
    if (rc != IVP_OK || !gathered) return rc;

    // ---- gather: shard columns into the batch-wide arrays on gather_device ----
    int n = 0, np = 0;
    rc = validate(c0, prob, B ? B : 1, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    MemberDesc md[kMembers];
    member_table(result_shape(prob, opt, n), md);
    for (int i = 0; i < n_shards; ++i) {
        ivp_shard_t &sh = shards[i];
        if (sh.count == 0) continue;
        HIP_TRY(c0, hipSetDevice(sh.ctx->device));
        hipStream_t s = (hipStream_t)sh.hip_stream;
        for (int k = 0; k < kMembers; ++k) {
            void *dst = member(gathered, md[k]);
            const void *src = member(&sh.out, md[k]);
            if (!dst || !src) continue;
            // a CSR step log is not an SoA member: it is gathered below
            if (sh.out.log_offsets && (md[k].off == offsetof(ivp_batch_result_t, t_log) || md[k].off == offsetof(ivp_batch_result_t, y_log))) continue;
            if (grids && (md[k].off == offsetof(ivp_batch_result_t, y_eval) || md[k].off == offsetof(ivp_batch_result_t, eval_idx))) {
                // samples on per-trajectory grids are time-major CSR records (include/ivp_hip.h): a shard's records are one
                // contiguous run that starts at record offsets[first] + first * e of the batch-wide arrays
                const EvalRun run = eval_run(opt, sh.first, sh.count, result_shape(prob, opt, n).nev > 0);
                const size_t rec = md[k].off == offsetof(ivp_batch_result_t, y_eval) ? 8u * (size_t)n : 4u;
                HIP_TRY(c0, copy_rows_peer((char *)dst + run.first * rec, gather_device, run.count * rec, src, sh.ctx->device, run.count * rec, 1,
                                           run.count * rec, 1, s));
                continue;
            }
            HIP_TRY(c0, copy_rows_peer((char *)dst + sh.first * md[k].elem, gather_device, B, src, sh.ctx->device, sh.count, md[k].elem, sh.count, md[k].rows, s));
        }
    }
    // ---- CSR step logs (Solution.t / Solution.y of every trajectory): every shard's records are one contiguous run
    // [total_k] / [total_k][n]; in trajectory order the shards' runs follow each other, and a shard's offsets (which
    // start at 0 in its own buffers) are re-based by the number of records of the shards before it ----
    if (gathered->log_offsets) {
        if (!gathered->t_log || !gathered->y_log) return fail(c0, IVP_ERR_BAD_ARGUMENT, "gathered.log_offsets needs gathered.t_log and gathered.y_log");
        std::vector<int> order;
        for (int i = 0; i < n_shards; ++i) if (shards[i].count) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return shards[a].first < shards[b].first; });
        unsigned long long base = 0;
        std::vector<unsigned long long> off;
        for (int i : order) {
            ivp_shard_t &sh = shards[i];
            if (!sh.out.log_offsets || !sh.out.t_log || !sh.out.y_log)
                return fail(c0, IVP_ERR_BAD_ARGUMENT, "shard %d has no CSR step log (out.log_offsets / t_log / y_log) to gather", i);
            HIP_TRY(c0, hipSetDevice(sh.ctx->device));
            hipStream_t s = (hipStream_t)sh.hip_stream;
            off.resize(sh.count + 1);
            HIP_TRY(c0, hipMemcpyAsync(off.data(), sh.out.log_offsets, (sh.count + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(c0, hipStreamSynchronize(s));
            const unsigned long long total = off[sh.count];
            HIP_TRY(c0, copy_rows_peer((char *)gathered->t_log + base * 8, gather_device, total, sh.out.t_log, sh.ctx->device, total, 8, total, 1, s));
            HIP_TRY(c0, copy_rows_peer((char *)gathered->y_log + base * 8 * (size_t)n, gather_device, total * (size_t)n, sh.out.y_log, sh.ctx->device,
                                       total * (size_t)n, 8, total * (size_t)n, 1, s));
            for (size_t k = 0; k <= sh.count; ++k) off[k] += base;
            const bool last = i == order.back();
            HIP_TRY(c0, hipMemcpyAsync((unsigned long long *)gathered->log_offsets + sh.first, off.data(), (sh.count + (last ? 1 : 0)) * sizeof(unsigned long long),
                                       hipMemcpyHostToDevice, s));
            HIP_TRY(c0, hipStreamSynchronize(s));   // `off` is reused by the next shard
            base += total;
        }
    }
    for (int i = 0; i < n_shards; ++i) {
        if (shards[i].count == 0) continue;
        HIP_TRY(c0, hipSetDevice(shards[i].ctx->device));
        HIP_TRY(c0, hipStreamSynchronize((hipStream_t)shards[i].hip_stream));
    }
    return IVP_OK;
}

int ivp_batch_solve_multi_host(ivp_ctx_t *const *ctxs, int32_t n_ctx, const ivp_problem_t *prob, size_t B, const double *y0,
                               const double *params, const double *t0, size_t t0_len, const double *t1, size_t t1_len,
                               const ivp_options_t *opt, ivp_batch_result_t *out)
{
    if (!ctxs || n_ctx <= 0 || n_ctx > 64 || !ctxs[0]) return IVP_ERR_BAD_ARGUMENT;
    DeviceGuard restore_device;
    ivp_ctx_t *c0 = ctxs[0];
    for (int i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return IVP_ERR_BAD_ARGUMENT;
        for (int k = 0; k < i; ++k) if (ctxs[k] == ctxs[i]) return fail(c0, IVP_ERR_BAD_ARGUMENT, "contexts %d and %d are the same", k, i);
    }
    c0->err.clear();
    int n = 0, np = 0;
    int rc = validate(c0, prob, B, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    if (!y0 || !t0 || !t1 || !out) return fail(c0, IVP_ERR_BAD_ARGUMENT, "null y0/t0/t1/out");
    if (np > 0 && !params) return fail(c0, IVP_ERR_BAD_ARGUMENT, "params required (n_params=%d)", np);
    if ((t0_len != 1 && t0_len != B) || (t1_len != 1 && t1_len != B)) return fail(c0, IVP_ERR_BAD_ARGUMENT, "t0/t1 length must be 1 or B");
    // CSR step log through host pointers: out->log_offsets [B + 1] is READ here (the caller's exclusive scan of a counting
    // pass's n_log); every shard gets device buffers of exactly its own record count and its slice of the offsets,
    // re-based to 0, and its records return to the host arrays at the batch-wide offsets
    const bool csr_log = out->log_offsets != nullptr;
    if (csr_log && (!out->t_log || !out->y_log)) return fail(c0, IVP_ERR_BAD_ARGUMENT, "out.log_offsets needs out.t_log and out.y_log");
    if (csr_log)
        for (size_t b = 0; b < B; ++b)
            if (out->log_offsets[b + 1] < out->log_offsets[b]) return fail(c0, IVP_ERR_BAD_ARGUMENT, "log_offsets must be non-decreasing");
    auto is_log_member = [&](const MemberDesc &d) { return csr_log && (d.off == offsetof(ivp_batch_result_t, t_log) || d.off == offsetof(ivp_batch_result_t, y_log)); };
    auto log_rec_bytes = [&](const MemberDesc &d) -> size_t { return d.off == offsetof(ivp_batch_result_t, y_log) ? 8u * (size_t)n : 8u; };
    std::vector<unsigned long long> log_local;
    const bool grids = opt->t_eval_offsets && opt->t_eval;
    if (opt->t_eval_offsets && !opt->t_eval) return fail(c0, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets without t_eval");
    if (grids) {   // checked HERE, before any staging size is derived from them (a malformed array must not underflow a size_t)
        if (opt->t_eval_offsets[0] != 0 || opt->t_eval_offsets[B] != (uint64_t)opt->n_eval)
            return fail(c0, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must run from 0 to n_eval");
        for (size_t b = 0; b < B; ++b) {
            if (opt->t_eval_offsets[b + 1] < opt->t_eval_offsets[b]) return fail(c0, IVP_ERR_BAD_ARGUMENT, "t_eval_offsets must be non-decreasing");
            if (opt->t_eval_offsets[b + 1] - opt->t_eval_offsets[b] > 0x7FFFFFFFull) return fail(c0, IVP_ERR_BAD_ARGUMENT, "a trajectory's t_eval grid is too long");
        }
    }
    const bool grid_events = grids && result_shape(prob, opt, n).nev > 0;
    auto is_eval_member = [&](const MemberDesc &d) { return grids && (d.off == offsetof(ivp_batch_result_t, y_eval) || d.off == offsetof(ivp_batch_result_t, eval_idx)); };
    auto eval_rec_bytes = [&](const MemberDesc &d) -> size_t { return d.off == offsetof(ivp_batch_result_t, y_eval) ? 8u * (size_t)n : 4u; };
    MemberDesc md[kMembers];
    member_table(result_shape(prob, opt, n), md);

    // contiguous balanced shards: the first B % n_ctx get one more trajectory
    std::vector<ivp_shard_t> sh(n_ctx);
    const size_t base = B / (size_t)n_ctx, extra = B % (size_t)n_ctx;
    size_t first = 0;
    for (int i = 0; i < n_ctx; ++i) {
        ivp_ctx *ctx = ctxs[i];
        ivp_shard_t &S = sh[i];
        std::memset(&S, 0, sizeof S);
        S.ctx = ctx;
        S.first = first;
        S.count = base + ((size_t)i < extra ? 1 : 0);
        first += S.count;
        if (S.count == 0) continue;
        HIP_TRY(c0, hipSetDevice(ctx->device));
        const size_t m = S.count;
        const size_t l0 = t0_len == 1 ? 1 : m, l1 = t1_len == 1 ? 1 : m;
        HIP_TRY(c0, ctx->st_y0.reserve(sizeof(double) * n * m));
        HIP_TRY(c0, ctx->st_params.reserve(sizeof(double) * std::max(np, 1) * m));
        HIP_TRY(c0, ctx->st_t0.reserve(sizeof(double) * l0));
        HIP_TRY(c0, ctx->st_t1.reserve(sizeof(double) * l1));
        // pageable host memory: these copies are staged by the runtime and return once the source has been read
        HIP_TRY(c0, copy_rows(ctx->st_y0.p, m, y0 + S.first, B, 8, m, (size_t)n, hipMemcpyHostToDevice, nullptr));
        if (np > 0) HIP_TRY(c0, copy_rows(ctx->st_params.p, m, params + S.first, B, 8, m, (size_t)np, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(c0, hipMemcpyAsync(ctx->st_t0.p, t0 + (t0_len == 1 ? 0 : S.first), sizeof(double) * l0, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(c0, hipMemcpyAsync(ctx->st_t1.p, t1 + (t1_len == 1 ? 0 : S.first), sizeof(double) * l1, hipMemcpyHostToDevice, nullptr));
        S.y0 = (const double *)ctx->st_y0.p;
        S.params = np > 0 ? (const double *)ctx->st_params.p : nullptr;
        S.t0 = (const double *)ctx->st_t0.p; S.t0_len = l0;
        S.t1 = (const double *)ctx->st_t1.p; S.t1_len = l1;
        for (int k = 0; k < kMembers; ++k) {   // device mirrors of every requested output
            const size_t bytes = is_eval_member(md[k]) ? std::max<size_t>(eval_run(opt, S.first, m, grid_events).count, 1) * eval_rec_bytes(md[k])
                                 : is_log_member(md[k]) ? std::max<size_t>((size_t)(out->log_offsets[S.first + m] - out->log_offsets[S.first]), 1) * log_rec_bytes(md[k])
                                                        : md[k].elem * md[k].rows * m;
            if (member(out, md[k]) && bytes) {
                HIP_TRY(c0, ctx->st_out[k].reserve(bytes));
                member(&S.out, md[k]) = ctx->st_out[k].p;
            }
        }
    }
    if (csr_log) {
        for (int i = 0; i < n_ctx; ++i) {
            ivp_shard_t &S = sh[i];
            if (S.count == 0) continue;
            HIP_TRY(c0, hipSetDevice(S.ctx->device));
            log_local.resize(S.count + 1);
            for (size_t k = 0; k <= S.count; ++k) log_local[k] = out->log_offsets[S.first + k] - out->log_offsets[S.first];
            HIP_TRY(c0, S.ctx->st_logoff.reserve(sizeof(unsigned long long) * (S.count + 1)));
            HIP_TRY(c0, hipMemcpy(S.ctx->st_logoff.p, log_local.data(), sizeof(unsigned long long) * (S.count + 1), hipMemcpyHostToDevice));
            S.out.log_offsets = (const uint64_t *)S.ctx->st_logoff.p;
        }
    }
    rc = ivp_batch_solve_multi(sh.data(), n_ctx, prob, B, opt, 0, nullptr);
    if (rc != IVP_OK) return rc;
    for (int i = 0; i < n_ctx; ++i) {
        ivp_shard_t &S = sh[i];
        if (S.count == 0) continue;
        HIP_TRY(c0, hipSetDevice(S.ctx->device));
        for (int k = 0; k < kMembers; ++k) {
            void *host = member(out, md[k]);
            const void *dev = member(&S.out, md[k]);
            if (host && dev && is_log_member(md[k])) {   // CSR step log: one contiguous run of records per shard
                const size_t rec = log_rec_bytes(md[k]);
                const size_t first = (size_t)out->log_offsets[S.first], count = (size_t)(out->log_offsets[S.first + S.count] - out->log_offsets[S.first]);
                if (count) HIP_TRY(c0, hipMemcpyAsync((char *)host + first * rec, dev, count * rec, hipMemcpyDeviceToHost, nullptr));
                continue;
            }
            if (host && dev && is_eval_member(md[k])) {   // CSR sample records: one contiguous run per shard
                const EvalRun run = eval_run(opt, S.first, S.count, grid_events);
                const size_t rec = eval_rec_bytes(md[k]);
                if (run.count) HIP_TRY(c0, hipMemcpyAsync((char *)host + run.first * rec, dev, run.count * rec, hipMemcpyDeviceToHost, nullptr));
            } else if (host && dev)
                HIP_TRY(c0, copy_rows((char *)host + S.first * md[k].elem, B, dev, S.count, md[k].elem, S.count, md[k].rows, hipMemcpyDeviceToHost, nullptr));
        }
    }
    for (int i = 0; i < n_ctx; ++i) {
        if (sh[i].count == 0) continue;
        HIP_TRY(c0, hipSetDevice(sh[i].ctx->device));
        HIP_TRY(c0, hipStreamSynchronize(nullptr));
    }
    return IVP_OK;
}

int ivp_rhs_compile(ivp_ctx_t *ctx, const char *ode_source, int32_t n, int32_t n_params, void **handle)
{
    return ivp_rhs_compile_events(ctx, ode_source, n, n_params, 0, handle);
}

int ivp_rhs_compile_events(ivp_ctx_t *ctx, const char *ode_source, int32_t n, int32_t n_params, int32_t n_events, void **handle)
{
    return ivp_rhs_compile_ex(ctx, ode_source, n, n_params, n_events, 0u, handle);
}

int ivp_rhs_compile_ex(ivp_ctx_t *ctx, const char *ode_source, int32_t n, int32_t n_params, int32_t n_events, uint32_t flags, void **handle)
{
    if (!ctx || !ode_source || !handle) return IVP_ERR_BAD_ARGUMENT;
    if (n < 1 || n > IVP_MAX_GROUP_N || n_params < 0 || n_params > IVP_MAX_P || n_events < 0 || n_events > 64)
        return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unsupported dimensions");
    if (flags & ~IVP_RHS_HAS_JAC) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "unknown flags 0x%x", flags);
    std::string log;
    int rc = ivp_jit_compile(ctx->device, ode_source, n, n_params, n_events, flags, handle, &log);
    if (rc != IVP_OK) ctx->err = log;
    return rc;
}

void ivp_rhs_free(void *handle) { ivp_jit_free(handle); }

}  // extern "C"
