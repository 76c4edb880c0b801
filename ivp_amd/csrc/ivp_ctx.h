// ivp_ctx.h -- PRIVATE host-side header of libivp_hip.so: the context object behind ivp_ctx_t and the helpers the
// translation units of the host library share (ivp_capi.cpp: validation, launch loop, SoA result plumbing; ivp_log.cpp:
// the one-pass accepted-step log).  Not part of the C ABI (include/ivp_hip.h is).
#pragma once
#include "../../include/ivp_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "ivp_kargs.h"

namespace ivp_host {

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace ivp_host

struct ivp_ctx {
    int device = 0;
    // the part's geometry (hipDeviceProp_t): the launch policy sizes everything from it.  CDNA compute units have
    // 4 SIMDs and wave64; multiProcessorCount is 256 on MI355X, 304 on MI300X, fewer on a partitioned device.
    uint32_t cus = 256, simds = 1024;
    uint32_t one_wave_per_simd() const { return simds * (uint32_t)IVP_WAVE; }   // lanes that fill every SIMD with one wave
    std::string err;
    // scratch (device)
    ivp_host::DevBuf k1, facold, hlamb, flags, perm[2], counts, slot, ran, teval, teval_off, evcfg, tolvec, zero_off;
    ivp_host::DevBuf sc_y, sc_x, sc_h, sc_status, sc_nfev, sc_nstep, sc_naccpt, sc_nrejct;
    ivp_host::DevBuf sc_next_idx, sc_n_filled, sc_n_log, sc_n_seg, sc_t_last;
    ivp_host::DevBuf bdf_d, bdf_jac, bdf_lu, bdf_piv, sc_njev, sc_nlu, prev_event, sc_n_ev;
    // staging for the host-pointer entry point
    ivp_host::DevBuf st_y0, st_params, st_t0, st_t1;
    ivp_host::DevBuf st_out[24];
    ivp_host::DevBuf st_logoff;   // this shard's CSR step-log offsets (host-pointer entry points)
    // ---- one-pass accepted-step log (ivp_log.cpp): the page pool the stepping kernels fill, the per-trajectory chain heads,
    // the offsets / block sums of the scan, staging for the records of the host-pointer and multi-device forms ----
    ivp_host::DevBuf log_pool, log_alloc, log_off, log_bsum, st_log_t, st_log_y;
    ivp_host::DevBuf def_rec;   // noted steps of the deferred t_eval sampling (kernel flavour 3)
    ivp_host::DevBuf evd_rec, evd_cnt;   // noted steps of the deferred event refinement (IvpKArgs.evd_rec) and their per-trajectory counts
    struct LogPlan {
        bool want = false;          // the next ivp_batch_submit_device on this context records into the page pool
        uint64_t reserve = 0;       // caller's estimate of the total number of records (0 = automatic)
    } log_plan;
    struct LogState {
        bool valid = false;         // the pool holds the complete log of the last solve (no overflow)
        bool overflow = false;      // ... the pool ran dry: n_log is exact, the records are not all there
        size_t B = 0;
        int n = 0;
        uint64_t pool_doubles = 0;  // capacity of the pool during that solve
        uint64_t region = 0;        // ... per sub-pool
        uint64_t pool_used = 0;     // doubles its pages took
        uint64_t region_used_max = 0;   // ... in the fullest sub-pool
        uint32_t subs = 1;          // sub-pools in use
        uint32_t max_arenas = 0;    // largest directory count among the sub-pools (grid of the gather)
        uint64_t total = 0;         // records of the last logged solve (sizes the next pool)
        size_t last_B = 0;          // batch size `total` belongs to
        const uint32_t *n_log = nullptr;   // device: the counts of the last logged solve (the caller's out->n_log, or scratch):
                                           // must stay untouched until the records have been fetched
    } log_state;
    uint32_t *pinned = nullptr;  // host-pinned: active count + misc
    unsigned long long *alloc_host = nullptr;   // host-pinned copy of the step-log sub-pool counters
    std::vector<hipEvent_t> events;
    ivp_run_stats_t stats{};
    // ---- the solve in flight (ivp_batch_submit_device .. ivp_batch_poll / ivp_batch_wait) ----
    struct Pending {
        bool active = false;
        IvpKArgs a;
        ivp_problem_t prob;
        int method = 0, fp_mode = 0, variant = 0, profile = 0, n = 0;
        int full = 0;   // kernel flavour (rk_launch.h): 0 end state, 1 whole DefaultSolOut, 2 log-only
        bool group = false, jit = false, coop_ok = false, has_settings = false, adaptive = false, lds_lu_ok = false, has_events = false;
        bool lu_dense = false;      // large-n BDF: the kernels' eliminations update most trailing columns (LDS factors pay at any batch size)
        bool lu_probed = false;     // ... and the short first launch that measures it has been enqueued
        uint32_t chunk = 64, lanes = 0;
        uint32_t chunk_now = 64;    // attempts per bulk launch of the next round (adaptive: follows the decay of the active set)
        uint32_t quiet_rounds = 0;  // consecutive rounds that retired (almost) nobody
        size_t B = 0;
        uint64_t c = 0;             // chunk launches so far
        bool spec = false;          // the last launch of the round in flight was a speculative cooperative one
        bool paged = false;         // this solve records its accepted steps into the page pool (one-pass step log)
        bool sampled = false;       // flavour 3 / deferred events: the second kernel has been enqueued (the solve is done when it is)
        bool err_checked = false;
        hipStream_t stream = nullptr;
        hipEvent_t round_done = nullptr;
        size_t ev_used = 0;
        hipEvent_t ev_t0 = nullptr, ev_init1 = nullptr;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> step_ev;
        std::vector<char> step_is_coop;
        std::vector<uint32_t> step_lanes;   // active count the host knew when it enqueued the launch (profiling trace)
    } pend;
};

namespace ivp_host {

inline int fail(ivp_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail((ctx), IVP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


// Every member of ivp_batch_result_t as (offset, element size, rows): a member is `rows` SoA rows of B elements.
struct MemberDesc { size_t off, elem, rows; };
struct ResultShape { size_t n, ne_rows, ml, nc, nev, mev; };
constexpr int kMembers = 24;
ResultShape result_shape(const ivp_problem_t *prob, const ivp_options_t *opt, int n);
void member_table(const ResultShape &r, MemberDesc (&m)[kMembers]);
inline void *&member(ivp_batch_result_t *r, const MemberDesc &d) { return *(void **)((char *)r + d.off); }
inline void *member(const ivp_batch_result_t *r, const MemberDesc &d) { return *(void *const *)((const char *)r + d.off); }
// rows x (count elements) between two SoA arrays of different stride
hipError_t copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t elem, size_t count, size_t rows,
                     hipMemcpyKind kind, hipStream_t s);
// device -> device, possibly across devices (peer-enabled 2-D copy over xGMI, else one hipMemcpyPeerAsync per row)
hipError_t copy_rows_peer(void *dst, int dst_dev, size_t dst_stride, const void *src, int src_dev, size_t src_stride, size_t elem,
                          size_t count, size_t rows, hipStream_t s);
// Options / problem validation shared by every entry point (ivp_capi.cpp)
int validate(ivp_ctx *ctx, const ivp_problem_t *prob, size_t B, const ivp_options_t *opt, int *n_out, int *p_out);
// restores the caller's current HIP device when a multi-device entry point returns
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace ivp_host
