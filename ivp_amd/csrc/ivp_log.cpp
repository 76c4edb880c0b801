// ivp_log.cpp -- the one-pass accepted-step log behind ivp_batch_solve_logged*(): Solution.t / Solution.y of a batch
// from ONE integration (the reference fills them while it integrates, src/solve/solout.rs:387-428, and returns them in
// src/solve/solve_ivp.rs:288-312).
//
// Flow of a logged solve on one context:
//   1. ivp_batch_submit_device with ctx->log_plan.want: the stepping kernels append every accepted step to wave pages in the
//      context's pool, chained per trajectory (ivp_kargs.h, so_log_open / so_push_log in rk_core.h), and count them in n_log;
//   2. exclusive scan of n_log -> offsets, total (log_gather.hip); the total travels to the host (8 bytes);
//   3. destination = the caller's buffers if they hold `total` records, else library-allocated;
//   4. the gather kernel walks the chains and writes the CSR log.
// If the pool ran dry on the way (IVP_ERRFLAG_LOG_OVERFLOW) the counts are still exact: step 4 is replaced by the counted
// FILL pass of the two-pass CSR log (a second integration writing straight to the destination), and the next logged
// solve of this batch size sizes its pool from the total it has learnt.  No arithmetic of the integration happens here.
#include "ivp_ctx.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "log_gather.h"

using namespace ivp_host;

namespace {

// the sub-pools' counters (arenas << 40 | doubles) of the context's last logged solve -- they arrived with the round that
// finished it (enqueue_round in ivp_capi.cpp): what the pages took, and the grid of the gather
void pool_stats(ivp_ctx *ctx)
{
    ivp_ctx::LogState &LS = ctx->log_state;
    LS.pool_used = 0;
    LS.region_used_max = 0;
    LS.max_arenas = 0;
    for (uint32_t q = 0; q < LS.subs; ++q) {
        const unsigned long long c = ctx->alloc_host[(size_t)q * IVP_LOG_ALLOC_STRIDE];
        const unsigned long long used = (c & ((1ull << 40) - 1ull)) + (c >> 40);
        LS.pool_used += used;
        LS.region_used_max = std::max<uint64_t>(LS.region_used_max, used);
        LS.max_arenas = std::max<uint32_t>(LS.max_arenas, (uint32_t)std::min<unsigned long long>(c >> 40, 0x3FFFFFFull));
    }
}

// offsets (device, [B + 1]) from the counts of the context's last logged solve, the total on its way to the host: enqueued
// on `s`, not waited for
int enqueue_scan(ivp_ctx *ctx, unsigned long long *offsets_dev, hipStream_t s)
{
    ivp_ctx::LogState &LS = ctx->log_state;
    HIP_TRY(ctx, ctx->log_bsum.reserve(ivp_log_scan_scratch_bytes(LS.B)));
    HIP_TRY(ctx, ivp_log_scan(LS.n_log, LS.B, offsets_dev, ctx->log_bsum.p, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned + 8, offsets_dev + LS.B, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    return IVP_OK;
}
uint64_t scanned_total(ivp_ctx *ctx)   // after the stream has been synchronised
{
    ivp_ctx::LogState &LS = ctx->log_state;
    unsigned long long t;
    std::memcpy(&t, ctx->pinned + 8, sizeof t);
    LS.total = t;
    LS.last_B = LS.B;
    return t;
}
// both, and the wait
int scan_counts(ivp_ctx *ctx, unsigned long long *offsets_dev, hipStream_t s, uint64_t *total)
{
    pool_stats(ctx);
    const int rc = enqueue_scan(ctx, offsets_dev, s);
    if (rc != IVP_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *total = scanned_total(ctx);
    return IVP_OK;
}

void fill_log_info(const ivp_ctx *ctx, ivp_step_log_t *log)
{
    const ivp_ctx::LogState &LS = ctx->log_state;
    log->total = LS.total;
    log->page_slots = IVP_LOG_SLOTS;
    log->pool_bytes = LS.pool_doubles * 8;
    log->pool_used_bytes = std::min<uint64_t>(LS.pool_used, LS.pool_doubles) * 8;
}

// the caller's buffers, or exactly `total` records of device memory owned by the log (released by ivp_step_log_free)
int device_destination(ivp_ctx *ctx, ivp_step_log_t *log, uint64_t total, int n, int device)
{
    if (!log->t && !log->y) {
        const size_t recs = (size_t)std::max<uint64_t>(total, 1);
        void *t = nullptr, *y = nullptr;
        HIP_TRY(ctx, hipMalloc(&t, recs * sizeof(double)));
        const hipError_t e = hipMalloc(&y, recs * sizeof(double) * (size_t)n);
        if (e != hipSuccess) { (void)hipFree(t); return fail(ctx, IVP_ERR_HIP, "hipMalloc of %zu log records: %s", recs, hipGetErrorString(e)); }
        log->t = (double *)t;
        log->y = (double *)y;
        log->capacity = recs;
        log->owned = 1;
        log->device = device;
        return IVP_OK;
    }
    if (!log->t || !log->y) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "ivp_step_log_t: t and y must both be given or both be NULL");
    if (log->capacity < total)
        return fail(ctx, IVP_ERR_LOG_CAPACITY, "the log has %llu records, t / y hold %llu", (unsigned long long)total, (unsigned long long)log->capacity);
    return IVP_OK;
}

int gather_pool(ivp_ctx *ctx, const unsigned long long *offsets_dev, uint64_t capacity, uint64_t dst_base, double *t, double *y, hipStream_t s)
{
    const ivp_ctx::LogState &LS = ctx->log_state;
    HIP_TRY(ctx, ivp_log_gather((const double *)ctx->log_pool.p, LS.region, (const unsigned long long *)ctx->log_alloc.p, LS.subs, LS.max_arenas, offsets_dev, LS.B, LS.n,
                                capacity, dst_base, t, y, s));
    return IVP_OK;
}

}  // namespace

extern "C" {

int ivp_batch_solve_logged_device(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0, const double *params,
                                  const double *t0, size_t t0_len, const double *t1, size_t t1_len, const ivp_options_t *opt,
                                  ivp_batch_result_t *out, ivp_step_log_t *log, void *hip_stream)
{
    if (!ctx) return IVP_ERR_BAD_ARGUMENT;
    if (!opt || !out || !log || !log->offsets) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null options / out / log / log->offsets");
    if (opt->t_eval) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "the accepted-step log is what solve_ivp records when t_eval is None");
    hipStream_t s = (hipStream_t)hip_stream;
    log->owned = 0; log->device = -1; log->passes = 0; log->total = 0; log->pool_bytes = 0; log->pool_used_bytes = 0; log->page_slots = 0;
    ivp_options_t o = *opt;
    o.count_log = 0;
    ivp_batch_result_t r = *out;
    r.t_log = nullptr; r.y_log = nullptr; r.log_offsets = nullptr;
    ctx->log_plan.want = true;
    ctx->log_plan.reserve = log->reserve;
    int rc = ivp_batch_solve_device(ctx, prob, B, y0, params, t0, t0_len, t1, t1_len, &o, &r, hip_stream);
    ctx->log_plan = ivp_ctx::LogPlan{};
    if (rc != IVP_OK) return rc;
    log->passes = 1;
    // offsets, and -- when the caller brought buffers -- the gather right behind the scan, without a round trip to the host in
    // between: the kernel itself leaves a log alone that does not fit (offsets[B] > capacity)
    pool_stats(ctx);
    rc = enqueue_scan(ctx, (unsigned long long *)log->offsets, s);
    if (rc != IVP_OK) return rc;
    const bool early = !log->defer && log->t && log->y && !ctx->log_state.overflow;
    if (early) {
        rc = gather_pool(ctx, (const unsigned long long *)log->offsets, log->capacity, 0, log->t, log->y, s);
        if (rc != IVP_OK) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    const uint64_t total = scanned_total(ctx);
    fill_log_info(ctx, log);
    if (log->defer) return IVP_OK;   // the caller fetches the records (ivp_step_log_fetch_device); after an overflow that fetch fails
    rc = device_destination(ctx, log, total, ctx->log_state.n, ctx->device);
    if (rc != IVP_OK) return rc;
    if (early) return IVP_OK;        // the records are there (device_destination has checked that they fitted)
    if (!ctx->log_state.overflow) {
        rc = gather_pool(ctx, (const unsigned long long *)log->offsets, log->capacity, 0, log->t, log->y, s);
        if (rc != IVP_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(s));
        return IVP_OK;
    }
    // the pool ran dry: the counts are exact, so this is the counted two-pass log whose first pass has just been done
    r.t_log = log->t; r.y_log = log->y; r.log_offsets = log->offsets;
    rc = ivp_batch_solve_device(ctx, prob, B, y0, params, t0, t0_len, t1, t1_len, &o, &r, hip_stream);
    if (rc != IVP_OK) return rc;
    log->passes = 2;
    return IVP_OK;
}

int ivp_step_log_fetch_device(ivp_ctx_t *ctx, ivp_step_log_t *log, void *hip_stream)
{
    if (!ctx) return IVP_ERR_BAD_ARGUMENT;
    if (!log) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null log");
    ivp_ctx::LogState &LS = ctx->log_state;
    if (!LS.valid || ctx->pend.active)
        return fail(ctx, IVP_ERR_BAD_ARGUMENT, LS.overflow ? "the page pool ran dry during the last logged solve: solve again (the pool is sized from the counted total now)"
                                                          : "no complete step log in this context's pool");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    log->owned = 0; log->device = -1;
    // the offsets of the pool's log: recomputed into context scratch (the caller's offsets array need not be passed again)
    HIP_TRY(ctx, ctx->log_off.reserve(sizeof(unsigned long long) * (LS.B + 1)));
    uint64_t total = 0;
    int rc = scan_counts(ctx, (unsigned long long *)ctx->log_off.p, s, &total);
    if (rc != IVP_OK) return rc;
    fill_log_info(ctx, log);
    rc = device_destination(ctx, log, total, LS.n, ctx->device);
    if (rc != IVP_OK) return rc;
    rc = gather_pool(ctx, (const unsigned long long *)ctx->log_off.p, log->capacity, 0, log->t, log->y, s);
    if (rc != IVP_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (!log->passes) log->passes = 1;
    return IVP_OK;
}

void ivp_step_log_free(ivp_step_log_t *log)
{
    if (!log || !log->owned) return;
    if (log->device >= 0) {
        DeviceGuard restore;
        (void)hipSetDevice(log->device);
        if (log->t) (void)hipFree(log->t);
        if (log->y) (void)hipFree(log->y);
    } else {
        std::free(log->t);
        std::free(log->y);
    }
    log->t = nullptr; log->y = nullptr; log->capacity = 0; log->owned = 0;
}

int ivp_batch_solve_logged(ivp_ctx_t *ctx, const ivp_problem_t *prob, size_t B, const double *y0, const double *params,
                           const double *t0, size_t t0_len, const double *t1, size_t t1_len, const ivp_options_t *opt,
                           ivp_batch_result_t *out, ivp_step_log_t *log)
{
    if (!ctx) return IVP_ERR_BAD_ARGUMENT;
    ctx->err.clear();
    if (!opt || !out || !log || !log->offsets) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null options / out / log / log->offsets");
    int n = 0, np = 0;
    int rc = validate(ctx, prob, B, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    if (!y0 || !t0 || !t1) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "null y0/t0/t1");
    if (np > 0 && !params) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "params required (n_params=%d)", np);
    if ((t0_len != 1 && t0_len != B) || (t1_len != 1 && t1_len != B)) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "t0/t1 length must be 1 or B");
    if ((log->t == nullptr) != (log->y == nullptr)) return fail(ctx, IVP_ERR_BAD_ARGUMENT, "ivp_step_log_t: t and y must both be given or both be NULL");
    DeviceGuard restore_device;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ---- stage the inputs and a device mirror of every requested SoA output ----
    const size_t l0 = t0_len == 1 ? 1 : B, l1 = t1_len == 1 ? 1 : B;
    HIP_TRY(ctx, ctx->st_y0.reserve(sizeof(double) * n * B));
    HIP_TRY(ctx, ctx->st_params.reserve(sizeof(double) * std::max(np, 1) * B));
    HIP_TRY(ctx, ctx->st_t0.reserve(sizeof(double) * l0));
    HIP_TRY(ctx, ctx->st_t1.reserve(sizeof(double) * l1));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st_y0.p, y0, sizeof(double) * n * B, hipMemcpyHostToDevice, nullptr));
    if (np > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->st_params.p, params, sizeof(double) * np * B, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st_t0.p, t0, sizeof(double) * l0, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st_t1.p, t1, sizeof(double) * l1, hipMemcpyHostToDevice, nullptr));
    MemberDesc md[kMembers];
    member_table(result_shape(prob, opt, n), md);
    ivp_batch_result_t dev;
    std::memset(&dev, 0, sizeof dev);
    auto is_log = [&](const MemberDesc &d) { return d.off == offsetof(ivp_batch_result_t, t_log) || d.off == offsetof(ivp_batch_result_t, y_log); };
    for (int k = 0; k < kMembers; ++k) {
        const size_t bytes = md[k].elem * md[k].rows * B;
        if (!member(out, md[k]) || !bytes || is_log(md[k])) continue;
        HIP_TRY(ctx, ctx->st_out[k].reserve(bytes));
        member(&dev, md[k]) = ctx->st_out[k].p;
    }
    HIP_TRY(ctx, ctx->st_logoff.reserve(sizeof(unsigned long long) * (B + 1)));
    // ---- integrate once; the records stay in the pool until their number is known ----
    ivp_step_log_t dl;
    std::memset(&dl, 0, sizeof dl);
    dl.offsets = (uint64_t *)ctx->st_logoff.p;
    dl.reserve = log->reserve;
    dl.defer = 1;
    rc = ivp_batch_solve_logged_device(ctx, prob, B, (const double *)ctx->st_y0.p, np > 0 ? (const double *)ctx->st_params.p : nullptr,
                                       (const double *)ctx->st_t0.p, l0, (const double *)ctx->st_t1.p, l1, opt, &dev, &dl, nullptr);
    if (rc != IVP_OK) return rc;
    const uint64_t total = dl.total;
    log->total = total; log->page_slots = dl.page_slots; log->pool_bytes = dl.pool_bytes; log->pool_used_bytes = dl.pool_used_bytes;
    log->owned = 0; log->device = -1; log->passes = 1;
    HIP_TRY(ctx, hipMemcpyAsync(log->offsets, ctx->st_logoff.p, sizeof(unsigned long long) * (B + 1), hipMemcpyDeviceToHost, nullptr));
    for (int k = 0; k < kMembers; ++k) {
        void *host = member(out, md[k]);
        const void *d = member(&dev, md[k]);
        if (host && d) HIP_TRY(ctx, hipMemcpyAsync(host, d, md[k].elem * md[k].rows * B, hipMemcpyDeviceToHost, nullptr));
    }
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    if (log->t && log->capacity < total)
        return fail(ctx, IVP_ERR_LOG_CAPACITY, "the log has %llu records, t / y hold %llu", (unsigned long long)total, (unsigned long long)log->capacity);
    // ---- records: pool -> device staging -> host ----
    const size_t recs = (size_t)std::max<uint64_t>(total, 1);
    HIP_TRY(ctx, ctx->st_log_t.reserve(recs * sizeof(double)));
    HIP_TRY(ctx, ctx->st_log_y.reserve(recs * sizeof(double) * (size_t)n));
    dl.t = (double *)ctx->st_log_t.p;
    dl.y = (double *)ctx->st_log_y.p;
    dl.capacity = recs;
    dl.defer = 0;
    if (!ctx->log_state.overflow) {
        rc = ivp_step_log_fetch_device(ctx, &dl, nullptr);
    } else {   // the pool ran dry: integrate again, now with a pool sized from the counted total (or through the counted fill pass)
        rc = ivp_batch_solve_logged_device(ctx, prob, B, (const double *)ctx->st_y0.p, np > 0 ? (const double *)ctx->st_params.p : nullptr,
                                           (const double *)ctx->st_t0.p, l0, (const double *)ctx->st_t1.p, l1, opt, &dev, &dl, nullptr);
        log->passes = 1 + dl.passes;
    }
    if (rc != IVP_OK) return rc;
    if (!log->t) {
        log->t = (double *)std::malloc(recs * sizeof(double));
        log->y = (double *)std::malloc(recs * sizeof(double) * (size_t)n);
        if (!log->t || !log->y) { std::free(log->t); std::free(log->y); log->t = log->y = nullptr; return fail(ctx, IVP_ERR_BAD_ARGUMENT, "out of host memory for %zu log records", recs); }
        log->capacity = recs;
        log->owned = 1;
    }
    if (total) {
        HIP_TRY(ctx, hipMemcpyAsync(log->t, dl.t, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(ctx, hipMemcpyAsync(log->y, dl.y, (size_t)total * sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, nullptr));
    }
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    return IVP_OK;
}

}  // extern "C"

namespace {

// trajectory order of the non-empty shards, each shard's record count and the record it starts at in the batch-wide log
struct MultiPlan { std::vector<int> order; std::vector<uint64_t> totals, bases; uint64_t total = 0; };
MultiPlan multi_plan(const ivp_shard_t *shards, int n_shards)
{
    MultiPlan mp;
    for (int i = 0; i < n_shards; ++i) if (shards[i].count) mp.order.push_back(i);
    std::sort(mp.order.begin(), mp.order.end(), [&](int a, int b) { return shards[a].first < shards[b].first; });
    mp.totals.assign(n_shards, 0);
    mp.bases.assign(n_shards, 0);
    for (int i : mp.order) {
        mp.totals[i] = shards[i].ctx->log_state.total;
        mp.bases[i] = mp.total;
        mp.total += mp.totals[i];
    }
    return mp;
}

// second half of a multi-context logged solve: every shard's records become one contiguous run of the batch-wide log on
// gather_device, starting at record bases[i] -- written there directly by the gather kernel for shards that live on that
// device, assembled in a staging buffer and sent as two peer copies (xGMI) otherwise
int multi_collect(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, const ivp_options_t *opt, int n, int32_t gather_device,
                  ivp_step_log_t *log)
{
    ivp_ctx_t *c0 = shards[0].ctx;
    const MultiPlan mp = multi_plan(shards, n_shards);
    log->total = mp.total;
    if (hipSetDevice(gather_device) != hipSuccess) return fail(c0, IVP_ERR_HIP, "hipSetDevice(%d)", gather_device);
    int rc = device_destination(c0, log, mp.total, n, gather_device);
    if (rc != IVP_OK) return rc;
    ivp_options_t o = *opt;
    o.count_log = 0;
    for (int i : mp.order) {
        ivp_shard_t &sh = shards[i];
        ivp_ctx *ctx = sh.ctx;
        auto bail = [&](int code) { if (ctx != c0) c0->err = ctx->err; return code; };
        if (hipSetDevice(ctx->device) != hipSuccess) return bail(fail(ctx, IVP_ERR_HIP, "hipSetDevice(%d)", ctx->device));
        hipStream_t s = (hipStream_t)sh.hip_stream;
        const bool local = ctx->device == gather_device;
        const uint64_t tot = mp.totals[i], base = mp.bases[i];
        const size_t recs = (size_t)std::max<uint64_t>(tot, 1);
        double *dt = log->t + base, *dy = log->y + base * (size_t)n;
        if (!local) {
            if (ctx->st_log_t.reserve(recs * sizeof(double)) != hipSuccess || ctx->st_log_y.reserve(recs * sizeof(double) * (size_t)n) != hipSuccess)
                return bail(fail(ctx, IVP_ERR_HIP, "staging for %zu log records", recs));
            dt = (double *)ctx->st_log_t.p;
            dy = (double *)ctx->st_log_y.p;
        }
        if (ctx->log_state.valid) {
            rc = gather_pool(ctx, (const unsigned long long *)ctx->log_off.p, tot, 0, dt, dy, s);
            if (rc != IVP_OK) return bail(rc);
        } else if (ctx->log_state.overflow) {   // this shard's pool ran dry: its counted fill pass, straight into the run
            ivp_batch_result_t r = sh.out;
            r.t_log = dt; r.y_log = dy; r.log_offsets = (const uint64_t *)ctx->log_off.p;
            rc = ivp_batch_solve_device(ctx, prob, sh.count, sh.y0, sh.params, sh.t0, sh.t0_len, sh.t1, sh.t1_len, &o, &r, sh.hip_stream);
            if (rc != IVP_OK) return bail(rc);
            log->passes = 2;
        } else {
            return bail(fail(ctx, IVP_ERR_BAD_ARGUMENT, "shard %d: no step log in its context's pool", i));
        }
        if (!local && tot) {
            hipError_t e = copy_rows_peer(log->t + base, gather_device, tot, dt, ctx->device, tot, 8, tot, 1, s);
            if (e == hipSuccess) e = copy_rows_peer(log->y + base * (size_t)n, gather_device, tot * (size_t)n, dy, ctx->device, tot * (size_t)n, 8, tot * (size_t)n, 1, s);
            if (e != hipSuccess) return bail(fail(ctx, IVP_ERR_HIP, "peer copy of the step log: %s", hipGetErrorString(e)));
        }
    }
    for (int i : mp.order) {
        if (hipSetDevice(shards[i].ctx->device) != hipSuccess || hipStreamSynchronize((hipStream_t)shards[i].hip_stream) != hipSuccess)
            return fail(c0, IVP_ERR_HIP, "waiting for shard %d's records", i);
    }
    return IVP_OK;
}

}  // namespace

extern "C" {

int ivp_batch_solve_logged_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B, const ivp_options_t *opt,
                                 int32_t gather_device, ivp_batch_result_t *gathered, ivp_step_log_t *log)
{
    if (!shards || n_shards <= 0 || n_shards > 64 || !shards[0].ctx) return IVP_ERR_BAD_ARGUMENT;
    ivp_ctx_t *c0 = shards[0].ctx;
    if (!opt || !log || !log->offsets) return fail(c0, IVP_ERR_BAD_ARGUMENT, "null options / log / log->offsets");
    if (opt->t_eval) return fail(c0, IVP_ERR_BAD_ARGUMENT, "the accepted-step log is what solve_ivp records when t_eval is None");
    if ((log->t == nullptr) != (log->y == nullptr)) return fail(c0, IVP_ERR_BAD_ARGUMENT, "ivp_step_log_t: t and y must both be given or both be NULL");
    DeviceGuard restore_device;
    log->owned = 0; log->device = -1; log->passes = 1; log->total = 0; log->pool_bytes = 0; log->pool_used_bytes = 0; log->page_slots = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || gather_device < 0 || gather_device >= ndev) return fail(c0, IVP_ERR_BAD_ARGUMENT, "gather_device %d", gather_device);
    ivp_options_t o = *opt;
    o.count_log = 0;
    // ---- one integration per shard, each into its own context's pool; the SoA members are gathered as usual ----
    std::vector<ivp_batch_result_t> saved(n_shards);
    for (int i = 0; i < n_shards; ++i) {
        if (!shards[i].ctx) return IVP_ERR_BAD_ARGUMENT;
        saved[i] = shards[i].out;
        shards[i].out.t_log = nullptr; shards[i].out.y_log = nullptr; shards[i].out.log_offsets = nullptr;
        shards[i].ctx->log_plan.want = shards[i].count != 0;
        shards[i].ctx->log_plan.reserve = log->reserve ? log->reserve * shards[i].count / std::max<size_t>(B, 1) + 1 : 0;
        shards[i].ctx->log_state.valid = false; shards[i].ctx->log_state.overflow = false;
    }
    ivp_batch_result_t g;
    if (gathered) { g = *gathered; g.t_log = nullptr; g.y_log = nullptr; g.log_offsets = nullptr; }
    int rc = ivp_batch_solve_multi(shards, n_shards, prob, B, &o, gather_device, gathered ? &g : nullptr);
    for (int i = 0; i < n_shards; ++i) { shards[i].ctx->log_plan = ivp_ctx::LogPlan{}; shards[i].out = saved[i]; }
    if (rc != IVP_OK) return rc;
    int n = 0, np = 0;
    rc = validate(c0, prob, B ? B : 1, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    // ---- per-shard offsets (they start at 0 in the shard's own scratch) and totals; batch-wide offsets on gather_device ----
    std::vector<int> order;
    for (int i = 0; i < n_shards; ++i) if (shards[i].count) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return shards[a].first < shards[b].first; });
    std::vector<unsigned long long> off;
    uint64_t total = 0;
    for (int i : order) {
        ivp_shard_t &sh = shards[i];
        ivp_ctx *ctx = sh.ctx;
        auto bail = [&](int code) { if (ctx != c0) c0->err = ctx->err; return code; };
        if (hipSetDevice(ctx->device) != hipSuccess) return bail(fail(ctx, IVP_ERR_HIP, "hipSetDevice(%d)", ctx->device));
        hipStream_t s = (hipStream_t)sh.hip_stream;
        if (ctx->log_off.reserve(sizeof(unsigned long long) * (sh.count + 1)) != hipSuccess) return bail(fail(ctx, IVP_ERR_HIP, "offsets scratch"));
        uint64_t tot = 0;
        rc = scan_counts(ctx, (unsigned long long *)ctx->log_off.p, s, &tot);
        if (rc != IVP_OK) return bail(rc);
        log->pool_bytes += ctx->log_state.pool_doubles * 8;
        log->pool_used_bytes += std::min<uint64_t>(ctx->log_state.pool_used, ctx->log_state.pool_doubles) * 8;
        log->page_slots = IVP_LOG_SLOTS;
        off.resize(sh.count + 1);
        if (hipMemcpyAsync(off.data(), ctx->log_off.p, (sh.count + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return bail(fail(ctx, IVP_ERR_HIP, "reading the shard's offsets"));
        for (size_t k = 0; k <= sh.count; ++k) off[k] += total;
        const bool last = i == order.back();
        if (hipMemcpyAsync((unsigned long long *)log->offsets + sh.first, off.data(), (sh.count + (last ? 1 : 0)) * sizeof(unsigned long long), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)   // `off` is reused by the next shard
            return bail(fail(ctx, IVP_ERR_HIP, "writing the batch-wide offsets"));
        total += tot;
    }
    log->total = total;
    if (order.empty()) {
        if (hipSetDevice(gather_device) != hipSuccess || hipMemset(log->offsets, 0, sizeof(unsigned long long) * (B + 1)) != hipSuccess) return fail(c0, IVP_ERR_HIP, "zeroing the offsets");
    }
    if (log->defer) return IVP_OK;   // records: ivp_step_log_fetch_multi, into buffers the caller sizes from `total`
    return multi_collect(shards, n_shards, prob, opt, n, gather_device, log);
}

int ivp_step_log_fetch_multi(ivp_shard_t *shards, int32_t n_shards, const ivp_problem_t *prob, size_t B, const ivp_options_t *opt,
                             int32_t gather_device, ivp_step_log_t *log)
{
    if (!shards || n_shards <= 0 || n_shards > 64 || !shards[0].ctx) return IVP_ERR_BAD_ARGUMENT;
    ivp_ctx_t *c0 = shards[0].ctx;
    if (!opt || !log) return fail(c0, IVP_ERR_BAD_ARGUMENT, "null options / log");
    for (int i = 0; i < n_shards; ++i) if (!shards[i].ctx) return IVP_ERR_BAD_ARGUMENT;
    DeviceGuard restore_device;
    int n = 0, np = 0;
    const int rc = validate(c0, prob, B ? B : 1, opt, &n, &np);
    if (rc != IVP_OK) return rc;
    log->owned = 0; log->device = -1;
    if (!log->passes) log->passes = 1;
    return multi_collect(shards, n_shards, prob, opt, n, gather_device, log);
}

}  // extern "C"
