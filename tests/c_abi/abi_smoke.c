/* A plain-C client of include/ivp_hip.h: proves the header is valid C and that the boundary needs nothing but
 * pointers and sizes.  Without a GPU it exercises the entry points that need none and exits 0 (printing "no device");
 * with one it integrates y' = -k y for four trajectories through the host-pointer entry point. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "ivp_hip.h"

int main(void)
{
    if (ivp_abi_version() != IVP_HIP_ABI_VERSION) { printf("abi mismatch\n"); return 2; }
    ivp_options_t opt;
    ivp_options_default(&opt);
    if (opt.method != IVP_DOPRI5 || opt.rtol != 1e-3 || opt.atol != 1e-6 || opt.max_steps != 0) return 3;
    if (ivp_options_method_defaults(&opt, IVP_DOP853) != IVP_OK || opt.scale_max != 6.0) return 4;
    ivp_options_default(&opt);
    int32_t n = 0, np = 0;
    if (ivp_rhs_dims(IVP_RHS_CR3BP, &n, &np) != IVP_OK || n != 6 || np != 1) return 5;
    if (ivp_rhs_dims(IVP_RHS_HEAT1D_256, &n, &np) != IVP_OK || n != 256 || np != 1) return 6;
    if (ivp_rhs_n_events(IVP_RHS_RATIONAL_EV) != 3) return 7;
    ivp_ctx_t *ctx = NULL;
    if (ivp_device_count() == 0) {
        if (ivp_ctx_create(&ctx, 0) != IVP_ERR_NO_DEVICE || ctx != NULL) return 8;   /* no CPU fallback */
        printf("no device: abi v%d ok\n", ivp_abi_version());
        return 0;
    }
    if (ivp_ctx_create(&ctx, 0) != IVP_OK) return 9;
    enum { B = 4 };
    const double y0[B] = {1.0, 2.0, 3.0, 4.0}, k[B] = {0.5, 1.0, 1.5, 2.0}, t0 = 0.0, t1[B] = {1.0, 2.0, 3.0, 0.0};
    double y_end[B], t_end[B];
    int32_t status[B];
    uint64_t naccpt[B];
    ivp_problem_t prob = {IVP_RHS_DECAY, 1, 1, NULL};
    ivp_batch_result_t out;
    memset(&out, 0, sizeof out);
    out.y_end = y_end; out.t_end = t_end; out.status = status; out.naccpt = naccpt;
    opt.rtol = 1e-9; opt.atol = 1e-12;
    int rc = ivp_batch_solve(ctx, &prob, B, y0, k, &t0, 1, t1, B, &opt, &out);
    if (rc != IVP_OK) { printf("solve failed: %d %s\n", rc, ivp_last_error_string(ctx)); return 10; }
    for (int b = 0; b < B; ++b) {
        const double want = y0[b] * exp(-k[b] * t1[b]);
        if (status[b] != IVP_STATUS_SUCCESS || fabs(y_end[b] - want) > 1e-8 || t_end[b] != t1[b]) {
            printf("trajectory %d: status %d y %.17g want %.17g\n", b, status[b], y_end[b], want);
            return 11;
        }
    }
    /* the same batch split over two contexts (here on one device; one per GPU in production): identical bits */
    {
        ivp_ctx_t *ctx2 = NULL;
        if (ivp_ctx_create(&ctx2, ivp_device_count() > 1 ? 1 : 0) != IVP_OK) return 13;
        ivp_ctx_t *both[2];
        both[0] = ctx; both[1] = ctx2;
        double y2[B], te2[B];
        int32_t st2[B];
        uint64_t na2[B];
        ivp_batch_result_t out2;
        memset(&out2, 0, sizeof out2);
        out2.y_end = y2; out2.t_end = te2; out2.status = st2; out2.naccpt = na2;
        rc = ivp_batch_solve_multi_host(both, 2, &prob, B, y0, k, &t0, 1, t1, B, &opt, &out2);
        if (rc != IVP_OK) { printf("multi solve failed: %d %s\n", rc, ivp_last_error_string(ctx)); return 14; }
        if (memcmp(y2, y_end, sizeof y2) || memcmp(te2, t_end, sizeof te2) || memcmp(st2, status, sizeof st2) || memcmp(na2, naccpt, sizeof na2)) {
            printf("multi-context result differs from the single-context one\n");
            return 15;
        }
        both[1] = ctx;
        if (ivp_batch_solve_multi_host(both, 2, &prob, B, y0, k, &t0, 1, t1, B, &opt, &out2) != IVP_ERR_BAD_ARGUMENT) return 16;
        ivp_ctx_destroy(ctx2);
        printf("multi-context solve ok\n");
    }
    /* Solution.t / Solution.y of the four calls in ONE call: every accepted step, CSR, t / y allocated by the library
     * like the Vecs the reference returns (src/solve/solve_ivp.rs:288-312), released with ivp_step_log_free */
    {
        uint64_t offsets[B + 1];
        uint32_t n_log[B];
        ivp_step_log_t log;
        memset(&log, 0, sizeof log);
        log.offsets = offsets;
        out.n_log = n_log;
        rc = ivp_batch_solve_logged(ctx, &prob, B, y0, k, &t0, 1, t1, B, &opt, &out, &log);
        if (rc != IVP_OK) { printf("logged solve failed: %d %s\n", rc, ivp_last_error_string(ctx)); return 17; }
        if (!log.owned || !log.t || !log.y || log.total != offsets[B] || offsets[0] != 0) return 18;
        for (int b = 0; b < B; ++b) {
            const uint64_t lo = offsets[b], hi = offsets[b + 1];
            /* the log starts at (t0, y0) and ends at the end state; trajectory 3 has a zero-length interval: one record */
            if (hi - lo != n_log[b] || hi <= lo || log.t[lo] != t0 || log.y[lo] != y0[b] || log.t[hi - 1] != t_end[b] || log.y[hi - 1] != y_end[b]) {
                printf("trajectory %d: bad step log [%llu, %llu)\n", b, (unsigned long long)lo, (unsigned long long)hi);
                return 19;
            }
            if (b < 3 && hi - lo != naccpt[b] + 1) return 20;
        }
        if (offsets[4] - offsets[3] != 1) return 21;
        ivp_step_log_free(&log);
        if (log.owned || log.t || log.y) return 22;
        out.n_log = NULL;
        printf("logged solve ok: %llu records\n", (unsigned long long)log.total);
    }
    opt.method = IVP_RADAU;
    if (ivp_batch_solve(ctx, &prob, B, y0, k, &t0, 1, t1, B, &opt, &out) != IVP_ERR_UNSUPPORTED_METHOD) return 12;
    ivp_ctx_destroy(ctx);
    printf("device solve ok: %llu %llu %llu %llu accepted steps\n", (unsigned long long)naccpt[0], (unsigned long long)naccpt[1],
           (unsigned long long)naccpt[2], (unsigned long long)naccpt[3]);
    return 0;
}
