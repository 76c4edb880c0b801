// emul.cpp -- TEST-ONLY harness: runs the per-lane kernel bodies of ivp_amd/csrc/rk_core.h on the
// CPU, lane by lane, with the same init -> chunk -> chunk ... schedule the GPU launch loop uses.
//
// It exists because the authoring container has no GPU: it lets the `-m "not gpu"` tests check the
// kernel logic (chunk boundaries, state save/restore, flag packing, DefaultSolOut on the device)
// bit for bit against the CPU oracle before any GPU minute is spent.  It is NOT part of the
// product: nothing in ivp_amd/ or libivp_hip.so links or loads it, and the product has no CPU path.
// Compiled with g++ -ffp-contract=off, once as is (strict) and once with -DIVP_FAST=1 (the FMA arithmetic mode:
// explicit fma() at the IVP_MA sites).
#include <cstdint>
#include <cstring>
#include <vector>

#define IVP_HD inline
#define IVP_NS ivp_emul
#include "../../ivp_amd/csrc/rk_core.h"
#include "../../ivp_amd/csrc/bdf_core.h"

using namespace ivp_emul;

template <int M, class R, int FULL>
static void run_all(IvpKArgs a, uint64_t *chunks_out)
{
    uint64_t chunks = 0;
    for (uint32_t j = 0; j < a.B; ++j) {
        int32_t st = any_init_body<M, R, FULL>(a, j);
        constexpr bool kHasCtl = M == M_RK23 || M == M_DOPRI5 || M == M_DOP853;
        while (st == IVP_RUNNING) {
            // same dispatch as launch_one() in rk_kernels.hip: run-time controller fields only for a direct method call
            if (kHasCtl && a.has_ctl) any_chunk_body<M, R, FULL, kHasCtl>(a, j, st);
            else any_chunk_body<M, R, FULL>(a, j, st);
            ++chunks;
        }
    }
    if (chunks_out) *chunks_out = chunks;
}

// full: kernel flavour as in rk_launch.h (0 end state, 1 whole DefaultSolOut, 2 log-only: adaptive explicit methods of
// problems without event functions, everything else runs it as 1 -- the dispatch of rk_kernels.hip)
template <int M, class R>
static void run_flavour(int full, const IvpKArgs &a, uint64_t *chunks)
{
    if constexpr (R::NE == 0 && (M == M_RK23 || M == M_DOPRI5 || M == M_DOP853)) {
        if (full == 2) { run_all<M, R, 2>(a, chunks); return; }
    }
    if constexpr (R::NE == 0 && M == M_DOP853) {
        if (full == 3) {   // deferred t_eval sampling: the stepping bodies note the sampled steps, then one "lane" per noted step
            run_all<M, R, 3>(a, chunks);
            for (uint32_t j = 0; j < a.B; ++j)
                for (uint32_t k = 0; k < a.n_seg[j] && k < a.def_cap; ++k) dop853_sample_body<R>(a, j, k);
            return;
        }
    }
    full ? run_all<M, R, 1>(a, chunks) : run_all<M, R, 0>(a, chunks);
    if constexpr (R::NE > 0 && M != M_BDF) {   // deferred event refinement (evd_rec set): one "lane" per noted step, as event_kernel_t does
        if (full && a.evd_rec != nullptr)
            for (uint32_t j = 0; j < a.B; ++j)
                for (uint32_t q = 0; q < a.evd_cnt[j] && q < a.evd_cap; ++q) so_events_deferred_body<M, R>(a, j, q);
    }
}

template <class R>
static int run_rhs(int method, int full, const IvpKArgs &a, uint64_t *chunks)
{
    switch (method) {
    case 0: run_flavour<0, R>(full, a, chunks); return 0;
    case 1: run_flavour<1, R>(full, a, chunks); return 0;
    case 2: run_flavour<2, R>(full, a, chunks); return 0;
    case 3: run_flavour<3, R>(full, a, chunks); return 0;
    case 5: run_flavour<5, R>(full, a, chunks); return 0;
    }
    return -1;
}

extern "C" int emul_is_fast(void) { return IVP_FAST; }

extern "C" int emul_solve(int method, int rhs_id, int full, IvpKArgs *args, uint64_t *chunks)
{
    IvpKArgs a = *args;
    const size_t B = a.B;
    // scratch the library would own
    std::vector<double> k1(8 * B), facold(B), hlamb(B), t_last(B);
    std::vector<uint32_t> flags(B);
    std::vector<int32_t> next_idx(B);
    uint32_t err_flag = 0;
    a.err_flag = &err_flag;
    std::vector<double> bdf_d(8 * 8 * B), bdf_jac(64 * B), bdf_lu(64 * B);
    std::vector<uint32_t> bdf_piv(B);
    std::vector<double> prev_event(4 * B);
    a.prev_event = prev_event.data();
    a.bdf_d = bdf_d.data(); a.bdf_jac = bdf_jac.data(); a.bdf_lu = bdf_lu.data(); a.bdf_piv = bdf_piv.data();
    a.k1 = k1.data(); a.facold = facold.data(); a.hlamb = hlamb.data(); a.flags = flags.data();
    a.t_last = t_last.data(); a.next_idx = next_idx.data();
    int rc = -1;
    switch (rhs_id) {
    case 0: rc = run_rhs<RhsDecay>(method, full, a, chunks); break;
    case 1: rc = run_rhs<RhsSho>(method, full, a, chunks); break;
    case 2: rc = run_rhs<RhsVdp>(method, full, a, chunks); break;
    case 3: rc = run_rhs<RhsCr3bp>(method, full, a, chunks); break;
    case 4: rc = run_rhs<RhsLorenz>(method, full, a, chunks); break;
    case 5: rc = run_rhs<RhsZero>(method, full, a, chunks); break;
    case 6: rc = run_rhs<RhsRational>(method, full, a, chunks); break;
    case 7: rc = run_rhs<RhsExp2>(method, full, a, chunks); break;
    case 8: rc = run_rhs<RhsLinear>(method, full, a, chunks); break;
    case 9: rc = run_rhs<RhsRobertson>(method, full, a, chunks); break;
    case 10: rc = run_rhs<RhsVdpEps>(method, full, a, chunks); break;
    case 11: rc = run_rhs<RhsShoEv>(method, full, a, chunks); break;
    case 12: rc = run_rhs<RhsBall>(method, full, a, chunks); break;
    case 13: rc = run_rhs<RhsCannon>(method, full, a, chunks); break;
    case 14: rc = run_rhs<RhsRationalEv>(method, full, a, chunks); break;
    case 15: rc = run_rhs<RhsRobertsonJac>(method, full, a, chunks); break;
    }
    if (rc == 0 && (err_flag & 0x1u)) return -5;  // IVP_ERR_INVALID_STEP_SIZE
    return rc;
}

// bdf_change_d (structured) next to bdf_change_d_generic (the literal restatement) on the same input: d is [8][n] row-major
extern "C" int emul_change_d(int n, int order, double factor, const double *d_in, double *d_fast, double *d_generic)
{
    using namespace IVP_NS;
    auto run = [&](auto tag) {
        constexpr int N = decltype(tag)::value;
        double a[8][N], b[8][N];
        for (int k = 0; k < 8; ++k)
            for (int c = 0; c < N; ++c) a[k][c] = b[k][c] = d_in[k * N + c];
        bdf_change_d<N>(a, order, factor);
        if (factor != 1.0) bdf_change_d_generic<N>(b, order > BDF_MAXO ? BDF_MAXO : order, factor);
        for (int k = 0; k < 8; ++k)
            for (int c = 0; c < N; ++c) { d_fast[k * N + c] = a[k][c]; d_generic[k * N + c] = b[k][c]; }
    };
    switch (n) {
    case 1: run(std::integral_constant<int, 1>{}); return 0;
    case 2: run(std::integral_constant<int, 2>{}); return 0;
    case 3: run(std::integral_constant<int, 3>{}); return 0;
    case 6: run(std::integral_constant<int, 6>{}); return 0;
    default: return -1;
    }
}

// ivp_div_small_const<C>(x) next to x / C
extern "C" long emul_div_small_const(int c, const double *x, long n)
{
    using namespace IVP_NS;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const double a = c == 3 ? x[i] / 3.0 : x[i] / 5.0;
        const double b = c == 3 ? ivp_div_small_const<3>(x[i]) : ivp_div_small_const<5>(x[i]);
        if (std::memcmp(&a, &b, 8) != 0) ++bad;
    }
    return bad;
}

// ivp_pow3 next to three ivp_pow calls
extern "C" void emul_pow3(const double *x, const double *e, double *r3, double *r1)
{
    using namespace IVP_NS;
    const double xs[3] = {x[0], x[1], x[2]}, es[3] = {e[0], e[1], e[2]};
    double out[3];
    ivp_pow3(xs, es, out);
    for (int i = 0; i < 3; ++i) { r3[i] = out[i]; r1[i] = ivp_pow(x[i], e[i]); }
}

extern "C" size_t emul_kargs_size(void) { return sizeof(IvpKArgs); }
