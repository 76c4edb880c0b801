// bdf_group.h -- variable-order BDF(1..5) for LARGE state dimensions (8 < n <= 512): one group of G lanes (a whole
// 64-lane wavefront for the built-in problems) integrates one trajectory.  Device only.
//
// Restates src/methods/bdf.rs:86-732 like bdf_core.h does for n <= 8 -- same control flow, same arithmetic per
// component -- with the state distributed over the group's lanes (lane l owns components l, l+G, ...: the mapping of
// rk_group.h, whose GroupRhs / NormOps / OutMap it reuses) and the per-trajectory n x n matrices in memory:
//   * J and LU = (I - cJ) live in global memory, one contiguous n*n block per trajectory, COLUMN-major, so that
//     "lane <-> row" makes every column operation a coalesced access;
//   * lu_decomp (src/matrix/lu.rs:37-125): right-looking elimination with partial pivoting, column by column.  The
//     pivot search is a group reduction that returns the FIRST row attaining the maximum (the reference's strict `>`
//     scan); multipliers stay in the owning lanes' registers; the trailing update a[i][j] += a[i][k] * t_j runs over
//     the columns with rows in lanes.  Every element sees the reference's operations in the reference's order (each
//     a[i][j] is only ever touched by the lane that owns row i), so the factors are bit-identical;
//   * lin_solve (src/matrix/linear.rs:55-96): the right-hand side sits in LDS; forward and backward substitution are
//     2n sequential pivot steps, each a broadcast of b[k] and one column axpy over the owning lanes;
//   * the forward-difference Jacobian (trait IVP::jac default, src/ivp.rs:67-107) perturbs one component at a time:
//     n + 1 evaluations of the component-form right-hand side through LDS.
// The weighted RMS norms use NormOps<GroupRhs>::sum (strict build: index-order sum, the reference's bits).
// All controller scalars are computed redundantly by every lane, so control flow is group-uniform.
// Outputs: end state, t_eval sampling, the accepted-step log and dense-output segments.
//
// LDS-resident factors (LDSLU kernels; BASELINE C5's "batched dense LU of per-trajectory Jacobians in LDS").  With one
// wavefront per trajectory and n <= 128 the n x n matrix LU = (I - cJ) fits the CU's 160 KiB of LDS (80 KB at n = 100:
// two trajectories per CU; 128 KB at n = 128: one), so the LDSLU instantiation keeps it THERE for the whole launch:
// lu_decomp and lin_solve are the very same functions (every access below is `a[col * NT + row]` on a pointer whose
// address space the compiler infers), but a pivot step now waits for LDS round trips (~64 cycles) instead of L2 / HBM
// round trips (~1-2 us for a lone wavefront), and the trailing update no longer leaves the CU.  J stays in global memory
// (it is read once per refactorisation, coalesced).  LDS does not survive a launch: factors that are current when a
// launch ends are written back to the trajectory's global block and fetched again by the next launch, so the
// refactorisation count (`nlu`) and every bit of the results equal the global-memory path's.
// All barriers in this file are workgroup barriers of a ONE-WAVEFRONT workgroup (blockDim.x == IVP_WAVE, enforced by the
// launch tables): groups of a wavefront may diverge around them (NGROUP > 1) without deadlock.
#pragma once

namespace IVP_NS {

// -DIVP_PHASE_CLOCKS: shader-clock cycles per phase of the attempt, summed per launch and printed by lane 0 of workgroup 0
// (a measuring build for tools/phase_clocks_large_n.sh; never part of libivp_hip.so)
#ifdef IVP_PHASE_CLOCKS
__device__ unsigned long long ivp_phase_clk[16];
#define IVP_CLK_T0() unsigned long long ivp_clk_t = clock64()
#define IVP_CLK(ph) do { const unsigned long long ivp_clk_n = clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) ivp_phase_clk[ph] += ivp_clk_n - ivp_clk_t; ivp_clk_t = ivp_clk_n; } while (0)
#else
#define IVP_CLK_T0() do { } while (0)
#define IVP_CLK(ph) do { } while (0)
#endif

template <class R, int G>
struct BdfG {
    using GR = GroupRhs<R, G>;
    enum { NT = R::N, C = GR::N, P = R::P, NGROUP = IVP_WAVE / G };
    static __device__ __forceinline__ int gl() { return GR::gl(); }
    static __device__ __forceinline__ int gi(int c) { return GR::gl() + G * c; }
    static __device__ __forceinline__ bool own(int c) { return gi(c) < NT; }
    static __device__ __forceinline__ int wl0() { return ((int)threadIdx.x / G) * G; }   // wave lane of the group's lane 0

    // weighted_rms_scaled (bdf.rs:659-667) over all NT components
    static __device__ __forceinline__ double wrms(const double *v, const double *scale)
    {
        double term[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double denom = scale[c] == 0.0 ? 2.220446049250313e-16 : scale[c];
            const double ratio = v[c] / denom;
            term[c] = own(c) ? ratio * ratio : 0.0;
        }
        return sqrt(NormOps<GR>::sum(term) / (double)NT);
    }

    // default IVP::jac (src/ivp.rs:67-107): forward differences; jac is column-major [col * NT + row].
    // One column needs f at the state with ONE component perturbed.  The component-form right-hand side reads its state from
    // LDS, so KJ perturbed copies of the state are kept there and KJ columns are evaluated per sweep: three barriers per KJ
    // columns instead of two per column, KJ x C independent evaluations between them (a lone wavefront's sweep is a chain of
    // LDS round trips and barriers: 1.5 us per column at n = 100, where the n + 1 evaluations of this function were 2/3 of a
    // BDF step).  Every entry is still (f_row(y + pert e_col) - f_row(y)) / pert with the reference's operations.
    // `work` (LDSWORK): the kernels that keep the factors of (I - cJ) in LDS have no room for KJ more vectors (two 80 KB
    // matrices fill a CU at n = 100) -- but the factors are dead whenever BDF evaluates a Jacobian (it refactorises afterwards,
    // bdf.rs:448-459, 596-606), so their LDS is the work space there.
    enum { KJ = 8 };
    template <bool LDSWORK>
    static __device__ __forceinline__ void fd_jac(double x, const double (&y)[C], const double *p, double *jac, double *work)
    {
        static_assert(!LDSWORK || NT >= KJ, "the factor matrix (n x n) must hold KJ state copies");
        double *st = work;
        if constexpr (!LDSWORK) {
            __shared__ double ivp_fd_states[NGROUP * KJ * NT];
            st = ivp_fd_states + (size_t)((int)threadIdx.x / G) * KJ * NT;
        }
        double fo[C];
        GR::ode(x, y, fo, p);
        const double eps = 1.4901161193847656e-08;   // f64::EPSILON.sqrt() = 2^-26
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KJ; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) if (own(c)) st[k * NT + gi(c)] = y[c];
#pragma unroll 1
        for (int col0 = 0; col0 < NT; col0 += KJ) {
            double pert[KJ];
            // the owner of component col0 + k perturbs it in copy k; everybody needs the perturbation for the quotient
#pragma unroll
            for (int k = 0; k < KJ; ++k) {
                const int col = col0 + k;
                const int ll = col % G, cc = col / G;
                double ysel = y[0];
#pragma unroll
                for (int q = 1; q < C; ++q) ysel = cc == q ? y[q] : ysel;
                const double yo = __shfl(ysel, wl0() + ll);
                pert[k] = eps * fmax(fabs(yo), 1.0);
                if (col < NT && gl() == ll) st[k * NT + col] = yo + pert[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KJ; ++k) {
                const int col = col0 + k;
                if (col < NT) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int i = gi(c);
                        if (i < NT) jac[(size_t)col * NT + i] = (R::ode_comp(i, x, st + k * NT, p) - fo[c]) / pert[k];
                    }
                }
            }
            __syncthreads();
            // the copies return to y for the next KJ columns
#pragma unroll
            for (int k = 0; k < KJ; ++k) {
                const int col = col0 + k;
                const int ll = col % G, cc = col / G;
                double ysel = y[0];
#pragma unroll
                for (int q = 1; q < C; ++q) ysel = cc == q ? y[q] : ysel;
                if (col < NT && gl() == ll) st[k * NT + col] = ysel;
            }
        }
        __syncthreads();
    }

    // f.jac(x, y, &mut j): the problem's own Jacobian when the functor has one -- the `impl IVP { fn jac }` override of
    // src/ivp.rs:67-107 in COLUMN form, static void jac_col(int col, double x, const double* y, double* column /* [n] */,
    // const double* p), which writes column `col` of dF/dy (entries it does not write keep their previous value: the
    // reference's Matrix persists between calls and starts zeroed, bdf.rs:152) -- else the forward differences above.
    // Lane l fills columns l, l + G, ...: every column of the column-major J is one contiguous run of memory.
    template <class RR, class = void>
    struct HasJacCol { enum { v = 0 }; };
    template <class RR>
    struct HasJacCol<RR, decltype((void)&RR::jac_col)> { enum { v = 1 }; };
    template <bool LDSWORK = false>
    static __device__ __forceinline__ void eval_jac(double x, const double (&y)[C], const double *p, double *jac, double *work = nullptr)
    {
        if constexpr (HasJacCol<R>::v) {
            double *st = GR::scratch();
            __syncthreads();
#pragma unroll
            for (int c = 0; c < C; ++c) if (own(c)) st[gi(c)] = y[c];
            __syncthreads();
            for (int col = gl(); col < NT; col += G) R::jac_col(col, x, st, jac + (size_t)col * NT, p);
            __syncthreads();
        } else {
            fd_jac<LDSWORK>(x, y, p, jac, work);
        }
    }

    // max over the group's lanes of v (v >= -1.0, no NaNs), in every lane.  One group per wavefront: the DPP row-shift /
    // row-broadcast reduction (6 dependent v_max with DPP operand moves), then a v_readlane of lane 63 -- no LDS round trips
    // (the shuffle butterfly's 6 dependent ds_bpermute stages were 45 % of a pivot step of lu_decomp on a lone wavefront).
    static __device__ __forceinline__ double group_max(double v)
    {
        if constexpr (G == IVP_WAVE) {
            constexpr int kIdLo = 0, kIdHi = (int)0xBFF00000;   // -1.0: below every candidate
            auto stage = [&](auto ctrl, auto row_mask) {
                const unsigned long long u = d2u(v);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(kIdLo, (int)(uint32_t)u, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(kIdHi, (int)(uint32_t)(u >> 32), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
                const double o = u2d(((unsigned long long)hi << 32) | lo);
                v = o > v ? o : v;
            };
            stage(IntC<0x111>{}, IntC<0xf>{});   // row_shr:1
            stage(IntC<0x112>{}, IntC<0xf>{});   // row_shr:2
            stage(IntC<0x114>{}, IntC<0xf>{});   // row_shr:4
            stage(IntC<0x118>{}, IntC<0xf>{});   // row_shr:8   (lane 15 of every row: the row's maximum)
            stage(IntC<0x142>{}, IntC<0xa>{});   // row_bcast:15 into rows 1 and 3
            stage(IntC<0x143>{}, IntC<0xc>{});   // row_bcast:31 into rows 2 and 3   (lane 63: the maximum)
            return lane_bcast(v, IVP_WAVE - 1);
        } else {
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) { const double ov = __shfl_xor(v, o); v = ov > v ? ov : v; }
            return v;
        }
    }
    template <int V> struct IntC { static constexpr int value = V; };

    // lu_decomp (src/matrix/lu.rs:37-125) in place on the column-major matrix a; pivots to piv[0..NT-2].
    // `density` (may be NULL): {trailing columns that needed an update, trailing columns looked at}, summed over the pivots of
    // this factorisation -- the host's choice between LDS-resident and global-memory factors follows it (ivp_capi.cpp)
    static __device__ __forceinline__ bool lu_decomp(double *a, uint32_t *piv, uint32_t *density = nullptr)
    {
        uint32_t n_flagged = 0, n_looked = 0;
        if (NT == 1) return a[0] != 0.0;
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < NT - 1; ++k) {
            // One memory round trip brings everything the pivot step needs from column k: this lane's rows (whole column,
            // no masks: rows >= n read a clamped address and never become candidates).  The diagonal entry and the pivot
            // come out of the registers by v_readlane, the pivot row by a maximum over the wavefront plus a ballot; the
            // multipliers are formed from the values already in registers and every lane stores its own rows of column k
            // (row k receives the pivot: that IS the row exchange within this column), so the step needs no barrier
            // before the trailing columns are touched.
            IVP_CLK_T0();
            double colk[C];
#pragma unroll
            for (int c = 0; c < C; ++c) colk[c] = a[(size_t)k * NT + (own(c) ? gi(c) : NT - 1)];
            const double akk = row_bcast(colk, k, 0);
            IVP_CLK(8);
            // first row >= k attaining max |a[row][k]| (NaNs never win)
            double av[C], lv = -1.0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = gi(c);
                const double v = fabs(colk[c]);
                av[c] = (i >= k && i < NT && v == v) ? v : -1.0;
                lv = av[c] > lv ? av[c] : lv;
            }
            const double vmax = group_max(lv);
            int li = NT;
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                unsigned long long hit = __ballot(av[c] == vmax);
                if (G < IVP_WAVE) hit = (hit >> wl0()) & ((1ull << (G & 63)) - 1ull);
                if (hit != 0ull) li = G * c + __ffsll((long long)hit) - 1;
            }
            int m = k;
            double pivot = akk;
            if (akk == akk && vmax >= 0.0) { m = li; pivot = row_bcast(colk, li, 0); }   // |a[k][k]| NaN: every `>` of the reference's scan is false
            if (gl() == 0) piv[k] = (uint32_t)m;
            if (pivot == 0.0) return false;
            IVP_CLK(9);
            const double t = 1.0 / pivot;
            double mult[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = gi(c);
                mult[c] = (i > k && i < NT) ? -((i == m) ? akk : colk[c]) * t : 0.0;    // row m holds row k's entry after the swap
                if (i >= k && i < NT) a[(size_t)k * NT + i] = i == k ? pivot : mult[c];
            }
            // Trailing update.  A column j > k changes only if its pivot-row entry t_j = a[m][j] is non-zero (the reference
            // guards the update with `t != 0`, lu.rs:88-104) or if the row exchange moves two different values; for a
            // banded or sparse Jacobian -- the method-of-lines systems this path exists for -- that is a handful of columns
            // per pivot instead of n - k.  One strided load per G columns finds them (lane <-> column), a ballot turns
            // them into a bit mask, and only flagged columns are read, updated and written, JB at a time (their loads
            // are all issued before the first dependent instruction: a lone wavefront working out of L2 pays a full
            // memory round trip for whatever it waits on).  Skipped columns are exactly those the reference leaves
            // bit-for-bit unchanged; a dense matrix flags every column and costs two extra loads per 64 columns.
            IVP_CLK(10);
            constexpr int JB = 4;
            struct Blk { double tj[JB], akj[JB], cur[JB][C]; };
            // Branch-free per element: a G-row chunk that lies entirely above the pivot row is skipped by a SCALAR test
            // (k is wave-uniform); inside the chunk that holds row k the rows above it are written back unchanged.  What
            // a row does -- receive row k's entry (the swap), take the update, become U[k][j] -- are selects on masks
            // that depend on the row only, so they are formed once per pivot, not once per element: per-element exec-mask
            // branches were 40 % of this kernel's instruction stream (SQ_INSTS_SALU + SQ_INSTS_BRANCH vs SQ_INSTS_VALU).
            bool is_m[C], below[C], is_k[C], in_range[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = gi(c);
                is_m[c] = i == m; below[c] = i > k; is_k[c] = i == k; in_range[c] = i < NT;
            }
            auto fetch = [&](const int (&jc)[JB], const bool (&act)[JB], Blk &q) {
#pragma unroll
                for (int b = 0; b < JB; ++b) {   // read before anything in these columns is written
                    q.tj[b] = 0.0; q.akj[b] = 0.0;
                    if (act[b]) { const double *col = a + (size_t)jc[b] * NT; q.tj[b] = col[m]; q.akj[b] = col[k]; }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (G * c + (G - 1) < k) continue;                        // whole chunk above the pivot row (scalar test)
#pragma unroll
                    for (int b = 0; b < JB; ++b) {
                        q.cur[b][c] = 0.0;
                        if (act[b] && ((c + 1) * G <= NT || in_range[c])) q.cur[b][c] = a[(size_t)jc[b] * NT + gi(c)];
                    }
                }
            };
            auto finish = [&](const int (&jc)[JB], const bool (&act)[JB], const Blk &q) {
                bool upd[JB];
#pragma unroll
                for (int b = 0; b < JB; ++b) upd[b] = q.tj[b] != 0.0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (G * c + (G - 1) < k) continue;
#pragma unroll
                    for (int b = 0; b < JB; ++b) {
                        const double v = is_m[c] ? q.akj[b] : q.cur[b][c];   // row m receives row k's entry (the swap)
                        const double w = IVP_MA(v, mult[c], q.tj[b]);
                        const double v2 = (upd[b] && below[c]) ? w : v;
                        const double out = is_k[c] ? q.tj[b] : v2;
                        if (act[b] && ((c + 1) * G <= NT || in_range[c])) a[(size_t)jc[b] * NT + gi(c)] = out;
                    }
                }
            };
            // the pivot-row entries of ALL trailing columns first (one round trip for up to 8 strided loads), then segment
            // by segment; processing a segment only writes that segment's columns
            constexpr int SEG = (NT + G - 1) / G;
            double tjs[SEG], aks[SEG];
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) {
                const int jj = k + 1 + sg * G + gl();
                tjs[sg] = 0.0; aks[sg] = 0.0;
                if (jj < NT) { tjs[sg] = a[(size_t)jj * NT + m]; aks[sg] = a[(size_t)jj * NT + k]; }
            }
            IVP_CLK(11);
#pragma unroll 1
            for (int sg = 0; sg < SEG; ++sg) {
                const int j0 = k + 1 + sg * G;
                if (j0 >= NT) break;
                const int jj = j0 + gl();
                double tjv = tjs[0], akv = aks[0];   // select chain instead of a run-time register index
#pragma unroll
                for (int q = 1; q < SEG; ++q) { tjv = sg == q ? tjs[q] : tjv; akv = sg == q ? aks[q] : akv; }
                const bool need = jj < NT && (tjv != 0.0 || (m != k && d2u(tjv) != d2u(akv)));
                unsigned long long todo = __ballot(need);
                if (G < IVP_WAVE) todo = (todo >> wl0()) & ((1ull << (G & 63)) - 1ull);   // this group's columns
                n_flagged += (uint32_t)__popcll(todo);
                n_looked += (uint32_t)((NT - j0) < G ? (NT - j0) : G);
                // DENSE segments (every column of the range flagged: a full Jacobian) go JB columns at a time with no
                // per-column bookkeeping -- the general loop below spends ~45 instructions and four taken branches per column
                // on finding its columns one bit at a time (dense 64-state system: 83 % of lu_decomp, tools/phase_clocks_large_n.sh)
                {
                    const int ncol = (NT - j0) < G ? (NT - j0) : G;
                    const unsigned long long all = ncol >= 64 ? ~0ull : ((1ull << ncol) - 1ull);
                    if (todo == all) {
                        int done = 0;
#pragma unroll 1
                        for (; done + JB <= ncol; done += JB) {
                            int jc[JB];
                            bool act[JB];
#pragma unroll
                            for (int b = 0; b < JB; ++b) { act[b] = true; jc[b] = j0 + done + b; }
                            Blk q;
                            fetch(jc, act, q);
                            finish(jc, act, q);
                        }
                        todo = done >= 64 ? 0ull : (all >> done) << done;   // the last ncol % JB columns: below
                    }
                }
#pragma unroll 1
                while (__any(todo != 0ull)) {
                    int jc[JB];
                    bool act[JB];
#pragma unroll
                    for (int b = 0; b < JB; ++b) {
                        act[b] = todo != 0ull;
                        jc[b] = j0 + (act[b] ? __ffsll((long long)todo) - 1 : 0);
                        todo &= todo - 1ull;
                    }
                    Blk q;
                    fetch(jc, act, q);
                    finish(jc, act, q);
                }
            }
            __syncthreads();   // column k+1 is complete before the next pivot search reads it
            IVP_CLK(12);
        }
        if (density != nullptr && gl() == 0) { atomicAdd(density, n_flagged); atomicAdd(density + 1, n_looked); }
        return a[(size_t)(NT - 1) * NT + (NT - 1)] != 0.0;
    }

    // Lane `l` (of this group) broadcasts a value to the whole group.  l is uniform within the group; with one group per
    // wavefront it is wave-uniform and the broadcast is a v_readlane (no LDS round trip, the result lands in SGPRs).
    static __device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int l)
    {
        if constexpr (G == IVP_WAVE) return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l));
        else return (uint32_t)__shfl((int)v, wl0() + l);
    }
    static __device__ __forceinline__ double lane_bcast(double v, int l)
    {
        if constexpr (G == IVP_WAVE) {
            const int lane = __builtin_amdgcn_readfirstlane(l);
            const unsigned long long u = d2u(v);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
            return u2d(((unsigned long long)hi << 32) | lo);
        } else {
            return __shfl(v, wl0() + l);
        }
    }
    // row r (uniform within the group, r >= G * c0; c0 a constant after unrolling) of a vector whose row i lives in lane
    // i % G, component i / G
    static __device__ __forceinline__ double row_bcast(const double (&v)[C], int r, int c0)
    {
        const int q = r / G;
        double sel = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (c == c0) sel = v[c];
            else if (c > c0) sel = q == c ? v[c] : sel;
        }
        return lane_bcast(sel, r % G);
    }

    // lin_solve (src/matrix/linear.rs:55-96): b (this lane's components) <- A^-1 b.
    // The right-hand side never leaves the registers: each of the 2n sequential pivot steps broadcasts ONE entry of b and
    // every lane updates the rows it owns -- the reference's operations on every entry, in its order.  A lone wavefront
    // retires one instruction every ~5 cycles, so what a pivot step costs is its instruction count (round 3 kept b in LDS:
    // two LDS round trips, two barriers and ~100 instructions per step = 525 cycles, 60 % of a BDF step at n = 100;
    // tools/phase_clocks_large_n.sh).  Hence:
    //   * the steps are walked in blocks of G (block KB: rows KB*G ...): components below the block are not touched at all,
    //     the block's own component is the only one that needs the row mask, which component holds b[k] is static;
    //   * whole columns of the factors are fetched, PF steps ahead, without masks: entries the step must not use are
    //     dropped by the row mask of the update (rows >= n read a clamped address; their results are never looked at);
    //   * the pivot indices sit in registers (one coalesced load) and are read with v_readlane.
    enum { PF = C <= 2 ? 8 : (C <= 4 ? 4 : 2) };
    static __device__ __forceinline__ void lin_solve(const double *a, const uint32_t *piv, double (&bl)[C])
    {
        if (NT == 1) { bl[0] = own(0) ? bl[0] / a[0] : 0.0; return; }
        int row[C];
        uint32_t pv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            row[c] = own(c) ? gi(c) : NT - 1;
            pv[c] = gi(c) < NT - 1 ? piv[gi(c)] : 0u;
        }
        // forward: b <- L^-1 P b (row exchange m <-> k, then b[i] += l[i][k] * b[k] for i > k; the multipliers are stored negated)
#pragma unroll
        for (int kb = 0; kb < C; ++kb) {
            const int kend = (kb + 1) * G < NT - 1 ? (kb + 1) * G : NT - 1;
            if (kb * G >= NT - 1) break;
            double cq[PF][C];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = kb * G + u < NT ? kb * G + u : NT - 1;
#pragma unroll
                for (int c = kb; c < C; ++c) cq[u][c] = a[(size_t)k * NT + row[c]];
            }
#pragma unroll 1
            for (int k0 = kb * G; k0 < kend; k0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int k = k0 + u;
                    if (k < kend) {
                        double ccur[C];
#pragma unroll
                        for (int c = kb; c < C; ++c) ccur[c] = cq[u][c];
                        const int kn = k + PF;
                        if (kn < kend) {
#pragma unroll
                            for (int c = kb; c < C; ++c) cq[u][c] = a[(size_t)kn * NT + row[c]];
                        }
                        const int m = (int)lane_bcast(pv[kb], k - kb * G);
                        const double bk_old = lane_bcast(bl[kb], k - kb * G);
                        const double t = row_bcast(bl, m, kb);
                        {
                            const int i = gi(kb);
                            double bi = bl[kb];
                            bi = i == m ? bk_old : bi;
                            bi = i == k ? t : bi;
                            bl[kb] = i > k ? IVP_MA(bi, ccur[kb], t) : bi;
                        }
#pragma unroll
                        for (int c = kb + 1; c < C; ++c) {
                            const double bi = gi(c) == m ? bk_old : bl[c];
                            bl[c] = IVP_MA(bi, ccur[c], t);
                        }
                    }
                }
            }
        }
        // backward: b <- U^-1 b (b[k] /= u[k][k], then b[i] -= u[i][k] * b[k] for i < k)
#pragma unroll
        for (int kb = C - 1; kb >= 0; --kb) {
            const int ktop = (kb + 1) * G - 1 < NT - 1 ? (kb + 1) * G - 1 : NT - 1;   // first (highest) row of the block
            const int kbot = kb == 0 ? 1 : kb * G;                                      // last row the loop handles (row 0: below)
            if (kb * G > NT - 1) continue;
            double cq[PF][C], dq[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = ktop - u >= 0 ? ktop - u : 0;
                dq[u] = a[(size_t)k * NT + k];
#pragma unroll
                for (int c = 0; c <= kb; ++c) cq[u][c] = a[(size_t)k * NT + row[c]];
            }
#pragma unroll 1
            for (int k0 = ktop; k0 >= kbot; k0 -= PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int k = k0 - u;
                    if (k >= kbot) {
                        double ccur[C];
                        const double dcur = dq[u];
#pragma unroll
                        for (int c = 0; c <= kb; ++c) ccur[c] = cq[u][c];
                        const int kn = k - PF;
                        if (kn >= kbot) {
                            dq[u] = a[(size_t)kn * NT + kn];
#pragma unroll
                            for (int c = 0; c <= kb; ++c) cq[u][c] = a[(size_t)kn * NT + row[c]];
                        }
                        const double bk = lane_bcast(bl[kb], k - kb * G) / dcur;
                        {
                            const int i = gi(kb);
                            const double bi = i == k ? bk : bl[kb];
                            bl[kb] = i < k ? IVP_MA(bi, ccur[kb], -bk) : bi;
                        }
#pragma unroll
                        for (int c = 0; c < kb; ++c) bl[c] = IVP_MA(bl[c], ccur[c], -bk);
                    }
                }
            }
        }
        const double b0 = lane_bcast(bl[0], 0) / a[0];
        if (gi(0) == 0) bl[0] = b0;
#pragma unroll
        for (int c = 0; c < C; ++c) bl[c] = own(c) ? bl[c] : 0.0;
    }
};

// smallest rtol over all components (bdf.rs:174-185)
template <class R, int G>
__device__ __forceinline__ double bdfg_rtol_min(const IvpKArgs &a)
{
    double r = a.rtol[0];
    if (a.rtol_dev) {
        r = u2d(0x7FF0000000000000ull);
        for (int i = 0; i < R::N; ++i) r = fmin(r, a.rtol_dev[i]);
    }
    return fmax(r, 2.220446049250313e-16);
}

template <class R, int FULL, int G>
__device__ __forceinline__ int32_t bdf_group_init_body(const IvpKArgs &a, uint32_t j)
{
    using BG = BdfG<R, G>;
    using GR = GroupRhs<R, G>;
    using MAP = typename OutMap<GR>::type;
    constexpr int C = BG::C, NT = BG::NT, P = R::P;
    const size_t B = a.B;
    Lane<C, P> L;
    double y[C], f0[C];
#pragma unroll
    for (int c = 0; c < C; ++c) y[c] = map_ld<MAP>(a.y0, c, B, j);
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    L.x0 = a.t0[(size_t)j * a.t0_stride];
    L.xend = a.t1[(size_t)j * a.t1_stride];
    L.flags = 0;
    L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0; L.n_log = 0;
    if (FULL && a.log_pool != nullptr) so_log_open<MAP>(a, j, L, 2u, 0u, 1u);   // the initial callback records at most twice
    auto store_so = [&]() {
        if (FULL) {
            a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_log[j] = L.n_log;
            a.n_seg[j] = L.n_seg; a.t_last[j] = L.t_last;
            if (a.log_pool != nullptr) so_log_flush<MAP>(a, L);
        }
    };
    a.nfev[j] = 0; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; a.njev[j] = 0; a.nlu[j] = 0;
    a.facold[j] = 0.0; a.hlamb[j] = 1.0;
#pragma unroll
    for (int c = 0; c < C; ++c) { map_st<MAP>(a.y, c, B, j, y[c]); map_st<MAP>(a.k1, c, B, j, 0.0); }

    if (fabs(L.xend - L.x0) < 1e-15) {  // solve_ivp.rs:110-145
        if (FULL) {
            if (a.n_eval >= 0) {
                const EvalGrid grid = so_grid(a, j);
                for (int32_t i = 0; i < grid.n; ++i)
                    if (fabs(grid.t[i] - L.x0) < 1e-12) so_emit_eval<M_BDF, C, P, MAP>(a, j, L, i, y);
            } else if (a.t_log != nullptr) {
                so_push_log<M_BDF, C, P, MAP>(a, j, L, L.x0, y);
            }
            if (a.collect_dense && a.max_log > 0) {  // ContinuousOutput::constant, BDF layout (cont.rs:44-51)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (!MAP::own(c)) continue;
                    const size_t g = (size_t)MAP::gi(c) * 7;
                    a.seg_cont[(g + 0) * B + j] = y[c];
                    for (int s7 = 1; s7 < 6; ++s7) a.seg_cont[(g + s7) * B + j] = 0.0;
                    a.seg_cont[(g + 6) * B + j] = 1.0;
                }
                a.seg_xold[j] = L.x0;
                a.seg_h[j] = 1e-15;
                L.n_seg = 1;
            }
        }
        store_so();
        a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 0;
        return 0;
    }
    if (L.x0 != L.x0 || L.xend != L.xend) {   // NaN interval: see init_body in rk_core.h
        store_so();
        a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 3;
        return 3;
    }
    const double direction = rs_signum(L.xend - L.x0);
    const double hmax = fabs(a.has_max_step ? a.max_step : fabs(L.xend - L.x0));
    GR::ode(L.x0, y, f0, L.p);
    double *jac = a.bdf_jac + (size_t)j * NT * NT;
    if constexpr (BG::template HasJacCol<R>::v) {   // the reference's jac storage starts zeroed (bdf.rs:152)
        for (int e = BG::gl(); e < NT * NT; e += G) jac[e] = 0.0;
    }
    BG::eval_jac(L.x0, y, L.p, jac);
    double h_abs;
    if (a.has_first_step) {
        if (a.first_step == 0.0) {   // Err(InvalidStepSize), bdf.rs:192-197
            ivp_flag_error(a, IVP_ERRFLAG_INVALID_STEP);
            store_so();
            a.x[j] = L.x0; a.h[j] = 0.0; a.flags[j] = 0; a.status[j] = 0;
            return 0;
        }
        h_abs = fabs(a.first_step);
    } else {
        double guess = hinit<GR>(a, L.x0, y, direction, f0, L.p, 1, hmax);
        const double max_h = fabs(L.xend - L.x0);
        if (fabs(guess) > max_h) guess = max_h * direction;
        h_abs = fabs(guess);
    }
    h_abs = fmin(h_abs, fmax(hmax, 2.2250738585072014e-308));
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (!MAP::own(c)) continue;
        const size_t g = (size_t)MAP::gi(c);
        a.bdf_d[(0 * (size_t)NT + g) * B + j] = y[c];
        a.bdf_d[(1 * (size_t)NT + g) * B + j] = f0[c] * h_abs * direction;
#pragma unroll
        for (int k = 2; k < 8; ++k) a.bdf_d[((size_t)k * NT + g) * B + j] = 0.0;
    }
    L.x = L.x0;
    if (FULL) (void)solout_full<M_BDF, GR>(a, j, L, L.x0, L.x0, y, (const double *)y, (const double *)nullptr, 0.0, L.x0);
    store_so();
    a.nfev[j] = 1; a.njev[j] = 1;
    a.x[j] = L.x0; a.h[j] = h_abs;
    a.flags[j] = (1u << IVP_BDF_ORDER_SHIFT) | (L.flags & IVP_F_FIRSTOUT);
    a.status[j] = IVP_RUNNING;
    return IVP_RUNNING;
}

template <int C>
struct BdfGLane {
    double y[C], d[8][C];
    double x, current_h, current_c, pending_factor, xend, x0, direction, hmax, hmin, rtol_min;
    uint32_t flags;
    int32_t status;
    uint32_t d_nfev, d_njev, d_nlu, d_nstep, d_naccpt, d_nrejct, budget;
    bool over;
};

// One pass of the main loop (bdf.rs:276-607). Returns false when the trajectory retired.
template <class R, int FULL, int G, bool LDSWORK = false>
__device__ __forceinline__ bool bdf_group_attempt(const IvpKArgs &a, uint32_t j, BdfGLane<BdfG<R, G>::C> &S, Lane<BdfG<R, G>::C, R::P> &L,
                                                  double *jac, double *lu, uint32_t *piv)
{
    using BG = BdfG<R, G>;
    using GR = GroupRhs<R, G>;
    constexpr int C = BG::C;
    constexpr BdfTables T{};
    constexpr double EPS = 2.220446049250313e-16, MIN_POSITIVE = 2.2250738585072014e-308;
    constexpr int newton_maxiter = 4;
    int order = (int)((S.flags >> IVP_BDF_ORDER_SHIFT) & 7u);
    int n_equal = (int)((S.flags >> IVP_BDF_NEQ_SHIFT) & 7u);
    bool lu_current = (S.flags & IVP_BDF_LU_CURRENT) != 0;
    auto pack = [&]() {
        S.flags = (S.flags & ~((7u << IVP_BDF_ORDER_SHIFT) | (7u << IVP_BDF_NEQ_SHIFT) | IVP_BDF_LU_CURRENT)) |
                  ((uint32_t)order << IVP_BDF_ORDER_SHIFT) | ((uint32_t)n_equal << IVP_BDF_NEQ_SHIFT) |
                  (lu_current ? IVP_BDF_LU_CURRENT : 0u);
    };

    IVP_CLK_T0();
    if (S.over || S.d_nstep >= S.budget) { S.status = 2; return false; }                 // steps.total >= nmax
    if (S.current_h < MIN_POSITIVE) { S.status = 3; return false; }
    double h_try = S.current_h;
    double h_signed = 0.0, x_new = 0.0;
    bool finished = false;
    // The four D-rescalings that may precede the predictor (a pending one from the previous attempt, the h_max clamp,
    // the h_min clamp, the last-step clamp; bdf.rs:276-340).  Their conditions only involve scalars, so the scalar
    // bookkeeping runs first, straight-line, and the rescalings follow in the reference's order through ONE instance of
    // change_d (it is the fattest helper; a factor of exactly 1.0 is its own early exit).
    double fpass[4] = {1.0, 1.0, 1.0, 1.0};
    if (S.flags & IVP_BDF_PENDING) { fpass[0] = S.pending_factor; S.flags &= ~IVP_BDF_PENDING; }
    if (h_try > S.hmax) {
        fpass[1] = S.hmax / h_try;
        h_try = S.hmax; S.current_h = h_try; n_equal = 0; lu_current = false;
    }
    if (h_try < S.hmin && S.hmin > 0.0) {
        fpass[2] = fmax(S.hmin / h_try, 1.0);
        h_try = S.hmin; S.current_h = h_try; n_equal = 0; lu_current = false;
    }
    h_signed = S.direction * h_try;
    x_new = S.x + h_signed;
    if (S.direction * (x_new - S.xend) > 0.0) {
        const double step_to_end = fabs(S.xend - S.x);
        if (step_to_end == 0.0) { finished = true; }
        else {
            fpass[3] = step_to_end / h_try;
            S.current_h *= fpass[3];
            h_try = S.current_h;
            h_signed = S.direction * h_try;
            x_new = S.x + h_signed;
            n_equal = 0; lu_current = false;
        }
    }
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const double factor = pass == 0 ? fpass[0] : (pass == 1 ? fpass[1] : (pass == 2 ? fpass[2] : fpass[3]));
        if (factor != 1.0) bdf_change_d<C>(S.d, order, factor);
    }
    if (finished) { pack(); S.status = 0; return false; }
    if ((S.x + 0.1 * fabs(h_signed)) == S.x) { pack(); S.status = 3; return false; }
    const double x_start = S.x;
    S.d_nstep += 1;

    double y_predict[C], scale[C], psi[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k <= order) sum += S.d[k][i];
        y_predict[i] = sum;
    }
#pragma unroll
    for (int i = 0; i < C; ++i) {
        scale[i] = IVP_MA(NormOps<GR>::atol(a, i), NormOps<GR>::rtol(a, i), fabs(y_predict[i]));
        if (scale[i] == 0.0) scale[i] = EPS;
    }
    const double alpha_o = bdf_sel6(T.alpha, order);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int jj = 1; jj < 6; ++jj) if (jj <= order) sacc = IVP_MA(sacc, T.gamma[jj], S.d[jj][i]);
        psi[i] = sacc / alpha_o;
    }
    const double c = h_signed / alpha_o;
    bool lu_failed = false;
    IVP_CLK(0);
    if (!lu_current || fabs(c - S.current_c) / fmax(fabs(c), 1.0) > 0.1) {
        __syncthreads();
        // (I - cJ), element by element over the n x n column-major block: coalesced, FB loads in flight per lane (a loop
        // over columns waited for one memory round trip per column: 64 000 cycles per refactorisation at n = 100)
        {
            constexpr int NE = BG::NT * BG::NT, FB = 8;
#pragma unroll 1
            for (int e0 = 0; e0 < NE; e0 += FB * G) {
                double jv[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) { const int e = e0 + u * G + BG::gl(); jv[u] = e < NE ? jac[e] : 0.0; }
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int e = e0 + u * G + BG::gl();
                    const int col = e / BG::NT, r = e - col * BG::NT;
                    const double v = r == col ? IVP_MA(1.0, -c, jv[u]) : -c * jv[u];   // (I - cJ): the diagonal is -c j + 1
                    if (e < NE) lu[e] = v;
                }
            }
        }
        S.d_nlu += 1;
        IVP_CLK(1);
        // (the first 64 trajectories report how dense their elimination was: a.err_flag[1..2], read by the host every round)
        if (BG::lu_decomp(lu, piv, j < 64u ? a.err_flag + 1 : nullptr)) { lu_current = true; S.current_c = c; }
        else lu_failed = true;
        __syncthreads();
        IVP_CLK(2);
    }
    if (lu_failed) {   // bdf.rs:373-381
        S.pending_factor = 0.5; S.flags |= IVP_BDF_PENDING;
        S.current_h *= 0.5; n_equal = 0; lu_current = false; S.d_nrejct += 1;
        pack();
        return true;
    }

    double y_new[C], delta[C], rhs[C];
#pragma unroll
    for (int i = 0; i < C; ++i) { y_new[i] = y_predict[i]; delta[i] = 0.0; }
    bool converged = false, has_prev = false;
    double dy_norm_prev = 0.0;
    int iters = 0;
    double newton_tol = fmax(10.0 * EPS / S.rtol_min, fmin(sqrt(S.rtol_min), 0.03));   // bdf.rs:174-185
    if (newton_tol <= 0.0) newton_tol = 1e-9;
#pragma unroll 1
    while (iters < newton_maxiter) {
        IVP_CLK(7);
        GR::ode(x_new, y_new, rhs, L.p);
        S.d_nfev += 1;
#pragma unroll
        for (int i = 0; i < C; ++i) rhs[i] = IVP_MB(c, rhs[i], psi[i]) - delta[i];
        IVP_CLK(3);
        BG::lin_solve(lu, piv, rhs);
        IVP_CLK(4);
        const double dy_norm = BG::wrms(rhs, scale);
        IVP_CLK(5);
        bool rate_condition = false;
        if (has_prev && dy_norm_prev > 0.0) {
            const double rate = dy_norm / dy_norm_prev;
            if (rate >= 1.0) rate_condition = true;
            else {
                // rate.powf(remaining), remaining = 3, 2, 1 iterations left: the product (oracle: orc_pow_small_int)
                static_assert(newton_maxiter == 4, "the contraction-rate power is unrolled for 1..3 iterations left");
                const int remaining = newton_maxiter - iters;
                double rate_pow = rate;
                rate_pow = remaining >= 2 ? rate_pow * rate : rate_pow;
                rate_pow = remaining >= 3 ? rate_pow * rate : rate_pow;
                const double estimate = rate_pow / (1.0 - rate) * dy_norm;
                if (estimate > newton_tol) rate_condition = true;
            }
        }
#pragma unroll
        for (int i = 0; i < C; ++i) { y_new[i] += rhs[i]; delta[i] += rhs[i]; }
        if (dy_norm == 0.0) { converged = true; break; }
        if (has_prev && dy_norm_prev > 0.0) {
            const double rate = dy_norm / dy_norm_prev;
            if (rate < 1.0) {
                const double estimate = rate / (1.0 - rate) * dy_norm;
                if (estimate < newton_tol) { converged = true; break; }
            }
        }
        if (rate_condition) break;
        dy_norm_prev = dy_norm; has_prev = true;
        iters += 1;
    }
    IVP_CLK(7);
    if (!converged) {   // bdf.rs:448-459: refresh the Jacobian at the predictor, halve the step
        BG::template eval_jac<LDSWORK>(x_new, y_predict, L.p, jac, lu);   // (lu_current goes false right below: the factors are dead)
        IVP_CLK(6);
        S.d_njev += 1;
        lu_current = false;
        S.pending_factor = 0.5; S.flags |= IVP_BDF_PENDING;
        S.current_h *= 0.5; n_equal = 0; S.d_nrejct += 1;
        pack();
        return true;
    }
    const double safety = 0.9 * (2.0 * (double)newton_maxiter + 1.0) / (2.0 * (double)newton_maxiter + (double)(iters + 1));
#pragma unroll
    for (int i = 0; i < C; ++i) {
        scale[i] = IVP_MA(NormOps<GR>::atol(a, i), NormOps<GR>::rtol(a, i), fabs(y_new[i]));
        if (scale[i] == 0.0) scale[i] = EPS;
    }
    const double ec_o = bdf_sel6(T.error_const, order);
#pragma unroll
    for (int i = 0; i < C; ++i) rhs[i] = ec_o * delta[i];
    const double error_norm = BG::wrms(rhs, scale);
    if (error_norm > 1.0) {   // bdf.rs:481-489
        double factor = safety * ivp_pow(error_norm, -1.0 / ((double)order + 1.0));
        factor = fmax(factor, 0.2);
        S.pending_factor = factor; S.flags |= IVP_BDF_PENDING;
        S.current_h *= factor; n_equal = 0; S.d_nrejct += 1;
        pack();
        return true;
    }

    S.d_naccpt += 1;
    n_equal += 1;
    S.x = x_new;
    double yold[C];
#pragma unroll
    for (int i = 0; i < C; ++i) { yold[i] = S.y[i]; S.y[i] = y_new[i]; }
#pragma unroll
    for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int k = 2; k < 8; ++k) {
            if (k == order + 2) S.d[k][i] = delta[i] - S.d[k - 1][i];
        }
#pragma unroll
        for (int k = 1; k < 7; ++k) {
            if (k == order + 1) S.d[k][i] = delta[i];
        }
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            if (k <= order) S.d[k][i] += S.d[k + 1][i];
        }
    }
    if (FULL) {
        double cont[7 * C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            cont[i * 7] = S.d[0][i];
#pragma unroll
            for (int k = 0; k < 5; ++k) cont[i * 7 + 1 + k] = (k + 1 <= order) ? S.d[k + 1][i] : 0.0;
            cont[i * 7 + 6] = (double)order;
        }
        L.x0 = S.x0;
        if (solout_full<M_BDF, GR>(a, j, L, S.x - h_signed, S.x, S.y, yold, cont, h_signed, x_start)) { pack(); S.status = 1; return false; }
    }
    if (S.direction * (S.x - S.xend) >= 0.0) { pack(); S.status = 0; return false; }

    if (n_equal >= order + 1) {   // order / step adaptation, bdf.rs:551-606
        double err_m = u2d(0x7FF0000000000000ull), err_p = u2d(0x7FF0000000000000ull);
        if (order > 1) {
            const double ecm = bdf_sel6(T.error_const, order - 1);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                double dv = S.d[1][i];
                IVP_OPAQUE_V(dv);
#pragma unroll
                for (int k = 2; k < 6; ++k) { double dk = S.d[k][i]; IVP_OPAQUE_V(dk); dv = (k == order) ? dk : dv; }
                rhs[i] = ecm * dv;
            }
            err_m = BG::wrms(rhs, scale);
        }
        if (order < BDF_MAXO) {
            const double ecp = bdf_sel6(T.error_const, order + 1);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                double dv = S.d[3][i];
                IVP_OPAQUE_V(dv);
#pragma unroll
                for (int k = 4; k < 8; ++k) { double dk = S.d[k][i]; IVP_OPAQUE_V(dk); dv = (k == order + 2) ? dk : dv; }
                rhs[i] = ecp * dv;
            }
            err_p = BG::wrms(rhs, scale);
        }
        double factors[3];
        const double errors[3] = {err_m, error_norm, err_p};
#pragma unroll 1
        for (int idx = 0; idx < 3; ++idx) {
            const double e = idx == 0 ? errors[0] : (idx == 1 ? errors[1] : errors[2]);
            const double v = ivp_pow(e, -1.0 / ((double)order + (double)idx));
            if (idx == 0) factors[0] = v; else if (idx == 1) factors[1] = v; else factors[2] = v;
        }
        int best = 0;   // Iterator::max_by keeps a later element unless the current maximum is strictly greater
        double bestv = factors[0];
        if (!(bestv > factors[1])) { best = 1; bestv = factors[1]; }
        if (!(bestv > factors[2])) { best = 2; bestv = factors[2]; }
        int new_order = order;
        if (best == 0 && order > 1) new_order -= 1;
        else if (best == 2 && order < BDF_MAXO) new_order += 1;
        double max_factor = fmax(fmax(fmax(0.0, factors[0]), factors[1]), factors[2]);
        const double step_factor = fmin(safety * max_factor, 10.0);
        const int old_order = order;
        S.pending_factor = step_factor; S.flags |= IVP_BDF_PENDING;   // change_d(d, new_order, step_factor)
        S.current_h *= step_factor;
        order = new_order;
        n_equal = 0;
        lu_current = false;
        IVP_CLK(7);
        if (new_order != old_order) { BG::template eval_jac<LDSWORK>(S.x, S.y, L.p, jac, lu); S.d_njev += 1; }
        IVP_CLK(6);
    }
    IVP_CLK(7);
    pack();
    return true;
}

// largest n whose factors the LDSLU kernels keep in LDS (128 x 128 doubles = 128 KiB of the CU's 160)
#define IVP_LDS_LU_MAX_N 128
template <class R, int FULL, int G, bool LDSLU>
__device__ __forceinline__ uint32_t bdf_group_chunk_body(const IvpKArgs &a, uint32_t j, int32_t &status_out)
{
    using BG = BdfG<R, G>;
    using GR = GroupRhs<R, G>;
    using MAP = typename OutMap<GR>::type;
    constexpr int C = BG::C, NT = BG::NT, P = R::P;
    constexpr bool kLds = LDSLU && BG::NGROUP == 1 && NT <= IVP_LDS_LU_MAX_N;
    static_assert(!LDSLU || kLds, "LDS-resident factors: one wavefront per trajectory, n <= IVP_LDS_LU_MAX_N");
    const size_t B = a.B;
    BdfGLane<C> S;
    Lane<C, P> L;
#pragma unroll
    for (int c = 0; c < C; ++c) S.y[c] = map_ld<MAP>(a.y, c, B, j);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) S.d[k][c] = MAP::own(c) ? a.bdf_d[((size_t)k * NT + MAP::gi(c)) * B + j] : 0.0;
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    double *jac = a.bdf_jac + (size_t)j * NT * NT, *lu_mem = a.bdf_lu + (size_t)j * NT * NT;
    uint32_t *piv_mem = a.bdf_piv + (size_t)j * NT;
    double *lu = lu_mem;
    uint32_t *piv = piv_mem;
    S.x = a.x[j];
    S.current_h = a.h[j];
    S.current_c = a.facold[j];
    S.pending_factor = a.hlamb[j];
    S.flags = a.flags[j];
    if constexpr (kLds) {
        __shared__ double ivp_lu_lds[NT * NT];
        __shared__ uint32_t ivp_piv_lds[NT];
        lu = ivp_lu_lds;
        piv = ivp_piv_lds;
        if (S.flags & IVP_BDF_LU_CURRENT) {   // factors computed by an earlier launch: fetch them (coalesced)
            for (int e = (int)threadIdx.x; e < NT * NT; e += IVP_WAVE) ivp_lu_lds[e] = lu_mem[e];
            for (int e = (int)threadIdx.x; e < NT; e += IVP_WAVE) ivp_piv_lds[e] = piv_mem[e];
        }
        __syncthreads();
    }
    S.x0 = a.t0[(size_t)j * a.t0_stride];
    S.xend = a.t1[(size_t)j * a.t1_stride];
    S.direction = rs_signum(S.xend - S.x0);
    S.hmax = fabs(a.has_max_step ? a.max_step : fabs(S.xend - S.x0));
    S.hmin = fabs(a.has_min_step ? a.min_step : 0.0);
    S.rtol_min = bdfg_rtol_min<R, G>(a);
    S.status = IVP_RUNNING;
    S.d_nfev = S.d_njev = S.d_nlu = S.d_nstep = S.d_naccpt = S.d_nrejct = 0;
    const uint64_t nstep0 = a.nstep[j];
    S.over = nstep0 >= a.nmax;
    const uint64_t left = S.over ? 0 : a.nmax - nstep0;
    S.budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    L.flags = S.flags & IVP_F_FIRSTOUT;
    L.x0 = S.x0;
    L.kz = 0;
    if (FULL) {
        L.next_idx = a.next_idx[j]; L.n_filled = a.n_filled[j]; L.n_log = a.n_log[j];
        L.n_seg = a.n_seg[j]; L.t_last = a.t_last[j];
    } else {
        L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    }
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0;
    uint32_t it = 0;
    bool run = true;
    while (run && it < a.chunk) {
        if (FULL && a.log_pool != nullptr) so_log_attempt<MAP, 1>(a, j, L, it);
        run = bdf_group_attempt<R, FULL, G, kLds>(a, j, S, L, jac, lu, piv);
        ++it;
    }
    if (FULL && a.log_pool != nullptr) so_log_flush<MAP>(a, L);
    __syncthreads();
#ifdef IVP_PHASE_CLOCKS
    if (threadIdx.x == 0 && j == 0) {
        printf("phase clocks (launch of %u attempts): pre %llu  form %llu  lu %llu  ode %llu  solve %llu  wrms %llu  jac %llu  rest %llu\n", it,
               ivp_phase_clk[0], ivp_phase_clk[1], ivp_phase_clk[2], ivp_phase_clk[3], ivp_phase_clk[4], ivp_phase_clk[5], ivp_phase_clk[6], ivp_phase_clk[7]);
        printf("  inside lu: column load %llu  pivot search %llu  swap + multipliers %llu  pivot-row loads %llu  trailing update + barrier %llu\n",
               ivp_phase_clk[8], ivp_phase_clk[9], ivp_phase_clk[10], ivp_phase_clk[11], ivp_phase_clk[12]);
        for (int q = 0; q < 16; ++q) ivp_phase_clk[q] = 0;
    }
#endif
    if constexpr (kLds) {
        if (S.flags & IVP_BDF_LU_CURRENT) {   // the next launch continues with these factors (same nlu as the global-memory path)
            for (int e = (int)threadIdx.x; e < NT * NT; e += IVP_WAVE) lu_mem[e] = lu[e];
            for (int e = (int)threadIdx.x; e < NT; e += IVP_WAVE) piv_mem[e] = piv[e];
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) map_st<MAP>(a.y, c, B, j, S.y[c]);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) if (MAP::own(c)) a.bdf_d[((size_t)k * NT + MAP::gi(c)) * B + j] = S.d[k][c];
    a.x[j] = S.x;
    a.h[j] = S.status == IVP_RUNNING ? S.current_h : S.direction * S.current_h;   // IntegrationResult.h, bdf.rs:609-614
    a.facold[j] = S.current_c;
    a.hlamb[j] = S.pending_factor;
    a.flags[j] = (S.flags & ~IVP_F_FIRSTOUT) | (L.flags & IVP_F_FIRSTOUT);
    a.status[j] = S.status;
    if (BG::gl() == 0) {   // counters are read-modify-write: one lane per trajectory
        a.nfev[j] += S.d_nfev; a.njev[j] += S.d_njev; a.nlu[j] += S.d_nlu;
        a.nstep[j] += S.d_nstep; a.naccpt[j] += S.d_naccpt; a.nrejct[j] += S.d_nrejct;
    }
    if (FULL) {
        a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_log[j] = L.n_log;
        a.n_seg[j] = L.n_seg; a.t_last[j] = L.t_last;
    }
    status_out = S.status;
    return it;
}

}  // namespace IVP_NS
