// rk_core.h -- per-lane bodies of the gfx950 batched explicit Runge-Kutta kernels.
//
// One GPU lane owns one trajectory (one reference solve_ivp() call): y, the FSAL derivative k1,
// x, h and the controller memory live in VGPRs for a whole chunk of step attempts; the k-stages are
// VGPR arrays indexed at compile time; tableau coefficients are literals (wave-uniform -> SGPRs /
// inline constants).  Nothing here touches LDS or other lanes: trajectories are independent.
//
// The code is written against IVP_HD so that tests/host_emul can run the very same bodies lane by
// lane on a CPU (no GPU in the authoring container); the product only ever compiles it for gfx950
// through rk_kernels.hip.
//
// Floating-point contract: in the STRICT build (-ffp-contract=off) every expression keeps the
// reference's left-to-right association (SURVEY.md section 7 "FP / semantics notes"), so the result is
// the same IEEE-754 operation sequence as the Rust crate evaluates; only the step-controller
// power function differs (ivp_pow below instead of libm pow).  The FMA build (IVP_FAST = 1, ivp_options_t.fp_mode =
// IVP_FP_FMA) is a DEFINED arithmetic as well, not "whatever the compiler contracts": it is compiled with
// -ffp-contract=off too, and every fused multiply-add is spelled out through IVP_MA / IVP_MS / ivp_lc below (the
// multiply-add sites of the stage combinations, error estimates, dense coefficients, interpolants, tolerance scales
// and of the built-in right-hand sides, whose FMA forms also share one reciprocal per primary / denominator).  The
// same sites are fused in oracle/ivp_oracle.c's ORC_FMA build, so FMA-mode results are bit-comparable with
// liboracle_fma.so and identical in every kernel variant (lean, resident, lane-cooperative, wave-per-trajectory).
//
// Reference citations are file:line in Ryan-D-Gast/ivp 0.5.1.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#include <math.h>
#include "ivp_kargs.h"
#endif

#ifndef IVP_HD
#define IVP_HD __host__ __device__ __forceinline__
#endif
#ifndef IVP_FAST
#define IVP_FAST 0
#endif
#ifndef IVP_NS
#define IVP_NS ivp
#endif
// KC(c): a wave-uniform f64 literal, materialised into an SGPR pair right where it is used.
// gfx950 VOP3 f64 instructions cannot encode a 64-bit literal, so every tableau / polynomial
// coefficient must sit in registers; left alone, LLVM hoists all ~70 of them out of the attempt
// loop (first into SGPRs, then, once those run out, into *vector* registers: 140+ VGPRs) and
// halves the occupancy.  XOR-ing the bit pattern with an opaque scalar zero that is re-defined
// inside the loop (KC_SCOPE) makes each constant loop-variant as far as LICM can tell: it costs
// two `s_xor_b32 sN, sZ, literal` SALU instructions per use (they co-issue with the other waves'
// VALU work) and keeps the live register set down to the integrator's real state.
// On the host (tests/host_emul) KC is the identity.  Build variant IVP_HOIST=1 also makes it the identity:
// that variant lets LLVM keep every coefficient resident in registers (~250 VGPRs, one or two waves per
// SIMD) and is the right trade once the active set no longer over-subscribes the chip, where the SALU
// moves of the lean variant sit on the critical path of a lone wave.
#ifndef IVP_HOIST
#define IVP_HOIST 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !IVP_HOIST
#define KC_SCOPE const uint64_t ivp_kz = IVP_NS::ivp_opaque_zero();
#define KC(c) IVP_NS::u2d(__builtin_bit_cast(uint64_t, (double)(c)) ^ ivp_kz)
#elif defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 2
// Build variant IVP_HOIST=2 (kernels in which a lone wave owns its SIMD: rk_coop.h, rk_group.h): every
// coefficient is PINNED in a vector register for the whole attempt loop.  The XOR partner is an opaque zero
// defined once before the attempt loop (Lane::kz), so loop-invariant code motion hoists the XOR and the
// result cannot be re-materialised as a literal.  A lone wave pays a full issue slot for every s_mov / v_mov
// that re-creates a constant (measured: ~100 of ~1350 slots per DOPRI5 attempt), and has registers to spare.
#define KC_SCOPE const uint64_t ivp_kz = 0;   /* code outside the attempt loop (init kernel): plain literals */
#define KC(c) IVP_NS::u2d(__builtin_bit_cast(uint64_t, (double)(c)) ^ ivp_kz)
#elif defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 1
// resident build: plain literals (LLVM keeps what fits in registers); ivp_kz only travels to the power core, whose
// coefficients are pinned in vector registers (ivp_pow_core_w)
#define KC_SCOPE const uint64_t ivp_kz = 0; (void)ivp_kz;
#define KC(c) (c)
#else
#define KC_SCOPE
#define KC(c) (c)
#endif
// KC_SCOPE_KZ(z): like KC_SCOPE inside the attempt loop; in the pinned variant the XOR partner is the opaque vector
// zero `z` that chunk_body defined BEFORE the loop (Lane::kz), so `literal ^ z` is loop-invariant and gets hoisted.
// IVP_KZ_ARG hands the partner on to helpers that have no Lane (ivp_pow).
#if defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 2
#define KC_SCOPE_KZ(z) const uint64_t ivp_kz = (z);
#define IVP_KZ_ARG ivp_kz
#elif defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 1
#define KC_SCOPE_KZ(z) const uint64_t ivp_kz = (z); (void)ivp_kz;
#define IVP_KZ_ARG ivp_kz
#else
#define KC_SCOPE_KZ(z) KC_SCOPE
#define IVP_KZ_ARG 0ull
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define IVP_OPAQUE_V(v) asm volatile("" : "+v"(v))
#else
#define IVP_OPAQUE_V(v) ((void)0)
#endif

// The multiply-add sites.  STRICT: two IEEE operations in the reference's association; FMA: one fused operation.
//   IVP_MA(acc, a, b) = acc + a * b        IVP_MS(acc, a, b) = acc - a * b        IVP_MB(a, b, c) = a * b - c
//   IVP_LC(c1, k1, c2, k2, ...) = ((c1 * k1 + c2 * k2) + c3 * k3) + ...   (the reference's left-to-right sums)
#if IVP_FAST
#define IVP_MA(acc, a, b) fma((a), (b), (acc))
#define IVP_MS(acc, a, b) fma(-(a), (b), (acc))
#define IVP_MB(a, b, c) fma((a), (b), -(c))
#else
#define IVP_MA(acc, a, b) ((acc) + (a) * (b))
#define IVP_MS(acc, a, b) ((acc) - (a) * (b))
#define IVP_MB(a, b, c) ((a) * (b) - (c))
#endif
#define IVP_LC(c1, k1, ...) IVP_NS::ivp_lc((c1) * (k1), __VA_ARGS__)

namespace IVP_NS {

IVP_HD double ivp_lc(double acc) { return acc; }
template <class... Ts>
IVP_HD double ivp_lc(double acc, double c, double k, Ts... rest) { return ivp_lc(IVP_MA(acc, c, k), rest...); }

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t ivp_opaque_zero()
{
    uint64_t z = 0;
    asm volatile("" : "+s"(z));
    return z;
}
__device__ __forceinline__ uint64_t ivp_opaque_zero_v()
{
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return ((uint64_t)z << 32) | z;
}
#endif
// whole-call validation failures that can only be detected per trajectory (the reference's Err(Error::Config))
#define IVP_ERRFLAG_INVALID_STEP 0x1u
// the page pool of the one-pass accepted-step log ran dry (IvpKArgs.log_pool): records were counted but not all stored
#define IVP_ERRFLAG_LOG_OVERFLOW 0x2u
IVP_HD void ivp_flag_error(const IvpKArgs &a, uint32_t bit)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(a.err_flag, bit);
#else
    *a.err_flag |= bit;
#endif
}
IVP_HD double rs_signum(double v) { return v != v ? v : copysign(1.0, v); }  // Rust f64::signum
IVP_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
IVP_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// Step-controller power x^e for x >= 0 (err^expo1, facold^beta, (0.01/der12)^(1/iord)):
// exp2(e*log2 x) from IEEE +,-,*,/,fma,rint and bit moves only, ~2 ulp.  Replaces f64::powf
// (dopri5.rs:351-353, dop853.rs:432-434, rk23.rs:289,303, mod.rs:276); deterministic on any IEEE
// machine, which is what makes the strict path bit-comparable with a CPU restatement.
// Two entry points with identical results: ivp_pow_full is the definition (every special case in order);
// ivp_pow takes the common case -- x a positive normal number, e finite and non-zero -- through one
// straight-line instruction stream (no exec-mask branches: the out-of-range results are selected at the
// end) and hands everything else to ivp_pow_full.
#if defined(__HIP_DEVICE_COMPILE__)
// a * b + c with the wave-uniform c read from an SGPR pair (three-address VOP3 form)
__device__ __forceinline__ double ivp_fma_sgpr_addend(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#endif
// W independent powers evaluated in lock step: statement by statement over all W operands, so that the W dependent chains
// are interleaved in the instruction stream as written (a lone wave waits 8.5 cycles for a dependent f64 result and
// issues every 5.5; the register-pressure-driven scheduler keeps source order).  Per operand the operations and their
// order are those of the scalar form: W = 1 IS the scalar form.
template <int W>
IVP_HD void ivp_pow_core_w(const double (&x)[W], const double (&e)[W], const int (&k0)[W], double (&res)[W], uint64_t kz)
{
    // Coefficients of the two Horner chains.  In the thread-per-trajectory kernels (IVP_HOIST = 0 / 1: two or more waves
    // per SIMD) they are kept OUT of the vector registers: with the constants pinned there LLVM
    // emits `v_mov_b64 tmp, c; v_fmac_f64 tmp, p, z` per step (a copy to protect the pinned value), whereas an SGPR
    // operand gives the single three-address `v_fma_f64 p, p, z, s[..]` and the s_xor that re-creates it issues on the
    // scalar port beside the other wave's vector work.
#if defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST == 1
    // resident build (at most two waves per SIMD, since the windowed launches mostly ONE): a lone wave pays a full issue
    // slot for each of the two s_xor that re-create a coefficient (~100 slots per DOPRI5 attempt), so here the 26
    // coefficients are pinned in vector registers (XOR with the opaque zero defined before the attempt loop: hoisted, never
    // re-materialised); the compiler picks the three-address v_fma_f64 for the Horner steps (no copy to protect the pinned
    // value, and -- unlike an inline-asm fma -- no s_nop between dependent steps)
#define KP(c) IVP_NS::u2d(__builtin_bit_cast(uint64_t, (double)(c)) ^ kz)
#define PFMA(a, b, c) fma((a), (b), (c))
#elif defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST != 2
    const uint64_t ivp_kzp = IVP_NS::ivp_opaque_zero();
#define KP(c) IVP_NS::u2d(__builtin_bit_cast(uint64_t, (double)(c)) ^ ivp_kzp)
    // ... and the fused multiply-add is spelled out, because instruction selection otherwise turns fma(p, z, c) with a
    // freshly made c into the two-address v_fmac_f64 (dst = c), which wants c in a VGPR: two v_mov_b32 per step.
#define PFMA(a, b, c) IVP_NS::ivp_fma_sgpr_addend((a), (b), (c))
#else
#define PFMA(a, b, c) fma((a), (b), (c))
    KC_SCOPE_KZ(kz)
#define KP(c) KC(c)
#endif
    (void)kz;
#define IVP_W for (int w = 0; w < W; ++w)
    int k[W];
    double t[W], z[W], p[W];
#pragma unroll
    IVP_W {
        const uint64_t u = d2u(x[w]);
        int ex = (int)(u >> 52) - 1023;
        const uint64_t mant = u & 0x000FFFFFFFFFFFFFull;
        const bool hi = mant > 0x6A09E667F3BCDull;
        const double m = u2d(mant | (hi ? 0x3FE0000000000000ull : 0x3FF0000000000000ull));
        ex += hi ? 1 : 0;
        k[w] = k0[w] + ex;
        t[w] = (m - 1.0) / (m + 1.0);
        z[w] = t[w] * t[w];
        p[w] = KP(1.0 / 25.0);
    }
#define IVP_PSTEP(c) _Pragma("unroll") IVP_W p[w] = PFMA(p[w], z[w], KP(c));
    IVP_PSTEP(1.0 / 23.0)
    IVP_PSTEP(1.0 / 21.0)
    IVP_PSTEP(1.0 / 19.0)
    IVP_PSTEP(1.0 / 17.0)
    IVP_PSTEP(1.0 / 15.0)
    IVP_PSTEP(1.0 / 13.0)
    IVP_PSTEP(1.0 / 11.0)
    IVP_PSTEP(1.0 / 9.0)
    IVP_PSTEP(1.0 / 7.0)
    IVP_PSTEP(1.0 / 5.0)
    IVP_PSTEP(1.0 / 3.0)
#undef IVP_PSTEP
    double wv[W], kd[W], v[W], q[W];
#pragma unroll
    IVP_W {
        p[w] = fma(p[w], z[w], 1.0);
        const double lnm = (2.0 * t[w]) * p[w];
        const double l2 = fma(lnm, KP(0x1.71547652b82fep+0), (double)k[w]);
        wv[w] = e[w] * l2;
        // w >= 1024 -> +inf and w <= -1022 -> 0 are selected at the end; a NaN w (e = +-inf with x == 1) travels through
        kd[w] = rint(wv[w]);
        const double r = wv[w] - kd[w];
        v[w] = r * KP(0x1.62e42fefa39efp-1);
        q[w] = KP(1.0 / 87178291200.0);
    }
#define IVP_QSTEP(c) _Pragma("unroll") IVP_W q[w] = PFMA(q[w], v[w], KP(c));
    IVP_QSTEP(1.0 / 6227020800.0)
    IVP_QSTEP(1.0 / 479001600.0)
    IVP_QSTEP(1.0 / 39916800.0)
    IVP_QSTEP(1.0 / 3628800.0)
    IVP_QSTEP(1.0 / 362880.0)
    IVP_QSTEP(1.0 / 40320.0)
    IVP_QSTEP(1.0 / 5040.0)
    IVP_QSTEP(1.0 / 720.0)
    IVP_QSTEP(1.0 / 120.0)
    IVP_QSTEP(1.0 / 24.0)
    IVP_QSTEP(1.0 / 6.0)
#undef IVP_QSTEP
#pragma unroll
    IVP_W q[w] = fma(q[w], v[w], 0.5);
#pragma unroll
    IVP_W q[w] = fma(q[w], v[w], 1.0);
#pragma unroll
    IVP_W q[w] = fma(q[w], v[w], 1.0);
#pragma unroll
    IVP_W {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ki = (int)kd[w];   // v_cvt_i32_f64 saturates; kd is out of range only where the result is replaced below
#else
        const int ki = (kd[w] >= -2000.0 && kd[w] <= 2000.0) ? (int)kd[w] : 0;
#endif
        double r = q[w] * u2d((uint64_t)((uint32_t)ki + 1023u) << 52);
        r = (wv[w] <= -1022.0) ? 0.0 : r;
        r = (wv[w] >= 1024.0) ? u2d(0x7FF0000000000000ull) : r;
        res[w] = r;
    }
#undef IVP_W
}
#undef KP
#undef PFMA
IVP_HD double ivp_pow_core(double x, double e, int k0, uint64_t kz)
{
    const double xs[1] = {x}, es[1] = {e};
    const int ks[1] = {k0};
    double r[1];
    ivp_pow_core_w<1>(xs, es, ks, r, kz);
    return r[0];
}
IVP_HD double ivp_pow_full(double x, double e, uint64_t kz)
{
    if (e == 0.0) return 1.0;
    if (x != x || e != e) return x + e;
    if (x < 0.0) return u2d(0x7FF8000000000000ull);
    if (x == 0.0) return e > 0.0 ? 0.0 : u2d(0x7FF0000000000000ull);
    if (x == u2d(0x7FF0000000000000ull)) return e > 0.0 ? x : 0.0;
    int k = 0;
    if ((d2u(x) >> 52) == 0) { x *= 0x1p54; k = -54; }
    const double r = ivp_pow_core(x, e, k, kz);
    // e = +-inf with x == 1 gives w = NaN: neither range test holds and the NaN travels through the polynomial
    return r;
}
IVP_HD double ivp_pow(double x, double e, uint64_t kz = 0)
{
    const uint64_t ux = d2u(x);
    const bool common = (ux - 0x0010000000000000ull) < 0x7FE0000000000000ull   // x positive, normal, finite
                        && fabs(e) < u2d(0x7FF0000000000000ull) && e != 0.0;     // e finite, non-zero
    if (common) return ivp_pow_core(x, e, 0, kz);
    return ivp_pow_full(x, e, kz);
}

// Three powers in ONE instruction stream (same values as ivp_pow, argument for argument).  BDF needs err^(-1/(order+k))
// for k = 0, 1, 2 at the same place (bdf.rs:482, 568-577); three dependent Horner chains one after the other leave a
// lone wave waiting on its own results (8.5 cycles per dependent f64 operation against 5.5 of issue), three interleaved
// ones do not.  The special cases of ivp_pow_full are applied as selects on top of the core's result, in reverse
// order of their priority; only a subnormal base still branches.
IVP_HD void ivp_pow3(const double (&x)[3], const double (&e)[3], double (&r)[3], uint64_t kz = 0)
{
    const double inf = u2d(0x7FF0000000000000ull);
    bool normal[3], subnormal = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        normal[i] = (d2u(x[i]) - 0x0010000000000000ull) < 0x7FE0000000000000ull;   // positive, normal, finite
        subnormal = subnormal || (x[i] > 0.0 && x[i] < 0x1p-1022 && e[i] == e[i] && e[i] != 0.0);
    }
    const double xs[3] = {normal[0] ? x[0] : 1.0, normal[1] ? x[1] : 1.0, normal[2] ? x[2] : 1.0};
    const int ks[3] = {0, 0, 0};
    ivp_pow_core_w<3>(xs, e, ks, r, kz);
    // the three results exist here, unconditionally: without this LLVM sinks each core into "if its result is selected",
    // which puts the three chains into three consecutive branches
#pragma unroll
    for (int i = 0; i < 3; ++i) IVP_OPAQUE_V(r[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double v = r[i];
        v = (x[i] == inf) ? (e[i] > 0.0 ? inf : 0.0) : v;
        v = (x[i] == 0.0) ? (e[i] > 0.0 ? 0.0 : inf) : v;
        v = (x[i] < 0.0) ? u2d(0x7FF8000000000000ull) : v;
        v = (x[i] != x[i] || e[i] != e[i]) ? x[i] + e[i] : v;
        v = (e[i] == 0.0) ? 1.0 : v;
        r[i] = v;
    }
    if (subnormal) {
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            const double xi = i == 0 ? x[0] : (i == 1 ? x[1] : x[2]), ei = i == 0 ? e[0] : (i == 1 ? e[1] : e[2]);
            if (xi > 0.0 && xi < 0x1p-1022 && ei == ei && ei != 0.0) {
                const double v = ivp_pow_full(xi, ei, kz);
                if (i == 0) r[0] = v; else if (i == 1) r[1] = v; else r[2] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Right-hand sides: the device-side `impl IVP for T { fn ode(&self, x, y, dydx) }` (src/ivp.rs:29).
// ------------------------------------------------------------------------------------------------
struct RhsDecay {    // examples/exponential_decay.rs:9-13
    enum { N = 1, P = 1, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *p) { d[0] = -p[0] * y[0]; }
};
struct RhsSho {      // tests/common.rs:3-9
    enum { N = 2, P = 0, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *) { d[0] = y[1]; d[1] = -y[0]; }
};
struct RhsVdp {      // benches/benchmark.py:22-27
    enum { N = 2, P = 1, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *p)
    {
        d[0] = y[1];
        d[1] = IVP_MB(p[0] * IVP_MS(1.0, y[0], y[0]), y[1], y[0]);   // p0 * (1 - y0 * y0) * y1 - y0
    }
};
struct RhsCr3bp {    // examples/cr3bp.rs:23-36
    enum { N = 6, P = 1, NE = 0 };
    static IVP_HD void ode(double, const double *s, double *d, const double *p)
    {
        const double mu = p[0];
        const double x = s[0], y = s[1], z = s[2], vx = s[3], vy = s[4], vz = s[5];
        const double a = x + mu;
        const double b = x - 1.0 + mu;
#if IVP_FAST
        // FMA form (oracle: rhs_cr3bp under ORC_FMA): one division per primary instead of three, every a * b + c fused
        const double d1 = fma(z, z, fma(y, y, a * a));
        const double d2 = fma(z, z, fma(y, y, b * b));
        const double r1 = sqrt(d1), r2 = sqrt(d2);
        const double g1 = (1.0 - mu) / (d1 * r1);
        const double g2 = mu / (d2 * r2);
        d[0] = vx; d[1] = vy; d[2] = vz;
        d[3] = fma(-g2, b, fma(-g1, a, fma(2.0, vy, x)));
        d[4] = fma(-g2, y, fma(-g1, y, fma(-2.0, vx, y)));
        d[5] = fma(-g2, z, fma(-g1, z, -0.0));
#else
        const double r1 = sqrt(a * a + y * y + z * z);
        const double r2 = sqrt(b * b + y * y + z * z);
        const double r13 = r1 * r1 * r1;  // powi(3)
        const double r23 = r2 * r2 * r2;
        d[0] = vx; d[1] = vy; d[2] = vz;
        d[3] = x + 2.0 * vy - (1.0 - mu) * (x + mu) / r13 - mu * (x - 1.0 + mu) / r23;
        d[4] = y - 2.0 * vx - (1.0 - mu) * y / r13 - mu * y / r23;
        d[5] = -(1.0 - mu) * z / r13 - mu * z / r23;
#endif
    }
#if defined(__HIPCC__)
    // Lane-cooperative form for rk_coop.h.  Layout inside the 8-lane group: lanes 0..2 hold x, y, z, lanes 4..6 hold
    // vx, vy, vz (lanes 3 and 7 idle), so that every exchange below is one DPP hop.  The two quads run ONE instruction
    // stream: the lower quad evaluates the attraction of the first primary (offset a = x + mu, coefficient 1 - mu), the
    // upper quad that of the second (b = x - 1 + mu, coefficient mu) -- one sqrt and one division per lane.  Lane i of
    // a quad handles numerator i of (a|b, y, z).  The upper quad then combines
    //     d = (lin - T1) - T2,   T1 = ((1-mu) q1) / r1^3 from the lane four below,   T2 = (mu q2) / r2^3 its own,
    // which is the reference expression of d[3], d[4], d[5] bit for bit: x + (-0.0) == x, x + (-1.0) == x - 1.0,
    // y + (-2.0) vx == y - 2.0 vx, and lin = -0.0 reproduces the leading unary minus of d[5] including the sign of a
    // zero result ((-s) * t == -(s * t) and (-s) / t == -(s / t) hold exactly in IEEE arithmetic).
    static constexpr int coop_lane_of(int c) { return c < 3 ? c : c + 1; }
    static __device__ __forceinline__ double ode_coop(double, double ys, const double *p);
#endif
};
struct RhsLorenz {   // benches/benchmark.py:30-37
    enum { N = 3, P = 3, NE = 0 };
    static IVP_HD void ode(double, const double *s, double *d, const double *p)
    {
        d[0] = p[0] * (s[1] - s[0]);
        d[1] = IVP_MB(s[0], p[1] - s[2], s[1]);     // s0 * (p1 - s2) - s1
        d[2] = IVP_MS(s[0] * s[1], p[2], s[2]);     // s0 * s1 - p2 * s2
    }
};
struct RhsZero {     // tests/ivp.rs:11-19
    enum { N = 3, P = 0, NE = 0 };
    static IVP_HD void ode(double, const double *, double *d, const double *) { d[0] = 0.0; d[1] = 0.0; d[2] = 0.0; }
};
struct RhsRational { // tests/test_helpers.py:23-25
    enum { N = 2, P = 0, NE = 0 };
    static IVP_HD void ode(double t, const double *y, double *d, const double *)
    {
        d[0] = y[1] / t;
        d[1] = y[1] * (IVP_MA(y[0], 2.0, y[1]) - 1.0) / (t * (y[0] - 1.0));
    }
};
struct RhsExp2 {     // tests/ivp.rs:291-298
    enum { N = 2, P = 0, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *) { d[0] = y[0]; d[1] = y[1]; }
};

struct RhsLinear {   // tests/test_helpers.py:11-12
    enum { N = 2, P = 0, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *)
    {
        d[0] = IVP_MS(-y[0], 5.0, y[1]);
        d[1] = y[0] + y[1];
    }
};
struct RhsRobertson {   // tests/test_ivp.py:327-333
    enum { N = 3, P = 0, NE = 0 };
    static IVP_HD void ode(double, const double *s, double *d, const double *)
    {
        const double x = s[0], y = s[1], z = s[2];
        d[0] = IVP_MA(-0.04 * x, 1e4 * y, z);                       // -0.04 x + 1e4 y z
        d[1] = IVP_MS(IVP_MS(0.04 * x, 1e4 * y, z), 3e7 * y, y);    // 0.04 x - 1e4 y z - 3e7 y y
        d[2] = 3e7 * y * y;
    }
};
// Robertson with an `impl IVP { fn jac(..) }` override (src/ivp.rs:67-107): the analytic Jacobian, j[row][col]
struct RhsRobertsonJac : RhsRobertson {
    static IVP_HD void jac(double, const double *s, double (&j)[3][3], const double *)
    {
        const double y = s[1], z = s[2];
        j[0][0] = -0.04;  j[0][1] = 1e4 * z;              j[0][2] = 1e4 * y;
        j[1][0] = 0.04;   j[1][1] = IVP_MS(-1e4 * z, 6e7, y);   j[1][2] = -1e4 * y;
        j[2][0] = 0.0;    j[2][1] = 6e7 * y;              j[2][2] = 0.0;
    }
};
struct RhsVdpEps {   // examples/van_der_pol.rs:9-14
    enum { N = 2, P = 1, NE = 0 };
    static IVP_HD void ode(double, const double *y, double *d, const double *p)
    {
        d[0] = y[1];
        d[1] = IVP_MB(IVP_MS(1.0, y[0], y[0]), y[1], y[0]) / p[0];   // ((1 - y0 y0) y1 - y0) / eps
    }
};

// Problems with event functions: trait IVP::events / n_events (src/ivp.rs:31-46). NE = n_events().
struct RhsShoEv {    // tests/ivp.rs:151-221: SHO with g = y0
    enum { N = 2, P = 0, NE = 1 };
    static IVP_HD void ode(double, const double *y, double *d, const double *) { d[0] = y[1]; d[1] = -y[0]; }
    static IVP_HD void events(double, const double *y, double *g, const double *) { g[0] = y[0]; }
};
struct RhsBall {     // examples/bouncing_ball.rs:5-31  p = {gravity, drag}
    enum { N = 2, P = 2, NE = 1 };
    static IVP_HD void ode(double, const double *s, double *d, const double *p)
    {
        const double vy = s[1];
        d[0] = vy;
        d[1] = IVP_MS(-p[0], p[1] * vy, fabs(vy));   // -g - drag vy |vy|
    }
    static IVP_HD void events(double, const double *y, double *g, const double *) { g[0] = y[0]; }
};
struct RhsCannon {   // tests/test_ivp.py:152-160
    enum { N = 2, P = 0, NE = 1 };
    static IVP_HD void ode(double, const double *y, double *d, const double *) { d[0] = y[1]; d[1] = -9.80665; }
    static IVP_HD void events(double, const double *y, double *g, const double *) { g[0] = y[0]; }
};
struct RhsRationalEv {   // tests/test_ivp.py:345-353
    enum { N = 2, P = 0, NE = 3 };
    static IVP_HD void ode(double t, const double *y, double *d, const double *)
    {
        d[0] = y[1] / t;
        d[1] = y[1] * (IVP_MA(y[0], 2.0, y[1]) - 1.0) / (t * (y[0] - 1.0));
    }
    static IVP_HD void events(double t, const double *y, double *g, const double *)
    {
        g[0] = y[0] - pow(y[1], 0.7);
        g[1] = pow(y[1], 0.6) - y[0];
        g[2] = t - 7.4;
    }
};

// ------------------------------------------------------------------------------------------------
// Customisation points for kernels in which a lane holds only SOME components of a trajectory (rk_group.h:
// one wavefront per trajectory).  For the ordinary functors they are the identity / a plain loop, so the
// thread-per-trajectory code below is unchanged by them.
//   IdMap<N>      local component c -> global component gi(c) of NT; own(c) = this lane really holds it
//   NormOps<R>    NT = number of components in the RMS norms, sum() = the reference's left-to-right sum,
//                 rtol(a, c) / atol(a, c) = tolerances of local component c
// ------------------------------------------------------------------------------------------------
template <int N>
struct IdMap {
    enum { NT = N };
    static IVP_HD int gi(int c) { return c; }
    static IVP_HD bool own(int) { return true; }
    // lanes that hold one trajectory together (rk_coop.h, rk_group.h) elect one of them for per-trajectory side effects
    // (a page allocation of the one-pass step log) and share its result; a lane that owns its trajectory is its own leader
    static IVP_HD bool leader() { return true; }
    static IVP_HD uint32_t bcast(uint32_t v) { return v; }
};
template <class MAP>
IVP_HD uint64_t map_bcast64(uint64_t v) { return ((uint64_t)MAP::bcast((uint32_t)(v >> 32)) << 32) | (uint64_t)MAP::bcast((uint32_t)v); }
template <class R, class = void>
struct OutMap { using type = IdMap<R::N>; };
template <class R, class = void>
struct NormOps {
    enum { NT = R::N };
    static IVP_HD double rtol(const IvpKArgs &a, int i) { return a.rtol[i]; }   // Tolerance index (mod.rs:194-204)
    static IVP_HD double atol(const IvpKArgs &a, int i) { return a.atol[i]; }
    template <int N>
    static IVP_HD double sum(const double (&t)[N])
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) s += t[i];
        return s;
    }
    // the step controller's two powers (dopri5.rs:351-352, dop853.rs:432-433); rk_coop.h evaluates them side by side
    static IVP_HD void pow2(double x1, double e1, double x2, double e2, double &r1, double &r2, uint64_t kz)
    {
        r1 = ivp_pow(x1, e1, kz);
        r2 = ivp_pow(x2, e2, kz);
    }
};
// state-array access through the map (component-major SoA: a[gi(c) * B + j])
template <class MAP>
IVP_HD double map_ld(const double *arr, int c, size_t B, uint32_t j) { return MAP::own(c) ? arr[(size_t)MAP::gi(c) * B + j] : 0.0; }
template <class MAP>
IVP_HD void map_st(double *arr, int c, size_t B, uint32_t j, double v) { if (MAP::own(c)) arr[(size_t)MAP::gi(c) * B + j] = v; }

// ------------------------------------------------------------------------------------------------
// Per-lane state
// ------------------------------------------------------------------------------------------------
template <int N, int P>
struct Lane {
    double y[N], k1[N], p[P > 0 ? P : 1];
    double x, h, facold, hlamb, xend, x0, posneg, hmax;
    uint32_t flags;
    int32_t status;
    uint32_t d_nfev, d_nstep, d_naccpt, d_nrejct;  // this chunk's increments
    uint32_t budget;      // attempts left before `steps.total > nmax` (saturating)
    uint32_t acc_small;   // min(naccpt, 2): `steps.accepted > 1` test (dopri5.rs:455)
    bool over;            // nmax already exceeded on entry
    uint64_t kz;          // opaque zero for KC() in the pinned-coefficient variant (IVP_HOIST = 2), else unused
    // DefaultSolOut registers (FULL kernels)
    int32_t next_idx, n_filled;
    uint32_t n_log, n_seg;
    double t_last;
    // one-pass step log (IvpKArgs.log_pool): the trajectory's column in the wave's current page
    // (page offset << 18 | (arena cols - 1) << 12 | (cols - 1) << 6 | column), the slots of it that hold a record, the slot
    // the next record of this attempt goes to
    uint64_t log_seg;
    uint32_t log_bits, log_slot;
    // deferred t_eval sampling (flavour 3): the next t_eval point of the trajectory (NaN: none left), so that an attempt whose
    // step holds no sample decides that from registers -- a load from the grid in every attempt sat on every wave's path
    double t_next;
};

// The t_eval grid of trajectory j: the batch's shared grid, or -- every reference solve_ivp() call has its own
// Options.t_eval (options.rs:75-123) -- its own slice of a CSR grid.
struct EvalGrid { const double *t; int32_t n; };
IVP_HD EvalGrid so_grid(const IvpKArgs &a, uint32_t j)
{
    if (a.teval_off != nullptr) {
        const unsigned long long lo = a.teval_off[j];
        return EvalGrid{a.t_eval + lo, (int32_t)(a.teval_off[j + 1] - lo)};
    }
    return EvalGrid{a.t_eval, a.n_eval};
}

template <class R>
IVP_HD void lane_load(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L, bool rk23, int full)
{
    constexpr int N = R::N, P = R::P;
    using MAP = typename OutMap<R>::type;
    const size_t B = a.B;
#pragma unroll
    for (int c = 0; c < N; ++c) { L.y[c] = map_ld<MAP>(a.y, c, B, j); L.k1[c] = map_ld<MAP>(a.k1, c, B, j); }
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    L.x = a.x[j];
    L.h = a.h[j];
    L.facold = a.facold[j];
    L.hlamb = a.hlamb[j];
    L.x0 = a.t0[(size_t)j * a.t0_stride];
    L.xend = a.t1[(size_t)j * a.t1_stride];
    L.posneg = rs_signum(L.xend - L.x0);
    L.flags = a.flags[j];
    L.status = IVP_RUNNING;
    L.d_nfev = L.d_nstep = L.d_naccpt = L.d_nrejct = 0;
    const uint64_t nstep0 = a.nstep[j];
    const uint64_t nacc0 = a.naccpt[j];
    if (rk23) {   // `steps.total >= nmax` (rk23.rs:191)
        L.over = nstep0 >= a.nmax;
        const uint64_t left = L.over ? 0 : a.nmax - nstep0;
        L.budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    } else {      // `steps.total > nmax` (dopri5.rs:268)
        L.over = nstep0 > a.nmax;
        const uint64_t left = L.over ? 0 : a.nmax - nstep0;
        L.budget = left > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)left;
    }
    L.acc_small = nacc0 > 2 ? 2u : (uint32_t)nacc0;
    // DefaultSolOut registers: only what the kernel flavour reads (a field that is neither loaded nor stored costs no
    // register across the attempt loop; the flavours that record are register-bound: BASELINE C3 runs 5 waves per SIMD at
    // 95 VGPRs and 4 at 97)
    L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    if (full == 1 || full == 3) { L.next_idx = a.next_idx[j]; L.n_filled = a.n_filled[j]; L.n_seg = a.n_seg[j]; }
    if (full == 1 || full == 2) { L.n_log = a.n_log[j]; L.t_last = a.t_last[j]; }
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0;
    L.t_next = 0.0;
    if (full == 3) {
        const EvalGrid grid = so_grid(a, j);
        L.t_next = L.next_idx < grid.n ? grid.t[L.next_idx] : u2d(0x7FF8000000000000ull);
    }
}

template <class R>
IVP_HD void lane_store(const IvpKArgs &a, uint32_t j, const Lane<R::N, R::P> &L, int full)
{
    constexpr int N = R::N;
    using MAP = typename OutMap<R>::type;
    const size_t B = a.B;
#pragma unroll
    for (int c = 0; c < N; ++c) { map_st<MAP>(a.y, c, B, j, L.y[c]); map_st<MAP>(a.k1, c, B, j, L.k1[c]); }
    a.x[j] = L.x;
    a.h[j] = L.h;
    a.facold[j] = L.facold;
    a.hlamb[j] = L.hlamb;
    a.flags[j] = L.flags;
    a.status[j] = L.status;
    a.nfev[j] += L.d_nfev;
    a.nstep[j] += L.d_nstep;
    a.naccpt[j] += L.d_naccpt;
    a.nrejct[j] += L.d_nrejct;
    if (full == 1 || full == 3) { a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_seg[j] = L.n_seg; }
    if (full == 1 || full == 2) { a.n_log[j] = L.n_log; a.t_last[j] = L.t_last; }
}

// ------------------------------------------------------------------------------------------------
// Where a step's dense-output coefficients live while the device DefaultSolOut consumes them.
// DOPRI5 / RK23 / RK4 / BDF: a VGPR array.  DOP853: 8 blocks x N components on top of 10 stage vectors do not fit the
// register file (8 x 6 x 2 = 96 VGPRs for CR3BP: 256 VGPRs, occupancy 1 and 400 B of scratch per lane), so the block is
// STAGED THROUGH LDS: NC doubles per lane, laid out [coefficient][lane] (a wave's access to one coefficient is one
// conflict-free 512-byte row; every lane reads only what it wrote, so there is no barrier), 4 KB x N per wave.
// The consumers (interpolate, so_sample, so_events, so_collect_dense) are templates over the container: anything with
// operator[] -- a plain pointer / array, or ContStage.  On the host (tests/host_emul) ContStage is a plain array.
// ------------------------------------------------------------------------------------------------
template <int NC>
struct ContStage {
#if defined(__HIP_DEVICE_COMPILE__)
    double *b;
    __device__ __forceinline__ ContStage()
    {
        // one row of IVP_WAVE doubles per coefficient, addressed by threadIdx.x alone: the kernels that instantiate this are
        // launched with one-wavefront workgroups (rk_kernels.hip / rk_coop.h: block(IVP_WAVE); a wider block would alias lanes).
        // LDS cost: NC x 512 B per wave = 32 KB at N = 8 -- 5 waves per CU of the 160 KB, below the VGPR limit of these kernels.
        __shared__ double ivp_cont_lds[NC * IVP_WAVE];
        __builtin_assume(blockDim.x == IVP_WAVE);
        b = ivp_cont_lds + threadIdx.x;
    }
    __device__ __forceinline__ double operator[](int c) const { return b[c * IVP_WAVE]; }
    __device__ __forceinline__ void set(int c, double v) const { b[c * IVP_WAVE] = v; }
#else
    mutable double s[NC];
    double operator[](int c) const { return s[c]; }
    void set(int c, double v) const { s[c] = v; }
#endif
};
template <int NC>
struct ContRegs {   // the same interface on a register array
    mutable double s[NC];
    IVP_HD double operator[](int c) const { return s[c]; }
    IVP_HD void set(int c, double v) const { s[c] = v; }
};
template <bool B, class T, class F> struct ivp_cond { using type = T; };
template <class T, class F> struct ivp_cond<false, T, F> { using type = F; };
IVP_HD bool ivp_isnull(const double *p) { return p == nullptr; }
template <int NC>
IVP_HD bool ivp_isnull(const ContStage<NC> &) { return false; }
template <int NC>
IVP_HD bool ivp_isnull(const ContRegs<NC> &) { return false; }

// ------------------------------------------------------------------------------------------------
// Dense-output polynomials (dopri5.rs:467-478, dop853.rs:659-670, rk23.rs:313-321)
// ------------------------------------------------------------------------------------------------
enum { M_RK23 = 0, M_DOPRI5 = 1, M_DOP853 = 2, M_RK4 = 3, M_RADAU = 4, M_BDF = 5 };   // Method order, options.rs:14-27
template <int M> struct NCoef { enum { v = (M == M_DOPRI5) ? 5 : (M == M_DOP853) ? 8 : (M == M_BDF) ? 7 : 4 }; };

// BDF dense output (bdf.rs:618-656); cont is per-state blocks [D0, D1..D5, order]
template <int N, class CP>
IVP_HD void bdf_interpolate(double xi, double *yi, const CP &cont, double xold, double h)
{
    if (h == 0.0) return;
    double ordf = rint(cont[6]);   // order is an exact small integer: round() == rint()
    ordf = ordf < 1.0 ? 1.0 : (ordf > 5.0 ? 5.0 : ordf);
    const int order = (int)ordf;
    const double x_new = xold + h;
    double pk[5];
    double prev = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double denom = h * ((double)k + 1.0);
        const double t_shift = x_new - h * (double)k;
        const double xf = (xi - t_shift) / denom;
        pk[k] = k == 0 ? xf : prev * xf;
        prev = pk[k];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sum = cont[i * 7];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < order) sum += cont[i * 7 + 1 + k] * pk[k];
        yi[i] = sum;
    }
}

template <int M, int N, class CP>
IVP_HD void interpolate(double xi, double *yi, const CP &cont, double xold, double h)
{
    if constexpr (M == M_BDF) {
        bdf_interpolate<N>(xi, yi, cont, xold, h);
    } else if constexpr (M == M_DOPRI5) {
        const double theta = (xi - xold) / h;
        const double theta1 = 1.0 - theta;
#pragma unroll
        for (int i = 0; i < N; ++i)
            yi[i] = IVP_MA(cont[i], theta, IVP_MA(cont[N + i], theta1, IVP_MA(cont[2 * N + i], theta, IVP_MA(cont[3 * N + i], theta1, cont[4 * N + i]))));
    } else if constexpr (M == M_DOP853) {
        const double s = (xi - xold) / h;
        const double s1 = 1.0 - s;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double conpar = IVP_MA(cont[4 * N + i], s, IVP_MA(cont[5 * N + i], s1, IVP_MA(cont[6 * N + i], s, cont[7 * N + i])));
            yi[i] = IVP_MA(cont[i], s, IVP_MA(cont[N + i], s1, IVP_MA(cont[2 * N + i], s, IVP_MA(cont[3 * N + i], s1, conpar))));
        }
    } else if constexpr (M == M_RK4) {   // rk4.rs:229-244
        const double t = (xi - xold) / h;
        const double t2 = t * t;
        const double t3 = t2 * t;
        const double h00 = 2.0 * t3 - 3.0 * t2 + 1.0;
        const double h10 = t3 - 2.0 * t2 + t;
        const double h01 = -2.0 * t3 + 3.0 * t2;
        const double h11 = t3 - t2;
#pragma unroll
        for (int i = 0; i < N; ++i)
            yi[i] = IVP_LC(h00, cont[i], h10 * h, cont[N + i], h01, cont[3 * N + i], h11 * h, cont[2 * N + i]);
    } else {
        const double xc = (xi - xold) / h;
        const double x2 = xc * xc;
        const double x3 = x2 * xc;
#pragma unroll
        for (int i = 0; i < N; ++i)
            yi[i] = IVP_MA(cont[i], h, IVP_LC(cont[N + i], xc, cont[2 * N + i], x2, cont[3 * N + i], x3));
    }
}

// ------------------------------------------------------------------------------------------------
// DefaultSolOut on the device (src/solve/solout.rs:127-431 without events): dense-segment
// collection, t_eval sampling, accepted-step recording with first_step enforcement.
// `cont == nullptr` is the initial callback (interpolant None).
// ------------------------------------------------------------------------------------------------
template <int M, int N, int P, class MAP = IdMap<N>>
IVP_HD void so_emit_eval(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, int32_t ti, const double *yv)
{
    const size_t B = a.B;
    const size_t k = (size_t)L.n_filled;
    if (a.teval_off != nullptr) {   // per-trajectory grids: time-major CSR records like Solution.y (Vec<Vec<f64>>)
        const size_t q = (size_t)a.teval_off[j] + (size_t)j * a.teval_extra + k;
#pragma unroll
        for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_eval[q * MAP::NT + MAP::gi(c)] = yv[c];
        if (a.eval_idx) a.eval_idx[q] = ti;
    } else {
#pragma unroll
        for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_eval[(k * MAP::NT + MAP::gi(c)) * B + j] = yv[c];
        if (a.eval_idx) a.eval_idx[k * B + j] = ti;
    }
    L.n_filled += 1;
}
// ---- one-pass step log: wave pages (layout and rationale: ivp_kargs.h) ----
// a lane's "segment": page offset (doubles) << 23 | (slots - 1) << 18 | (arena cols - 1) << 12 | (cols - 1) << 6 | column
#define IVP_SEG_BASE(seg) ((size_t)((seg) >> 23))
#define IVP_SEG_SLOTS(seg) ((size_t)(((seg) >> 18) & 0x1Fu) + 1u)
#define IVP_SEG_ACOLS(seg) ((size_t)(((seg) >> 12) & 0x3Fu) + 1u)
#define IVP_SEG_COLS(seg) ((size_t)(((seg) >> 6) & 0x3Fu) + 1u)
#define IVP_SEG_COL(seg) ((size_t)((seg) & 0x3Fu))
// doubles of a page of `cols` columns and `slots` slots: header, column headers, ceil(cols / W) column groups
IVP_HD unsigned long long ivp_log_page_doubles(unsigned long long cols, unsigned long long slots, unsigned long long np1)
{
    const unsigned long long w = IVP_LOG_GROUP(np1);
    return IVP_LOG_HDR(cols) + ((((cols + w - 1u) / w) * slots * w * np1 + 15u) & ~15ull);
}
// so_log_flush: the slots of the current page column that hold a record go to its header (a column nobody recorded into
// keeps the 0 it was opened with).
template <class MAP, int N, int P>
IVP_HD void so_log_flush(const IvpKArgs &a, const Lane<N, P> &L)
{
    if (L.log_seg != IVP_NO_SEG && L.log_bits != 0u && MAP::leader())
        ((uint32_t *)(a.log_pool + IVP_SEG_BASE(L.log_seg) + 1u + 2u * IVP_SEG_COL(L.log_seg)))[2] = L.log_bits;
}
// so_log_open: every trajectory the wave is stepping gets a column of ONE fresh page of `slots` record slots.  Called at a
// point all of them reach together (top of the attempt loop, top of the init body) with the same `page_no` (the page's number
// within this launch) and `arena` (pages drawn per allocation: 1, 2 or 4).  The lanes present form the columns (ballot ->
// popcount = cols, rank among the set bits = column).  Every `arena`-th page the first of them draws the next arena with ONE
// atomicAdd on its sub-pool's counter, writes the arena's directory entry and marks its pages "not opened"; the pages in
// between follow at the arena's page stride (and may be narrower: trajectories retire).  Lanes that hold a trajectory
// together (rk_coop.h, rk_group.h) act through their leader.  Region exhausted: IVP_NO_SEG -- the records are still counted
// (n_log), the host sees IVP_ERRFLAG_LOG_OVERFLOW and falls back to the counted two-pass log.
template <class MAP, int N, int P>
IVP_HD void so_log_open(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, uint32_t slots, uint32_t page_no, uint32_t arena)
{
    so_log_flush<MAP>(a, L);
    uint64_t seg = IVP_NO_SEG;
    if (MAP::leader()) {
        constexpr unsigned long long np1 = MAP::NT + 1;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long m = __ballot(true);
        const uint32_t lane = __lane_id();
        const uint32_t first = (uint32_t)(__ffsll((long long)m) - 1);
        const uint32_t cols = (uint32_t)__popcll(m), col = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t sub = blockIdx.x & a.log_sub_mask;
#else
        const uint32_t cols = 1, col = 0, lane = 0, first = 0, sub = j & a.log_sub_mask;
#endif
        const uint32_t p_local = page_no & (arena - 1u);
        unsigned long long base = 0;
        uint32_t acols = cols;
        bool ok;
        if (p_local == 0u) {   // a new arena: `arena` pages of the width the wave has now
            const unsigned long long stride = ivp_log_page_doubles(cols, slots, np1), need = (unsigned long long)arena * stride;
            unsigned long long old = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            if (lane == first) old = atomicAdd(a.log_alloc + (size_t)sub * IVP_LOG_ALLOC_STRIDE, (1ull << 40) | need);
            old = ((unsigned long long)(uint32_t)__shfl((int)(old >> 32), (int)first) << 32) | (uint32_t)__shfl((int)(uint32_t)old, (int)first);
#else
            old = a.log_alloc[(size_t)sub * IVP_LOG_ALLOC_STRIDE];
            a.log_alloc[(size_t)sub * IVP_LOG_ALLOC_STRIDE] = old + ((1ull << 40) | need);
#endif
            const unsigned long long used = old & ((1ull << 40) - 1ull), dir_idx = old >> 40;
            ok = used + need + dir_idx + 1u <= a.log_region;   // pages grow up from the region's start, the directory down from its end
            base = (unsigned long long)sub * a.log_region + used;
            if (ok && lane == first) {
                ((unsigned long long *)a.log_pool)[(size_t)(sub + 1u) * a.log_region - 1u - (size_t)dir_idx] =
                    (base << 8) | ((unsigned long long)(cols - 1u) << 2) | (unsigned long long)(arena - 1u);
                for (uint32_t q = 1; q < arena; ++q) *(uint32_t *)(a.log_pool + (size_t)(base + q * stride)) = 0u;   // "not opened"
            }
        } else {               // the next page of the arena this lane's current page belongs to
            ok = L.log_seg != IVP_NO_SEG;
            acols = (uint32_t)IVP_SEG_ACOLS(L.log_seg);
            base = IVP_SEG_BASE(L.log_seg) + ivp_log_page_doubles(acols, slots, np1);
        }
        if (!ok) {
            ivp_flag_error(a, IVP_ERRFLAG_LOG_OVERFLOW);
        } else {
            seg = ((uint64_t)base << 23) | ((uint64_t)(slots - 1u) << 18) | ((uint64_t)(acols - 1u) << 12) | ((uint64_t)(cols - 1u) << 6) | (uint64_t)col;
            if (lane == first) {   // page header: opened -- this many columns, this many slots
                uint32_t *ph = (uint32_t *)(a.log_pool + (size_t)base);
                ph[0] = cols;
                ph[1] = slots;
            }
            uint32_t *hdr = (uint32_t *)(a.log_pool + (size_t)base + 1u + 2u * (size_t)col);
            hdr[0] = j;            // the trajectory,
            hdr[1] = L.n_log;      // its record count so far: where this column's records go in its log,
            hdr[2] = 0u;           // the slots that hold a record (so_log_flush)
        }
    }
    L.log_seg = map_bcast64<MAP>(seg);
    L.log_bits = 0;
}
// so_log_attempt: top of every step attempt of a recording kernel -- all trajectories the wave is still stepping pass here
// together with the same attempt index `it`.  An attempt records at most once in the log-only flavour (FULL == 2:
// so_log_accepted) and at most twice in the full one (first_step enforcement, solout.rs:392-421), so a page of
// IVP_LOG_SLOTS slots lasts 32 / 16 attempts.
template <class MAP, int FULL, int N, int P>
IVP_HD void so_log_attempt(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, uint32_t it)
{
    constexpr uint32_t kPerAttempt = FULL == 2 ? 1u : 2u, kAttempts = IVP_LOG_SLOTS / kPerAttempt;
    const uint32_t q = it & (kAttempts - 1u);
    if (q == 0u) {
        // pages per allocation: what this launch can use (a.chunk attempts), at most four (a wave that retires early leaves
        // the rest of its arena unopened)
        const uint32_t want = (a.chunk + kAttempts - 1u) / kAttempts;
        // (a launch shorter than a page -- chunk_attempts < 32 -- opens pages of just the slots it can use; arenas of more than
        // one page exist only for launches of more than a page, whose pages all have IVP_LOG_SLOTS slots: the gather relies on it)
        const uint32_t slots = want >= 2u ? IVP_LOG_SLOTS : (a.chunk * kPerAttempt < IVP_LOG_SLOTS ? a.chunk * kPerAttempt : IVP_LOG_SLOTS);
        so_log_open<MAP>(a, j, L, slots, it / kAttempts, want >= 4u ? 4u : (want >= 2u ? 2u : 1u));
    }
    L.log_slot = q * kPerAttempt;
}
template <int M, int N, int P, class MAP = IdMap<N>>
IVP_HD void so_push_log(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, double t, const double *yv)
{
    const size_t B = a.B;
    if (a.log_pool != nullptr) {
        // one-pass log: the record goes to this attempt's slot of the trajectory's column in the wave's current page
        // (ivp_kargs.h) -- next to the records the other trajectories of the wave write in this attempt
        if (L.log_seg != IVP_NO_SEG) {
            // column groups of W trajectories (ivp_kargs.h): record (slot, col) of group g = col / W sits at
            // body + ((g * slots + slot) * W + col % W) * (n + 1): the W records of a group and slot are one contiguous run
            constexpr size_t np1 = MAP::NT + 1, W = IVP_LOG_GROUP(np1);
            const size_t cols = IVP_SEG_COLS(L.log_seg), col = IVP_SEG_COL(L.log_seg), slots = IVP_SEG_SLOTS(L.log_seg);
            double *rec = a.log_pool + IVP_SEG_BASE(L.log_seg) + IVP_LOG_HDR(cols) + (((col / W) * slots + (size_t)L.log_slot) * W + (col % W)) * np1;
            if (MAP::leader()) rec[0] = t;
#pragma unroll
            for (int c = 0; c < N; ++c) if (MAP::own(c)) rec[1 + MAP::gi(c)] = yv[c];
            L.log_bits |= 1u << L.log_slot;
        }
        L.log_slot += 1;
    } else if (a.log_off != nullptr) {
        // CSR log (two-pass count / fill): record k of trajectory j lives at log_off[j] + k, memory = sum of the counts
        const unsigned long long lo = a.log_off[j], cap = a.log_off[j + 1] - lo;
        if (L.n_log < cap) {
            const size_t q = (size_t)(lo + L.n_log);
            a.t_log[q] = t;
#pragma unroll
            for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_log[q * MAP::NT + MAP::gi(c)] = yv[c];
        }
    } else if (L.n_log < a.max_log) {
        const size_t k = L.n_log;
        a.t_log[k * B + j] = t;
#pragma unroll
        for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_log[(k * MAP::NT + MAP::gi(c)) * B + j] = yv[c];
    }
    L.n_log += 1;
    L.t_last = t;
}

template <int M, int N, int P, class MAP = IdMap<N>, class CP>
IVP_HD void so_collect_dense(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, double xold, double x,
                             const CP &cont, double h, double ixold)
{
    constexpr int NC = NCoef<M>::v * N;
    constexpr int NCT = NCoef<M>::v * MAP::NT;   // coefficient blocks are [coef][component] (cont.rs:16-28)
    const size_t B = a.B;
    // dense collection, solout.rs:141-146
    if (a.collect_dense && x != xold && !ivp_isnull(cont) && h != 0.0) {
        if (L.n_seg < a.max_log) {
            const size_t k = L.n_seg;
            if (M == M_BDF) {   // per-state blocks [D0, D1..D5, order] (cont.rs:44-51): local (state i, slot s) -> global state
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (MAP::own(c / 7)) a.seg_cont[(k * NCT + (size_t)MAP::gi(c / 7) * 7 + (size_t)(c % 7)) * B + j] = cont[c];
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (MAP::own(c % N)) a.seg_cont[(k * NCT + (size_t)(c / N) * MAP::NT + MAP::gi(c % N)) * B + j] = cont[c];
            }
            a.seg_xold[k * B + j] = ixold;
            a.seg_h[k * B + j] = h;
        }
        L.n_seg += 1;
    }
}

// t_eval sampling / accepted-step recording (solout.rs:344-428). `xold` is the callback's first argument,
// `ixold`/`h` the interpolant's own anchor (StepInterpolant.xold/.h): identical for the RK methods, different for
// BDF (bdf.rs:518-519).
template <int M, int N, int P, class MAP = IdMap<N>, class CP>
IVP_HD void so_sample(const IvpKArgs &a, uint32_t j, Lane<N, P> &L, double xold, double x,
                      const double *y, const CP &cont, double h, double ixold)
{
    const double tol = 1e-12;
    (void)tol;
    double yi[N];
    if (a.n_eval >= 0) {  // Mode 1, solout.rs:344-386
        int32_t i = L.next_idx;
        const EvalGrid grid = so_grid(a, j);
        const int32_t ne = grid.n;
        if (fabs(xold - x) <= tol) {
            while (i < ne && fabs(grid.t[i] - x) <= tol) { so_emit_eval<M, N, P, MAP>(a, j, L, i, y); ++i; }
        } else if (x > xold) {
            while (i < ne && grid.t[i] <= x + tol) {
                const double te = grid.t[i];
                if (te >= xold - tol) { interpolate<M, N>(te, yi, cont, ixold, h); so_emit_eval<M, N, P, MAP>(a, j, L, i, yi); }
                ++i;
            }
        } else {
            while (i < ne && grid.t[i] >= x - tol) {
                const double te = grid.t[i];
                if (te <= xold + tol) { interpolate<M, N>(te, yi, cont, ixold, h); so_emit_eval<M, N, P, MAP>(a, j, L, i, yi); }
                ++i;
            }
        }
        L.next_idx = i;
    } else if (a.t_log != nullptr) {  // Mode 2, solout.rs:387-428
        if (a.has_first_step) {
            if (!(L.flags & IVP_F_FIRSTOUT) && fabs(xold - x) > tol) {
                const double direction = rs_signum(x - xold);
                const double target = L.x0 + direction * a.first_step;
                if (direction * (x - target) >= -tol) {
                    if (!ivp_isnull(cont)) {
                        interpolate<M, N>(target, yi, cont, ixold, h);
                        so_push_log<M, N, P, MAP>(a, j, L, target, yi);
                        L.flags |= IVP_F_FIRSTOUT;
                    }
                    if (fabs(x - target) > tol) so_push_log<M, N, P, MAP>(a, j, L, x, y);
                }
                return;
            }
        }
        if (L.n_log == 0 || fabs(L.t_last - x) > tol) so_push_log<M, N, P, MAP>(a, j, L, x, y);
    }
}


// One root of event function i in the step [xold, x] (solout.rs:214-289): Brent's method on the step interpolant (xtol
// 2e-12, rtol eps, <= 100 iterations).  g_prev / g_cur are the event values at the two ends; et / ymid receive the event point.
template <int M, class R, class YP, class CP>
IVP_HD void so_event_root(int i, double xold, double x, double g_prev, double g_cur, const double *y, const YP &yold, const CP &cont,
                          double h, double ixold, const double *p, double &et, double (&ymid)[R::N])
{
    constexpr int N = R::N, NE = R::NE > 0 ? R::NE : 1;
    const double XTOL = 2e-12, RTOL = 2.220446049250313e-16;
    double ea = xold, eb = x, fa = g_prev, fb = g_cur;
    double gmid[NE];
    if (fabs(fa) <= XTOL) {
        et = ea;
#pragma unroll
        for (int c = 0; c < N; ++c) ymid[c] = yold[c];
    } else if (fabs(fb) <= XTOL) {
        et = eb;
#pragma unroll
        for (int c = 0; c < N; ++c) ymid[c] = y[c];
    } else {
        double ec = ea, fc = fa, ed = eb - ea, ee = ed;
#pragma unroll 1
        for (int it = 0; it < 100; ++it) {
            if (fb * fc > 0.0) { ec = ea; fc = fa; ed = eb - ea; ee = ed; }
            if (fabs(fc) < fabs(fb)) { ea = eb; eb = ec; ec = ea; fa = fb; fb = fc; fc = fa; }
            const double tol1 = 2.0 * RTOL * fabs(eb) + 0.5 * XTOL;
            const double xm = 0.5 * (ec - eb);
            if (fabs(xm) <= tol1 || fb == 0.0) break;
            if (fabs(ee) >= tol1 && fabs(fa) > fabs(fb)) {
                double sq, pp, qq;
                if (ea == ec) {
                    sq = fb / fa;
                    pp = 2.0 * xm * sq;
                    qq = 1.0 - sq;
                } else {
                    const double q_val = fa / fc, rr = fb / fc;
                    sq = fb / fa;
                    pp = sq * (2.0 * xm * q_val * (q_val - rr) - (eb - ea) * (rr - 1.0));
                    qq = (q_val - 1.0) * (rr - 1.0) * (sq - 1.0);
                }
                if (qq > 0.0) pp = -pp; else qq = -qq;
                if (2.0 * pp < fmin(3.0 * xm * qq - fabs(tol1 * qq), fabs(ee * qq))) { ee = ed; ed = pp / qq; }
                else { ed = xm; ee = ed; }
            } else { ed = xm; ee = ed; }
            ea = eb; fa = fb;
            if (fabs(ed) > tol1) eb += ed;
            else eb += xm > 0.0 ? tol1 : -tol1;
            interpolate<M, N>(eb, ymid, cont, ixold, h);
            R::events(eb, ymid, gmid, p);
            double fnew = gmid[0];
#pragma unroll
            for (int q = 1; q < NE; ++q) fnew = (q == i) ? gmid[q] : fnew;
            fb = fnew;
        }
        interpolate<M, N>(eb, ymid, cont, ixold, h);
        et = eb;
    }
}

// did event function values g_prev -> g_cur cross zero in the watched direction? (solout.rs:186-199)
IVP_HD bool so_event_crossed(int dir, double g_prev, double g_cur)
{
    if (dir == 0) return (g_prev <= 0.0 && g_cur >= 0.0) || (g_prev >= 0.0 && g_cur <= 0.0);
    if (dir > 0) return g_prev < 0.0 && g_cur >= 0.0;
    return g_prev > 0.0 && g_cur <= 0.0;
}

// DEFERRED event refinement (IvpKArgs.evd_rec != nullptr; the host asks for it only when NO event is terminal, the method
// is an explicit one and the problem runs in the thread-per-trajectory / lane-cooperative kernels).  A root search in one
// lane stalls the other 63 of its wavefront (BASELINE C2 with one event function: 0.9 % of the steps hold a crossing, 44 %
// of the wave-attempts waited for a Brent iteration: 2.9 -> 4.8 ms).  Without terminal events nothing the integration
// does depends on the roots, so the stepping kernel only NOTES a step with a crossing -- its interpolant, the event values
// at both ends and the output slot of every crossing (event_hits counts on as before) -- and event_kernel_t finds the
// roots afterwards, one lane per noted step, with the very same so_event_root.  Record (SoA over the batch like every
// other array): evd_rec[(q * F + f) * B + j], F = 4 + 3 NE + n + NCoef n fields: xold, x, interpolant anchor, h,
// slot[NE] (-1: no crossing, or beyond max_events: counted, not stored), g_prev[NE], g_cur[NE], y[n], cont[NCoef n].
// At most NE * max_events noted steps per trajectory (each fills at least one slot).
template <int M, int NT, int NE>
struct EvdRec { enum { SLOT = 4, GPREV = 4 + NE, GCUR = 4 + 2 * NE, Y = 4 + 3 * NE, CONT = 4 + 3 * NE + NT, F = 4 + 3 * NE + NT + NCoef<M>::v * NT }; };

template <int M, class R, class CP>
IVP_HD void so_events_note(const IvpKArgs &a, uint32_t j, double xold, double x, const double *y, const CP &cont, double h, double ixold,
                           const double *g_curr)
{
    constexpr int N = R::N, NE = R::NE > 0 ? R::NE : 1, NC = NCoef<M>::v * N;
    using MAP = typename OutMap<R>::type;
    using F = EvdRec<M, MAP::NT, NE>;
    const size_t B = a.B;
    double slot[NE], g_prev[NE];
    bool store = false;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        g_prev[i] = a.prev_event[(size_t)i * B + j];
        const int dir = a.ev_direction_dev ? a.ev_direction_dev[i] : a.ev_direction[i < 4 ? i : 3];
        slot[i] = -1.0;
        if (so_event_crossed(dir, g_prev[i], g_curr[i])) {
            const uint32_t k = a.n_ev[(size_t)i * B + j];
            a.n_ev[(size_t)i * B + j] = k + 1;   // event_hits
            if (k < a.max_events) { slot[i] = (double)k; store = true; }
        }
    }
    if (store) {
        const uint32_t q = a.evd_cnt[j];
        if (q < a.evd_cap) {
            double *rec = a.evd_rec + (size_t)q * (size_t)F::F * B + j;
            if (MAP::leader()) {
                rec[0] = xold; rec[B] = x; rec[2 * B] = ixold; rec[3 * B] = h;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    rec[(size_t)(F::SLOT + i) * B] = slot[i]; rec[(size_t)(F::GPREV + i) * B] = g_prev[i]; rec[(size_t)(F::GCUR + i) * B] = g_curr[i];
                }
            }
#pragma unroll
            for (int c = 0; c < N; ++c) if (MAP::own(c)) rec[(size_t)(F::Y + MAP::gi(c)) * B] = y[c];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (MAP::own(c % N)) rec[(size_t)(F::CONT + (c / N) * MAP::NT + MAP::gi(c % N)) * B] = cont[c];
        }
        a.evd_cnt[j] = q + 1;
    }
}

// one noted step (thread-per-trajectory: plain functor R, all n components in this lane)
template <int M, class R>
IVP_HD void so_events_deferred_body(const IvpKArgs &a, uint32_t j, uint32_t q)
{
    constexpr int N = R::N, P = R::P, NE = R::NE > 0 ? R::NE : 1, NC = NCoef<M>::v * N;
    using F = EvdRec<M, N, NE>;
    const size_t B = a.B;
    const double *rec = a.evd_rec + (size_t)q * (size_t)F::F * B + j;
    const double xold = rec[0], x = rec[B], ixold = rec[2 * B], h = rec[3 * B];
    double p[P > 0 ? P : 1], y[N];
    ContRegs<NC> cont;
#pragma unroll
    for (int c = 0; c < P; ++c) p[c] = a.params[c * B + j];
#pragma unroll
    for (int c = 0; c < N; ++c) y[c] = rec[(size_t)(F::Y + c) * B];
#pragma unroll
    for (int c = 0; c < NC; ++c) cont.set(c, rec[(size_t)(F::CONT + c) * B]);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const double ks = rec[(size_t)(F::SLOT + i) * B];
        if (ks >= 0.0) {
            double et, ymid[N];
            // the explicit methods pass their coefficient block as yold too (cont[0 .. n) = y at xold; dopri5 / dop853 / rk23 / rk4_attempt)
            so_event_root<M, R>(i, xold, x, rec[(size_t)(F::GPREV + i) * B], rec[(size_t)(F::GCUR + i) * B], y, cont, cont, h, ixold, p, et, ymid);
            const size_t k = (size_t)ks;
            a.t_events[((size_t)i * a.max_events + k) * B + j] = et;
#pragma unroll
            for (int c = 0; c < N; ++c) a.y_events[(((size_t)i * a.max_events + k) * N + c) * B + j] = ymid[c];
        }
    }
}

// Event detection (solout.rs:158-331): zero crossings of R::events between the previous and the current accepted
// point, refined with Brent's method on the step interpolant (so_event_root), processed in chronological order; a
// terminal event appends its point to the output and interrupts the integration.
template <int M, class R, class YP, class CP>
IVP_HD bool so_events(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L, double xold, double x,
                      const double *y, const YP &yold, const CP &cont, double h, double ixold)
{
    constexpr int N = R::N, P = R::P, NE = R::NE > 0 ? R::NE : 1;
    using MAP = typename OutMap<R>::type;
    const size_t B = a.B;
    double g_curr[NE];
    R::events(x, y, g_curr, L.p);
    if (ivp_isnull(cont)) {   // initial callback: DefaultSolOut.yold is still empty (solout.rs:163-164)
#pragma unroll
        for (int i = 0; i < NE; ++i) a.prev_event[(size_t)i * B + j] = g_curr[i];
        return false;
    }
    if constexpr (M != M_BDF) {
        if (a.evd_rec != nullptr) {
            so_events_note<M, R>(a, j, xold, x, y, cont, h, ixold, g_curr);
#pragma unroll
            for (int i = 0; i < NE; ++i) a.prev_event[(size_t)i * B + j] = g_curr[i];
            return false;
        }
    }
    double det_t[NE], det_y[NE][N];
    int det_i[NE];
    int ndet = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const double g_prev = a.prev_event[(size_t)i * B + j], g_cur = g_curr[i];
        const int dir = a.ev_direction_dev ? a.ev_direction_dev[i] : a.ev_direction[i < 4 ? i : 3];
        if (so_event_crossed(dir, g_prev, g_cur)) {
            double ymid[N];
            double et;
            so_event_root<M, R>(i, xold, x, g_prev, g_cur, y, yold, cont, h, ixold, L.p, et, ymid);
            // append to the detected list (ndet is a run-time count: guarded static slots)
#pragma unroll
            for (int q = 0; q < NE; ++q)
                if (q == ndet) {
                    det_t[q] = et; det_i[q] = i;
#pragma unroll
                    for (int c = 0; c < N; ++c) det_y[q][c] = ymid[c];
                }
            ndet += 1;
        }
    }
    // stable insertion sort by time (ascending forward, descending backward), solout.rs:297-303
    const bool forward = x > xold;
#pragma unroll
    for (int u = 1; u < NE; ++u) {
        bool moving = u < ndet;
#pragma unroll
        for (int v = u; v > 0; --v) {
            const bool sw = moving && (forward ? (det_t[v] < det_t[v - 1]) : (det_t[v] > det_t[v - 1]));
            if (sw) {
                const double tt = det_t[v]; det_t[v] = det_t[v - 1]; det_t[v - 1] = tt;
                const int ti = det_i[v]; det_i[v] = det_i[v - 1]; det_i[v - 1] = ti;
#pragma unroll
                for (int c = 0; c < N; ++c) { const double ty = det_y[v][c]; det_y[v][c] = det_y[v - 1][c]; det_y[v - 1][c] = ty; }
            } else moving = false;
        }
    }
    bool interrupt = false;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        if (u < ndet && !interrupt) {
            const int i = det_i[u];
            const uint32_t k = a.n_ev[(size_t)i * B + j];
            if (k < a.max_events) {
                a.t_events[((size_t)i * a.max_events + k) * B + j] = det_t[u];
#pragma unroll
                for (int c = 0; c < N; ++c)
                    if (MAP::own(c)) a.y_events[(((size_t)i * a.max_events + k) * MAP::NT + MAP::gi(c)) * B + j] = det_y[u][c];
            }
            a.n_ev[(size_t)i * B + j] = k + 1;   // event_hits
            const uint32_t term = a.ev_terminal_dev ? a.ev_terminal_dev[i] : a.ev_terminal[i < 4 ? i : 3];
            if (term != 0 && k + 1 >= term) {
                // the terminal event point is appended to Solution.t / Solution.y (solout.rs:316-319)
                if (a.n_eval >= 0) { so_emit_eval<M, N, P, MAP>(a, j, L, -1, det_y[u]); if (a.t_term) a.t_term[j] = det_t[u]; }
                else if (a.t_log != nullptr) so_push_log<M, N, P, MAP>(a, j, L, det_t[u], det_y[u]);
                interrupt = true;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) a.prev_event[(size_t)i * B + j] = g_curr[i];
    return interrupt;
}

// DefaultSolOut::solout on the device. Returns true for ControlFlag::Interrupt (terminal event).
template <int M, class R, class YP, class CP>
IVP_HD bool solout_full(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L, double xold, double x,
                        const double *y, const YP &yold, const CP &cont, double h, double ixold)
{
    using MAP = typename OutMap<R>::type;
    so_collect_dense<M, R::N, R::P, MAP>(a, j, L, xold, x, cont, h, ixold);
    if constexpr (R::NE > 0) {
        if (so_events<M, R>(a, j, L, xold, x, y, yold, cont, h, ixold)) return true;
    }
    so_sample<M, R::N, R::P, MAP>(a, j, L, xold, x, y, cont, h, ixold);
    return false;
}

// FULL == 2, the LOG-ONLY flavour of the kernels: the reference's default output contract and nothing else -- no t_eval, no
// first_step, no dense-output collection, no event functions (the host checks all four before it picks this flavour).
// DefaultSolOut then reduces to "record (x, y) of every accepted step unless it repeats the last record" (solout.rs:422-427),
// which needs no interpolant: the stepping kernels skip the dense-output coefficients altogether and keep the register
// budget (and the occupancy) of the end-state kernels.
template <class R>
IVP_HD void so_log_accepted(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L, double x, const double *y)
{
    using MAP = typename OutMap<R>::type;
    if (L.n_log == 0 || fabs(L.t_last - x) > 1e-12) so_push_log<0, R::N, R::P, MAP>(a, j, L, x, y);
}

// FULL == 3, the DEFERRED-SAMPLING flavour (DOP853 with Options.t_eval and nothing else asked of DefaultSolOut: no event
// functions, no dense-output collection).  DOP853's interpolant costs three more right-hand-side evaluations per step that is
// sampled (dop853.rs:474-560); in lock-step a WAVE pays them whenever ANY of its 64 trajectories has a sample in the step --
// with 128 samples per ~450 steps that is nearly every step (BASELINE C3: 12.8 -> 23.8 ms).  Here the stepping kernel only
// NOTES a sampled step -- (x, h, y), the t_eval indices it consumes and where its samples go (so_defer_samples) -- and
// a second kernel with one lane per noted step (dop853_sample_body) redoes that step from (x, h, y, k1) with the very same
// expressions, evaluates the dense stages and the polynomial and writes the samples: the same arithmetic per sampled step
// as the reference, no union over a wave, every lane busy.  Record layout: def_rec[(k * F + f) * B + j], F = n + 4 fields
// x, h, {first, end} t_eval index (two u32 in one 8-byte field), first output position, y[n]; L.n_seg counts a trajectory's
// noted steps.  k1 = f(x, y) is not stored: the stepping kernel's k1 IS R::ode(x, y) -- the last stage of the previous step,
// dop853.rs:443, or init's f(x0, y0) -- so the sample kernel evaluates it again (same function, same inputs, same bits) and
// the notes of BASELINE C3 with 128 samples shrink from 9.2 GB to 6.1 GB.
template <int N, class MAP>
IVP_HD void so_emit_eval_at(const IvpKArgs &a, uint32_t j, size_t k, int32_t ti, const double *yv)
{
    const size_t B = a.B;
    if (a.teval_off != nullptr) {   // per-trajectory grids: time-major CSR records like Solution.y (Vec<Vec<f64>>)
        const size_t q = (size_t)a.teval_off[j] + (size_t)j * a.teval_extra + k;
#pragma unroll
        for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_eval[q * MAP::NT + MAP::gi(c)] = yv[c];
        if (a.eval_idx) a.eval_idx[q] = ti;
    } else {
#pragma unroll
        for (int c = 0; c < N; ++c) if (MAP::own(c)) a.y_eval[(k * MAP::NT + MAP::gi(c)) * B + j] = yv[c];
        if (a.eval_idx) a.eval_idx[k * B + j] = ti;
    }
}
template <class R>
IVP_HD void so_defer_samples(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L, double xold, double x,
                             const double *yold, const double *ynew, double h)
{
    constexpr int N = R::N;
    using MAP = typename OutMap<R>::type;
    const double tol = 1e-12;
    const EvalGrid grid = so_grid(a, j);
    const int32_t ne = grid.n;
    int32_t i = L.next_idx;
    if (fabs(xold - x) <= tol) {   // solout.rs:349-356: no interpolant involved
        while (i < ne && fabs(grid.t[i] - x) <= tol) { so_emit_eval<M_DOP853, N, R::P, MAP>(a, j, L, i, ynew); ++i; }
        L.next_idx = i;
        return;
    }
    // no sample in this step (the usual case): decided from the cached next point (a NaN -- grid exhausted -- fails both tests)
    if (x > xold ? !(L.t_next <= x + tol) : !(L.t_next >= x - tol)) return;
    const int32_t i0 = i;
    int32_t cnt = 0;
    if (x > xold) {                // the tests of so_sample (solout.rs:357-385), without the interpolation
        while (i < ne && grid.t[i] <= x + tol) { if (grid.t[i] >= xold - tol) ++cnt; ++i; }
    } else {
        while (i < ne && grid.t[i] >= x - tol) { if (grid.t[i] <= xold + tol) ++cnt; ++i; }
    }
    if (cnt > 0) {
        if (L.n_seg < a.def_cap) {
            const size_t B = a.B;
            double *rec = a.def_rec + (size_t)L.n_seg * (size_t)(MAP::NT + 4) * B + j;
            if (MAP::leader()) {
                rec[0] = xold; rec[B] = h; rec[2 * B] = u2d((uint64_t)(uint32_t)i0 | ((uint64_t)(uint32_t)i << 32)); rec[3 * B] = (double)L.n_filled;
            }
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (MAP::own(c)) rec[(size_t)(4 + MAP::gi(c)) * B] = yold[c];
        }
        L.n_seg += 1;
        L.n_filled += cnt;
    }
    L.next_idx = i;
    L.t_next = i < ne ? grid.t[i] : u2d(0x7FF8000000000000ull);
}

// does the accepted step [xold, xph] need dense coefficients? (lazy DOP853 dense stages)
template <int N, int P>
IVP_HD bool so_needs_dense(const IvpKArgs &a, uint32_t j, const Lane<N, P> &L, double xold, double xph)
{
    const double tol = 1e-12;
    if (a.collect_dense) return true;
    if (a.n_eval >= 0) {
        const EvalGrid grid = so_grid(a, j);
        if (L.next_idx >= grid.n) return false;
        const double te = grid.t[L.next_idx];
        return xph > xold ? (te <= xph + tol) : (te >= xph - tol);
    }
    return a.t_log != nullptr && a.has_first_step && !(L.flags & IVP_F_FIRSTOUT);
}

// Solution.njev / Solution.nlu of an explicit method are 0 (tests/test_ivp.py:214-216): written by the init kernel when
// the caller asked for them, instead of two fill kernels per solve
IVP_HD void so_zero_implicit_counters(const IvpKArgs &a, uint32_t j)
{
    if (a.njev) a.njev[j] = 0;
    if (a.nlu) a.nlu[j] = 0;
}

// ------------------------------------------------------------------------------------------------
// hinit (src/methods/mod.rs:217-281)
// ------------------------------------------------------------------------------------------------
template <class R>
IVP_HD double hinit(const IvpKArgs &a, double x, const double *y, double posneg, const double *f0,
                    const double *p, int iord, double hmax)
{
    constexpr int N = R::N;
    double t_dnf[N], t_dny[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double sk = IVP_MA(NormOps<R>::atol(a, i), NormOps<R>::rtol(a, i), fabs(y[i]));
        t_dnf[i] = (f0[i] / sk) * (f0[i] / sk);
        t_dny[i] = (y[i] / sk) * (y[i] / sk);
    }
    const double dnf = NormOps<R>::sum(t_dnf), dny = NormOps<R>::sum(t_dny);
    double h;
    if (dnf <= 1e-10 || dny <= 1e-10) h = 1.0e-6;
    else h = sqrt(dny / dnf) * 0.01;
    if (h > fabs(hmax)) h = fabs(hmax);
    h = fabs(h) * rs_signum(posneg);
    double y1[N], f1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, f0[i]);
    R::ode(x + h, y1, f1, p);
    double t_der2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double sk = IVP_MA(NormOps<R>::atol(a, i), NormOps<R>::rtol(a, i), fabs(y[i]));
        const double df = (f1[i] - f0[i]) / sk;
        t_der2[i] = df * df;
    }
    double der2 = NormOps<R>::sum(t_der2);
    der2 = sqrt(der2) / fabs(h);
    const double der12 = fmax(fabs(der2), sqrt(dnf));
    double h1;
    if (der12 <= 1.0e-15) h1 = fmax(1.0e-6, fabs(h) * 1.0e-3);
    else h1 = ivp_pow(0.01 / der12, 1.0 / (double)iord);
    const double hf = fmin(fmin(fmin(fabs(h), 100.0 * fabs(h)), h1), fabs(hmax));  // mod.rs:279 as written
    return fabs(hf) * rs_signum(posneg);
}

// ------------------------------------------------------------------------------------------------
// init body: the part of solve_ivp()/XXX::solve() before the main loop
//   (solve_ivp.rs:110-145 zero-interval short-circuit; dopri5.rs:201-263 / dop853.rs:196-269 /
//    rk23.rs:138-186: f0, hinit or first_step, initial SolOut call)
// ------------------------------------------------------------------------------------------------
template <int M, class R, int FULL>
IVP_HD int32_t init_body(const IvpKArgs &a, uint32_t j)
{
    constexpr int N = R::N, P = R::P;
    using MAP = typename OutMap<R>::type;
    const size_t B = a.B;
    Lane<N, P> L;
#pragma unroll
    for (int c = 0; c < N; ++c) L.y[c] = map_ld<MAP>(a.y0, c, B, j);
#pragma unroll
    for (int c = 0; c < P; ++c) L.p[c] = a.params[c * B + j];
    L.x0 = a.t0[(size_t)j * a.t0_stride];
    L.xend = a.t1[(size_t)j * a.t1_stride];
    L.x = L.x0;
    L.flags = 0;
    L.facold = 1e-4;
    L.hlamb = 0.0;
    L.next_idx = 0; L.n_filled = 0; L.n_log = 0; L.n_seg = 0; L.t_last = 0.0;
    L.log_seg = IVP_NO_SEG; L.log_bits = 0; L.log_slot = 0;
    if (FULL && a.log_pool != nullptr) so_log_open<MAP>(a, j, L, 2u, 0u, 1u);   // the initial callback records at most twice
    uint64_t nfev = 0;

    if (fabs(L.xend - L.x0) < 1e-15) {  // solve_ivp.rs:110-145
#pragma unroll
        for (int c = 0; c < N; ++c) { map_st<MAP>(a.y, c, B, j, L.y[c]); map_st<MAP>(a.k1, c, B, j, 0.0); }
        if (FULL) {
            if (a.n_eval >= 0) {
                const EvalGrid grid = so_grid(a, j);
                for (int32_t i = 0; i < grid.n; ++i)
                    if (fabs(grid.t[i] - L.x0) < 1e-12) so_emit_eval<M, N, P, MAP>(a, j, L, i, L.y);
            } else if (a.t_log != nullptr) {
                so_push_log<M, N, P, MAP>(a, j, L, L.x0, L.y);
            }
            if (a.collect_dense && a.max_log > 0) {  // ContinuousOutput::constant, cont.rs:32-64
                constexpr int NC = NCoef<M>::v * N;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (MAP::own(c % N)) a.seg_cont[((size_t)(c / N) * MAP::NT + MAP::gi(c % N)) * B + j] = c < N ? L.y[c] : 0.0;
                a.seg_xold[j] = L.x0;
                a.seg_h[j] = 1e-15;
                L.n_seg = 1;
            }
            a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_log[j] = L.n_log;
            a.n_seg[j] = L.n_seg; a.t_last[j] = L.t_last;
            if (a.log_pool != nullptr) so_log_flush<MAP>(a, L);
        }
        a.x[j] = L.x0; a.h[j] = 0.0; a.facold[j] = L.facold; a.hlamb[j] = 0.0; a.flags[j] = 0;
        a.status[j] = 0;
        a.nfev[j] = 0; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; so_zero_implicit_counters(a, j);
        return 0;
    }

    if (L.x0 != L.x0 || L.xend != L.xend) {
        // A NaN interval end makes every comparison of the reference's step loop false: it never terminates
        // (with the default max_steps = None).  A GPU lane must retire: StepSizeTooSmall, nothing integrated.
#pragma unroll
        for (int c = 0; c < N; ++c) { map_st<MAP>(a.y, c, B, j, L.y[c]); map_st<MAP>(a.k1, c, B, j, 0.0); }
        a.x[j] = L.x0; a.h[j] = 0.0; a.facold[j] = L.facold; a.hlamb[j] = 0.0; a.flags[j] = 0;
        a.status[j] = 3;
        a.nfev[j] = 0; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; so_zero_implicit_counters(a, j);
        if (FULL) { a.next_idx[j] = 0; a.n_filled[j] = 0; a.n_log[j] = 0; a.n_seg[j] = 0; a.t_last[j] = 0.0; }
        return 3;
    }
    L.posneg = rs_signum(L.xend - L.x0);
    // h_max: dopri5.rs:180 keeps the sign of max_step, dop853.rs:172-175 / rk23.rs:135 take |.|
    if (a.has_max_step) L.hmax = (M == M_DOPRI5) ? a.max_step : fabs(a.max_step);
    else L.hmax = fabs(L.xend - L.x);

    R::ode(L.x, L.y, L.k1, L.p);
    if constexpr (M == M_RK4) {
        // solve_ivp.rs:184-196: h = first_step or (xend - x0)/100; RK4::solve rejects h == 0 or a sign that
        // does not match xend - x0 (rk4.rs:81-87, Err(InvalidStepSize)): flagged for the host, lane parked.
        // The initial evaluation is not counted in nfev (rk4.rs:119).
        L.h = a.has_first_step ? a.first_step : (L.xend - L.x0) / 100.0;
        if (L.h == 0.0 || rs_signum(L.h) != L.posneg) {
            ivp_flag_error(a, IVP_ERRFLAG_INVALID_STEP);
#pragma unroll
            for (int c = 0; c < N; ++c) { map_st<MAP>(a.y, c, B, j, L.y[c]); map_st<MAP>(a.k1, c, B, j, 0.0); }
            a.x[j] = L.x0; a.h[j] = L.h; a.facold[j] = 0.0; a.hlamb[j] = 0.0; a.flags[j] = 0;
            a.status[j] = 0;
            a.nfev[j] = 0; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; so_zero_implicit_counters(a, j);
            if (FULL) { a.next_idx[j] = 0; a.n_filled[j] = 0; a.n_log[j] = 0; a.n_seg[j] = 0; a.t_last[j] = 0.0; }
            return 0;
        }
    } else if (a.has_first_step) {
        nfev += 1;
        L.h = fabs(a.first_step) * L.posneg;
    } else {
        nfev += 2;  // f(x0, y0) and hinit's Euler probe
        L.h = hinit<R>(a, L.x, L.y, L.posneg, L.k1, L.p, M == M_DOPRI5 ? 5 : (M == M_DOP853 ? 8 : 3), L.hmax);
    }
    if (FULL == 1) (void)solout_full<M, R>(a, j, L, L.x, L.x, L.y, (const double *)L.y, (const double *)nullptr, 0.0, L.x);
    else if (FULL == 2) so_log_accepted<R>(a, j, L, L.x, L.y);
    else if (FULL == 3) so_sample<M, N, P, MAP>(a, j, L, L.x, L.x, L.y, (const double *)nullptr, 0.0, L.x);   // xold == x: no interpolant

#pragma unroll
    for (int c = 0; c < N; ++c) { map_st<MAP>(a.y, c, B, j, L.y[c]); map_st<MAP>(a.k1, c, B, j, L.k1[c]); }
    a.x[j] = L.x; a.h[j] = L.h; a.facold[j] = L.facold; a.hlamb[j] = 0.0; a.flags[j] = L.flags;
    a.status[j] = IVP_RUNNING;
    a.nfev[j] = nfev; a.nstep[j] = 0; a.naccpt[j] = 0; a.nrejct[j] = 0; so_zero_implicit_counters(a, j);
    if (FULL) {
        a.next_idx[j] = L.next_idx; a.n_filled[j] = L.n_filled; a.n_log[j] = L.n_log;
        a.n_seg[j] = L.n_seg; a.t_last[j] = L.t_last;
        if (a.log_pool != nullptr) so_log_flush<MAP>(a, L);
    }
    return IVP_RUNNING;
}

// Shared tail of an accepted/rejected attempt for DOPRI5 and DOP853 (dopri5.rs:434-460, dop853.rs:626-652).
#define IVP_STIFF_BOOKKEEPING(THRESH)                                                              \
    {                                                                                              \
        uint32_t iasti = (L.flags >> IVP_F_IASTI_SHIFT) & 0xFu;                                    \
        uint32_t nonstiff = (L.flags >> IVP_F_NONSTIFF_SHIFT) & 0xFu;                              \
        if (stden > 0.0) L.hlamb = fabs(h) * sqrt(stnum / stden);                                  \
        if (L.hlamb > (THRESH)) {                                                                  \
            nonstiff = 0;                                                                          \
            iasti += 1;                                                                            \
            if (iasti == 15) stiff_break = true;                                                   \
        } else {                                                                                   \
            nonstiff += 1;                                                                         \
            if (nonstiff == 6) iasti = 0;                                                          \
        }                                                                                          \
        L.flags = (L.flags & ~((0xFu << IVP_F_IASTI_SHIFT) | (0xFu << IVP_F_NONSTIFF_SHIFT))) |    \
                  ((iasti & 0xFu) << IVP_F_IASTI_SHIFT) | ((nonstiff & 0xFu) << IVP_F_NONSTIFF_SHIFT); \
    }

// stiffness-test cadence: `steps.accepted % nstiff == 0` (dopri5.rs:364) kept as a counter in flags bits 12..31.
// CTL = false: the struct default nstiff = 1000.  CTL = true: a.ctl_nstiff; a stiff_test >= 2^20 falls back to the
// 64-bit modulo on the stored accepted count.
template <bool CTL>
IVP_HD bool stiff_tick(const IvpKArgs &a, uint32_t j, uint32_t &flags, uint32_t d_naccpt)
{
    bool hit;
    if (!CTL || a.ctl_nstiff < (1ull << 20)) {
        uint32_t sc = flags >> IVP_F_STIFFCTR_SHIFT;
        sc += 1;
        hit = sc == (CTL ? (uint32_t)a.ctl_nstiff : 1000u);
        if (hit) sc = 0;
        flags = (flags & ((1u << IVP_F_STIFFCTR_SHIFT) - 1u)) | (sc << IVP_F_STIFFCTR_SHIFT);
    } else {
        hit = ((a.naccpt[j] + d_naccpt) % a.ctl_nstiff) == 0;
    }
    return hit || ((flags >> IVP_F_IASTI_SHIFT) & 0xFu) > 0;
}
// controller field: the compile-time struct default (CTL = false: solve_ivp()'s path) or the run-time value of a
// direct method call (CTL = true)
#define IVP_CTL(field, dflt) (CTL ? a.field : KC(dflt))

// ------------------------------------------------------------------------------------------------
// DOPRI5 attempt (dopri5.rs:266-461).  Returns false when the trajectory retired.
// ------------------------------------------------------------------------------------------------
template <class R, int FULL, bool CTL = false>
IVP_HD bool dopri5_attempt(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L)
{
    KC_SCOPE_KZ(L.kz)
    constexpr int N = R::N;
    // tableau, dopri5.rs:482-520
    constexpr double C2 = 0.2, C3 = 0.3, C4 = 0.8, C5 = 8.0 / 9.0;
    constexpr double A21 = 0.2, A31 = 3.0 / 40.0, A32 = 9.0 / 40.0;
    constexpr double A41 = 44.0 / 45.0, A42 = -56.0 / 15.0, A43 = 32.0 / 9.0;
    constexpr double A51 = 19372.0 / 6561.0, A52 = -25360.0 / 2187.0, A53 = 64448.0 / 6561.0, A54 = -212.0 / 729.0;
    constexpr double A61 = 9017.0 / 3168.0, A62 = -355.0 / 33.0, A63 = 46732.0 / 5247.0, A64 = 49.0 / 176.0, A65 = -5103.0 / 18656.0;
    constexpr double A71 = 35.0 / 384.0, A73 = 500.0 / 1113.0, A74 = 125.0 / 192.0, A75 = -2187.0 / 6784.0, A76 = 11.0 / 84.0;
    constexpr double E1 = 71.0 / 57600.0, E3 = -71.0 / 16695.0, E4 = 71.0 / 1920.0, E5 = -17253.0 / 339200.0, E6 = 22.0 / 525.0, E7 = -1.0 / 40.0;
    constexpr double D1 = -12715105075.0 / 11282082432.0, D3 = 87487479700.0 / 32700410799.0, D4 = -10690763975.0 / 1880347072.0,
                     D5 = 701980252875.0 / 199316789632.0, D6 = -1453857185.0 / 822651844.0, D7 = 69997945.0 / 29380423.0;
    // struct defaults (dopri5.rs:34-72); solve_ivp overrides only max_step/first_step/max_steps
    constexpr double d_uround = 2.3e-16, d_safety = 0.9, d_beta = 0.04;
    constexpr double d_facc1 = 1.0 / 0.2, d_facc2 = 1.0 / 10.0;
    constexpr double d_expo1 = 0.2 - d_beta * 0.75;

    if (L.over || L.d_nstep > L.budget) { L.status = 2; return false; }               // NeedLargerNMax
    double h = L.h;
    const double x = L.x;
    if (KC(0.1) * fabs(h) <= fabs(x) * IVP_CTL(ctl_uround, d_uround)) { L.status = 3; return false; }             // StepSizeTooSmall
    bool last = (L.flags & IVP_F_LAST) != 0;
    if ((x + KC(1.01) * h - L.xend) * L.posneg > 0.0) { h = L.xend - x; last = true; }
    L.d_nstep += 1;

    const double *y = L.y, *k1 = L.k1, *p = L.p;
    double k2[N], k3[N], k4[N], k5[N], k6[N], y1[N];
{ const double cA21 = KC(A21);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h * cA21, k1[i]);
}
    R::ode(x + KC(C2) * h, y1, k2, p);
{ const double cA31 = KC(A31), cA32 = KC(A32);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA31, k1[i], cA32, k2[i]));
}
    R::ode(x + KC(C3) * h, y1, k3, p);
{ const double cA41 = KC(A41), cA42 = KC(A42), cA43 = KC(A43);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA41, k1[i], cA42, k2[i], cA43, k3[i]));
}
    R::ode(x + KC(C4) * h, y1, k4, p);
{ const double cA51 = KC(A51), cA52 = KC(A52), cA53 = KC(A53), cA54 = KC(A54);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA51, k1[i], cA52, k2[i], cA53, k3[i], cA54, k4[i]));
}
    R::ode(x + KC(C5) * h, y1, k5, p);
{ const double cA61 = KC(A61), cA62 = KC(A62), cA63 = KC(A63), cA64 = KC(A64), cA65 = KC(A65);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA61, k1[i], cA62, k2[i], cA63, k3[i], cA64, k4[i], cA65, k5[i]));
}
    const double xph = x + h;
    R::ode(xph, y1, k6, p);
{ const double cA71 = KC(A71), cA73 = KC(A73), cA74 = KC(A74), cA75 = KC(A75), cA76 = KC(A76);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA71, k1[i], cA73, k3[i], cA74, k4[i], cA75, k5[i], cA76, k6[i]));
}
    R::ode(xph, y1, k2, p);  // k7 -> k2 (FSAL)
    L.d_nfev += 6;

    double cont[FULL == 1 ? 5 * N : 1];
    if (FULL == 1) {
{ const double cD1 = KC(D1), cD3 = KC(D3), cD4 = KC(D4), cD5 = KC(D5), cD6 = KC(D6), cD7 = KC(D7);
#pragma unroll
        for (int i = 0; i < N; ++i)
            cont[4 * N + i] = h * IVP_LC(cD1, k1[i], cD3, k3[i], cD4, k4[i], cD5, k5[i], cD6, k6[i], cD7, k2[i]);
}
    }
{ const double cE1 = KC(E1), cE3 = KC(E3), cE4 = KC(E4), cE5 = KC(E5), cE6 = KC(E6), cE7 = KC(E7);
#pragma unroll
    for (int i = 0; i < N; ++i)
        k4[i] = IVP_LC(cE1, k1[i], cE3, k3[i], cE4, k4[i], cE5, k5[i], cE6, k6[i], cE7, k2[i]) * h;
}
    double t_err[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double sk = IVP_MA(NormOps<R>::atol(a, i), NormOps<R>::rtol(a, i), fmax(fabs(y[i]), fabs(y1[i])));
        t_err[i] = (k4[i] / sk) * (k4[i] / sk);
    }
    double err = NormOps<R>::sum(t_err);
    err = sqrt(err / (double)NormOps<R>::NT);

    double fac11, facb;
    NormOps<R>::pow2(err, IVP_CTL(ctl_expo1, d_expo1), L.facold, CTL ? a.ctl_beta : d_beta, fac11, facb, IVP_KZ_ARG);
    double fac = fac11 / facb;
    fac = fmax(IVP_CTL(ctl_facc2, d_facc2), fmin(IVP_CTL(ctl_facc1, d_facc1), fac / IVP_CTL(ctl_safety, d_safety)));
    double hnew = h / fac;

    if (err <= 1.0) {
        L.facold = fmax(err, KC(1.0e-4));
        L.d_naccpt += 1;
        if (L.acc_small < 2) L.acc_small += 1;
        if (stiff_tick<CTL>(a, j, L.flags, L.d_naccpt)) {  // dopri5.rs:364-391 (rare: every 1000 accepted steps)
            double t_num[N], t_den[N];
            bool stiff_break = false;
{ const double cA61 = KC(A61), cA62 = KC(A62), cA63 = KC(A63), cA64 = KC(A64), cA65 = KC(A65);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double d1 = k2[i] - k6[i];
                const double ysti = IVP_MA(y[i], h, IVP_LC(cA61, k1[i], cA62, k2[i], cA63, k3[i], cA64, k4[i], cA65, k5[i]));
                const double d2 = y1[i] - ysti;
                t_num[i] = d1 * d1;
                t_den[i] = d2 * d2;
            }
}
            const double stnum = NormOps<R>::sum(t_num), stden = NormOps<R>::sum(t_den);
            IVP_STIFF_BOOKKEEPING(3.25)
            if (stiff_break) { L.h = h; L.status = 4; return false; }                  // ProbablyStiff
        }
        if (FULL == 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double ydiff = y1[i] - y[i];
                const double bspl = IVP_MB(h, k1[i], ydiff);
                cont[i] = y[i];
                cont[N + i] = ydiff;
                cont[2 * N + i] = bspl;
                cont[3 * N + i] = IVP_MA(ydiff, -h, k2[i]) - bspl;   // -h k7 + ydiff - bspl
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) { L.k1[i] = k2[i]; L.y[i] = y1[i]; }
        L.x = xph;
        if (FULL == 1) {   // ControlFlag::Interrupt => status = UserInterrupt, break before h = hnew (dopri5.rs:418-421)
            if (solout_full<M_DOPRI5, R>(a, j, L, x, xph, L.y, cont, cont, h, x)) { L.h = h; L.status = 1; return false; }
        } else if (FULL == 2) so_log_accepted<R>(a, j, L, xph, L.y);
        if (last) { L.h = hnew; L.status = 0; return false; }                          // Success
        if (fabs(hnew) > fabs(L.hmax)) hnew = L.posneg * fabs(L.hmax);
        if (L.flags & IVP_F_REJECT) { hnew = L.posneg * fmin(fabs(hnew), fabs(h)); L.flags &= ~IVP_F_REJECT; }
    } else {
        hnew = h / fmin(IVP_CTL(ctl_facc1, d_facc1), fac11 / IVP_CTL(ctl_safety, d_safety));
        L.flags |= IVP_F_REJECT;
        if (L.acc_small > 1) L.d_nrejct += 1;
        last = false;
    }
    L.flags = last ? (L.flags | IVP_F_LAST) : (L.flags & ~IVP_F_LAST);
    L.h = hnew;
    return true;
}

// Hairer's DOP853 coefficients (dop853.rs:674-848): shared by the stepping attempt and by the sample-parallel t_eval kernel
// (dop853_sample_body), which redoes a recorded step with the very same expressions
namespace dop853_tab {
constexpr double C2 = 0.526001519587677318785587544488e-01, C3 = 0.789002279381515978178381316732e-01,
                 C4 = 0.118350341907227396726757197510e+00, C5 = 0.281649658092772603273242802490e+00,
                 C6 = 0.333333333333333333333333333333e+00, C7 = 0.25e+00, C8 = 0.307692307692307692307692307692e+00,
                 C9 = 0.651282051282051282051282051282e+00, C10 = 0.6e+00, C11 = 0.857142857142857142857142857142e+00,
                 C14 = 0.1e+00, C15 = 0.2e+00, C16 = 7.777777777777778e-1;
constexpr double A21 = 5.26001519587677318785587544488e-2;
constexpr double A31 = 1.97250569845378994544595329183e-2, A32 = 5.91751709536136983633785987549e-2;
constexpr double A41 = 2.95875854768068491816892993775e-2, A43 = 8.87627564304205475450678981324e-2;
constexpr double A51 = 2.41365134159266685502369798665e-1, A53 = -8.84549479328286085344864962717e-1, A54 = 9.24834003261792003115737966543e-1;
constexpr double A61 = 3.7037037037037037037037037037e-2, A64 = 1.70828608729473871279604482173e-1, A65 = 1.25467687566822425016691814123e-1;
constexpr double A71 = 3.7109375e-2, A74 = 1.70252211019544039314978060272e-1, A75 = 6.02165389804559606850219397283e-2, A76 = -1.7578125e-2;
constexpr double A81 = 3.70920001185047927108779319836e-2, A84 = 1.70383925712239993810214054705e-1, A85 = 1.07262030446373284651809199168e-1,
                 A86 = -1.53194377486244017527936158236e-2, A87 = 8.27378916381402288758473766002e-3;
constexpr double A91 = 6.24110958716075717114429577812e-1, A94 = -3.36089262944694129406857109825e0, A95 = -8.68219346841726006818189891453e-1,
                 A96 = 2.75920996994467083049415600797e1, A97 = 2.01540675504778934086186788979e1, A98 = -4.34898841810699588477366255144e1;
constexpr double A101 = 4.77662536438264365890433908527e-1, A104 = -2.48811461997166764192642586468e0, A105 = -5.90290826836842996371446475743e-1,
                 A106 = 2.12300514481811942347288949897e1, A107 = 1.52792336328824235832596922938e1, A108 = -3.32882109689848629194453265587e1,
                 A109 = -2.03312017085086261358222928593e-2;
constexpr double A111 = -9.3714243008598732571704021658e-1, A114 = 5.18637242884406370830023853209e0, A115 = 1.09143734899672957818500254654e0,
                 A116 = -8.14978701074692612513997267357e0, A117 = -1.85200656599969598641566180701e1, A118 = 2.27394870993505042818970056734e1,
                 A119 = 2.49360555267965238987089396762e0, A1110 = -3.0467644718982195003823669022e0;
constexpr double A121 = 2.27331014751653820792359768449e0, A124 = -1.05344954667372501984066689879e1, A125 = -2.00087205822486249909675718444e0,
                 A126 = -1.79589318631187989172765950534e1, A127 = 2.79488845294199600508499808837e1, A128 = -2.85899827713502369474065508674e0,
                 A129 = -8.87285693353062954433549289258e0, A1210 = 1.23605671757943030647266201528e1, A1211 = 6.43392746015763530355970484046e-1;
constexpr double B1 = 5.42937341165687622380535766363e-2, B6 = 4.45031289275240888144113950566e0, B7 = 1.89151789931450038304281599044e0,
                 B8 = -5.8012039600105847814672114227e0, B9 = 3.1116436695781989440891606237e-1, B10 = -1.52160949662516078556178806805e-1,
                 B11 = 2.01365400804030348374776537501e-1, B12 = 4.47106157277725905176885569043e-2;
constexpr double BH1 = 0.244094488188976377952755905512e+00, BH2 = 0.733846688281611857341361741547e+00, BH3 = 0.220588235294117647058823529412e-01;
constexpr double ER1 = 0.1312004499419488073250102996e-01, ER6 = -0.1225156446376204440720569753e+01, ER7 = -0.4957589496572501915214079952e+00,
                 ER8 = 0.1664377182454986536961530415e+01, ER9 = -0.3503288487499736816886487290e+00, ER10 = 0.3341791187130174790297318841e+00,
                 ER11 = 0.8192320648511571246570742613e-01, ER12 = -0.2235530786388629525884427845e-01;
constexpr double A141 = 5.61675022830479523392909219681e-2, A147 = 2.53500210216624811088794765333e-1, A148 = -2.46239037470802489917441475441e-1,
                 A149 = -1.24191423263816360469010140626e-1, A1410 = 1.5329179827876569731206322685e-1, A1411 = 8.20105229563468988491666602057e-3,
                 A1412 = 7.56789766054569976138603589584e-3, A1413 = -8.298e-3;
constexpr double A151 = 3.18346481635021405060768473261e-2, A156 = 2.83009096723667755288322961402e-2, A157 = 5.35419883074385676223797384372e-2,
                 A158 = -5.49237485713909884646569340306e-2, A1511 = -1.08347328697249322858509316994e-4, A1512 = 3.82571090835658412954920192323e-4,
                 A1513 = -3.40465008687404560802977114492e-4, A1514 = 1.41312443674632500278074618366e-1;
constexpr double A161 = -4.28896301583791923408573538692e-1, A166 = -4.69762141536116384314449447206e0, A167 = 7.68342119606259904184240953878e0,
                 A168 = 4.06898981839711007970213554331e0, A169 = 3.56727187455281109270669543021e-1, A1613 = -1.39902416515901462129418009734e-3,
                 A1614 = 2.9475147891527723389556272149e0, A1615 = -9.15095847217987001081870187138e0;
constexpr double D41 = -0.84289382761090128651353491142e+01, D46 = 0.56671495351937776962531783590e+00, D47 = -0.30689499459498916912797304727e+01,
                 D48 = 0.23846676565120698287728149680e+01, D49 = 0.21170345824450282767155149946e+01, D410 = -0.87139158377797299206789907490e+00,
                 D411 = 0.22404374302607882758541771650e+01, D412 = 0.63157877876946881815570249290e+00, D413 = -0.88990336451333310820698117400e-01,
                 D414 = 0.18148505520854727256656404962e+02, D415 = -0.91946323924783554000451984436e+01, D416 = -0.44360363875948939664310572000e+01;
constexpr double D51 = 0.10427508642579134603413151009e+02, D56 = 0.24228349177525818288430175319e+03, D57 = 0.16520045171727028198505394887e+03,
                 D58 = -0.37454675472269020279518312152e+03, D59 = -0.22113666853125306036270938578e+02, D510 = 0.77334326684722638389603898808e+01,
                 D511 = -0.30674084731089398182061213626e+02, D512 = -0.93321305264302278729567221706e+01, D513 = 0.15697238121770843886131091075e+02,
                 D514 = -0.31139403219565177677282850411e+02, D515 = -0.93529243588444783865713862664e+01, D516 = 0.35816841486394083752465898540e+02;
constexpr double D61 = 0.19985053242002433820987653617e+02, D66 = -0.38703730874935176555105901742e+03, D67 = -0.18917813819516756882830838328e+03,
                 D68 = 0.52780815920542364900561016686e+03, D69 = -0.11573902539959630126141871134e+02, D610 = 0.68812326946963000169666922661e+01,
                 D611 = -0.10006050966910838403183860980e+01, D612 = 0.77771377980534432092869265740e+00, D613 = -0.27782057523535084065932004339e+01,
                 D614 = -0.60196695231264120758267380846e+02, D615 = 0.84320405506677161018159903784e+02, D616 = 0.11992291136182789328035130030e+02;
constexpr double D71 = -0.25693933462703749003312586129e+02, D76 = -0.15418974869023643374053993627e+03, D77 = -0.23152937917604549567536039109e+03,
                 D78 = 0.35763911791061412378285349910e+03, D79 = 0.93405324183624310003907691704e+02, D710 = -0.37458323136451633156875139351e+02,
                 D711 = 0.10409964950896230045147246184e+03, D712 = 0.29840293426660503123344363579e+02, D713 = -0.43533456590011143754432175058e+02,
                 D714 = 0.96324553959188282948394950600e+02, D715 = -0.39177261675615439165231486172e+02, D716 = -0.14972683625798562581422125276e+03;
}  // namespace dop853_tab

// ------------------------------------------------------------------------------------------------
// One noted step of the deferred-sampling flavour (FULL == 3), redone by ONE lane: the stages, the eighth-order
// solution and the new derivative exactly as dop853_attempt computes them (same expressions, same order: identical bits),
// then the dense-output coefficients with their three extra stages (dop853.rs:474-592) and the samples the step holds
// (solout.rs:357-385), written to the positions the stepping kernel reserved for them.
// ------------------------------------------------------------------------------------------------
template <class R>
IVP_HD void dop853_sample_body(const IvpKArgs &a, uint32_t j, uint32_t kd)
{
    KC_SCOPE
    constexpr int N = R::N, P = R::P;
    using namespace dop853_tab;
    const size_t B = a.B;
    const double *rec = a.def_rec + (size_t)kd * (size_t)(N + 4) * B + j;
    const double x = rec[0], h = rec[B];
    const uint64_t idx = d2u(rec[2 * B]);
    const int32_t i0 = (int32_t)(uint32_t)idx, i1 = (int32_t)(uint32_t)(idx >> 32);
    size_t pos = (size_t)rec[3 * B];
    double y[N], k1[N], p[P > 0 ? P : 1];
#pragma unroll
    for (int c = 0; c < N; ++c) y[c] = rec[(size_t)(4 + c) * B];
#pragma unroll
    for (int c = 0; c < P; ++c) p[c] = a.params[c * B + j];
    R::ode(x, y, k1, p);   // the step's first stage: what the stepping kernel held as k1 (see the record layout above)
    double k2[N], k3[N], k4[N], k5[N], k6[N], k7[N], k8[N], k9[N], k10[N], y1[N];
{ const double cA21 = KC(A21);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h * cA21, k1[i]);
}
    R::ode(x + KC(C2) * h, y1, k2, p);
{ const double cA31 = KC(A31), cA32 = KC(A32);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA31, k1[i], cA32, k2[i]));
}
    R::ode(x + KC(C3) * h, y1, k3, p);
{ const double cA41 = KC(A41), cA43 = KC(A43);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA41, k1[i], cA43, k3[i]));
}
    R::ode(x + KC(C4) * h, y1, k4, p);
{ const double cA51 = KC(A51), cA53 = KC(A53), cA54 = KC(A54);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA51, k1[i], cA53, k3[i], cA54, k4[i]));
}
    R::ode(x + KC(C5) * h, y1, k5, p);
{ const double cA61 = KC(A61), cA64 = KC(A64), cA65 = KC(A65);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA61, k1[i], cA64, k4[i], cA65, k5[i]));
}
    R::ode(x + KC(C6) * h, y1, k6, p);
{ const double cA71 = KC(A71), cA74 = KC(A74), cA75 = KC(A75), cA76 = KC(A76);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA71, k1[i], cA74, k4[i], cA75, k5[i], cA76, k6[i]));
}
    R::ode(x + KC(C7) * h, y1, k7, p);
{ const double cA81 = KC(A81), cA84 = KC(A84), cA85 = KC(A85), cA86 = KC(A86), cA87 = KC(A87);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA81, k1[i], cA84, k4[i], cA85, k5[i], cA86, k6[i], cA87, k7[i]));
}
    R::ode(x + KC(C8) * h, y1, k8, p);
{ const double cA91 = KC(A91), cA94 = KC(A94), cA95 = KC(A95), cA96 = KC(A96), cA97 = KC(A97), cA98 = KC(A98);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA91, k1[i], cA94, k4[i], cA95, k5[i], cA96, k6[i], cA97, k7[i], cA98, k8[i]));
}
    R::ode(x + KC(C9) * h, y1, k9, p);
{ const double cA101 = KC(A101), cA104 = KC(A104), cA105 = KC(A105), cA106 = KC(A106), cA107 = KC(A107), cA108 = KC(A108), cA109 = KC(A109);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA101, k1[i], cA104, k4[i], cA105, k5[i], cA106, k6[i], cA107, k7[i], cA108, k8[i], cA109, k9[i]));
}
    R::ode(x + KC(C10) * h, y1, k10, p);
{ const double cA111 = KC(A111), cA114 = KC(A114), cA115 = KC(A115), cA116 = KC(A116), cA117 = KC(A117), cA118 = KC(A118), cA119 = KC(A119), cA1110 = KC(A1110);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA111, k1[i], cA114, k4[i], cA115, k5[i], cA116, k6[i], cA117, k7[i], cA118, k8[i], cA119, k9[i], cA1110, k10[i]));
}
    R::ode(x + KC(C11) * h, y1, k2, p);
    const double xph = x + h;
{ const double cA121 = KC(A121), cA124 = KC(A124), cA125 = KC(A125), cA126 = KC(A126), cA127 = KC(A127), cA128 = KC(A128), cA129 = KC(A129), cA1210 = KC(A1210), cA1211 = KC(A1211);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA121, k1[i], cA124, k4[i], cA125, k5[i], cA126, k6[i], cA127, k7[i], cA128, k8[i], cA129, k9[i],
                                       cA1210, k10[i], cA1211, k2[i]));
}
    R::ode(xph, y1, k3, p);

{ const double cB1 = KC(B1), cB6 = KC(B6), cB7 = KC(B7), cB8 = KC(B8), cB9 = KC(B9), cB10 = KC(B10), cB11 = KC(B11), cB12 = KC(B12);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        k4[i] = IVP_LC(cB1, k1[i], cB6, k6[i], cB7, k7[i], cB8, k8[i], cB9, k9[i], cB10, k10[i], cB11, k2[i], cB12, k3[i]);
        k5[i] = IVP_MA(y[i], h, k4[i]);
    }
}
    R::ode(xph, k5, k4, p);   // the new derivative (dop853.rs:443)
    ContRegs<8 * N> cont;
{ const double cD41 = KC(D41), cD46 = KC(D46), cD47 = KC(D47), cD48 = KC(D48), cD49 = KC(D49), cD410 = KC(D410), cD411 = KC(D411), cD412 = KC(D412), cD51 = KC(D51), cD56 = KC(D56), cD57 = KC(D57), cD58 = KC(D58), cD59 = KC(D59), cD510 = KC(D510), cD511 = KC(D511), cD512 = KC(D512), cD61 = KC(D61), cD66 = KC(D66), cD67 = KC(D67), cD68 = KC(D68), cD69 = KC(D69), cD610 = KC(D610), cD611 = KC(D611), cD612 = KC(D612), cD71 = KC(D71), cD76 = KC(D76), cD77 = KC(D77), cD78 = KC(D78), cD79 = KC(D79), cD710 = KC(D710), cD711 = KC(D711), cD712 = KC(D712);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cont.set(i, y[i]);
        const double ydiff = k5[i] - y[i];
        cont.set(N + i, ydiff);
        const double bspl = IVP_MB(h, k1[i], ydiff);
        cont.set(2 * N + i, bspl);
        cont.set(3 * N + i, IVP_MS(ydiff, h, k4[i]) - bspl);
        cont.set(4 * N + i, IVP_LC(cD41, k1[i], cD46, k6[i], cD47, k7[i], cD48, k8[i], cD49, k9[i], cD410, k10[i], cD411, k2[i], cD412, k3[i]));
        cont.set(5 * N + i, IVP_LC(cD51, k1[i], cD56, k6[i], cD57, k7[i], cD58, k8[i], cD59, k9[i], cD510, k10[i], cD511, k2[i], cD512, k3[i]));
        cont.set(6 * N + i, IVP_LC(cD61, k1[i], cD66, k6[i], cD67, k7[i], cD68, k8[i], cD69, k9[i], cD610, k10[i], cD611, k2[i], cD612, k3[i]));
        cont.set(7 * N + i, IVP_LC(cD71, k1[i], cD76, k6[i], cD77, k7[i], cD78, k8[i], cD79, k9[i], cD710, k10[i], cD711, k2[i], cD712, k3[i]));
    }
}
{ const double cA141 = KC(A141), cA147 = KC(A147), cA148 = KC(A148), cA149 = KC(A149), cA1410 = KC(A1410), cA1411 = KC(A1411), cA1412 = KC(A1412), cA1413 = KC(A1413);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA141, k1[i], cA147, k7[i], cA148, k8[i], cA149, k9[i], cA1410, k10[i], cA1411, k2[i], cA1412, k3[i], cA1413, k4[i]));
}
    R::ode(x + KC(C14) * h, y1, k10, p);
{ const double cA151 = KC(A151), cA156 = KC(A156), cA157 = KC(A157), cA158 = KC(A158), cA1511 = KC(A1511), cA1512 = KC(A1512), cA1513 = KC(A1513), cA1514 = KC(A1514);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA151, k1[i], cA156, k6[i], cA157, k7[i], cA158, k8[i], cA1511, k2[i], cA1512, k3[i], cA1513, k4[i], cA1514, k10[i]));
}
    R::ode(x + KC(C15) * h, y1, k2, p);
{ const double cA161 = KC(A161), cA166 = KC(A166), cA167 = KC(A167), cA168 = KC(A168), cA169 = KC(A169), cA1613 = KC(A1613), cA1614 = KC(A1614), cA1615 = KC(A1615);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA161, k1[i], cA166, k6[i], cA167, k7[i], cA168, k8[i], cA169, k9[i], cA1613, k4[i], cA1614, k10[i], cA1615, k2[i]));
}
    R::ode(x + KC(C16) * h, y1, k3, p);
{ const double cD413 = KC(D413), cD414 = KC(D414), cD415 = KC(D415), cD416 = KC(D416), cD513 = KC(D513), cD514 = KC(D514), cD515 = KC(D515), cD516 = KC(D516), cD613 = KC(D613), cD614 = KC(D614), cD615 = KC(D615), cD616 = KC(D616), cD713 = KC(D713), cD714 = KC(D714), cD715 = KC(D715), cD716 = KC(D716);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cont.set(4 * N + i, h * IVP_NS::ivp_lc(cont[4 * N + i], cD413, k4[i], cD414, k10[i], cD415, k2[i], cD416, k3[i]));
        cont.set(5 * N + i, h * IVP_NS::ivp_lc(cont[5 * N + i], cD513, k4[i], cD514, k10[i], cD515, k2[i], cD516, k3[i]));
        cont.set(6 * N + i, h * IVP_NS::ivp_lc(cont[6 * N + i], cD613, k4[i], cD614, k10[i], cD615, k2[i], cD616, k3[i]));
        cont.set(7 * N + i, h * IVP_NS::ivp_lc(cont[7 * N + i], cD713, k4[i], cD714, k10[i], cD715, k2[i], cD716, k3[i]));
    }
}
    // the samples of this step, in t_eval order (the tests of so_sample; the stepping kernel counted with the same ones)
    const double tol = 1e-12;
    const EvalGrid grid = so_grid(a, j);
    double yi[N];
    for (int32_t i = i0; i < i1; ++i) {
        const double te = grid.t[i];
        if (xph > x ? (te >= x - tol) : (te <= x + tol)) {
            interpolate<M_DOP853, N>(te, yi, cont, x, h);
            so_emit_eval_at<N, IdMap<N>>(a, j, pos, i, yi);
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DOP853 attempt (dop853.rs:272-653)
// ------------------------------------------------------------------------------------------------
template <class R, int FULL, bool CTL = false>
IVP_HD bool dop853_attempt(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L)
{
    KC_SCOPE_KZ(L.kz)
    constexpr int N = R::N, P = R::P;
    using namespace dop853_tab;
    // struct defaults (dop853.rs:34-63)
    constexpr double d_uround = 2.3e-16, d_safety = 0.9, d_beta = 0.0;   // dop853.rs:34-63
    constexpr double d_facc1 = 1.0 / 0.333, d_facc2 = 1.0 / 6.0;
    constexpr double d_expo1 = 1.0 / 8.0 - d_beta * 0.2;

    if (L.over || L.d_nstep > L.budget) { L.status = 2; return false; }
    double h = L.h;
    const double x = L.x;
    if (KC(0.1) * fabs(h) <= fabs(x) * IVP_CTL(ctl_uround, d_uround)) { L.status = 3; return false; }
    bool last = (L.flags & IVP_F_LAST) != 0;
    if ((x + KC(1.01) * h - L.xend) * L.posneg > 0.0) { h = L.xend - x; last = true; }
    L.d_nstep += 1;

    const double *y = L.y, *k1 = L.k1, *p = L.p;
    double k2[N], k3[N], k4[N], k5[N], k6[N], k7[N], k8[N], k9[N], k10[N], y1[N];
{ const double cA21 = KC(A21);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h * cA21, k1[i]);
}
    R::ode(x + KC(C2) * h, y1, k2, p);
{ const double cA31 = KC(A31), cA32 = KC(A32);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA31, k1[i], cA32, k2[i]));
}
    R::ode(x + KC(C3) * h, y1, k3, p);
{ const double cA41 = KC(A41), cA43 = KC(A43);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA41, k1[i], cA43, k3[i]));
}
    R::ode(x + KC(C4) * h, y1, k4, p);
{ const double cA51 = KC(A51), cA53 = KC(A53), cA54 = KC(A54);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA51, k1[i], cA53, k3[i], cA54, k4[i]));
}
    R::ode(x + KC(C5) * h, y1, k5, p);
{ const double cA61 = KC(A61), cA64 = KC(A64), cA65 = KC(A65);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA61, k1[i], cA64, k4[i], cA65, k5[i]));
}
    R::ode(x + KC(C6) * h, y1, k6, p);
{ const double cA71 = KC(A71), cA74 = KC(A74), cA75 = KC(A75), cA76 = KC(A76);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA71, k1[i], cA74, k4[i], cA75, k5[i], cA76, k6[i]));
}
    R::ode(x + KC(C7) * h, y1, k7, p);
{ const double cA81 = KC(A81), cA84 = KC(A84), cA85 = KC(A85), cA86 = KC(A86), cA87 = KC(A87);
#pragma unroll
    for (int i = 0; i < N; ++i) y1[i] = IVP_MA(y[i], h, IVP_LC(cA81, k1[i], cA84, k4[i], cA85, k5[i], cA86, k6[i], cA87, k7[i]));
}
    R::ode(x + KC(C8) * h, y1, k8, p);
{ const double cA91 = KC(A91), cA94 = KC(A94), cA95 = KC(A95), cA96 = KC(A96), cA97 = KC(A97), cA98 = KC(A98);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA91, k1[i], cA94, k4[i], cA95, k5[i], cA96, k6[i], cA97, k7[i], cA98, k8[i]));
}
    R::ode(x + KC(C9) * h, y1, k9, p);
{ const double cA101 = KC(A101), cA104 = KC(A104), cA105 = KC(A105), cA106 = KC(A106), cA107 = KC(A107), cA108 = KC(A108), cA109 = KC(A109);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA101, k1[i], cA104, k4[i], cA105, k5[i], cA106, k6[i], cA107, k7[i], cA108, k8[i], cA109, k9[i]));
}
    R::ode(x + KC(C10) * h, y1, k10, p);
{ const double cA111 = KC(A111), cA114 = KC(A114), cA115 = KC(A115), cA116 = KC(A116), cA117 = KC(A117), cA118 = KC(A118), cA119 = KC(A119), cA1110 = KC(A1110);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA111, k1[i], cA114, k4[i], cA115, k5[i], cA116, k6[i], cA117, k7[i], cA118, k8[i], cA119, k9[i], cA1110, k10[i]));
}
    R::ode(x + KC(C11) * h, y1, k2, p);
    const double xph = x + h;
{ const double cA121 = KC(A121), cA124 = KC(A124), cA125 = KC(A125), cA126 = KC(A126), cA127 = KC(A127), cA128 = KC(A128), cA129 = KC(A129), cA1210 = KC(A1210), cA1211 = KC(A1211);
#pragma unroll
    for (int i = 0; i < N; ++i)
        y1[i] = IVP_MA(y[i], h, IVP_LC(cA121, k1[i], cA124, k4[i], cA125, k5[i], cA126, k6[i], cA127, k7[i], cA128, k8[i], cA129, k9[i],
                                       cA1210, k10[i], cA1211, k2[i]));
}
    R::ode(xph, y1, k3, p);
    L.d_nfev += 11;

{ const double cB1 = KC(B1), cB6 = KC(B6), cB7 = KC(B7), cB8 = KC(B8), cB9 = KC(B9), cB10 = KC(B10), cB11 = KC(B11), cB12 = KC(B12);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        k4[i] = IVP_LC(cB1, k1[i], cB6, k6[i], cB7, k7[i], cB8, k8[i], cB9, k9[i], cB10, k10[i], cB11, k2[i], cB12, k3[i]);
        k5[i] = IVP_MA(y[i], h, k4[i]);
    }
}
    double t_err[N], t_err2[N];
{ const double cBH1 = KC(BH1), cBH2 = KC(BH2), cBH3 = KC(BH3), cER1 = KC(ER1), cER6 = KC(ER6), cER7 = KC(ER7), cER8 = KC(ER8), cER9 = KC(ER9), cER10 = KC(ER10), cER11 = KC(ER11), cER12 = KC(ER12);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double sk = IVP_MA(NormOps<R>::atol(a, i), NormOps<R>::rtol(a, i), fmax(fabs(y[i]), fabs(k5[i])));
        double erri = IVP_MS(IVP_MS(IVP_MS(k4[i], cBH1, k1[i]), cBH2, k9[i]), cBH3, k3[i]);
        double q = erri / sk;
        t_err2[i] = q * q;
        erri = IVP_LC(cER1, k1[i], cER6, k6[i], cER7, k7[i], cER8, k8[i], cER9, k9[i], cER10, k10[i], cER11, k2[i], cER12, k3[i]);
        q = erri / sk;
        t_err[i] = q * q;
    }
}
    double err = NormOps<R>::sum(t_err);
    const double err2 = NormOps<R>::sum(t_err2);
    double deno = IVP_MA(err, 0.01, err2);
    if (deno <= 0.0) deno = 1.0;
    err = fabs(h) * err * sqrt(1.0 / ((double)NormOps<R>::NT * deno));

    double fac11, facb;
    NormOps<R>::pow2(err, IVP_CTL(ctl_expo1, d_expo1), L.facold, CTL ? a.ctl_beta : d_beta, fac11, facb, IVP_KZ_ARG);
    double fac = fac11 / facb;
    fac = fmax(IVP_CTL(ctl_facc2, d_facc2), fmin(IVP_CTL(ctl_facc1, d_facc1), fac / IVP_CTL(ctl_safety, d_safety)));
    double hnew = h / fac;

    if (err <= 1.0) {
        L.facold = fmax(err, KC(1.0e-4));
        L.d_naccpt += 1;
        if (L.acc_small < 2) L.acc_small += 1;
        R::ode(xph, k5, k4, p);
        L.d_nfev += 1;
        if (stiff_tick<CTL>(a, j, L.flags, L.d_naccpt)) {  // dop853.rs:447-472
            double t_num[N], t_den[N];
            bool stiff_break = false;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double d1 = k4[i] - k3[i];
                const double d2 = k5[i] - y1[i];
                t_num[i] = d1 * d1;
                t_den[i] = d2 * d2;
            }
            const double stnum = NormOps<R>::sum(t_num), stden = NormOps<R>::sum(t_den);
            IVP_STIFF_BOOKKEEPING(6.1)
            if (stiff_break) { L.h = h; L.status = 4; return false; }
        }
        // Dense output (dop853.rs:476-592).  The reference computes it on every accepted step
        // (struct default dense_output = true) and counts its 3 evaluations in nfev; the
        // coefficients feed nothing but the interpolant, so the kernel evaluates them only when
        // this step's interpolant is actually consumed, and always counts the 3 evaluations.
        L.d_nfev += 3;
        // the 8 x N dense block: staged through LDS when a lane holds more than one component (see ContStage)
        typename ivp_cond<(FULL == 1 && N >= 2), ContStage<8 * N>, ContRegs<(FULL == 1 ? 8 * N : 1)>>::type cont;
        const bool need_dense = FULL == 1 && (R::NE > 0 || so_needs_dense<N, P>(a, j, L, x, xph));
        if (FULL == 1 && need_dense) {
{ const double cD41 = KC(D41), cD46 = KC(D46), cD47 = KC(D47), cD48 = KC(D48), cD49 = KC(D49), cD410 = KC(D410), cD411 = KC(D411), cD412 = KC(D412), cD51 = KC(D51), cD56 = KC(D56), cD57 = KC(D57), cD58 = KC(D58), cD59 = KC(D59), cD510 = KC(D510), cD511 = KC(D511), cD512 = KC(D512), cD61 = KC(D61), cD66 = KC(D66), cD67 = KC(D67), cD68 = KC(D68), cD69 = KC(D69), cD610 = KC(D610), cD611 = KC(D611), cD612 = KC(D612), cD71 = KC(D71), cD76 = KC(D76), cD77 = KC(D77), cD78 = KC(D78), cD79 = KC(D79), cD710 = KC(D710), cD711 = KC(D711), cD712 = KC(D712);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                cont.set(i, y[i]);
                const double ydiff = k5[i] - y[i];
                cont.set(N + i, ydiff);
                const double bspl = IVP_MB(h, k1[i], ydiff);
                cont.set(2 * N + i, bspl);
                cont.set(3 * N + i, IVP_MS(ydiff, h, k4[i]) - bspl);
                cont.set(4 * N + i, IVP_LC(cD41, k1[i], cD46, k6[i], cD47, k7[i], cD48, k8[i], cD49, k9[i], cD410, k10[i], cD411, k2[i], cD412, k3[i]));
                cont.set(5 * N + i, IVP_LC(cD51, k1[i], cD56, k6[i], cD57, k7[i], cD58, k8[i], cD59, k9[i], cD510, k10[i], cD511, k2[i], cD512, k3[i]));
                cont.set(6 * N + i, IVP_LC(cD61, k1[i], cD66, k6[i], cD67, k7[i], cD68, k8[i], cD69, k9[i], cD610, k10[i], cD611, k2[i], cD612, k3[i]));
                cont.set(7 * N + i, IVP_LC(cD71, k1[i], cD76, k6[i], cD77, k7[i], cD78, k8[i], cD79, k9[i], cD710, k10[i], cD711, k2[i], cD712, k3[i]));
            }
}
{ const double cA141 = KC(A141), cA147 = KC(A147), cA148 = KC(A148), cA149 = KC(A149), cA1410 = KC(A1410), cA1411 = KC(A1411), cA1412 = KC(A1412), cA1413 = KC(A1413);
#pragma unroll
            for (int i = 0; i < N; ++i)
                y1[i] = IVP_MA(y[i], h, IVP_LC(cA141, k1[i], cA147, k7[i], cA148, k8[i], cA149, k9[i], cA1410, k10[i], cA1411, k2[i], cA1412, k3[i], cA1413, k4[i]));
}
            R::ode(x + KC(C14) * h, y1, k10, p);
{ const double cA151 = KC(A151), cA156 = KC(A156), cA157 = KC(A157), cA158 = KC(A158), cA1511 = KC(A1511), cA1512 = KC(A1512), cA1513 = KC(A1513), cA1514 = KC(A1514);
#pragma unroll
            for (int i = 0; i < N; ++i)
                y1[i] = IVP_MA(y[i], h, IVP_LC(cA151, k1[i], cA156, k6[i], cA157, k7[i], cA158, k8[i], cA1511, k2[i], cA1512, k3[i], cA1513, k4[i], cA1514, k10[i]));
}
            R::ode(x + KC(C15) * h, y1, k2, p);
{ const double cA161 = KC(A161), cA166 = KC(A166), cA167 = KC(A167), cA168 = KC(A168), cA169 = KC(A169), cA1613 = KC(A1613), cA1614 = KC(A1614), cA1615 = KC(A1615);
#pragma unroll
            for (int i = 0; i < N; ++i)
                y1[i] = IVP_MA(y[i], h, IVP_LC(cA161, k1[i], cA166, k6[i], cA167, k7[i], cA168, k8[i], cA169, k9[i], cA1613, k4[i], cA1614, k10[i], cA1615, k2[i]));
}
            R::ode(x + KC(C16) * h, y1, k3, p);
{ const double cD413 = KC(D413), cD414 = KC(D414), cD415 = KC(D415), cD416 = KC(D416), cD513 = KC(D513), cD514 = KC(D514), cD515 = KC(D515), cD516 = KC(D516), cD613 = KC(D613), cD614 = KC(D614), cD615 = KC(D615), cD616 = KC(D616), cD713 = KC(D713), cD714 = KC(D714), cD715 = KC(D715), cD716 = KC(D716);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                cont.set(4 * N + i, h * IVP_NS::ivp_lc(cont[4 * N + i], cD413, k4[i], cD414, k10[i], cD415, k2[i], cD416, k3[i]));
                cont.set(5 * N + i, h * IVP_NS::ivp_lc(cont[5 * N + i], cD513, k4[i], cD514, k10[i], cD515, k2[i], cD516, k3[i]));
                cont.set(6 * N + i, h * IVP_NS::ivp_lc(cont[6 * N + i], cD613, k4[i], cD614, k10[i], cD615, k2[i], cD616, k3[i]));
                cont.set(7 * N + i, h * IVP_NS::ivp_lc(cont[7 * N + i], cD713, k4[i], cD714, k10[i], cD715, k2[i], cD716, k3[i]));
            }
}
        }
        if (FULL == 3) so_defer_samples<R>(a, j, L, x, xph, y, k5, h);   // before y (= L.y) moves on
#pragma unroll
        for (int i = 0; i < N; ++i) { L.k1[i] = k4[i]; L.y[i] = k5[i]; }
        L.x = xph;
        if (FULL == 1) {
            // `cont` is passed even when its dense rows were not computed (need_dense false): every reader is guarded
            // by exactly the conditions so_needs_dense() tests
            if (solout_full<M_DOP853, R>(a, j, L, x, xph, L.y, cont, cont, h, x)) { L.h = h; L.status = 1; return false; }
        } else if (FULL == 2) so_log_accepted<R>(a, j, L, xph, L.y);
        if (last) { L.h = hnew; L.status = 0; return false; }
        if (fabs(hnew) > fabs(L.hmax)) hnew = L.posneg * fabs(L.hmax);
        if (L.flags & IVP_F_REJECT) { hnew = L.posneg * fmin(fabs(hnew), fabs(h)); L.flags &= ~IVP_F_REJECT; }
    } else {
        hnew = h / fmin(IVP_CTL(ctl_facc1, d_facc1), fac11 / IVP_CTL(ctl_safety, d_safety));
        L.flags |= IVP_F_REJECT;
        if (L.acc_small > 1) L.d_nrejct += 1;
        last = false;
    }
    L.flags = last ? (L.flags | IVP_F_LAST) : (L.flags & ~IVP_F_LAST);
    L.h = hnew;
    return true;
}

// ------------------------------------------------------------------------------------------------
// RK23 attempt (rk23.rs:189-307)
// ------------------------------------------------------------------------------------------------
template <class R, int FULL, bool CTL = false>
IVP_HD bool rk23_attempt(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L)
{
    KC_SCOPE_KZ(L.kz)
    constexpr int N = R::N;
    // tableau, rk23.rs:325-347
    constexpr double C2 = 0.5, C3 = 0.75, A21 = 0.5, A32 = 0.75;
    constexpr double B1 = 2.0 / 9.0, B2 = 1.0 / 3.0, B3 = 4.0 / 9.0;
    constexpr double E1 = 5.0 / 72.0, E2 = -1.0 / 12.0, E3 = -1.0 / 9.0, E4 = 1.0 / 8.0;
    constexpr double D21 = -4.0 / 3.0, D22 = 1.0, D23 = 4.0 / 3.0, D24 = -1.0;
    constexpr double D31 = 5.0 / 9.0, D32 = -2.0 / 3.0, D33 = -8.0 / 9.0, D34 = 1.0;
    constexpr double d_safety = 0.9, d_scale_min = 0.2, d_scale_max = 10.0;  // rk23.rs:16-36
    constexpr double error_exponent = -1.0 / 3.0;

    if (L.over || L.d_nstep >= L.budget) { L.status = 2; return false; }   // `steps.total >= nmax`
    double h = L.h;
    const double x = L.x;
    if ((x + h - L.xend) * L.posneg > 0.0) h = L.xend - x;

    const double *y = L.y, *k1 = L.k1, *p = L.p;
    double k2[N], k3[N], k4[N], yt[N], ye[N];
{ const double cA21 = KC(A21);
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h * cA21, k1[i]);
}
    R::ode(x + KC(C2) * h, yt, k2, p);
{ const double cA32 = KC(A32);
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h * cA32, k2[i]);
}
    R::ode(x + KC(C3) * h, yt, k3, p);
{ const double cB1 = KC(B1), cB2 = KC(B2), cB3 = KC(B3);
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h, IVP_LC(cB1, k1[i], cB2, k2[i], cB3, k3[i]));
}
    R::ode(x + h, yt, k4, p);
    L.d_nfev += 3;
{ const double cE1 = KC(E1), cE2 = KC(E2), cE3 = KC(E3), cE4 = KC(E4);
#pragma unroll
    for (int i = 0; i < N; ++i) ye[i] = h * IVP_LC(cE1, k1[i], cE2, k2[i], cE3, k3[i], cE4, k4[i]);
}
    double t_err[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double tl = IVP_MA(NormOps<R>::atol(a, i), NormOps<R>::rtol(a, i), fmax(fabs(yt[i]), fabs(y[i])));
        const double q = ye[i] / tl;
        t_err[i] = q * q;
    }
    double err = NormOps<R>::sum(t_err);
    err = sqrt(err / (double)NormOps<R>::NT);

    if (err <= 1.0) {
        L.d_nstep += 1;
        L.d_naccpt += 1;
        const double xnew = x + h;
        if (FULL == 1) {
            double cont[4 * N];
{ const double cD21 = KC(D21), cD22 = KC(D22), cD23 = KC(D23), cD24 = KC(D24), cD31 = KC(D31), cD32 = KC(D32), cD33 = KC(D33), cD34 = KC(D34);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                cont[i] = y[i];
                cont[N + i] = k1[i];
                cont[2 * N + i] = IVP_LC(cD21, k1[i], cD22, k2[i], cD23, k3[i], cD24, k4[i]);
                cont[3 * N + i] = IVP_LC(cD31, k1[i], cD32, k2[i], cD33, k3[i], cD34, k4[i]);
            }
}
#pragma unroll
            for (int i = 0; i < N; ++i) L.y[i] = yt[i];
            L.x = xnew;
            if (solout_full<M_RK23, R>(a, j, L, x, xnew, L.y, cont, cont, h, x)) { L.h = h; L.status = 1; return false; }   // rk23.rs:266-269
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) L.y[i] = yt[i];
            L.x = xnew;
            if (FULL == 2) so_log_accepted<R>(a, j, L, xnew, L.y);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) L.k1[i] = k4[i];
        h *= fmax(fmin(IVP_CTL(ctl_safety, d_safety) * ivp_pow(err, KC(error_exponent), IVP_KZ_ARG), IVP_CTL(ctl_scale_max, d_scale_max)), IVP_CTL(ctl_scale_min, d_scale_min));
        if (fabs(h) > L.hmax) h = L.hmax * L.posneg;
        L.h = h;
        if (xnew == L.xend) { L.status = 0; return false; }
    } else {
        L.d_nrejct += 1;
        h *= fmax(fmin(IVP_CTL(ctl_safety, d_safety) * ivp_pow(err, KC(error_exponent), IVP_KZ_ARG), 1.0), IVP_CTL(ctl_scale_min, d_scale_min));
        L.h = h;
        // The reference never terminates from here when err is NaN (h *= 1.0 forever) or once h has
        // collapsed to 0 (rk23.rs:300-306 has no underflow test).  A GPU lane must retire: report
        // StepSizeTooSmall, the status DOPRI5/DOP853 give for the same situation.
        if (err != err || h == 0.0) { L.status = 3; return false; }
    }
    if (L.h == 0.0 && L.x != L.xend) { L.status = 3; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------
// RK4 step (rk4.rs:137-226): fixed step, no error control.  Quirks kept: steps.accepted stays 0, the last
// step is not shortened (the final x is x0 + k*h), nfev counts 4 per step and not the initial evaluation.
// ------------------------------------------------------------------------------------------------
template <class R, int FULL>
IVP_HD bool rk4_attempt(const IvpKArgs &a, uint32_t j, Lane<R::N, R::P> &L)
{
    KC_SCOPE_KZ(L.kz)
    constexpr int N = R::N;
    constexpr double C2 = 0.5, C3 = 0.5, A21 = 0.5, A32 = 0.5;   // rk4.rs:247-257 (C4 = A43 = 1)
    constexpr double B1 = 1.0 / 6.0, B2 = 1.0 / 3.0, B3 = 1.0 / 3.0, B4 = 1.0 / 6.0;

    if (L.over || L.d_nstep >= L.budget) { L.status = 2; return false; }   // `steps.total >= nmax`
    const double h = L.h;
    const double x = L.x;
    const bool last = (x + KC(1.01) * h - L.xend) * rs_signum(h) > 0.0;

    const double *y = L.y, *k1 = L.k1, *p = L.p;
    double k2[N], k3[N], k4[N], yt[N], yold[N];
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h * A21, k1[i]);
    R::ode(x + C2 * h, yt, k2, p);
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h * A32, k2[i]);
    R::ode(x + C3 * h, yt, k3, p);
#pragma unroll
    for (int i = 0; i < N; ++i) yt[i] = IVP_MA(y[i], h * 1.0, k3[i]);
    R::ode(x + 1.0 * h, yt, k4, p);
    const double xnew = x + h;
    {
        const double cB1 = KC(B1), cB2 = KC(B2), cB3 = KC(B3), cB4 = KC(B4);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            yold[i] = y[i];
            L.y[i] = IVP_MA(y[i], h, IVP_LC(cB1, k1[i], cB2, k2[i], cB3, k3[i], cB4, k4[i]));
        }
    }
    R::ode(xnew, L.y, L.k1, p);
    L.x = xnew;
    L.d_nfev += 4;
    L.d_nstep += 1;
    if (FULL == 2) so_log_accepted<R>(a, j, L, xnew, L.y);
    if (FULL == 1) {
        double cont[4 * N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cont[i] = yold[i];
            cont[N + i] = k4[i];
            cont[2 * N + i] = L.k1[i];
            cont[3 * N + i] = L.y[i];
        }
        if (solout_full<M_RK4, R>(a, j, L, x, xnew, L.y, cont, cont, h, x)) { L.status = 1; return false; }   // rk4.rs:201-204
    }
    if (last) { L.status = 0; return false; }
    return true;
}

// One chunk of step attempts for one lane. Every lane leaves after at most `chunk` attempts.
template <int M, class R, int FULL, bool CTL = false>
IVP_HD uint32_t chunk_body(const IvpKArgs &a, uint32_t j, int32_t &status_out)
{
    Lane<R::N, R::P> L;
    lane_load<R>(a, j, L, M == M_RK23 || M == M_RK4, FULL);
#if defined(__HIP_DEVICE_COMPILE__) && IVP_HOIST >= 1
    L.kz = ivp_opaque_zero_v();
#else
    L.kz = 0;
#endif
    if (a.has_max_step) L.hmax = (M == M_DOPRI5) ? a.max_step : fabs(a.max_step);
    else L.hmax = fabs(L.xend - L.x0);
    uint32_t it = 0;
    bool run = true;
    while (run && it < a.chunk) {
        if ((FULL == 1 || FULL == 2) && a.log_pool != nullptr) so_log_attempt<typename OutMap<R>::type, FULL>(a, j, L, it);
        if constexpr (M == M_DOPRI5) run = dopri5_attempt<R, FULL, CTL>(a, j, L);
        else if constexpr (M == M_DOP853) run = dop853_attempt<R, FULL, CTL>(a, j, L);
        else if constexpr (M == M_RK4) run = rk4_attempt<R, FULL>(a, j, L);
        else run = rk23_attempt<R, FULL, CTL>(a, j, L);
        ++it;
    }
    if ((FULL == 1 || FULL == 2) && a.log_pool != nullptr) so_log_flush<typename OutMap<R>::type>(a, L);
    // Re-derive the store addresses from an opaque copy of j: otherwise the ~2 VGPRs per state array
    // that the loads' address arithmetic produced stay live across the whole attempt loop.
    uint32_t js = j;
    IVP_OPAQUE_V(js);
    lane_store<R>(a, js, L, FULL);
    status_out = L.status;
    return it;
}

}  // namespace IVP_NS
