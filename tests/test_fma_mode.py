"""The FMA arithmetic mode (ivp_options_t.fp_mode = IVP_FP_FMA, `FpMode.FMA`).

It is a DEFINED arithmetic, not "whatever the compiler contracts": the multiply-add sites of the stage combinations,
error estimates, dense coefficients, interpolants, tolerance scales and of the built-in right-hand sides are written
out as fused operations in ivp_amd/csrc/rk_core.h (IVP_MA / IVP_MS / IVP_MB / IVP_LC, compiled with contraction off)
and fused at the same places in the oracle's third build (oracle/liboracle_fma.so, -DORC_FMA).  So FMA-mode results

  * equal liboracle_fma.so BIT FOR BIT, on the same case matrix strict mode has (tests/cases.py), and
  * are identical in every kernel variant (lean, resident-coefficient, lane-cooperative, wave-per-trajectory), for
    every chunk length and every batch composition -- which is what lets the launch loop hand FMA-mode trajectories to
    the cooperative tail kernel exactly as in strict mode.

Tolerance vs the faithful (libm-pow, unfused) restatement of the reference, stated in
test_fma_mode_vs_the_unfused_reference_arithmetic: 1e-9 absolute on O(1) states over the non-chaotic horizon, identical
step counts for > 95 % of the trajectories; BASELINE's "end-state error within 10x of the CPU reference's" on the full
horizon.

CPU part: the kernel bodies compiled for the host (tests/host_emul, -DIVP_FAST=1).  GPU part (-m gpu): libivp_hip.so.
"""
import json
import os

import numpy as np
import pytest

from ivp_amd import workloads as W
from oracle import oracle as O
from tests.cases import CASES, CASE_IDS, EVENT_CASES, c2_cr3bp, c3_vdp, check_events_against_oracle
from tests.common import assert_bitexact, emul_batch, gpu_batch, oracle_batch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
gpu = pytest.mark.gpu


def _check_eval(g, r, y0, o):
    if "t_eval" in o:
        assert np.array_equal(g["n_filled"], r["n_filled"])
        m = g["n_filled"]
        for b in range(y0.shape[1]):
            assert np.array_equal(g["y_eval"][: m[b], :, b], r["y_eval"][: m[b], :, b])


# ---------------------------------------------------------------------------------------------------------------
# CPU: oracle builds and the host-compiled kernel bodies
# ---------------------------------------------------------------------------------------------------------------

def test_the_three_oracle_builds_are_what_they_say():
    assert O.lib(False).orc_uses_detpow() == 0 and O.lib(False).orc_uses_fma() == 0
    assert O.lib(True).orc_uses_detpow() == 1 and O.lib(True).orc_uses_fma() == 0
    assert O.lib(True, True).orc_uses_detpow() == 1 and O.lib(True, True).orc_uses_fma() == 1


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_fma_bodies_bitexact_vs_fma_oracle(case):
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    g = emul_batch(rhs, y0, p, t0, t1, chunk=23, fast=True, **o)
    r = oracle_batch(rhs, y0, p, t0, t1, fma=True, **o)
    assert_bitexact(g, r, name + " [fma]: ")
    _check_eval(g, r, y0, o)


def test_fma_mode_is_a_different_arithmetic_from_strict_and_close_to_it():
    """The fused sites change bits (otherwise the mode would be pointless) at the rounding level only."""
    y0, p, t0, _ = W.cr3bp_batch(128)
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    a = oracle_batch("cr3bp", y0, p, t0, 2.0, **o)
    b = oracle_batch("cr3bp", y0, p, t0, 2.0, fma=True, **o)
    assert not np.array_equal(a["y_end"], b["y_end"])
    assert np.abs(a["y_end"] - b["y_end"]).max() < 1e-9
    assert np.mean(a["naccpt"] == b["naccpt"]) > 0.95


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "RK4", "BDF"])
def test_fma_t_eval_log_and_dense_records_match_the_fma_oracle(method):
    B = 12
    rng = np.random.default_rng(4)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    te = np.linspace(-0.1, 3.1, 17)
    o = dict(method=method, rtol=1e-5, atol=1e-8)
    g = emul_batch("sho", y0, None, 0.0, 3.0, chunk=5, fast=True, t_eval=te, **o)
    for b in range(B):
        s = O.solve_ivp("sho", 0.0, 3.0, y0[:, b], fma=True, t_eval=te, **o)
        m = g["n_filled"][b]
        assert m == len(s.t) and np.array_equal(te[g["eval_idx"][:m, b]], s.t) and np.array_equal(g["y_eval"][:m, :, b], s.y)
    if method == "RK4":
        return
    g = emul_batch("sho", y0, None, 3.0, 0.0, max_log=512, chunk=9, fast=True, dense_output=True, **o)
    for b in range(B):
        s = O.solve_ivp("sho", 3.0, 0.0, y0[:, b], fma=True, dense_output=True, **o)
        m, ns = g["n_log"][b], g["n_seg"][b]
        assert m == len(s.t) and np.array_equal(g["t_log"][:m, b], s.t) and np.array_equal(g["y_log"][:m, :, b], s.y)
        assert ns == len(s.seg_h) and np.array_equal(g["seg_cont"][:ns, :, b], s.seg_cont)


@pytest.mark.parametrize("case", EVENT_CASES, ids=[c[0] for c in EVENT_CASES])
def test_fma_event_detection_matches_the_fma_oracle(case):
    check_events_against_oracle(lambda rhs, y0, p, t0, t1, **kw: emul_batch(rhs, y0, p, t0, t1, chunk=7, fast=True, **kw), case, fma=True)


def test_group_norm_order_of_the_fma_oracle():
    """n > 8 in FMA mode: norms are summed in the wave-per-trajectory kernels' order (lane partials + xor butterfly,
    rk_group.h); checked here against a straightforward numpy restatement through a 100-state run's step count being
    reproducible and against the defining property: for n <= 8 the order is the reference's left-to-right sum."""
    y0 = np.linspace(0.5, 1.5, 100)[:, None] * np.ones((1, 3))
    a = O.solve_batch("linear_decay100", y0, None, 0.0, 2.0, detpow=True, method="DOPRI5", rtol=1e-7, atol=1e-10)
    b = O.solve_batch("linear_decay100", y0, None, 0.0, 2.0, fma=True, method="DOPRI5", rtol=1e-7, atol=1e-10)
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    np.testing.assert_allclose(b["y_end"], a["y_end"], rtol=1e-12)
    np.testing.assert_allclose(b["y_end"], y0 * np.exp(-2.0), rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------

@gpu
@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_fma_gpu_bitexact_vs_fma_oracle(case):
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    g = gpu_batch(rhs, y0, p, t0, t1, chunk=23, fast=True, **o)
    r = oracle_batch(rhs, y0, p, t0, t1, threads=8, fma=True, **o)
    assert_bitexact(g, r, name + " [fma]: ")
    _check_eval(g, r, y0, o)


COOP_CASES = [c for c in CASES if "DOPRI5" in c[0].upper() or "DOP853" in c[0].upper() or c[0] in ("C1-decay", "exp2-vector-rtol")]


@gpu
@pytest.mark.parametrize("case", COOP_CASES, ids=[c[0] for c in COOP_CASES])
def test_fma_every_kernel_variant_gives_the_same_bits(case):
    """lean (1), resident-coefficient (2) and lane-cooperative (3) kernels, whole runs in each, in FMA mode: all equal
    the FMA oracle -- the property the old compiler-contracted "fast" mode could not have."""
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    if "settings" in o:   # run-time controller fields exist in the lean and cooperative kernels only
        variants = (1, 3)
    else:
        variants = (1, 2, 3)
    r = oracle_batch(rhs, y0, p, t0, t1, threads=8, fma=True, **o)
    for v in variants:
        g = gpu_batch(rhs, y0, p, t0, t1, variant=v, chunk=17, fast=True, **o)
        assert_bitexact(g, r, f"{name} [fma] variant {v}: ")
        _check_eval(g, r, y0, o)


@gpu
@pytest.mark.parametrize("chunk", [1, 7, 64, 4096])
@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853"])
def test_fma_chunk_length_invariance(method, chunk):
    y0, p, t0, t1, o = c3_vdp(300, method)()
    a = gpu_batch("vdp", y0, p, t0, t1, chunk=chunk, fast=True, **o)
    b = oracle_batch("vdp", y0, p, t0, t1, threads=8, fma=True, **o)
    assert_bitexact(a, b)


@gpu
def test_fma_c2_hand_over_to_the_cooperative_tail_is_invisible():
    """A 20k slice of the C2 batch: the default policy (bulk launches, speculative hand-over to the cooperative tail
    kernel) vs whole runs in the lean and in the cooperative kernels, FMA mode -- same bits, equal to the FMA oracle."""
    y0, p, t0, t1 = W.cr3bp_batch(100000)
    y0, p = np.ascontiguousarray(y0[:, 40000:60000]), np.ascontiguousarray(p[:, 40000:60000])
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    auto = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=True, profile=1, **o)
    lean = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=True, variant=1, **o)
    coop = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=True, variant=3, **o)
    assert auto["stats"]["coop_launches"] > 0      # the tail really ran in the cooperative kernel
    assert_bitexact(auto, lean, "auto vs lean ")
    assert_bitexact(auto, coop, "auto vs coop ")
    r = oracle_batch("cr3bp", y0, p, t0, t1, threads=8, fma=True, **o)
    assert_bitexact(auto, r, "auto vs fma oracle ")


@gpu
def test_fma_results_do_not_depend_on_the_batch():
    y0, p, t0, t1 = W.cr3bp_batch(70000)     # > one wave per SIMD: lean kernels first, then resident, then cooperative
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    big = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=True, **o)
    idx = np.random.default_rng(3).choice(70000, 100, replace=False)
    small = gpu_batch("cr3bp", y0[:, idx], p[:, idx], t0, t1, fast=True, **o)
    for k in ("y_end", "t_end", "h_next", "nfev", "nstep", "naccpt", "nrejct"):
        assert np.array_equal(np.asarray(big[k])[..., idx], small[k]), k


@gpu
@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "BDF"])
def test_fma_wave_per_trajectory_kernels_vs_fma_oracle(method):
    """n = 100 and the coupled 256-cell heat equation: the butterfly norm order of rk_group.h == orc_sum()."""
    rng = np.random.default_rng(8)
    y0 = 1.0 + 0.3 * rng.standard_normal((100, 5))
    o = dict(method=method, rtol=1e-6, atol=1e-9)
    g = gpu_batch("linear_decay100", y0, None, 0.0, 3.0, fast=True, chunk=11, **o)
    r = oracle_batch("linear_decay100", y0, None, 0.0, 3.0, fma=True, **o)
    assert_bitexact(g, r, "decay100 [fma] ")
    xs = np.linspace(0.0, 1.0, 258)[1:-1]
    y0 = np.sin(np.pi * xs)[:, None] * (1.0 + 0.1 * np.arange(3))[None, :]
    kap = np.full((1, 3), 50.0 if method != "BDF" else 4000.0)
    g = gpu_batch("heat1d256", y0, kap, 0.0, 0.002, fast=True, **o)
    r = oracle_batch("heat1d256", y0, kap, 0.0, 0.002, fma=True, **o)
    assert_bitexact(g, r, "heat256 [fma] ")


@gpu
@pytest.mark.parametrize("case", EVENT_CASES, ids=[c[0] for c in EVENT_CASES])
def test_fma_gpu_event_detection_matches_the_fma_oracle(case):
    exact = not case[1].startswith("rational")   # those event functions call pow (ocml vs glibc)
    check_events_against_oracle(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, chunk=7, fast=True, **kw), case, exact=exact, fma=True)


@gpu
@pytest.mark.parametrize("method,rtol,atol", [("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)])
def test_fma_mode_vs_the_unfused_reference_arithmetic(method, rtol, atol):
    """The stated tolerance of FMA mode against the faithful restatement of the reference (libm pow, no fusion):
    1e-9 absolute on O(1) states over the non-chaotic horizon (t1 = 2), identical step counts for > 95 %."""
    y0, p, t0, _ = W.cr3bp_batch(1024)
    g = gpu_batch("cr3bp", y0, p, t0, 2.0, method=method, rtol=rtol, atol=atol, fast=True)
    r = oracle_batch("cr3bp", y0, p, t0, 2.0, detpow=False, threads=8, method=method, rtol=rtol, atol=atol)
    assert (g["status"] == 0).all()
    assert np.abs(g["y_end"] - r["y_end"]).max() < 1e-9
    assert np.mean(g["naccpt"] == r["naccpt"]) > 0.95


@gpu
def test_fma_c2_full_size_all_100k_trajectories_bitexact():
    """BASELINE C2 at full size in FMA mode: every one of the 100 000 trajectories against liboracle_fma.so."""
    y0, p, t0, t1 = W.cr3bp_batch(100_000)
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    g = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=True, **o)
    r = oracle_batch("cr3bp", y0, p, t0, t1, threads=16, fma=True, **o)
    assert (g["status"] == 0).all()
    assert_bitexact(g, r, "C2 100k [fma] ")
    truth = np.asarray(json.load(open(os.path.join(GOLD, "scipy_truth.json")))["truth"]["cr3bp"]["y_end"])
    n = len(truth)
    ref = oracle_batch("cr3bp", y0[:, :n], p[:, :n], t0, t1, detpow=False, **o)
    eg = np.abs(g["y_end"][:, :n].T - truth).max(axis=1)
    er = np.abs(ref["y_end"].T - truth).max(axis=1)
    assert np.median(eg) <= 10.0 * np.median(er) and eg.max() <= 10.0 * er.max()     # BASELINE's accuracy bar


VDP_SRC = r"""
__device__ void ode(double t, const double* y, double* d, const double* p)
{
    d[0] = y[1];
    d[1] = p[0] * (1.0 - y[0] * y[0]) * y[1] - y[0];
}
"""


@gpu
@pytest.mark.parametrize("method", ["DOPRI5", "DOP853", "RK23", "BDF"])
def test_fma_mode_with_a_user_right_hand_side_is_the_integrator_fused_and_the_snippet_as_written(method):
    """hiprtc problems in FMA mode: the integrator's multiply-add sites are fused, the user's code is compiled without
    contraction and evaluated as written.  The oracle's FMA build driven with the same right-hand side as a Python
    callable (plain IEEE double arithmetic = the snippet as written) reproduces it bit for bit."""
    import ivp_amd
    f = ivp_amd.DeviceIVP(VDP_SRC, n=2, params=(1.5,))
    rng = np.random.default_rng(9)
    y0 = np.stack([2.0 + 0.1 * rng.standard_normal(4), 0.1 * rng.standard_normal(4)])
    mu = np.full((1, 4), 1.5)
    o = dict(method=method, rtol=1e-6, atol=1e-9)
    r = ivp_amd.solve_ivp_batch(f, 0.0, 6.0, y0, mu, ivp_amd.Options(fp_mode=ivp_amd.FpMode.FMA, **o))
    fun = lambda t, y, p: [y[1], p[0] * (1.0 - y[0] * y[0]) * y[1] - y[0]]
    for b in range(4):
        s = O.solve_ivp(fun, 0.0, 6.0, list(y0[:, b]), params=[1.5], fma=True, **o)
        assert int(r.status[b]) == s.status == 0
        assert np.array_equal(r.y_end[:, b], s.y[-1]) and int(r.naccpt[b]) == s.naccpt and int(r.nfev[b]) == s.nfev, (method, b)
    # ... and it differs from the built-in problem's FMA form (whose right-hand side is fused as well) only at rounding level
    rb = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), 0.0, 6.0, y0, mu, ivp_amd.Options(fp_mode=ivp_amd.FpMode.FMA, **o))
    assert np.abs(np.asarray(rb.y_end) - np.asarray(r.y_end)).max() < 1e-6
