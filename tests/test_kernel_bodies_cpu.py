"""CPU checks of the kernel LOGIC: the per-lane bodies of ivp_amd/csrc/rk_core.h, compiled for the host by the
test-only harness tests/host_emul and driven with the GPU launch loop's init -> chunk -> chunk schedule,
must reproduce the CPU oracle bit for bit (strict build; both sides use the portable step-controller pow).

These run without a GPU; the same cases run against libivp_hip.so in tests/test_gpu_parity.py (-m gpu).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.cases import CASES, CASE_IDS, EVENT_CASES, check_events_against_oracle
from tests.common import assert_bitexact, emul_batch, oracle_batch


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_strict_bodies_bitexact_vs_oracle(case):
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    g = emul_batch(rhs, y0, p, t0, t1, chunk=23, **o)
    r = oracle_batch(rhs, y0, p, t0, t1, **o)
    assert_bitexact(g, r, name + ": ")
    if "t_eval" in o:
        assert np.array_equal(g["n_filled"], r["n_filled"])
        m = g["n_filled"]
        for b in range(y0.shape[1]):
            assert np.array_equal(g["y_eval"][: m[b], :, b], r["y_eval"][: m[b], :, b])


@pytest.mark.parametrize("chunk", [1, 2, 7, 64, 100000])
@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853"])
def test_chunk_size_does_not_change_results(method, chunk):
    """State save/restore at launch boundaries is lossless: any chunk length gives identical bits."""
    from tests.cases import c3_vdp
    y0, p, t0, t1, o = c3_vdp(48, method)()
    a = emul_batch("vdp", y0, p, t0, t1, chunk=chunk, **o)
    b = oracle_batch("vdp", y0, p, t0, t1, **o)
    assert_bitexact(a, b)


def test_stiffness_detector_fires_like_the_reference():
    from tests.cases import stiff_vdp
    y0, p, t0, t1, o = stiff_vdp("DOPRI5")()
    g = emul_batch("vdp", y0, p, t0, t1, **o)
    assert (g["status"] == 4).any(), g["status"]     # ProbablyStiff somewhere in the sweep
    assert (g["status"] == 0).any()                  # and the mild ones finish


def test_nan_and_inf_lanes_retire_without_poisoning_neighbours():
    """a NaN trajectory is rejected forever, h shrinks until StepSizeTooSmall; other lanes unaffected."""
    B = 16
    y0 = np.tile(np.array([[1.0], [0.0]]), (1, B))
    y0[0, 3] = np.nan
    y0[1, 9] = np.inf
    t0 = np.full(B, 0.5)
    for method in ("DOPRI5", "DOP853"):
        g = emul_batch("sho", y0, None, t0, 3.0, method=method, rtol=1e-6, atol=1e-9)
        r = oracle_batch("sho", y0, None, t0, 3.0, method=method, rtol=1e-6, atol=1e-9)
        assert g["status"][3] == 3 and g["status"][9] == 3
        ok = np.ones(B, bool); ok[[3, 9]] = False
        assert (g["status"][ok] == 0).all()
        assert_bitexact(g, r, method + ": ")
    # RK23: the reference never terminates on a NaN error estimate (rk23.rs:300-306); the kernels retire the
    # lane with StepSizeTooSmall instead (documented deviation), neighbours still bit-exact.
    g = emul_batch("sho", y0, None, t0, 3.0, method="RK23", rtol=1e-6, atol=1e-9)
    assert g["status"][3] == 3 and g["status"][9] == 3
    good = np.ones(B, bool); good[[3, 9]] = False
    r = oracle_batch("sho", y0[:, good], None, t0[good], 3.0, method="RK23", rtol=1e-6, atol=1e-9)
    for k in ("y_end", "t_end", "h_next"):
        assert np.array_equal(np.asarray(g[k])[..., good], r[k])


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "RK4", "BDF"])
@pytest.mark.parametrize("direction", ["fwd", "bwd"])
def test_t_eval_sampling_matches_oracle(method, direction):
    """DefaultSolOut mode 1 on the device (solout.rs:344-386): emitted samples, their order and values."""
    B = 40
    rng = np.random.default_rng(2)
    y0 = np.stack([np.full(B, 1 / 3), np.full(B, 2 / 9)]) * (1 + 1e-3 * rng.standard_normal((2, B)))
    if direction == "fwd":
        t0, t1 = 5.0, 9.0
        te = np.array([4.0, 5.0, 5.01, 5.5, 7.0, 8.0, 8.01, 9.0, 9.5])   # before-start and past-end points too
    else:
        t0, t1 = 5.0, 1.0
        te = np.array([5.0, 4.99, 3.0, 1.5, 1.1, 1.0, 0.5])
    o = dict(method=method, rtol=1e-3, atol=1e-6, t_eval=te)
    g = emul_batch("rational", y0, None, t0, t1, chunk=5, **o)
    for b in range(0, B, 7):
        s = O.solve_ivp("rational", t0, t1, y0[:, b], detpow=True, **o)
        m = g["n_filled"][b]
        assert m == len(s.t)
        assert np.array_equal(te[g["eval_idx"][:m, b]], s.t)
        assert np.array_equal(g["y_eval"][:m, :, b], s.y)


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "BDF"])
@pytest.mark.parametrize("first_step", [None, 0.1])
def test_step_log_and_dense_segments_match_oracle(method, first_step):
    """DefaultSolOut mode 2 (solout.rs:387-428, incl. first_step enforcement) and dense-segment collection
    (solout.rs:141-146): Solution.t / Solution.y / ContinuousOutput, record for record."""
    B = 12
    rng = np.random.default_rng(4)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    for (t0, t1) in ((0.0, 3.0), (3.0, 0.0)):
        o = dict(method=method, rtol=1e-5, atol=1e-8, dense_output=True)
        if first_step is not None:
            o["first_step"] = first_step
        g = emul_batch("sho", y0, None, t0, t1, max_log=512, chunk=9, **o)
        for b in range(B):
            s = O.solve_ivp("sho", t0, t1, y0[:, b], detpow=True, **o)
            m = g["n_log"][b]
            assert m == len(s.t)
            assert np.array_equal(g["t_log"][:m, b], s.t)
            assert np.array_equal(g["y_log"][:m, :, b], s.y)
            ns = g["n_seg"][b]
            assert ns == len(s.seg_h)
            assert np.array_equal(g["seg_xold"][:ns, b], s.seg_xold)
            assert np.array_equal(g["seg_h"][:ns, b], s.seg_h)
            assert np.array_equal(g["seg_cont"][:ns, :, b], s.seg_cont)


def test_log_overflow_is_counted_not_written():
    y0 = np.array([[1.0], [0.0]])
    g = emul_batch("sho", y0, None, 0.0, 20.0, method="DOPRI5", rtol=1e-8, atol=1e-8, max_log=8)
    assert g["n_log"][0] > 8
    assert not np.isnan(g["t_log"][:, 0]).any()


@pytest.mark.parametrize("method,rtol,atol", [("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)])
def test_fast_mode_tracks_strict_mode(method, rtol, atol):
    """FMA contraction / reciprocal sharing changes results at the 1e-16-per-operation level only: on the smooth
    short-horizon problem end states agree to 1e-9 and step counts match almost everywhere."""
    y0, p, t0, _ = __import__("ivp_amd").workloads.cr3bp_batch(128)
    a = emul_batch("cr3bp", y0, p, t0, 2.0, method=method, rtol=rtol, atol=atol)
    b = emul_batch("cr3bp", y0, p, t0, 2.0, method=method, rtol=rtol, atol=atol, fast=True)
    assert (b["status"] == 0).all()
    assert np.abs(a["y_end"] - b["y_end"]).max() < 1e-9   # states are O(1)
    assert np.mean(a["naccpt"] == b["naccpt"]) > 0.95


@pytest.mark.parametrize("case", EVENT_CASES, ids=[c[0] for c in EVENT_CASES])
def test_event_detection_matches_oracle(case):
    """Device event detection (Brent on the step interpolant, direction filter, chronological processing, terminal
    counts, the appended terminal sample in both output modes) record for record, all five methods."""
    check_events_against_oracle(lambda rhs, y0, p, t0, t1, **kw: emul_batch(rhs, y0, p, t0, t1, chunk=7, **kw), case)


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "RK4", "BDF"])
def test_nan_interval_retires_the_lane(method):
    """Documented deviation: with a NaN t0/t1 the reference's step loop never terminates; the kernels retire the
    lane with StepSizeTooSmall and leave every other trajectory untouched."""
    B = 5
    y0 = np.tile(np.array([[1.0], [0.0]]), (1, B))
    t1 = np.full(B, 2.0); t1[2] = np.nan
    t0 = np.zeros(B); t0[4] = np.nan
    g = emul_batch("sho", y0, None, t0, t1, method=method, rtol=1e-6, atol=1e-9)
    assert list(g["status"]) == [0, 0, 3, 0, 3] and g["nfev"][2] == 0
    r = oracle_batch("sho", y0[:, :2], None, 0.0, 2.0, method=method, rtol=1e-6, atol=1e-9)
    assert np.array_equal(g["y_end"][:, :2], r["y_end"])


@pytest.mark.parametrize("fast", [False, True])
def test_bdf_change_d_structured_equals_the_literal_form(fast):
    """bdf_change_d exploits the structure of U = compute_r(order, 1) and of R's first row / column (bdf_core.h); the
    literal restatement of bdf.rs:669-732 stays in the header as bdf_change_d_generic.  Same bits on random and on
    adversarial inputs: factors that zero an R entry, signed zeros / infinities / NaNs in D, non-finite factors."""
    import ctypes as C
    from tests.host_emul import emul
    lib = emul.lib(fast)      # fast: the FMA arithmetic mode (the multiply-add sites of both forms are fused)
    lib.emul_change_d.argtypes = [C.c_int, C.c_int, C.c_double] + [np.ctypeslib.ndpointer(np.float64, flags="C")] * 3
    lib.emul_change_d.restype = C.c_int
    rng = np.random.default_rng(20260207)
    special = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e300, -1e300, 5e-324]
    factors = [0.5, 1.0 / 3.0, 0.25, 0.2, 2.0, 10.0, 1e-3, 0.75, 1.5, 4.0, 0.0, -0.0, -0.5, 1.0, 3.0, 1e49, 1e51, 1e300,
               np.inf, -np.inf, np.nan, 1.0 + 2 ** -52]
    n_checked = 0
    for n in (1, 2, 3, 6):
        for trial in range(400):
            order = int(rng.integers(1, 7))          # 6 exercises the clamp to MAX_ORDER
            factor = factors[trial % len(factors)] if trial < 3 * len(factors) else float(rng.uniform(0.05, 12.0))
            d = rng.standard_normal((8, n)) * 10.0 ** rng.integers(-8, 8, size=(8, 1))
            if trial % 3 == 0:
                idx = rng.integers(0, 8 * n, size=3)
                d.reshape(-1)[idx] = rng.choice(special, size=3)
            if trial % 7 == 0:
                d[rng.integers(0, 8)] = 0.0
            a, b = np.empty_like(d), np.empty_like(d)
            assert lib.emul_change_d(n, order, factor, np.ascontiguousarray(d), a, b) == 0
            same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (n, order, factor, d, a, b)
            n_checked += 1
    assert n_checked == 1600


def test_pow3_equals_three_pow_calls():
    """ivp_pow3 (three interleaved powers, special cases as selects; rk_core.h) returns ivp_pow's bits argument for argument."""
    import ctypes as C
    from tests.host_emul import emul
    lib = emul.lib()
    arr = np.ctypeslib.ndpointer(np.float64, flags="C")
    lib.emul_pow3.argtypes = [arr] * 4
    lib.emul_pow3.restype = None
    rng = np.random.default_rng(20260208)
    xs_special = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 5e-324, 1e-310, 2.2250738585072014e-308, 1.7976931348623157e308,
                  0.5, 2.0, 1e-300, 1e300]
    es_special = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, -0.5, -1.0 / 3.0, -0.25, -0.2, -1.0 / 6.0, -1.0 / 7.0, 1e3, -1e3, 1e-320]
    r3, r1 = np.empty(3), np.empty(3)
    n = 0
    for trial in range(6000):
        if trial < 3000:
            x = np.array(rng.choice(xs_special, 3), dtype=np.float64)
            e = np.array(rng.choice(es_special, 3), dtype=np.float64)
            if trial % 2:
                x[rng.integers(0, 3)] = 10.0 ** rng.uniform(-12, 6)
        else:
            x = 10.0 ** rng.uniform(-14, 8, 3)
            e = -1.0 / rng.integers(1, 8, 3).astype(np.float64)
        lib.emul_pow3(np.ascontiguousarray(x), np.ascontiguousarray(e), r3, r1)
        same = (r3.view(np.uint64) == r1.view(np.uint64)) | (np.isnan(r3) & np.isnan(r1))
        assert same.all(), (x, e, r3, r1)
        n += 1
    assert n == 6000


def test_div_by_small_constant_is_ieee_division():
    """ivp_div_small_const<3|5> (bdf_core.h: multiply by the rounded reciprocal plus one exact-remainder correction) is
    the IEEE quotient on the operand range change_d feeds it (zero, or 2^-52 <= |x| <= 1e51) -- and well beyond."""
    import ctypes as C
    from tests.host_emul import emul
    lib = emul.lib()
    lib.emul_div_small_const.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C"), C.c_long]
    lib.emul_div_small_const.restype = C.c_long
    rng = np.random.default_rng(20260209)
    n = 2_000_000
    mant = rng.integers(0, 1 << 52, n, dtype=np.uint64)
    expo = rng.integers(1023 - 900, 1023 + 900, n, dtype=np.uint64)
    sign = rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)
    x = (sign | (expo << np.uint64(52)) | mant).view(np.float64)
    structured = np.concatenate([np.arange(-40000, 40001) * 0.125, 2.0 - rng.uniform(0.2, 12.0, 100000) * rng.integers(1, 6, 100000),
                                 4.0 - rng.uniform(0.2, 12.0, 100000) * rng.integers(1, 6, 100000), [0.0, 2.0 ** -52, 1e51, -1e51]])   # not -0.0: see the helper's comment
    for c in (3, 5):
        assert lib.emul_div_small_const(c, np.ascontiguousarray(x), n) == 0
        assert lib.emul_div_small_const(c, np.ascontiguousarray(structured), structured.size) == 0


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "RK4"])
def test_deferred_event_refinement_equals_the_inline_search(method):
    """Without terminal events the stepping bodies only NOTE the steps that hold a crossing (so_events_note) and the roots are
    found afterwards, one noted step at a time (so_events_deferred_body) -- same so_event_root, same inputs: every event
    record, every counter and every other output bit for bit as with Brent inside the step; hits beyond max_events are
    counted but neither stored nor refined."""
    B = 9
    rng = np.random.default_rng(12)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    kw = dict(method=method, max_log=4096, chunk=5, event_direction=[0], event_terminal=[0])
    if method != "RK4":
        kw.update(rtol=1e-7, atol=1e-9)
    for (t0, t1, max_events) in ((0.0, 20.0, 16), (20.0, 0.0, 3)):
        a = emul_batch("sho_ev", y0, None, t0, t1, max_events=max_events, defer_events=False, **kw)
        d = emul_batch("sho_ev", y0, None, t0, t1, max_events=max_events, defer_events=True, **kw)
        assert "evd_rec" in d and "evd_rec" not in a
        assert (d["n_ev"] >= 5).all() and np.array_equal(a["n_ev"], d["n_ev"])
        assert np.array_equal(a["t_events"], d["t_events"], equal_nan=True) and np.array_equal(a["y_events"], d["y_events"], equal_nan=True)
        assert (d["evd_cnt"] == np.minimum(d["n_ev"][0], max_events)).all()
        for k in ("y_end", "t_log", "y_log", "n_log", "nfev", "naccpt", "status", "h_next"):
            assert np.array_equal(a[k], d[k], equal_nan=True), k
    # three event functions, two of them crossing in the same step now and then
    y0 = np.array([[4 / 9], [20 / 81]]).repeat(3, axis=1) * np.array([1.0, 1.001, 0.999])
    kw = dict(method=method, max_log=4096, event_direction=[0, 0, 0], event_terminal=[0, 0, 0])
    a = emul_batch("rational_ev", y0, None, 8.0, 5.0, defer_events=False, **kw)
    d = emul_batch("rational_ev", y0, None, 8.0, 5.0, defer_events=True, **kw)
    assert d["n_ev"].sum() > 0 and np.array_equal(a["n_ev"], d["n_ev"])
    assert np.array_equal(a["t_events"], d["t_events"], equal_nan=True) and np.array_equal(a["y_events"], d["y_events"], equal_nan=True)
