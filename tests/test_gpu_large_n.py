"""Wave-per-trajectory kernels (rk_group.h: 8 < n <= 512, one wavefront per trajectory) against the oracle.

STRICT: the per-component arithmetic is the reference's and the error-norm sum runs in index order, so results are
bit-exact with the oracle's detpow build.  FAST uses a __shfl_xor butterfly for the norm: compared by tolerance."""
import numpy as np
import pytest

from tests.common import assert_bitexact, gpu_batch, oracle_batch

pytestmark = pytest.mark.gpu


def _decay_batch(B, seed=3):
    rng = np.random.default_rng(seed)
    y0 = rng.uniform(-2.0, 2.0, (100, B))
    t1 = rng.uniform(0.5, 12.0, B)
    return y0, None, 0.0, t1


def _heat_batch(B, seed=5):
    rng = np.random.default_rng(seed)
    x = np.arange(1, 257) / 257.0
    modes = rng.integers(1, 6, B)
    y0 = np.sin(np.pi * x[:, None] * modes[None, :]) + 0.1 * rng.standard_normal((256, B))
    kappa = rng.uniform(20.0, 400.0, (1, B))
    t1 = rng.uniform(0.05, 0.5, B)
    return y0, kappa, 0.0, t1


@pytest.mark.parametrize("tol", [(1e-3, 1e-6), (1e-6, 1e-9), (1e-10, 1e-12)])
def test_linear_decay100_bitexact(tol):
    """benches/benchmark.py:139-148 (N = 100, RK45) generalised to a batch with ragged horizons."""
    y0, p, t0, t1 = _decay_batch(300)
    ref = oracle_batch("linear_decay100", y0, p, t0, t1, method="DOPRI5", rtol=tol[0], atol=tol[1])
    got = gpu_batch("linear_decay100", y0, p, t0, t1, method="DOPRI5", rtol=tol[0], atol=tol[1])
    assert_bitexact(got, ref, "decay100 ")
    assert (got["status"] == 0).all()
    np.testing.assert_allclose(got["y_end"], y0 * np.exp(-t1)[None, :], rtol=0, atol=50 * tol[0])


@pytest.mark.parametrize("chunk", [1, 7, 64, 0])
def test_heat1d256_bitexact_any_chunk(chunk):
    y0, p, t0, t1 = _heat_batch(200)
    ref = oracle_batch("heat1d256", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9)
    got = gpu_batch("heat1d256", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, chunk=chunk)
    assert_bitexact(got, ref, f"heat chunk={chunk} ")
    assert ref["nrejct"].sum() > 0          # the case exercises the reject branch


def test_heat1d256_stiffness_detection_matches():
    """kappa large enough that DOPRI5's stiffness test fires (dopri5.rs:364-391): status ProbablyStiff at the same
    step, same state."""
    y0, p, t0, _ = _heat_batch(64)
    p[:] = 4000.0
    ref = oracle_batch("heat1d256", y0, p, t0, 20.0, method="DOPRI5", rtol=1e-4, atol=1e-7)
    got = gpu_batch("heat1d256", y0, p, t0, 20.0, method="DOPRI5", rtol=1e-4, atol=1e-7)
    assert_bitexact(got, ref, "heat stiff ")
    assert (ref["status"] == 4).any()


def test_large_n_options_first_step_max_step_max_steps_backward():
    y0, p, t0, t1 = _decay_batch(64, seed=9)
    for kw in (dict(first_step=1e-3), dict(max_step=0.05), dict(max_steps=20), dict(first_step=0.2, max_step=0.3)):
        ref = oracle_batch("linear_decay100", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, **kw)
        got = gpu_batch("linear_decay100", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, **kw)
        assert_bitexact(got, ref, f"{kw} ")
    # backward in time and a zero-length interval
    t1b = -np.abs(t1) * 0.3
    t1b[::7] = 0.0
    ref = oracle_batch("linear_decay100", y0, p, 0.0, t1b, method="DOPRI5", rtol=1e-6, atol=1e-9)
    got = gpu_batch("linear_decay100", y0, p, 0.0, t1b, method="DOPRI5", rtol=1e-6, atol=1e-9)
    assert_bitexact(got, ref, "backward ")


def test_large_n_device_arrays_and_fast_mode():
    y0, p, t0, t1 = _heat_batch(500, seed=11)
    ref = oracle_batch("heat1d256", y0, p, t0, t1, method="DOPRI5", rtol=1e-8, atol=1e-10)
    got = gpu_batch("heat1d256", y0, p, t0, t1, method="DOPRI5", rtol=1e-8, atol=1e-10, device_arrays=True)
    assert_bitexact(got, ref, "heat device ")
    fast = gpu_batch("heat1d256", y0, p, t0, t1, method="DOPRI5", rtol=1e-8, atol=1e-10, device_arrays=True, fast=True)
    assert (fast["status"] == 0).all()
    # FMA contraction + tree-order norm: the step sequence of this stability-limited problem shifts slightly, the
    # answers agree at the level of the requested tolerance (rtol 1e-8 on O(1) states)
    np.testing.assert_allclose(fast["y_end"], ref["y_end"], rtol=0, atol=1e-7)
    assert abs(int(fast["naccpt"].sum()) - int(ref["naccpt"].sum())) <= 0.02 * ref["naccpt"].sum()


def test_large_n_unsupported_requests_fail_loudly():
    import ivp_amd
    y0 = np.ones((100, 4))
    f = ivp_amd.LinearDecay100()
    with pytest.raises(ivp_amd.ConfigError) as e:   # RADAU is not on the accelerated path for any n
        ivp_amd.solve_ivp_batch(f, 0.0, 1.0, y0, None, ivp_amd.Options(method="RADAU"))
    assert e.value.code == -101
    with pytest.raises(ivp_amd.ConfigError) as e:   # Tolerance::Vector of the wrong length (mod.rs:156-161)
        ivp_amd.solve_ivp_batch(f, 0.0, 1.0, y0, None, ivp_amd.Options(method="DOPRI5", rtol=[1e-6] * 99))
    assert e.value.code == -4


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853"])
def test_large_n_vector_tolerances(method):
    """Tolerance::Vector (mod.rs:104-214) with n = 256: per-component rtol / atol travel as device arrays."""
    y0, p, t0, t1 = _heat_batch(48, seed=41)
    rng = np.random.default_rng(42)
    rt = 10.0 ** rng.uniform(-8, -4, 256)
    at = 10.0 ** rng.uniform(-11, -7, 256)
    for kw in (dict(rtol=rt, atol=at), dict(rtol=rt, atol=1e-9), dict(rtol=1e-6, atol=at)):
        ref = oracle_batch("heat1d256", y0, p, t0, t1, method=method, **kw)
        got = gpu_batch("heat1d256", y0, p, t0, t1, method=method, **kw)
        assert_bitexact(got, ref, f"vector tol {method} ")


@pytest.mark.parametrize("method,tol", [("RK23", (1e-4, 1e-7)), ("DOP853", (1e-9, 1e-11)), ("DOP853", (1e-4, 1e-6)), ("RK4", None)])
def test_large_n_other_methods_bitexact(method, tol):
    """The same attempt bodies as the thread-per-trajectory kernels, instantiated per lane-slice: RK23, DOP853, RK4."""
    kw = dict(method=method) if tol is None else dict(method=method, rtol=tol[0], atol=tol[1])
    y0, p, t0, t1 = _heat_batch(96, seed=21)
    ref = oracle_batch("heat1d256", y0, p, t0, t1, **kw)
    got = gpu_batch("heat1d256", y0, p, t0, t1, **kw)
    assert_bitexact(got, ref, f"heat {method} ")
    y0, p, t0, t1 = _decay_batch(64, seed=22)
    ref = oracle_batch("linear_decay100", y0, p, t0, t1, **kw)
    got = gpu_batch("linear_decay100", y0, p, t0, t1, chunk=9, **kw)
    assert_bitexact(got, ref, f"decay {method} ")


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853"])
def test_large_n_t_eval_log_and_dense_match_oracle(method):
    """DefaultSolOut on the wave-per-trajectory path: t_eval samples, accepted-step log and dense segments record for
    record (solout.rs:344-428), including points outside the span and a backward run."""
    from oracle import oracle as O
    import ivp_amd
    rng = np.random.default_rng(31)
    x = np.arange(1, 257) / 257.0
    B = 12
    y0 = np.sin(np.pi * x[:, None] * rng.integers(1, 4, B)[None, :]) + 0.05 * rng.standard_normal((256, B))
    kappa = rng.uniform(30.0, 120.0, (1, B))
    tol = dict(RK23=(1e-4, 1e-7), DOPRI5=(1e-6, 1e-9), DOP853=(1e-8, 1e-10))[method]
    te = np.concatenate([[-0.1], np.linspace(0.0, 0.3, 17), [0.31]])
    got = gpu_batch("heat1d256", y0, kappa, 0.0, 0.3, method=method, rtol=tol[0], atol=tol[1], t_eval=te)
    ref = oracle_batch("heat1d256", y0, kappa, 0.0, 0.3, method=method, rtol=tol[0], atol=tol[1], t_eval=te)
    assert_bitexact(got, ref, f"t_eval {method} ")
    assert np.array_equal(got["n_filled"], ref["n_filled"]) and (got["n_filled"] == 17).all()
    for b in range(B):
        assert np.array_equal(got["y_eval"][:17, :, b], ref["y_eval"][:17, :, b])
        assert np.array_equal(got["eval_idx"][:17, b], np.arange(1, 18))
    # step log + dense segments for one trajectory through the single-solve front end, against the oracle's Solution
    f = ivp_amd.Heat1D256(float(kappa[0, 0]))
    s = ivp_amd.solve_ivp(f, 0.0, 0.3, y0[:, 0], ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1], dense_output=True))
    o = O.solve_ivp("heat1d256", 0.0, 0.3, list(y0[:, 0]), params=[float(kappa[0, 0])], method=method, rtol=tol[0], atol=tol[1],
                    dense_output=True, detpow=True)
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
    assert (s.nfev, s.naccpt, s.nrejct) == (o.nfev, o.naccpt, o.nrejct)
    for tq in (0.0, 0.0123, 0.15, 0.2999, 0.3):
        assert np.array_equal(s.sol(tq), o.sol(tq))
    # backward in time with first_step (step-record mode enforces the first output, solout.rs:390-417)
    s = ivp_amd.solve_ivp(ivp_amd.LinearDecay100(), 1.0, 0.2, np.linspace(0.5, 1.5, 100), ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1], first_step=0.05))
    o = O.solve_ivp("linear_decay100", 1.0, 0.2, list(np.linspace(0.5, 1.5, 100)), method=method, rtol=tol[0], atol=tol[1], first_step=0.05, detpow=True)
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)


def test_large_n_single_solve_ivp_and_jit_component_form():
    """`impl IVP` for n = 40 as a device snippet in component form: a ring of coupled oscillators."""
    import ivp_amd
    s = ivp_amd.solve_ivp(ivp_amd.LinearDecay100(), 0.0, 5.0, np.linspace(0, 1, 100), ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    from oracle import oracle as O
    o = O.solve_ivp("linear_decay100", 0.0, 5.0, list(np.linspace(0, 1, 100)), method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True)
    assert (s.naccpt, s.nrejct, s.nfev, int(s.status)) == (o.naccpt, o.nrejct, o.nfev, o.status)
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y) and s.t[-1] == 5.0

    src = r'''
    __device__ double ode_comp(int i, double t, const double* y, const double* p)
    {   // 20 masses on a ring: y[0..20) positions, y[20..40) velocities
        if (i < 20) return y[20 + i];
        const int k = i - 20, l = (k + 19) % 20, r = (k + 1) % 20;
        return p[0] * (y[l] - 2.0 * y[k] + y[r]);
    }'''
    f = ivp_amd.DeviceIVP(src, n=40, params=(3.0,))

    def ring(t, y, p):
        d = np.empty(40)
        d[:20] = y[20:]
        q = y[:20]
        d[20:] = p[0] * (np.roll(q, 1) - 2.0 * q + np.roll(q, -1))
        return d
    rng = np.random.default_rng(2)
    y0 = rng.standard_normal(40)
    for method in ("DOPRI5", "DOP853"):
        s = ivp_amd.solve_ivp(f, 0.0, 4.0, y0, ivp_amd.Options(method=method, rtol=1e-7, atol=1e-9, t_eval=[1.0, 2.5, 4.0]))
        o = O.solve_ivp(ring, 0.0, 4.0, list(y0), params=[3.0], method=method, rtol=1e-7, atol=1e-9, t_eval=[1.0, 2.5, 4.0], detpow=True)
        assert int(s.status) == 0 and (s.naccpt, s.nrejct, s.nfev) == (o.naccpt, o.nrejct, o.nfev)
        # numpy's roll-based RHS adds in the same order as the snippet: bit-exact samples
        assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)


@pytest.mark.parametrize("method", ["DOPRI5", "DOP853", "RK23"])
def test_large_n_events_match_oracle(method):
    """Event detection (solout.rs:158-331) on the wave-per-trajectory path: the user's event functions see the whole
    state (LDS copy); detected times / states, direction filter and a terminal count agree with the oracle bit for bit."""
    import ivp_amd
    from oracle import oracle as O
    src = r'''
    __device__ double ode_comp(int i, double t, const double* y, const double* p)
    {   // 20 masses on a ring: y[0..20) positions, y[20..40) velocities
        if (i < 20) return y[20 + i];
        const int k = i - 20, l = (k + 19) % 20, r = (k + 1) % 20;
        return p[0] * (y[l] - 2.0 * y[k] + y[r]);
    }
    __device__ void events(double t, const double* y, double* g, const double* p)
    {
        g[0] = y[0];                 // mass 0 passes the origin
        g[1] = y[5] - y[15];         // two masses at the same displacement
    }'''

    def ring(t, y, p):
        d = np.empty(40)
        d[:20] = y[20:]
        q = y[:20]
        d[20:] = p[0] * (np.roll(q, 1) - 2.0 * q + np.roll(q, -1))
        return d

    ev = lambda t, y, p: [y[0], y[5] - y[15]]
    rng = np.random.default_rng(6)
    y0 = rng.standard_normal(40)
    tol = dict(RK23=(1e-5, 1e-8), DOPRI5=(1e-7, 1e-9), DOP853=(1e-9, 1e-11))[method]
    for cfgs, okw in (([ivp_amd.EventConfig(), ivp_amd.EventConfig()], dict(event_direction=[0, 0], event_terminal=[0, 0])),
                      ([ivp_amd.EventConfig().positive(), ivp_amd.EventConfig(ivp_amd.Direction.Negative, 2)],
                       dict(event_direction=[1, -1], event_terminal=[0, 2]))):
        f = ivp_amd.DeviceIVP(src, n=40, params=(3.0,), events=cfgs)
        s = ivp_amd.solve_ivp(f, 0.0, 6.0, y0, ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1]))
        o = O.solve_ivp(ring, 0.0, 6.0, list(y0), params=[3.0], method=method, rtol=tol[0], atol=tol[1], detpow=True,
                        events=ev, n_events=2, **okw)
        assert int(s.status) == o.status and (s.naccpt, s.nrejct, s.nfev) == (o.naccpt, o.nrejct, o.nfev)
        assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
        for i in range(2):
            assert np.array_equal(s.t_events[i], o.t_events[i]), (i, s.t_events[i], o.t_events[i])
            assert np.array_equal(s.y_events[i], o.y_events[i])
        assert sum(len(t) for t in s.t_events) >= 3


@pytest.mark.parametrize("K", [6, 12, 20])          # n = 12 (16 lanes per trajectory), 24 (32 lanes), 40 (whole wavefront)
@pytest.mark.parametrize("method", ["DOPRI5", "DOP853", "RK23"])
def test_group_width_follows_the_system_size(K, method):
    """hiprtc modules of systems with n <= 16 / n <= 32 put 4 / 2 trajectories into one wavefront (own LDS region each,
    groups diverge by predication); results are the oracle's bit for bit for every trajectory of a ragged batch,
    including t_eval samples and a terminal event."""
    import ivp_amd
    from oracle import oracle as O
    n = 2 * K
    src = r'''
    #define K %d
    __device__ double ode_comp(int i, double t, const double* y, const double* p)
    {   // K masses on a ring: y[0..K) positions, y[K..2K) velocities
        if (i < K) return y[K + i];
        const int k = i - K, l = (k + K - 1) %% K, r = (k + 1) %% K;
        return p[0] * (y[l] - 2.0 * y[k] + y[r]);
    }
    __device__ void events(double t, const double* y, double* g, const double* p) { g[0] = y[0] - y[K / 2]; }
    ''' % K

    def ring(t, y, p):
        d = np.empty(n)
        d[:K] = y[K:]
        q = y[:K]
        d[K:] = p[0] * (np.roll(q, 1) - 2.0 * q + np.roll(q, -1))
        return d

    rng = np.random.default_rng(10 + K)
    B = 11
    y0 = rng.standard_normal((n, B))
    par = rng.uniform(1.0, 4.0, (1, B))
    t1 = rng.uniform(1.0, 5.0, B)
    tol = dict(RK23=(1e-5, 1e-8), DOPRI5=(1e-7, 1e-9), DOP853=(1e-9, 1e-11))[method]
    # end states of the whole batch (no events: a problem without event functions)
    f = ivp_amd.DeviceIVP(src.replace("__device__ void events", "__device__ void unused_events"), n=n, params=(3.0,))
    r = ivp_amd.solve_ivp_batch(f, 0.0, t1, y0, par, ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1], chunk_attempts=7))
    for b in range(B):
        o = O.solve_ivp(ring, 0.0, t1[b], list(y0[:, b]), params=[par[0, b]], method=method, rtol=tol[0], atol=tol[1], detpow=True)
        assert np.array_equal(r.y_end[:, b], o.y[-1]) and (int(r.naccpt[b]), int(r.nrejct[b]), int(r.nfev[b])) == (o.naccpt, o.nrejct, o.nfev), b
    # one trajectory with t_eval and a terminal event (second crossing)
    fe = ivp_amd.DeviceIVP(src, n=n, params=(3.0,), events=[ivp_amd.EventConfig(ivp_amd.Direction.All, 2)])
    te = np.linspace(0.0, 6.0, 13)
    s = ivp_amd.solve_ivp(fe, 0.0, 6.0, y0[:, 0], ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1], t_eval=te))
    o = O.solve_ivp(ring, 0.0, 6.0, list(y0[:, 0]), params=[3.0], method=method, rtol=tol[0], atol=tol[1], detpow=True, t_eval=te,
                    events=lambda t, y, p: [y[0] - y[K // 2]], n_events=1, event_direction=[0], event_terminal=[2])
    assert int(s.status) == o.status == 1
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
    assert np.array_equal(s.t_events[0], o.t_events[0]) and np.array_equal(s.y_events[0], o.y_events[0])


# ---- BDF for large n (bdf_group.h): LU of the per-trajectory n x n matrices by the trajectory's wavefront ----------
def test_bdf_linear_decay100_equals_the_oracle():
    """`bdf.rs:86` is generic in n: the reference benchmark's N = 100 system through BDF (forward-difference Jacobian,
    100 x 100 LU with partial pivoting, Newton) equals the oracle bit for bit, counters included."""
    y0, p, t0, t1 = _decay_batch(40)
    for chunk in (0, 3):
        ref = oracle_batch("linear_decay100", y0, p, t0, t1, method="BDF", rtol=1e-5, atol=1e-8)
        got = gpu_batch("linear_decay100", y0, p, t0, t1, method="BDF", rtol=1e-5, atol=1e-8, chunk=chunk)
        assert_bitexact(got, ref, f"bdf decay100 chunk={chunk} ")
        assert (got["status"] == 0).all() and (got["nlu"] > 0).all()
    np.testing.assert_allclose(got["y_end"], y0 * np.exp(-t1)[None, :], rtol=0, atol=2e-3)


def test_bdf_heat1d256_stiff_equals_the_oracle():
    """The stiff case the explicit methods give up on (ProbablyStiff above): kappa = 4000 on 256 cells, t = 20.  BDF
    takes a few dozen steps; pivoting, LU reuse and Jacobian refreshes follow the oracle's path exactly."""
    y0, p, t0, _ = _heat_batch(12)
    p[:] = 4000.0
    ref = oracle_batch("heat1d256", y0, p, t0, 20.0, method="BDF", rtol=1e-4, atol=1e-7)
    got = gpu_batch("heat1d256", y0, p, t0, 20.0, method="BDF", rtol=1e-4, atol=1e-7)
    assert_bitexact(got, ref, "bdf heat ")
    assert (got["status"] == 0).all() and int(got["nstep"].max()) < 2000
    # outputs: t_eval sampling with the BDF interpolant, vector tolerances
    te = np.linspace(0.0, 20.0, 9)
    rt = np.full(256, 1e-4); rt[::2] = 1e-5
    ref = oracle_batch("heat1d256", y0[:, :4], p[:, :4], t0, 20.0, method="BDF", rtol=rt, atol=1e-7, t_eval=te)
    got = gpu_batch("heat1d256", y0[:, :4], p[:, :4], t0, 20.0, method="BDF", rtol=rt, atol=1e-7, t_eval=te)
    assert_bitexact(got, ref, "bdf heat t_eval ")
    assert np.array_equal(got["n_filled"], ref["n_filled"]) and np.array_equal(got["y_eval"][:9], ref["y_eval"][:9])


def test_bdf_large_n_dense_output_equals_the_oracle():
    """Dense-output segments of the wave-per-trajectory BDF (per-state blocks [D0, D1..D5, order], cont.rs:44-51): step
    record, interpolant and counters of a stiff 256-cell heat equation and of the N = 100 decay system, bit for bit."""
    import ivp_amd
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    x = np.linspace(0.0, 1.0, 258)[1:-1]
    y0 = np.sin(np.pi * x) + 0.05 * rng.standard_normal(256)
    s = ivp_amd.solve_ivp(ivp_amd.Heat1D256(800.0), 0.0, 0.05, y0, ivp_amd.Options(method="BDF", rtol=1e-5, atol=1e-8, dense_output=True))
    o = O.solve_ivp("heat1d256", 0.0, 0.05, list(y0), params=[800.0], method="BDF", rtol=1e-5, atol=1e-8, dense_output=True, detpow=True)
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
    assert (s.nfev, s.njev, s.nlu, s.naccpt, s.nrejct) == (o.nfev, o.njev, o.nlu, o.naccpt, o.nrejct)
    assert s.sol_span() == o.sol_span()
    for tq in (0.0, 1e-4, 0.0123, 0.04999, 0.05):
        assert np.array_equal(s.sol(tq), o.sol(tq))
    y1 = np.linspace(0.5, 1.5, 100)
    s = ivp_amd.solve_ivp(ivp_amd.LinearDecay100(), 2.0, 0.5, y1, ivp_amd.Options(method="BDF", rtol=1e-6, atol=1e-9, dense_output=True, t_eval=[2.0, 1.5, 1.0, 0.5]))
    o = O.solve_ivp("linear_decay100", 2.0, 0.5, list(y1), method="BDF", rtol=1e-6, atol=1e-9, dense_output=True, t_eval=[2.0, 1.5, 1.0, 0.5], detpow=True)
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
    for tq in (2.0, 1.77, 0.9, 0.5):
        assert np.array_equal(s.sol(tq), o.sol(tq))
    # a zero-length interval inside a batch: the constant interpolant (cont.rs:44-51)
    yb = np.stack([y1, y1 * 2.0], axis=1)
    r = ivp_amd.solve_ivp_batch(ivp_amd.LinearDecay100(), np.array([1.0, 1.0]), np.array([1.0, 1.2]), yb, None,
                                ivp_amd.Options(method="BDF", dense_output=True, max_log=64))
    assert int(r.n_seg[0]) == 1 and float(r.seg_h[0, 0]) == 1e-15 and float(r.seg_xold[0, 0]) == 1.0
    blk = np.asarray(r.seg_cont[0, :, 0]).reshape(100, 7)
    assert np.array_equal(blk[:, 0], y1) and (blk[:, 1:6] == 0.0).all() and (blk[:, 6] == 1.0).all()


@pytest.mark.parametrize("n", [12, 24])
def test_bdf_small_groups_dense_jacobian_equals_the_oracle(n):
    """Wave-per-trajectory BDF with SEVERAL trajectories per wavefront (n = 12: groups of 16 lanes, n = 24: groups of 32):
    a dense, differently scaled linear system per trajectory, so that the groups of one wave pivot differently and flag
    different trailing columns in `lu_decomp`; every trajectory equals the oracle driven by the same formula."""
    import ivp_amd
    from oracle import oracle as O
    src = f"""
__device__ double ode_comp(int i, double t, const double* y, const double* p)
{{
    double s = 0.0;
    for (int j = 0; j < {n}; ++j) s += ((double)((i * 7 + j * 13) % 11 - 5) / 4.0 - (i == j ? ((i & 1) ? 0.25 : 6.0) : 0.0)) * y[j];
    return p[0] * s + (double)(i % 3 - 1) * 0.5;
}}
"""
    A = [[((i * 7 + j * 13) % 11 - 5) / 4.0 - ((0.25 if i & 1 else 6.0) if i == j else 0.0) for j in range(n)] for i in range(n)]

    def fun(t, y, p):
        out = []
        for i in range(n):
            s = 0.0
            for j in range(n):
                s += A[i][j] * float(y[j])
            out.append(float(p[0]) * s + float(i % 3 - 1) * 0.5)
        return out

    B = 7
    rng = np.random.default_rng(n)
    y0 = rng.standard_normal((n, B))
    scale = np.array([[1.0, 40.0, 0.3, 12.0, 7.0, 25.0, 2.5]])   # odd rows are not diagonally dominant: I - cJ needs row exchanges
    f = ivp_amd.DeviceIVP(src, n=n, params=(1.0,))
    r = ivp_amd.solve_ivp_batch(f, 0.0, 0.25, y0, scale, ivp_amd.Options(method="BDF", rtol=1e-6, atol=1e-9))
    for b in range(B):
        o = O.solve_ivp(fun, 0.0, 0.25, list(y0[:, b]), params=[float(scale[0, b])], method="BDF", rtol=1e-6, atol=1e-9, detpow=True)
        assert int(r.status[b]) == int(o.status)
        assert (int(r.nfev[b]), int(r.njev[b]), int(r.nlu[b]), int(r.naccpt[b]), int(r.nrejct[b])) == (o.nfev, o.njev, o.nlu, o.naccpt, o.nrejct), b
        assert np.array_equal(np.asarray(r.y_end)[:, b], o.y[-1]), b


# ---- LDS-resident LU (BASELINE C5: "batched dense LU of per-trajectory Jacobians in LDS") ------------------------------

def _dense64_batch(B, seed=12):
    rng = np.random.default_rng(seed)
    y0 = 1.0 + 0.5 * rng.standard_normal((64, B))
    k = np.full((1, B), 3.0) * (1.0 + 0.2 * rng.uniform(-1, 1, (1, B)))
    return y0, k


@pytest.mark.parametrize("fast", [False, True])
def test_bdf_dense_64_state_system_lds_and_global_factors_bitexact_vs_oracle(fast):
    """A FULL 64 x 64 Jacobian (every pivot step updates every trailing column) through BDF on the wave-per-trajectory
    path: factors of (I - cJ) resident in LDS (variant 0 = default for n <= 128) and in global memory (variant 1) -- the
    same lu_decomp / lin_solve code on two address spaces -- both bit-exact vs the oracle (src/matrix/lu.rs:37-125,
    linear.rs:55-96), including njev / nlu, for chunk lengths that make the factors travel LDS -> memory -> LDS between
    launches while they are still current."""
    y0, k = _dense64_batch(7)
    o = dict(method="BDF", rtol=1e-6, atol=1e-9)
    ref = oracle_batch("dense64", y0, k, 0.0, 0.6, fma=fast, **o)
    assert (ref["status"] == 0).all() and (ref["nlu"] > 3).all()
    for variant in (0, 1):
        for chunk in (0, 1, 5):
            got = gpu_batch("dense64", y0, k, 0.0, 0.6, variant=variant, chunk=chunk, fast=fast, **o)
            assert_bitexact(got, ref, f"dense64 BDF variant {variant} chunk {chunk}: ")
    # and the explicit methods on the same system (wave-per-trajectory RK kernels)
    for method in ("DOPRI5", "DOP853"):
        ref = oracle_batch("dense64", y0, k, 0.0, 0.6, fma=fast, method=method, rtol=1e-7, atol=1e-10)
        got = gpu_batch("dense64", y0, k, 0.0, 0.6, fast=fast, method=method, rtol=1e-7, atol=1e-10)
        assert_bitexact(got, ref, f"dense64 {method}: ")


def test_bdf_n100_lds_factors_equal_global_factors_and_the_oracle():
    y0, p, t0, t1 = _decay_batch(9)
    o = dict(method="BDF", rtol=1e-5, atol=1e-8)
    ref = oracle_batch("linear_decay100", y0, p, t0, t1, **o)
    for variant in (0, 1):
        for chunk in (0, 7):
            got = gpu_batch("linear_decay100", y0, p, t0, t1, variant=variant, chunk=chunk, **o)
            assert_bitexact(got, ref, f"N=100 BDF variant {variant} chunk {chunk}: ")


@pytest.mark.parametrize("rhs", ["dense64", "linear_decay100"])
def test_bdf_large_batches_automatic_factor_placement_changes_no_bit(rhs):
    """Batches beyond two wavefronts per CU: the launch loop starts with ONE short launch, reads how dense the eliminations
    are (the counters behind err_flag) and keeps the factors in LDS for a dense Jacobian, in global memory for a sparse one
    (ivp_capi.cpp, enqueue_round).  Whatever it picks, launch by launch: the bits of variant 1 (global) and variant 2 (LDS), and
    of the oracle on a sample of the trajectories."""
    B = 1500
    if rhs == "dense64":
        y0, p = _dense64_batch(B)
        t0, t1, o = 0.0, 0.3, dict(method="BDF", rtol=1e-6, atol=1e-9)
    else:
        y0, p, t0, t1 = _decay_batch(B)
        o = dict(method="BDF", rtol=1e-5, atol=1e-8)
    runs = {v: gpu_batch(rhs, y0, p, t0, t1, variant=v, **o) for v in (0, 1, 2)}
    for v in (1, 2):
        for k in ("y_end", "t_end", "h_next", "status", "nfev", "njev", "nlu", "naccpt", "nrejct"):
            assert np.array_equal(runs[0][k], runs[v][k]), (rhs, v, k)
    idx = np.arange(0, B, 250)
    ref = oracle_batch(rhs, y0[:, idx], None if p is None else p[:, idx], t0, t1 if np.ndim(t1) == 0 else t1[idx], **o)
    sub = {k: (v[..., idx] if isinstance(v, np.ndarray) and v.shape and v.shape[-1] == B else v) for k, v in runs[0].items()}
    assert_bitexact(sub, ref, f"{rhs} automatic factor placement: ")
