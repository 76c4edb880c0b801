"""The reference's Python test-suite (tests/test_ivp.py, adapted there from SciPy's) restated against
``ivp_amd.pyfront.solve_ivp``: same arguments, same assertions, with the Python callables written as device code."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal

from ivp_amd import api
from ivp_amd.pyfront import Event, OdeResult, solve_ivp

gpu = pytest.mark.gpu
EXPLICIT = ["RK23", "RK45", "DOP853"]
METHODS = EXPLICIT + ["BDF"]

FUN_RATIONAL = "dydx[0] = y[1] / x; dydx[1] = y[1] * (y[0] + 2 * y[1] - 1) / (x * (y[0] - 1));"   # test_ivp.py:44-46
JAC_RATIONAL = ("j[0] = 0; j[1] = 1 / x; j[2] = -2 * y[1] * y[1] / (x * (y[0] - 1) * (y[0] - 1));"
                "j[3] = (y[0] + 4 * y[1] - 1) / (x * (y[0] - 1));")                                 # test_ivp.py:55-60
FUN_LINEAR = "dydx[0] = -y[0] - 5 * y[1]; dydx[1] = y[0] + y[1];"                                  # test_ivp.py:31-32


def sol_rational(t):
    return np.asarray((t / (t + 10), 10 * t / (t + 10) ** 2))


def sol_linear(t):
    return np.vstack((-5 * np.sin(2 * t), 2 * np.cos(2 * t) + np.sin(2 * t)))


def compute_error(y, y_true, rtol, atol):   # test_ivp.py:146-148
    e = (y - y_true) / (atol + rtol * np.abs(y_true))
    return np.linalg.norm(e, axis=0) / np.sqrt(e.shape[0])


# ---- host-only cases: these never reach an integrator, so they run without a GPU --------------------------------
@pytest.mark.parametrize("method", METHODS)
def test_no_integration(method):   # test_ivp.py:704-709
    sol = solve_ivp("dydx[0] = -y[0]; dydx[1] = -y[1];", [4, 4], [2, 3], method=method, dense_output=True)
    assert_equal(sol.sol(4), [2, 3])
    assert_equal(sol.sol([4, 5, 6]), [[2, 2, 2], [3, 3, 3]])


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("tf", [10, np.inf])
def test_empty(method, tf):   # test_ivp.py:712-728
    sol = solve_ivp("", [0, tf], np.zeros((0,)), method=method, dense_output=True)
    assert_equal(sol.sol(10), np.zeros((0,)))
    assert_equal(sol.sol([1, 2, 3]), np.zeros((0, 3)))


@pytest.mark.parametrize("method", METHODS)
def test_zero_interval(method):   # test_ivp.py:874-883
    res = solve_ivp("dydx[0] = 2 * y[0];", (0.0, 0.0), np.array([1.0]), method=method)
    assert res.success
    assert_allclose(res.y[0, -1], 1.0)
    assert res.y.shape == (1, 1) and res.t_events is None and res.sol is None


def test_result_object_and_argument_checks():   # result.rs:60-99, solve.rs:150-166
    res = solve_ivp("dydx[0] = 0;", (1.0, 1.0), [3.0], events=Event("y[0] - 1", terminal=True))
    assert isinstance(res, OdeResult)
    assert res["status"] == res.status == 0 and res["message"] == "Success" and res["success"] is True
    assert res["sol"] is None and res["t_events"] is res.t_events and len(res.t_events) == 1
    assert "message: Success" in repr(res) and "nfev: 0" in repr(res)
    with pytest.raises(KeyError):
        res["nope"]
    with pytest.raises(NotImplementedError):
        solve_ivp("dydx[0] = 0;", (0, 1), [1.0], jac_sparsity=np.eye(1))
    with pytest.raises(TypeError):
        solve_ivp(lambda t, y: -y, (0, 1), [1.0])
    with pytest.raises(ValueError):
        solve_ivp(api.ExponentialDecay(0.5), (0, 0), [1.0], args=(1.0,))


def test_device_source_assembly(monkeypatch):
    """What pyfront hands to DeviceIVP: the ode body wrapped, one `events` function from the Event expressions with the
    reference's attribute semantics (solve.rs:246-289: `terminal` only as a bool, `direction` truncated to its sign), a
    constant Jacobian written out row-major, `args` as parameter values; the Jacobian is dropped for explicit methods."""
    from ivp_amd import pyfront
    seen = {}

    class Recorder(api.IVP):
        def __init__(self, source, n, params=(), ctx=None, events=(), jac=False):
            seen.update(source=source, n=n, params=params, events=list(events), jac=jac)
            raise api.ConfigError(-100, "recorded")

    monkeypatch.setattr(api, "DeviceIVP", Recorder)
    evs = [Event("y[0]", direction=-1), Event("y[1] - p[0]", terminal=True, direction=2.9), Event("x - 3", terminal=1)]
    # a problem that cannot be built surfaces like any other solver failure: RuntimeError("Solver failed: ..") (solve.rs:216-221)
    with pytest.raises(RuntimeError, match="Solver failed"):
        solve_ivp("dydx[0] = y[1]; dydx[1] = -p[0] * y[0];", (0, 1), [1.0, 0.0], method="BDF", events=evs, args=(4,),
                  jac=np.array([[0, 1], [-4, 0]]))
    assert seen["n"] == 2 and seen["params"] == (4.0,) and seen["jac"] is True
    src = seen["source"]
    assert "__device__ void ode(double x, const double* y, double* dydx, const double* p)" in src
    assert "g[0] = (y[0]);" in src and "g[1] = (y[1] - p[0]);" in src and "g[2] = (x - 3);" in src
    assert "j[0] = 0.0;" in src and "j[1] = 1.0;" in src and "j[2] = -4.0;" in src and "j[3] = 0.0;" in src
    cfg = seen["events"]
    assert (cfg[0].direction, cfg[0].terminal_count) == (api.Direction.Negative, None)
    assert (cfg[1].direction, cfg[1].terminal_count) == (api.Direction.Positive, 1)
    assert (cfg[2].direction, cfg[2].terminal_count) == (api.Direction.All, None)      # terminal = 1 is not a bool
    with pytest.raises(RuntimeError, match="Solver failed"):
        solve_ivp("dydx[0] = -y[0];", (0, 1), [1.0], method="RK45", jac="j[0] = -1;")
    assert seen["jac"] is False and "void jac" not in seen["source"]
    with pytest.raises(ValueError):
        solve_ivp("dydx[0] = -y[0];", (0, 1), [1.0], method="BDF", jac=np.zeros((2, 2)))


# ---- the path proper ------------------------------------------------------------------------------------------------
@gpu
@pytest.mark.parametrize("method", METHODS)
def test_integration_zero_rhs(method):   # test_ivp.py:844-849
    result = solve_ivp("dydx[0] = 0; dydx[1] = 0; dydx[2] = 0;", [0, 10], np.ones(3), method=method)
    assert result.success
    assert_equal(result.status, 0)
    assert_allclose(result.y, 1.0, rtol=1e-15)
    assert result.y.shape == (3, result.t.size)


@gpu
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("t_span", [[5, 9], [5, 1]])
@pytest.mark.parametrize("jac", [None, JAC_RATIONAL])
def test_integration(method, t_span, jac):   # test_ivp.py:173-241 (vectorized / sparse-jac legs collapse: one device RHS)
    rtol, atol = 1e-3, 1e-6
    res = solve_ivp(FUN_RATIONAL, t_span, [1 / 3, 2 / 9], rtol=rtol, atol=atol, method=method, dense_output=True,
                    jac=jac, vectorized=False)
    assert_equal(res.t[0], t_span[0])
    assert res.t_events is None and res.y_events is None
    assert res.success
    assert_equal(res.status, 0)
    if method == "DOP853":
        assert res.nfev < 50
    if method in EXPLICIT:
        assert_equal(res.njev, 0)
        assert_equal(res.nlu, 0)
    else:
        assert 0 < res.njev
        assert 0 < res.nlu
    e = compute_error(res.y, sol_rational(res.t), rtol, atol)
    assert np.all(e < 5)
    tc = np.linspace(*t_span)
    e = compute_error(res.sol(tc), sol_rational(tc), rtol, atol)
    assert np.all(e < 5)
    tc = (t_span[0] + t_span[-1]) / 2
    e = compute_error(res.sol(tc), sol_rational(tc), rtol, atol)
    assert np.all(e < 5)
    assert_allclose(res.sol(res.t), res.y, rtol=1e-15, atol=1e-15)


@gpu
def test_integration_const_jac():   # test_ivp.py:272-317 (BDF leg; Radau is outside the path)
    rtol, atol = 1e-3, 1e-6
    t_span = [0, 2]
    res = solve_ivp(FUN_LINEAR, t_span, [0, 2], rtol=rtol, atol=atol, method="BDF", dense_output=True,
                    jac=np.array([[-1, -5], [1, 1]]))
    assert_equal(res.t[0], t_span[0])
    assert res.t_events is None and res.y_events is None and res.success
    assert_equal(res.status, 0)
    assert res.nfev < 100
    assert_equal(res.njev, 0)
    e = compute_error(res.y, sol_linear(res.t), rtol, atol)
    assert np.all(e < 10)
    tc = np.linspace(*t_span)
    e = compute_error(res.sol(tc), sol_linear(tc), rtol, atol)
    assert np.all(e < 60)
    assert_allclose(res.sol(res.t), res.y, rtol=1e-14, atol=1e-14)


@gpu
def test_integration_stiff():   # test_ivp.py:319-342
    res = solve_ivp("dydx[0] = -0.04 * y[0] + 1e4 * y[1] * y[2];"
                    "dydx[1] = 0.04 * y[0] - 1e4 * y[1] * y[2] - 3e7 * y[1] * y[1];"
                    "dydx[2] = 3e7 * y[1] * y[1];", [0, 1e8], [1e4, 0, 0], rtol=1e-6, atol=1e-6, method="BDF")
    assert res.nfev < 5000
    assert res.njev < 200
    assert res.success


@gpu
def test_duplicate_timestamps():   # test_ivp.py:152-170
    sol = solve_ivp("dydx[0] = y[1]; dydx[1] = -9.80665;", [0, np.inf], [0, 0.01], max_step=0.05 * 0.001 / 9.80665,
                    events=Event("y[0]", terminal=True, direction=-1), dense_output=True)
    assert_allclose(sol.sol(0.01), np.asarray([-0.00039033, -0.08806632]), rtol=1e-5, atol=1e-8)
    assert_allclose(sol.t_events[0], np.asarray([0.00203943]), rtol=1e-5, atol=1e-8)
    assert sol.success
    assert_equal(sol.status, 1)
    assert sol.message == "UserInterrupt"


@gpu
def test_t_eval():   # test_ivp.py:586-645
    rtol, atol = 1e-3, 1e-6
    y0 = [1 / 3, 2 / 9]
    cases = [([5, 9], np.linspace(5, 9, 10), True), ([5, 1], np.linspace(5, 1, 10), True),
             ([5, 9], [5, 5.01, 7, 8, 8.01, 9], True), ([5, 1], [5, 4.99, 3, 1.5, 1.1, 1.01, 1], False),
             ([5, 9], [5.01, 7, 8, 8.01], True), ([5, 1], [4.99, 3, 1.5, 1.1, 1.01], False)]
    for t_span, t_eval, check_error in cases:
        res = solve_ivp(FUN_RATIONAL, t_span, y0, rtol=rtol, atol=atol, t_eval=t_eval)
        assert_equal(res.t, t_eval)
        assert res.t_events is None and res.success
        assert_equal(res.status, 0)
        if check_error:
            e = compute_error(res.y, sol_rational(res.t), rtol, atol)
            assert np.all(e < 5)


@gpu
def test_t_eval_dense_output():   # test_ivp.py:648-672
    rtol, atol = 1e-3, 1e-6
    t_span = [5, 9]
    t_eval = np.linspace(t_span[0], t_span[1], 10)
    res = solve_ivp(FUN_RATIONAL, t_span, [1 / 3, 2 / 9], rtol=rtol, atol=atol, t_eval=t_eval)
    res_d = solve_ivp(FUN_RATIONAL, t_span, [1 / 3, 2 / 9], rtol=rtol, atol=atol, t_eval=t_eval, dense_output=True)
    assert_equal(res.t, t_eval)
    assert res.t_events is None and res.success
    assert_equal(res.t, res_d.t)
    assert_equal(res.y, res_d.y)
    assert res_d.t_events is None and res_d.success
    assert_equal(res_d.status, 0)
    e = compute_error(res.y, sol_rational(res.t), rtol, atol)
    assert np.all(e < 5)


@gpu
@pytest.mark.parametrize("method", METHODS)
def test_t_eval_early_event(method):   # test_ivp.py:675-701
    early_event = Event("x - 7", terminal=True)
    res = solve_ivp(FUN_RATIONAL, [5, 9], [1 / 3, 2 / 9], rtol=1e-3, atol=1e-6, method=method,
                    t_eval=np.linspace(7.5, 9, 16), events=early_event, jac=JAC_RATIONAL)
    assert res.success
    assert res.status == 1
    assert len(res.t_events) == 1
    assert res.t_events[0].size == 1
    assert res.t_events[0][0] == 7


@gpu
@pytest.mark.parametrize("method", ["DOP853", "BDF"])
def test_args(method):   # test_ivp.py:731-820 (Radau in the reference: DOP853 at its tolerances, BDF with the Jacobian)
    omega, k, tfinal, zfinal = 2, 4, 5, 0.99
    z0 = np.exp(-k * tfinal) / ((1 - zfinal) / zfinal + np.exp(-k * tfinal))
    tight = method == "DOP853"
    sol = solve_ivp("dydx[0] = -p[0] * y[1]; dydx[1] = p[0] * y[0]; dydx[2] = p[1] * y[2] * (1 - y[2]);",
                    [0, 2 * tfinal], [0, -1, z0],
                    events=[Event("y[0]", direction=-1), Event("y[1]", direction=1), Event("y[2] - p[2]", terminal=True)],
                    dense_output=True, args=(omega, k, zfinal), method=method,
                    jac="j[0] = 0; j[1] = -p[0]; j[2] = 0; j[3] = p[0]; j[4] = 0; j[5] = 0;"
                        "j[6] = 0; j[7] = 0; j[8] = p[1] * (1 - 2 * y[2]);",
                    rtol=1e-10 if tight else 1e-8, atol=1e-13 if tight else 1e-11)
    x0events_t, y0events_t, zfinalevents_t = sol.t_events
    tol = dict(rtol=1e-7) if tight else dict(rtol=1e-4)
    assert_allclose(x0events_t, [0.5 * np.pi, 1.5 * np.pi], **tol)
    assert_allclose(y0events_t, [0.25 * np.pi, 1.25 * np.pi], **tol)
    assert_allclose(zfinalevents_t, [tfinal], rtol=1e-5, atol=1e-5)
    assert sol.status == 1
    t = np.linspace(0, zfinalevents_t[0], 250)
    w = sol.sol(t)
    tol = dict(rtol=1e-5, atol=1e-6) if tight else dict(rtol=1e-3, atol=1e-4)
    assert_allclose(w[0], np.sin(omega * t), **tol)
    assert_allclose(w[1], -np.cos(omega * t), **tol)
    assert_allclose(w[2], 1 / (((1 - z0) / z0) * np.exp(-k * t) + 1), **tol)
    x0events, y0events, zfinalevents = sol.sol(x0events_t), sol.sol(y0events_t), sol.sol(zfinalevents_t)
    a0, a1 = (1e-10, 1e-6) if tight else (1e-6, 1e-4)
    assert_allclose(x0events[0], np.zeros_like(x0events[0]), atol=a0)
    assert_allclose(x0events[1], np.ones_like(x0events[1]), atol=a1)
    assert_allclose(y0events[0], np.ones_like(y0events[0]), atol=a1)
    assert_allclose(y0events[1], np.zeros_like(y0events[1]), atol=a0)
    assert_allclose(zfinalevents[2], [zfinal], atol=a1)
    # y_events rows are the states at the event times (solve.rs:383-400)
    assert sol.y_events[0].shape == (2, 3) and sol.y_events[2].shape == (1, 3)
    assert_allclose(sol.y_events[2][0, 2], zfinal, atol=1e-9)


@gpu
def test_array_rtol():   # test_ivp.py:824-841
    f = "dydx[0] = y[0]; dydx[1] = y[1];"
    sol = solve_ivp(f, (0, 1), [1., 1.], rtol=[1e-1, 1e-1])
    err1 = np.abs(np.linalg.norm(sol.y[:, -1] - np.exp(1)))
    sol = solve_ivp(f, (0, 1), [1., 1.], rtol=[1e-1, 1e-16])
    err2 = np.abs(np.linalg.norm(sol.y[:, -1] - np.exp(1)))
    assert err2 < err1


@gpu
def test_args_single_value():   # test_ivp.py:852-861
    sol = solve_ivp("dydx[0] = p[0] * y[0];", (0, 0.1), [1], args=(-1,))
    assert_allclose(sol.y[0, -1], np.exp(-0.1))


@gpu
@pytest.mark.parametrize("method", METHODS)
def test_max_step_first_step(method):   # test_ivp.py:521-583
    rtol, atol = 1e-3, 1e-6
    y0 = [1 / 3, 2 / 9]
    for t_span in ([5, 9], [5, 1]):
        res = solve_ivp(FUN_RATIONAL, t_span, y0, rtol=rtol, max_step=0.5, atol=atol, method=method, dense_output=True)
        assert_equal(res.t[0], t_span[0])
        assert_equal(res.t[-1], t_span[-1])
        assert np.all(np.abs(np.diff(res.t)) <= 0.5 + 1e-15)
        assert res.t_events is None and res.success
        e = compute_error(res.y, sol_rational(res.t), rtol, atol)
        assert np.all(e < 5)
        tc = np.linspace(*t_span)
        e = compute_error(res.sol(tc), sol_rational(tc), rtol, atol)
        assert np.all(e < 5)
        assert_allclose(res.sol(res.t), res.y, rtol=1e-15, atol=1e-15)
        first_step = 0.1
        res = solve_ivp(FUN_RATIONAL, t_span, y0, rtol=rtol, max_step=0.5, atol=atol, method=method, dense_output=True,
                        first_step=first_step)
        assert_allclose(first_step, np.abs(res.t[1] - 5))
        assert res.t_events is None and res.success
        e = compute_error(res.y, sol_rational(res.t), rtol, atol)
        assert np.all(e < 5)


@gpu
@pytest.mark.parametrize("method", METHODS)
def test_tbound_respected(method):   # test_ivp.py:885-949
    res = solve_ivp("dydx[0] = 1 / sqrt(1 - x);", (0.0, 1 - 1e-8), [0.0], method=method, rtol=1e-6, atol=1e-9)
    assert res.success and res.t[-1] <= 1 - 1e-8
    res = solve_ivp("dydx[0] = y[1]; dydx[1] = -y[0];", (0.0, 2 * np.pi), [1.0, 0.0], method=method, rtol=1e-8, atol=1e-10)
    assert res.success and res.t[-1] == 2 * np.pi
    assert_allclose(res.y[:, -1], [1.0, 0.0], atol=2e-4 if method == "BDF" else 1e-5)


@gpu
def test_builtin_problem_and_failure_status():   # an IVP instance as `fun`; status -1 / message (solve.rs:405-428)
    res = solve_ivp(api.ExponentialDecay(0.5), (0, 10), [2.0], t_eval=[0, 5, 10])
    assert_allclose(res.y[0], 2 * np.exp(-0.5 * np.array([0, 5, 10.0])), rtol=5e-3)
    res = solve_ivp(api.BouncingBall(), (0, 10), [10.0, 0.0])
    assert res.status == 1 and res.t_events[0].size == 1 and res.y_events[0].shape == (1, 2)
    res = solve_ivp(api.VanDerPol(1.0), (0, 100), [2.0, 0.0], max_steps=5)
    assert res.status == -1 and not res.success and res.message == "NeedLargerNMax"


@gpu
def test_nine_args():
    """`args` longer than the four values of round 1 (the fields of the user's struct are per-trajectory parameters, up to
    16 of them): a 3 x 3 linear system whose matrix arrives through `args`, against expm and, bit for bit, the oracle."""
    from scipy.linalg import expm
    from oracle import oracle as O
    A = np.array([[-0.5, 2.0, 0.1], [-2.0, -0.3, 0.4], [0.2, -0.1, -1.5]])
    src = ("dydx[0] = p[0] * y[0] + p[1] * y[1] + p[2] * y[2];"
           "dydx[1] = p[3] * y[0] + p[4] * y[1] + p[5] * y[2];"
           "dydx[2] = p[6] * y[0] + p[7] * y[1] + p[8] * y[2];")
    y0 = [1.0, -0.5, 0.25]
    res = solve_ivp(src, (0, 3), y0, method="DOP853", args=tuple(A.ravel()), rtol=1e-10, atol=1e-12)
    assert res.success
    assert_allclose(res.y[:, -1], expm(3.0 * A) @ np.array(y0), rtol=1e-8, atol=1e-10)

    def fun(t, y, p):
        return [p[0] * y[0] + p[1] * y[1] + p[2] * y[2], p[3] * y[0] + p[4] * y[1] + p[5] * y[2], p[6] * y[0] + p[7] * y[1] + p[8] * y[2]]

    ref = O.solve_ivp(fun, 0.0, 3.0, y0, params=list(A.ravel()), method="DOP853", rtol=1e-10, atol=1e-12, detpow=True)
    assert np.array_equal(res.t, ref.t) and np.array_equal(res.y.T, ref.y) and res.nfev == ref.nfev


@pytest.mark.gpu
def test_statement_form_with_more_than_eight_states_and_a_statement_jacobian():
    """The statement form of `fun` (and of `jac`) for a 12-state system: wrapped into the component / column forms the
    wave-per-trajectory kernels ask for.  RK45 and BDF agree with the closed form; a snippet that does not compile
    surfaces as the reference's RuntimeError("Solver failed: ...") (src/python/solve.rs:216-221)."""
    from ivp_amd.pyfront import solve_ivp
    n = 12
    body = "\n".join(f"dydx[{i}] = -p[0] * {i + 1}.0 * y[{i}];" for i in range(n))
    jac = "\n".join(f"j[{i * n + i}] = -p[0] * {i + 1}.0;" for i in range(n))
    y0 = np.linspace(1.0, 2.0, n)
    exact = y0 * np.exp(-0.3 * np.arange(1, n + 1) * 2.0)
    r = solve_ivp(body, (0.0, 2.0), y0, method="RK45", args=(0.3,), rtol=1e-8, atol=1e-10)
    assert r.status == 0 and r.y.shape[0] == n
    np.testing.assert_allclose(r.y[:, -1], exact, rtol=1e-6)
    b = solve_ivp(body, (0.0, 2.0), y0, method="BDF", args=(0.3,), jac=jac, rtol=1e-7, atol=1e-10)
    assert b.status == 0 and b.njev >= 1
    np.testing.assert_allclose(b.y[:, -1], exact, rtol=1e-4)
    with pytest.raises(RuntimeError, match="Solver failed"):
        solve_ivp("dydx[0] = this does not compile;", (0.0, 1.0), np.ones(n), method="RK45")


@pytest.mark.gpu
def test_large_statement_and_matrix_jacobians_cost_no_private_matrix():
    """The advertised range of the statement form (n up to 512) without an n x n private array per lane: a 120-state
    chain y_i' = -k (i + 1) y_i + c y_{i-1} through BDF with (a) the statement-form Jacobian (O(1) proxy: a write lands in
    the requested column or in a sink), (b) the same Jacobian as a constant matrix (one switch case per column), (c) the
    default forward differences.  (a) and (b) describe the same matrix, so BDF takes identical steps; all agree with the
    closed-form first component and with each other to the tolerance of the run."""
    from ivp_amd.pyfront import solve_ivp
    n = 120
    body = "dydx[0] = -p[0] * y[0];\n" + "\n".join(f"dydx[{i}] = -p[0] * {i + 1}.0 * y[{i}] + p[1] * y[{i - 1}];" for i in range(1, n))
    jac_s = "j[0] = -p[0];\n" + "\n".join(f"j[{i * n + i}] = -p[0] * {i + 1}.0; j[{i * n + i - 1}] = p[1];" for i in range(1, n))
    k, c = 0.05, 0.02
    J = np.diag(-k * np.arange(1, n + 1)) + np.diag(np.full(n - 1, c), -1)
    y0 = np.linspace(1.0, 2.0, n)
    kw = dict(method="BDF", args=(k, c), rtol=1e-6, atol=1e-9)
    a = solve_ivp(body, (0.0, 3.0), y0, jac=jac_s, **kw)
    b = solve_ivp(body, (0.0, 3.0), y0, jac=J, **kw)
    d = solve_ivp(body, (0.0, 3.0), y0, **kw)
    assert a.status == 0 and b.status == 0 and d.status == 0 and a.njev >= 1
    assert np.array_equal(a.t, b.t) and np.array_equal(a.y, b.y) and a.nfev == b.nfev and a.nlu == b.nlu
    np.testing.assert_allclose(a.y[0, -1], y0[0] * np.exp(-k * 3.0), rtol=1e-4)
    np.testing.assert_allclose(a.y[:, -1], d.y[:, -1], rtol=1e-4, atol=1e-8)
    assert d.njev == a.njev    # same step sequence up to the difference quotient's rounding: same number of Jacobian calls
