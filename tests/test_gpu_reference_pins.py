"""The reference's own explicit-RK tests, restated against the PRODUCT API (ivp_amd.solve_ivp -> libivp_hip.so ->
gfx950 kernels) so that they read like the crate's tests (tests/accuracy.rs, tests/ivp.rs,
tests/backward_and_bounds.rs and the explicit-RK cases of tests/test_ivp.py / test_step_control.py /
test_t_eval.py).  Run with -m gpu."""
import numpy as np
import pytest

import ivp_amd
from ivp_amd import (SHO, ExponentialDecay, Exp2, LinearSystem, Method, Options, Rational, Robertson, Status,
                     StiffVanDerPol, VanDerPol, ZeroRhs, solve_ivp)

pytestmark = pytest.mark.gpu
EXPLICIT = [Method.RK23, Method.DOPRI5, Method.DOP853]


def default_opts(method, **kw):  # tests/common.rs:21-28
    return Options(method=method, rtol=1e-9, atol=1e-9, **kw)


def sol_rational(t):
    t = np.asarray(t)
    return np.asarray((t / (t + 10), 10 * t / (t + 10) ** 2))


def compute_error(y, y_true, rtol, atol):
    e = (y - y_true) / (atol + rtol * np.abs(y_true))
    return np.linalg.norm(e, axis=0) / np.sqrt(e.shape[0])


@pytest.mark.parametrize("method", EXPLICIT + [Method.RK4])
def test_harmonic_accuracy_end_state(method):  # tests/accuracy.rs:18-48
    if method == Method.RK4:   # fixed step chosen to land on the period (accuracy.rs:24-31)
        opts = Options(method=method, first_step=2 * np.pi / 2000.0)
    else:
        opts = default_opts(method)
    sol = solve_ivp(SHO(), 0.0, 2 * np.pi, [1.0, 0.0], opts)
    y_end = sol.y[-1]
    assert abs(y_end[0] - 1.0) < 1e-5 and abs(y_end[1]) < 1e-5


@pytest.mark.parametrize("method", EXPLICIT + [Method.RK4])
def test_t_eval_sampling_exact_times(method):  # tests/accuracy.rs:51-77
    t_eval = [i / 10.0 for i in range(11)]
    sol = solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(method=method, rtol=1e-9, atol=1e-9, t_eval=t_eval))
    for te in t_eval:
        assert np.any(np.abs(sol.t - te) <= 1e-9)
    assert len(sol.y) == len(sol.t)


def test_iterate_samples():  # tests/accuracy.rs:80-89
    sol = solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], default_opts(Method.DOPRI5))
    for t, y in sol.iter():
        assert 0.0 <= t <= 1.0 and len(y) == 2


@pytest.mark.parametrize("method", EXPLICIT)
def test_integration_zero_rhs(method):  # tests/ivp.rs:21-46
    t_eval = [10.0 * i / 20.0 for i in range(21)]
    sol = solve_ivp(ZeroRhs(), 0.0, 10.0, [1.0, 1.0, 1.0], Options(method=method, rtol=1e-9, atol=1e-12, t_eval=t_eval))
    assert np.array_equal(sol.t, t_eval)
    assert np.abs(sol.y - 1.0).max() <= 1e-12


@pytest.mark.parametrize("method", EXPLICIT)
def test_max_step_and_first_step_controls(method):  # tests/ivp.rs:49-104
    sol = solve_ivp(SHO(), 0.0, 3.0, [1.0, 0.0], Options(method=method, rtol=1e-6, atol=1e-9, max_step=0.05))
    assert np.abs(np.diff(sol.t)).max() <= 0.05 + 1e-12
    sol = solve_ivp(SHO(), 0.0, 3.0, [1.0, 0.0], Options(method=method, rtol=1e-3, atol=1e-6, first_step=0.1))
    assert len(sol.t) >= 2 and abs(abs(sol.t[1] - sol.t[0]) - 0.1) <= 1e-6


@pytest.mark.parametrize("method", EXPLICIT)
def test_dense_output_matches_discrete_samples(method):  # tests/ivp.rs:107-136
    sol = solve_ivp(SHO(), 0.0, 2.0, [1.0, 0.0], Options(method=method, rtol=1e-8, atol=1e-10, dense_output=True))
    assert sol.sol_span() is not None
    ys = sol.sol_many(sol.t)
    assert ys.shape == sol.y.shape and np.abs(ys - sol.y).max() <= 1e-8


def test_dense_output_out_of_range_errors():  # tests/ivp.rs:139-149
    sol = solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], default_opts(Method.DOPRI5, dense_output=True))
    t0, t1 = sol.sol_span()
    with pytest.raises(ivp_amd.InterpolationError):
        sol.sol(t0 - 0.1)
    with pytest.raises(ivp_amd.InterpolationError):
        sol.sol(t1 + 0.1)


@pytest.mark.parametrize("method", EXPLICIT)
def test_zero_interval_returns_initial_state(method):  # tests/ivp.rs:278-289
    sol = solve_ivp(SHO(), 1.23, 1.23, [2.0, 3.0], default_opts(method))
    assert len(sol.t) >= 1 and np.abs(sol.y[-1] - [2.0, 3.0]).max() <= 1e-12


def test_vector_rtol_componentwise_control():  # tests/ivp.rs:291-334
    loose = solve_ivp(Exp2(), 0.0, 1.0, [1.0, 1.0], Options(method=Method.DOPRI5, rtol=[1e-2, 1e-2], atol=1e-10))
    tight = solve_ivp(Exp2(), 0.0, 1.0, [1.0, 1.0], Options(method=Method.DOPRI5, rtol=[1e-2, 1e-10], atol=1e-10))
    e = np.e
    assert abs(tight.y[-1][1] - e) < abs(loose.y[-1][1] - e) * 0.5
    assert abs(tight.y[-1][0] - e) <= 10.0 * abs(loose.y[-1][0] - e)


@pytest.mark.parametrize("method", EXPLICIT)
def test_backward_integration_works(method):  # tests/backward_and_bounds.rs:7-32
    sol = solve_ivp(SHO(), 2 * np.pi, 0.0, [1.0, 0.0], default_opts(method, dense_output=True))
    t0, t1 = sol.sol_span()
    assert t0 > t1
    mid = 0.5 * (t0 + t1)
    y_mid = sol.sol(mid)
    assert abs(y_mid[0] - np.cos(mid)) < 1e-6 and abs(y_mid[1] + np.sin(mid)) < 1e-6


@pytest.mark.parametrize("method", EXPLICIT)
@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_integration_rational(method, t_span):  # tests/test_ivp.py:173-241
    rtol, atol = 1e-3, 1e-6
    res = solve_ivp(Rational(), t_span[0], t_span[1], [1 / 3, 2 / 9],
                    Options(method=method, rtol=rtol, atol=atol, dense_output=True))
    assert res.t[0] == t_span[0] and res.status == Status.Success
    if method == Method.DOP853:
        assert res.nfev < 50
    assert res.njev == 0 and res.nlu == 0
    assert np.all(compute_error(res.y.T, sol_rational(res.t), rtol, atol) < 5)
    tc = np.linspace(*t_span)
    yc = np.array([res.continuous_sol.evaluate_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_rational(tc), rtol, atol) < 5)
    ys = np.array([res.continuous_sol.evaluate_extrapolate(t) for t in res.t])
    np.testing.assert_allclose(ys, res.y, rtol=1e-15, atol=1e-15)


@pytest.mark.parametrize("method", EXPLICIT)
@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_max_step_and_first_step_python(method, t_span):  # tests/test_ivp.py:521-583
    res = solve_ivp(Rational(), t_span[0], t_span[1], [1 / 3, 2 / 9],
                    Options(method=method, rtol=1e-3, atol=1e-6, max_step=0.5, first_step=0.1, dense_output=True))
    assert res.t[0] == t_span[0] and res.t[-1] == t_span[-1]
    assert np.all(np.abs(np.diff(res.t)) <= 0.5 + 1e-15)
    np.testing.assert_allclose(0.1, abs(res.t[1] - 5.0))
    assert res.status == Status.Success


@pytest.mark.parametrize("method", EXPLICIT)
def test_max_steps_parameter(method):  # tests/test_step_control.py:93-109
    res = solve_ivp(ExponentialDecay(1.0), 0.0, 10.0, [1.0], Options(method=method, max_steps=1))
    assert res.status == Status.NeedLargerNMax and not res.status.is_success()
    with pytest.raises(ivp_amd.ConfigError) as e:       # nmax == 0 => Err(Config(MustBePositive)), dopri5.rs:183-189
        solve_ivp(ExponentialDecay(1.0), 0.0, 10.0, [1.0], Options(method=method, max_steps=0))
    assert e.value.code == -1


@pytest.mark.parametrize("method", EXPLICIT)
def test_default_max_steps_is_unlimited(method):  # tests/test_step_control.py:130-159
    res = solve_ivp(ExponentialDecay(0.001), 0.0, 1e5, [1.0], Options(method=method, rtol=1e-8, atol=1e-10))
    assert res.status == Status.Success and res.t[-1] == 1e5
    assert abs(res.y[-1][0] - np.exp(-100.0)) < 1e-8      # y has decayed below atol by then


@pytest.mark.parametrize("method", EXPLICIT)
def test_t_eval_python(method):  # tests/test_ivp.py:586-672, tests/test_t_eval.py:9-134
    y0 = [1 / 3, 2 / 9]
    for t_span in ((5.0, 9.0), (5.0, 1.0)):
        te = np.linspace(*t_span, 10)
        res = solve_ivp(Rational(), t_span[0], t_span[1], y0, Options(method=method, rtol=1e-3, atol=1e-6, t_eval=te))
        assert np.array_equal(res.t, te) and res.status == Status.Success
        assert np.all(compute_error(res.y.T, sol_rational(res.t), 1e-3, 1e-6) < 5)
        resd = solve_ivp(Rational(), t_span[0], t_span[1], y0,
                         Options(method=method, rtol=1e-3, atol=1e-6, t_eval=te, dense_output=True))
        assert np.array_equal(resd.y, res.y)
    res = solve_ivp(Rational(), 5.0, 9.0, y0, Options(method=method, t_eval=[5.01, 7.0, 8.0, 8.01]))
    assert np.array_equal(res.t, [5.01, 7.0, 8.0, 8.01])
    res = solve_ivp(Rational(), 5.0, 1.0, y0, Options(method=method, t_eval=[4.99, 3.0, 1.5, 1.1]))
    assert np.array_equal(res.t, [4.99, 3.0, 1.5, 1.1])


@pytest.mark.parametrize("method", EXPLICIT)
def test_no_integration_and_empty(method):  # tests/test_ivp.py:704-728
    sol = solve_ivp(Rational(), 4.0, 4.0, [2.0, 4.0], Options(method=method, dense_output=True))
    np.testing.assert_array_equal(sol.continuous_sol.evaluate_extrapolate(4.0), [2.0, 4.0])
    np.testing.assert_array_equal(sol.continuous_sol.evaluate_extrapolate(6.0), [2.0, 4.0])
    sol = solve_ivp(SHO(), 0.0, 10.0, [], Options(method=method, dense_output=True))
    assert np.array_equal(sol.t, [0.0, 10.0]) and sol.y.shape == (2, 0)


def test_c1_exponential_decay():  # BASELINE config C1: README.md:74,93-101 and examples/exponential_decay.rs:16-26
    sol = solve_ivp(ExponentialDecay(0.5), 0.0, 10.0, [1.0], Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    assert sol.status == Status.Success and sol.t[0] == 0.0 and sol.t[-1] == 10.0
    assert abs(sol.y[-1][0] - np.exp(-5.0)) < 1e-6
    te = np.arange(11.0)
    sol = solve_ivp(ExponentialDecay(0.5), 0.0, 10.0, [10.0], Options(method="DOPRI5", rtol=1e-8, atol=1e-10, t_eval=te))
    assert np.array_equal(sol.t, te)
    assert np.abs(sol.y[:, 0] - 10.0 * np.exp(-0.5 * te)).max() < 1e-6


def test_rk4_invalid_step_size_is_a_config_error():  # rk4.rs:81-87
    with pytest.raises(ivp_amd.ConfigError) as e:
        solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(method=Method.RK4, first_step=-0.1))
    assert e.value.code == -5
    sol = solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(method=Method.RK4))      # default h = (xend - x0)/100
    assert sol.nstep == 100 and sol.nfev == 400 and sol.naccpt == 0 and len(sol.t) == 101


# ---- BDF (src/methods/bdf.rs): the reference's tests that include Method::BDF ----------------------------------

def sol_linear(t):
    return np.vstack((-5 * np.sin(2 * t), 2 * np.cos(2 * t) + np.sin(2 * t)))


def test_bdf_harmonic_accuracy_backward_t_eval_max_step():  # accuracy.rs:18-77, backward_and_bounds.rs:7-32, ivp.rs:49-76
    sol = solve_ivp(SHO(), 0.0, 2 * np.pi, [1.0, 0.0], default_opts(Method.BDF))
    assert abs(sol.y[-1][0] - 1.0) < 1e-5 and abs(sol.y[-1][1]) < 1e-5 and sol.njev > 0 and sol.nlu > 0
    sol = solve_ivp(SHO(), 2 * np.pi, 0.0, [1.0, 0.0], default_opts(Method.BDF, dense_output=True))
    t0, t1 = sol.sol_span()
    mid = 0.5 * (t0 + t1)
    assert t0 > t1 and abs(sol.sol(mid)[0] - np.cos(mid)) < 1e-6
    te = [i / 10.0 for i in range(11)]
    sol = solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(method=Method.BDF, rtol=1e-9, atol=1e-9, t_eval=te))
    assert all(np.any(np.abs(sol.t - t) <= 1e-9) for t in te)
    sol = solve_ivp(SHO(), 0.0, 3.0, [1.0, 0.0], Options(method=Method.BDF, rtol=1e-6, atol=1e-9, max_step=0.05))
    assert np.abs(np.diff(sol.t)).max() <= 0.05 + 1e-12


@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_bdf_integration_rational(t_span):  # tests/test_ivp.py:173-241, tests/test_basic_integration.py:89-104
    res = solve_ivp(Rational(), t_span[0], t_span[1], [1 / 3, 2 / 9], Options(method="BDF", rtol=1e-3, atol=1e-6, dense_output=True))
    assert res.t[0] == t_span[0] and res.status == Status.Success and 0 < res.njev and 0 < res.nlu
    assert np.all(compute_error(res.y.T, sol_rational(res.t), 1e-3, 1e-6) < 5)
    ys = np.array([res.continuous_sol.evaluate_extrapolate(t) for t in res.t])
    np.testing.assert_allclose(ys, res.y, rtol=1e-15, atol=1e-15)


def test_bdf_const_jac_linear_and_robertson():  # tests/test_ivp.py:273-342, tests/test_stiff.py:35-53
    res = solve_ivp(LinearSystem(), 0.0, 2.0, [0.0, 2.0], Options(method="BDF", rtol=1e-3, atol=1e-6, dense_output=True))
    assert res.status == Status.Success and res.nfev < 100
    assert np.all(compute_error(res.y.T, sol_linear(res.t), 1e-3, 1e-6) < 10)
    tc = np.linspace(0.0, 2.0)
    yc = np.array([res.continuous_sol.evaluate_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_linear(tc), 1e-3, 1e-6) < 60)
    res = solve_ivp(Robertson(), 0.0, 1e8, [1e4, 0.0, 0.0], Options(method="BDF", rtol=1e-6, atol=1e-6))
    assert res.status == Status.Success and res.nfev < 5000 and res.njev < 200


def test_bdf_examples_van_der_pol():  # examples/van_der_pol.rs:16-41 and benches/benchmark.py:118-126
    te = [i * 0.1 for i in range(21)]
    sol = solve_ivp(StiffVanDerPol(1e-3), 0.0, 2.0, [2.0, 0.0], Options(method=Method.BDF, rtol=1e-6, atol=1e-8, t_eval=te))
    assert sol.status == Status.Success and np.array_equal(sol.t, te)
    assert np.abs(sol.y[-1] - [1.7632345402033993, -0.8356886816853318]).max() < 1e-4     # SciPy Radau @1e-10
    sol = solve_ivp(VanDerPol(1000.0), 0.0, 3000.0, [2.0, 0.0], Options(method="BDF", rtol=1e-4, atol=1e-6))
    assert sol.status == Status.Success and np.abs(sol.y[-1] - [-1.5106069367440045, 0.0011783800007311195]).max() < 1e-2


def test_unsupported_methods_and_bad_tolerances_are_config_errors():
    with pytest.raises(ivp_amd.ConfigError) as e:
        solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(method="RADAU"))
    assert e.value.code == -101
    with pytest.raises(ivp_amd.ConfigError) as e:      # Tolerance::Vector length mismatch (mod.rs:156-161)
        solve_ivp(SHO(), 0.0, 1.0, [1.0, 0.0], Options(rtol=[1e-3, 1e-3, 1e-3]))
    assert e.value.code == -4
    assert Method.from_str("rk45") == Method.DOPRI5 and Method.from_str("nonsense") == Method.DOPRI5  # options.rs:61-73


# ---- events (src/solve/solout.rs:158-331) ------------------------------------------------------------------------
from ivp_amd import BouncingBall, Cannon, Direction, EventConfig, RationalEvents, SHOZeroEvent  # noqa: E402


def test_event_detection_all_and_directional():  # tests/ivp.rs:223-275
    cfg = EventConfig(); cfg.all(); cfg.terminal_count = 2
    sol = solve_ivp(SHOZeroEvent(cfg), 0.0, 6.0, [1.0, 0.0], default_opts(Method.DOPRI5))
    zeros = [t for t, y in zip(sol.t_events[0], sol.y_events[0]) if abs(y[0]) <= 1e-8]
    assert len(zeros) >= 2 and abs(zeros[0] - np.pi / 2) < 5e-3 and abs(zeros[-1] - 3 * np.pi / 2) < 5e-3
    assert sol.status == Status.UserInterrupt and sol.status.is_success()
    sol = solve_ivp(SHOZeroEvent(EventConfig().positive().terminal()), 0.0, 6.0, [1.0, 0.0], default_opts(Method.DOPRI5))
    assert abs(sol.t_events[0][0] - 3 * np.pi / 2) < 5e-3
    sol = solve_ivp(SHOZeroEvent(EventConfig().negative().terminal()), 0.0, 6.0, [1.0, 0.0], default_opts(Method.DOPRI5))
    assert abs(sol.t_events[0][0] - np.pi / 2) < 5e-3


def test_duplicate_timestamps_known_answers():  # tests/test_ivp.py:152-170: the reference's golden numbers
    sol = solve_ivp(Cannon(EventConfig(Direction.Negative, 1)), 0.0, np.inf, [0.0, 0.01],
                    Options(method="RK45", max_step=0.05 * 0.001 / 9.80665, dense_output=True))
    np.testing.assert_allclose(sol.continuous_sol.evaluate_extrapolate(0.01), [-0.00039033, -0.08806632], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(sol.t_events[0], [0.00203943], rtol=1e-5, atol=1e-8)
    assert sol.status == Status.UserInterrupt


@pytest.mark.parametrize("method", EXPLICIT + [Method.BDF])
def test_events_rational(method):  # tests/test_ivp.py:345-460
    ev1 = lambda t, y: y[0] - y[1] ** 0.7
    ev3 = lambda t, y: t - 7.4
    y0 = [1 / 3, 2 / 9]
    res = solve_ivp(RationalEvents(), 5.0, 8.0, y0, Options(method=method))
    assert res.status == Status.Success and len(res.t_events[0]) == 1 and len(res.t_events[1]) == 1
    assert 5.3 < res.t_events[0][0] < 5.7 and 7.3 < res.t_events[1][0] < 7.7
    assert res.y_events[0].shape == (1, 2) and abs(ev1(res.t_events[0][0], res.y_events[0][0])) < 1e-5
    res = solve_ivp(RationalEvents(EventConfig().positive(), EventConfig().positive()), 5.0, 8.0, y0, Options(method=method))
    assert len(res.t_events[0]) == 1 and len(res.t_events[1]) == 0
    res = solve_ivp(RationalEvents(EventConfig().negative(), EventConfig().negative()), 5.0, 8.0, y0, Options(method=method))
    assert len(res.t_events[0]) == 0 and len(res.t_events[1]) == 1
    res = solve_ivp(RationalEvents(EventConfig(), EventConfig(), EventConfig().terminal()), 5.0, 8.0, y0,
                    Options(method=method, dense_output=True))
    assert res.status == Status.UserInterrupt
    assert len(res.t_events[0]) == 1 and len(res.t_events[1]) == 0 and len(res.t_events[2]) == 1
    assert 7.3 < res.t_events[2][0] < 7.5 and abs(ev3(res.t_events[2][0], res.y_events[2][0])) < 1e-5
    np.testing.assert_allclose(sol_rational(res.t_events[0][0]), res.y_events[0][0], rtol=1e-3, atol=1e-6)
    tc = np.linspace(res.t[0], res.t[-1])
    yc = np.array([res.continuous_sol.evaluate_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_rational(tc), 1e-3, 1e-6) < 5)
    res = solve_ivp(RationalEvents(), 8.0, 5.0, [4 / 9, 20 / 81], Options(method=method))      # backward
    assert len(res.t_events[0]) == 1 and len(res.t_events[1]) == 1


def test_bouncing_ball_example_and_user_defined_events():  # examples/bouncing_ball.rs
    sol = solve_ivp(BouncingBall(9.81, 0.02), 0.0, 10.0, [10.0, 5.0], Options(method=Method.DOPRI5, rtol=1e-8, atol=1e-10))
    assert sol.status == Status.UserInterrupt and len(sol.t_events[0]) == 1 and abs(sol.y_events[0][0][0]) < 1e-9
    assert sol.t[-1] == sol.t_events[0][0]
    # the same system as a user-defined (hiprtc) problem with its own event function
    src = r"""
    __device__ void ode(double t, const double* s, double* d, const double* p)
    { const double vy = s[1]; d[0] = vy; d[1] = -p[0] - p[1] * vy * fabs(vy); }
    __device__ void events(double t, const double* s, double* g, const double* p) { g[0] = s[0]; }
    """
    f = ivp_amd.DeviceIVP(src, n=2, params=(9.81, 0.02), events=[EventConfig().terminal().negative()])
    s2 = solve_ivp(f, 0.0, 10.0, [10.0, 5.0], Options(method=Method.DOPRI5, rtol=1e-8, atol=1e-10))
    assert s2.status == Status.UserInterrupt and np.array_equal(s2.t_events[0], sol.t_events[0])
    assert np.array_equal(s2.y, sol.y)


@pytest.mark.parametrize("method", EXPLICIT)
def test_tbound_respected(method):  # tests/test_ivp.py:885-949, tests/test_edge_cases.py:55-121
    """The right-hand side is never evaluated outside [t0, tf]: the device RHS returns NaN there, which would poison
    the state (all stage abscissae are <= 1 and the last step lands exactly on tf)."""
    src = r"""
    __device__ void ode(double t, const double* y, double* d, const double* p)
    { const double lo = fmin(p[0], p[1]), hi = fmax(p[0], p[1]);
      d[0] = (t < lo || t > hi) ? nan("") : -y[0]; }
    """
    for a, b in ((0.0, 1.0), (1.0, 0.0), (0.0, 1e-3)):
        f = ivp_amd.DeviceIVP(src, n=1, params=(a, b))
        s = solve_ivp(f, a, b, [1.0], Options(method=method, rtol=1e-6, atol=1e-9))
        assert s.status == Status.Success and s.t[-1] == b
        assert np.isfinite(s.y).all() and abs(s.y[-1][0] - np.exp(-(b - a))) < 1e-4
