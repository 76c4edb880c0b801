"""Every known-answer / tolerance assertion the reference's own tests make for the explicit-RK
path, restated against the CPU oracle.  Each test cites the reference test it
restates.  The event-detection pins (tests/ivp.rs:151-275, tests/test_events.py) are outside the
hot path's scope and are not restated.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O

EXPLICIT = ["RK23", "DOPRI5", "DOP853"]


def sol_rational(t):
    t = np.asarray(t)
    return np.asarray((t / (t + 10), 10 * t / (t + 10) ** 2))


def compute_error(y, y_true, rtol, atol):
    e = (y - y_true) / (atol + rtol * np.abs(y_true))
    return np.linalg.norm(e, axis=0) / np.sqrt(e.shape[0])


# ---- tests/accuracy.rs ---------------------------------------------------------------------------

@pytest.mark.parametrize("method", EXPLICIT + ["RK4"])
def test_harmonic_accuracy_end_state(method):  # tests/accuracy.rs:18-48
    kw = dict(first_step=2 * np.pi / 2000.0) if method == "RK4" else dict(rtol=1e-9, atol=1e-9)
    s = O.solve_ivp("sho", 0.0, 2 * np.pi, [1.0, 0.0], method=method, **kw)
    assert abs(s.y[-1, 0] - 1.0) < 1e-5 and abs(s.y[-1, 1]) < 1e-5


@pytest.mark.parametrize("method", EXPLICIT + ["RK4"])
def test_t_eval_sampling_exact_times(method):  # tests/accuracy.rs:51-77
    te = np.arange(11) / 10.0
    s = O.solve_ivp("sho", 0.0, 1.0, [1.0, 0.0], method=method, rtol=1e-9, atol=1e-9, t_eval=te)
    for t in te:
        assert np.any(np.abs(s.t - t) <= 1e-9)
    assert len(s.y) == len(s.t)


def test_iterate_samples():  # tests/accuracy.rs:80-89
    s = O.solve_ivp("sho", 0.0, 1.0, [1.0, 0.0], method="DOPRI5", rtol=1e-9, atol=1e-9)
    assert np.all((s.t >= 0.0) & (s.t <= 1.0)) and s.y.shape[1] == 2


# ---- tests/ivp.rs --------------------------------------------------------------------------------

@pytest.mark.parametrize("method", EXPLICIT)
def test_integration_zero_rhs(method):  # tests/ivp.rs:21-46, tests/test_ivp.py:844-849
    te = np.array([10.0 * i / 20.0 for i in range(21)])
    s = O.solve_ivp("zero", 0.0, 10.0, [1.0, 1.0, 1.0], method=method, rtol=1e-9, atol=1e-12, t_eval=te)
    assert np.array_equal(s.t, te)
    assert np.abs(s.y - 1.0).max() <= 1e-12


@pytest.mark.parametrize("method", EXPLICIT)
def test_max_step_and_first_step_controls(method):  # tests/ivp.rs:49-104
    s = O.solve_ivp("sho", 0.0, 3.0, [1.0, 0.0], method=method, rtol=1e-6, atol=1e-9, max_step=0.05)
    assert np.abs(np.diff(s.t)).max() <= 0.05 + 1e-12
    s = O.solve_ivp("sho", 0.0, 3.0, [1.0, 0.0], method=method, rtol=1e-3, atol=1e-6, first_step=0.1)
    assert len(s.t) >= 2
    assert abs(abs(s.t[1] - s.t[0]) - 0.1) <= 1e-6


@pytest.mark.parametrize("method", EXPLICIT)
def test_dense_output_matches_discrete_samples(method):  # tests/ivp.rs:107-136
    s = O.solve_ivp("sho", 0.0, 2.0, [1.0, 0.0], method=method, rtol=1e-8, atol=1e-10, dense_output=True)
    assert s.sol_span() is not None
    for t, y in zip(s.t, s.y):
        assert np.abs(s.sol(t) - y).max() <= 1e-8


def test_dense_output_out_of_range_errors():  # tests/ivp.rs:139-149
    s = O.solve_ivp("sho", 0.0, 1.0, [1.0, 0.0], method="DOPRI5", rtol=1e-9, atol=1e-9, dense_output=True)
    t0, t1 = s.sol_span()
    with pytest.raises(ValueError):
        s.sol(t0 - 0.1)
    with pytest.raises(ValueError):
        s.sol(t1 + 0.1)


@pytest.mark.parametrize("method", EXPLICIT)
def test_zero_interval_returns_initial_state(method):  # tests/ivp.rs:278-289
    s = O.solve_ivp("sho", 1.23, 1.23, [2.0, 3.0], method=method, rtol=1e-9, atol=1e-9)
    assert len(s.t) >= 1 and np.abs(s.y[-1] - [2.0, 3.0]).max() <= 1e-12
    assert s.nfev == 0 and s.status == 0


def test_vector_rtol_componentwise_control():  # tests/ivp.rs:291-334
    loose = O.solve_ivp("exp2", 0.0, 1.0, [1.0, 1.0], method="DOPRI5", rtol=[1e-2, 1e-2], atol=1e-10)
    tight = O.solve_ivp("exp2", 0.0, 1.0, [1.0, 1.0], method="DOPRI5", rtol=[1e-2, 1e-10], atol=1e-10)
    e = np.e
    assert abs(tight.y[-1, 1] - e) < abs(loose.y[-1, 1] - e) * 0.5
    assert abs(tight.y[-1, 0] - e) <= 10.0 * abs(loose.y[-1, 0] - e)


# ---- tests/backward_and_bounds.rs ----------------------------------------------------------------

@pytest.mark.parametrize("method", EXPLICIT)
def test_backward_integration_works(method):  # tests/backward_and_bounds.rs:7-32
    s = O.solve_ivp("sho", 2 * np.pi, 0.0, [1.0, 0.0], method=method, rtol=1e-9, atol=1e-9, dense_output=True)
    t0, t1 = s.sol_span()
    assert t0 > t1
    mid = 0.5 * (t0 + t1)
    ym = s.sol(mid)
    assert abs(ym[0] - np.cos(mid)) < 1e-6 and abs(ym[1] + np.sin(mid)) < 1e-6


# ---- tests/test_ivp.py (explicit-RK cases) ---------------------------------------------------------

@pytest.mark.parametrize("method", EXPLICIT)
@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_integration_rational(method, t_span):  # tests/test_ivp.py:173-241
    rtol, atol = 1e-3, 1e-6
    s = O.solve_ivp("rational", t_span[0], t_span[1], [1 / 3, 2 / 9], method=method, rtol=rtol, atol=atol,
                    dense_output=True)
    assert s.t[0] == t_span[0] and s.status == 0
    if method == "DOP853":
        assert s.nfev < 50
    assert s.njev == 0 and s.nlu == 0
    e = compute_error(s.y.T, sol_rational(s.t), rtol, atol)
    assert np.all(e < 5)
    tc = np.linspace(*t_span)
    yc = np.array([s.sol_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_rational(tc), rtol, atol) < 5)
    # res.sol(res.t) == res.y to 1e-15
    ys = np.array([s.sol_extrapolate(t) for t in s.t])
    np.testing.assert_allclose(ys, s.y, rtol=1e-15, atol=1e-15)


@pytest.mark.parametrize("method", EXPLICIT)
@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_max_step_python(method, t_span):  # tests/test_ivp.py:521-552, tests/test_step_control.py:9-50
    s = O.solve_ivp("rational", t_span[0], t_span[1], [1 / 3, 2 / 9], method=method, rtol=1e-3, atol=1e-6,
                    max_step=0.5, dense_output=True)
    assert s.t[0] == t_span[0] and s.t[-1] == t_span[-1]
    assert np.all(np.abs(np.diff(s.t)) <= 0.5 + 1e-15)
    assert s.status == 0
    assert np.all(compute_error(s.y.T, sol_rational(s.t), 1e-3, 1e-6) < 5)


@pytest.mark.parametrize("method", EXPLICIT)
@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_first_step_python(method, t_span):  # tests/test_ivp.py:555-583, tests/test_step_control.py:53-90
    first_step = 0.1
    s = O.solve_ivp("rational", t_span[0], t_span[1], [1 / 3, 2 / 9], method=method, rtol=1e-3, atol=1e-6,
                    max_step=0.5, first_step=first_step, dense_output=True)
    assert s.t[0] == t_span[0] and s.t[-1] == t_span[-1]
    np.testing.assert_allclose(first_step, abs(s.t[1] - 5.0))
    assert s.status == 0
    assert np.all(compute_error(s.y.T, sol_rational(s.t), 1e-3, 1e-6) < 5)


@pytest.mark.parametrize("method", EXPLICIT)
def test_max_steps_parameter(method):  # tests/test_step_control.py:93-109
    s = O.solve_ivp("decay", 0.0, 10.0, [1.0], params=[1.0], method=method, rtol=1e-3, atol=1e-6, max_steps=1)
    assert s.status == 2  # NeedLargerNMax  (python status -1)


@pytest.mark.parametrize("method", EXPLICIT)
def test_default_max_steps_is_unlimited(method):  # tests/test_step_control.py:130-159
    s = O.solve_ivp("decay", 0.0, 1e5, [1.0], params=[0.001], method=method, rtol=1e-8, atol=1e-10)
    assert s.status == 0 and s.t[-1] == 1e5


@pytest.mark.parametrize("method", EXPLICIT)
def test_t_eval_python(method):  # tests/test_ivp.py:586-672, tests/test_t_eval.py:9-134
    rtol, atol = 1e-3, 1e-6
    y0 = [1 / 3, 2 / 9]
    for t_span in ((5.0, 9.0), (5.0, 1.0)):
        te = np.linspace(*t_span, 10)
        s = O.solve_ivp("rational", t_span[0], t_span[1], y0, method=method, rtol=rtol, atol=atol, t_eval=te)
        assert np.array_equal(s.t, te) and s.status == 0
        assert np.all(compute_error(s.y.T, sol_rational(s.t), rtol, atol) < 5)
        sd = O.solve_ivp("rational", t_span[0], t_span[1], y0, method=method, rtol=rtol, atol=atol, t_eval=te,
                         dense_output=True)
        assert np.array_equal(sd.y, s.y)  # t_eval with/without dense gives identical y
    te = np.array([5.01, 7.0, 8.0, 8.01])
    s = O.solve_ivp("rational", 5.0, 9.0, y0, method=method, rtol=rtol, atol=atol, t_eval=te)
    assert np.array_equal(s.t, te)
    te = np.array([4.99, 3.0, 1.5, 1.1])
    s = O.solve_ivp("rational", 5.0, 1.0, y0, method=method, rtol=rtol, atol=atol, t_eval=te)
    assert np.array_equal(s.t, te)


@pytest.mark.parametrize("method", EXPLICIT)
def test_no_integration_and_empty(method):  # tests/test_ivp.py:704-728
    s = O.solve_ivp("rational", 4.0, 4.0, [2.0, 4.0], method=method, dense_output=True)
    np.testing.assert_array_equal(s.sol_extrapolate(4.0), [2.0, 4.0])
    np.testing.assert_array_equal(s.sol_extrapolate(6.0), [2.0, 4.0])
    s = O.solve_ivp("sho", 0.0, 10.0, [], method=method, dense_output=True)
    assert np.array_equal(s.t, [0.0, 10.0]) and s.y.shape == (2, 0)


def test_args_single_value():  # tests/test_ivp.py:852-861: y(0.1) = exp(-0.1) at default tolerances
    s = O.solve_ivp("decay", 0.0, 0.1, [1.0], params=[1.0])
    np.testing.assert_allclose(s.y[-1, 0], np.exp(-0.1), rtol=1e-3)


@pytest.mark.parametrize("method", EXPLICIT)
def test_tbound_respected(method):  # tests/test_ivp.py:885-949, tests/test_edge_cases.py:55-121
    seen = []

    def f(t, y, p):
        seen.append(t)
        return [-y[0]]

    for a, b in ((0.0, 1.0), (1.0, 0.0), (0.0, 1e-3)):
        seen.clear()
        s = O.solve_ivp(f, a, b, [1.0], method=method, rtol=1e-6, atol=1e-9)
        lo, hi = min(a, b), max(a, b)
        assert s.status == 0 and s.t[-1] == b
        assert min(seen) >= lo and max(seen) <= hi


# ---- BDF ("next" row): the reference's pins for src/methods/bdf.rs -------------------

def sol_linear(t):
    return np.vstack((-5 * np.sin(2 * t), 2 * np.cos(2 * t) + np.sin(2 * t)))


def test_bdf_harmonic_accuracy_and_backward():  # tests/accuracy.rs:18-48, tests/backward_and_bounds.rs:7-32
    s = O.solve_ivp("sho", 0.0, 2 * np.pi, [1.0, 0.0], method="BDF", rtol=1e-9, atol=1e-9)
    assert abs(s.y[-1, 0] - 1.0) < 1e-5 and abs(s.y[-1, 1]) < 1e-5
    s = O.solve_ivp("sho", 2 * np.pi, 0.0, [1.0, 0.0], method="BDF", rtol=1e-9, atol=1e-9, dense_output=True)
    t0, t1 = s.sol_span()
    mid = 0.5 * (t0 + t1)
    ym = s.sol(mid)
    assert t0 > t1 and abs(ym[0] - np.cos(mid)) < 1e-6 and abs(ym[1] + np.sin(mid)) < 1e-6


def test_bdf_t_eval_max_step_zero_interval():  # tests/accuracy.rs:51-77, tests/ivp.rs:49-76,278-289
    te = np.arange(11) / 10.0
    s = O.solve_ivp("sho", 0.0, 1.0, [1.0, 0.0], method="BDF", rtol=1e-9, atol=1e-9, t_eval=te)
    assert all(np.any(np.abs(s.t - t) <= 1e-9) for t in te)
    s = O.solve_ivp("sho", 0.0, 3.0, [1.0, 0.0], method="BDF", rtol=1e-6, atol=1e-9, max_step=0.05)
    assert np.abs(np.diff(s.t)).max() <= 0.05 + 1e-12
    s = O.solve_ivp("sho", 1.23, 1.23, [2.0, 3.0], method="BDF", rtol=1e-9, atol=1e-9)
    assert np.abs(s.y[-1] - [2.0, 3.0]).max() <= 1e-12


@pytest.mark.parametrize("t_span", [(5.0, 9.0), (5.0, 1.0)])
def test_bdf_integration_rational(t_span):  # tests/test_ivp.py:173-241, tests/test_basic_integration.py:89-104
    rtol, atol = 1e-3, 1e-6
    s = O.solve_ivp("rational", t_span[0], t_span[1], [1 / 3, 2 / 9], method="BDF", rtol=rtol, atol=atol, dense_output=True)
    assert s.t[0] == t_span[0] and s.status == 0
    assert 0 < s.njev and 0 < s.nlu
    assert np.all(compute_error(s.y.T, sol_rational(s.t), rtol, atol) < 5)
    tc = np.linspace(*t_span)
    yc = np.array([s.sol_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_rational(tc), rtol, atol) < 5)
    ys = np.array([s.sol_extrapolate(t) for t in s.t])
    np.testing.assert_allclose(ys, s.y, rtol=1e-15, atol=1e-15)


def test_bdf_integration_const_jac():  # tests/test_ivp.py:273-317, tests/test_stiff.py:35-53 (FD Jacobian here)
    rtol, atol = 1e-3, 1e-6
    s = O.solve_ivp("linear", 0.0, 2.0, [0.0, 2.0], method="BDF", rtol=rtol, atol=atol, dense_output=True)
    assert s.t[0] == 0.0 and s.status == 0 and s.nfev < 100
    assert np.all(compute_error(s.y.T, sol_linear(s.t), rtol, atol) < 10)
    tc = np.linspace(0.0, 2.0)
    yc = np.array([s.sol_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_linear(tc), rtol, atol) < 60)
    ys = np.array([s.sol_extrapolate(t) for t in s.t])
    np.testing.assert_allclose(ys, s.y, rtol=1e-14, atol=1e-14)


def test_bdf_integration_stiff_robertson():  # tests/test_ivp.py:320-342
    s = O.solve_ivp("robertson", 0.0, 1e8, [1e4, 0.0, 0.0], method="BDF", rtol=1e-6, atol=1e-6)
    assert s.status == 0 and s.nfev < 5000 and s.njev < 200


def test_bdf_against_independent_stiff_truth():
    """SciPy Radau @1e-10 (tests/golden/scipy_stiff_truth.json) for the BDF workloads: BASELINE C5's Van der Pol
    mu=1000 on [0,3000] (benches/benchmark.py:118-126), Robertson, and examples/van_der_pol.rs."""
    import json, os
    tr = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scipy_stiff_truth.json")))["truth"]
    s = O.solve_ivp("vdp", 0.0, 3000.0, [2.0, 0.0], params=[1000.0], method="BDF", rtol=1e-4, atol=1e-6)
    assert s.status == 0 and np.abs(s.y[-1] - tr["vdp_mu1000_t3000"]).max() < 1e-2
    s = O.solve_ivp("robertson", 0.0, 1e8, [1e4, 0.0, 0.0], method="BDF", rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(s.y[-1], tr["robertson_t1e8"], rtol=1e-4, atol=1e-6)
    te = np.arange(21) * 0.1
    s = O.solve_ivp("vdp_eps", 0.0, 2.0, [2.0, 0.0], params=[1e-3], method="BDF", rtol=1e-6, atol=1e-8, t_eval=te)
    assert np.array_equal(s.t, te) and np.abs(s.y[-1] - tr["vdp_eps1e-3_t2"]).max() < 1e-4


# ---- events ("next" row): src/solve/solout.rs:158-331 -----------------------------------

def test_event_detection_all_and_directional():  # tests/ivp.rs:223-275
    kw = dict(method="DOPRI5", rtol=1e-9, atol=1e-9)
    s = O.solve_ivp("sho_ev", 0.0, 6.0, [1.0, 0.0], event_direction=[0], event_terminal=[2], **kw)
    assert len(s.t_events[0]) >= 2 and np.abs(s.y_events[0][:, 0]).max() <= 1e-8
    assert abs(s.t_events[0][0] - np.pi / 2) < 5e-3 and abs(s.t_events[0][-1] - 3 * np.pi / 2) < 5e-3
    s = O.solve_ivp("sho_ev", 0.0, 6.0, [1.0, 0.0], event_direction=[1], event_terminal=[1], **kw)
    assert abs(s.t_events[0][0] - 3 * np.pi / 2) < 5e-3
    s = O.solve_ivp("sho_ev", 0.0, 6.0, [1.0, 0.0], event_direction=[-1], event_terminal=[1], **kw)
    assert abs(s.t_events[0][0] - np.pi / 2) < 5e-3


def test_duplicate_timestamps_known_answers():  # tests/test_ivp.py:152-170 -- the reference's golden numbers
    s = O.solve_ivp("cannon", 0.0, np.inf, [0.0, 0.01], method="DOPRI5", max_step=0.05 * 0.001 / 9.80665,
                    event_direction=[-1], event_terminal=[1], dense_output=True)
    np.testing.assert_allclose(s.sol_extrapolate(0.01), [-0.00039033, -0.08806632], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(s.t_events[0], [0.00203943], rtol=1e-5, atol=1e-8)
    assert s.status == 1   # terminal event


@pytest.mark.parametrize("method", EXPLICIT + ["BDF"])
def test_events_rational(method):  # tests/test_ivp.py:345-445
    ev1 = lambda t, y: y[0] - y[1] ** 0.7
    ev2 = lambda t, y: y[1] ** 0.6 - y[0]
    kw = dict(method=method)
    s = O.solve_ivp("rational_ev", 5.0, 8.0, [1 / 3, 2 / 9], event_direction=[0, 0, 0], event_terminal=[0, 0, 0], **kw)
    # third event (t - 7.4, non-terminal here) also fires; the reference test registers only the first two
    assert s.status == 0 and len(s.t_events[0]) == 1 and len(s.t_events[1]) == 1
    assert 5.3 < s.t_events[0][0] < 5.7 and 7.3 < s.t_events[1][0] < 7.7
    assert abs(ev1(s.t_events[0][0], s.y_events[0][0])) < 1e-5 and abs(ev2(s.t_events[1][0], s.y_events[1][0])) < 1e-5
    s = O.solve_ivp("rational_ev", 5.0, 8.0, [1 / 3, 2 / 9], event_direction=[1, 1, 0], **kw)
    assert len(s.t_events[0]) == 1 and len(s.t_events[1]) == 0
    s = O.solve_ivp("rational_ev", 5.0, 8.0, [1 / 3, 2 / 9], event_direction=[-1, -1, 0], **kw)
    assert len(s.t_events[0]) == 0 and len(s.t_events[1]) == 1
    s = O.solve_ivp("rational_ev", 5.0, 8.0, [1 / 3, 2 / 9], event_direction=[0, 0, 0], event_terminal=[0, 0, 1],
                    dense_output=True, **kw)
    assert s.status == 1 and len(s.t_events[0]) == 1 and len(s.t_events[1]) == 0 and len(s.t_events[2]) == 1
    assert 7.3 < s.t_events[2][0] < 7.5
    np.testing.assert_allclose(sol_rational(s.t_events[0][0]), s.y_events[0][0], rtol=1e-3, atol=1e-6)
    tc = np.linspace(s.t[0], s.t[-1])
    yc = np.array([s.sol_extrapolate(t) for t in tc]).T
    assert np.all(compute_error(yc, sol_rational(tc), 1e-3, 1e-6) < 5)
    # backward, tests/test_ivp.py:441-460
    s = O.solve_ivp("rational_ev", 8.0, 5.0, [4 / 9, 20 / 81], event_direction=[0, 0, 0], **kw)
    assert len(s.t_events[0]) == 1 and len(s.t_events[1]) == 1
    assert 5.3 < s.t_events[0][0] < 5.7 and 7.3 < s.t_events[1][0] < 7.7


def test_bouncing_ball_example():  # examples/bouncing_ball.rs
    s = O.solve_ivp("ball", 0.0, 10.0, [10.0, 5.0], params=[9.81, 0.02], method="DOPRI5", rtol=1e-8, atol=1e-10,
                    event_direction=[-1], event_terminal=[1])
    assert s.status == 1 and len(s.t_events[0]) == 1 and abs(s.y_events[0][0][0]) < 1e-9
    assert s.t[-1] == s.t_events[0][0]      # terminal event point is appended to the output (solout.rs:317-319)


# ---- large-n problems (wave-per-trajectory kernels on the GPU side) -------------------------------------------------
def test_benchmark_problem4_large_linear_system():
    """benches/benchmark.py:139-148: y' = -y, N = 100, y0 = ones, t in [0, 10], RK45, rtol 1e-6 / atol 1e-8."""
    s = O.solve_ivp("linear_decay100", 0.0, 10.0, [1.0] * 100, method="DOPRI5", rtol=1e-6, atol=1e-8)
    assert s.status == 0 and s.t[-1] == 10.0
    assert np.abs(s.y[-1] - np.exp(-10.0)).max() < 1e-6
    # all components are identical, so the RMS norm equals the scalar problem's: same step sequence as n = 1
    s1 = O.solve_ivp("decay", 0.0, 10.0, [1.0], params=[1.0], method="DOPRI5", rtol=1e-6, atol=1e-8)
    assert (s.naccpt, s.nrejct) == (s1.naccpt, s1.nrejct)
    np.testing.assert_allclose(s.y[-1], s1.y[-1, 0], rtol=1e-13)


def test_heat1d256_oracle_against_closed_form():
    """Eigenmode of the discrete Laplacian decays as exp(-kappa (2 - 2 cos(pi m / 257)) t)."""
    x = np.arange(1, 257) / 257.0
    for m, kappa in ((1, 100.0), (3, 40.0)):
        y0 = np.sin(np.pi * m * x)
        s = O.solve_ivp("heat1d256", 0.0, 0.5, list(y0), params=[kappa], method="DOPRI5", rtol=1e-8, atol=1e-11)
        lam = kappa * (2.0 - 2.0 * np.cos(np.pi * m / 257.0))
        assert s.status == 0
        assert np.abs(s.y[-1] - y0 * np.exp(-lam * 0.5)).max() < 1e-7
