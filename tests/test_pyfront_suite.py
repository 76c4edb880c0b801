"""The topic files of the reference's Python test-suite (tests/test_events.py, test_step_control.py, test_edge_cases.py,
test_basic_integration.py, test_stiff.py, test_args.py; tests/test_t_eval.py repeats tests/test_ivp.py:586-701, which
tests/test_pyfront.py already restates) against ``ivp_amd.pyfront.solve_ivp`` -- same arguments and assertions, the
Python callables written as device code.  Radau legs are outside the path; the sparse-Jacobian legs run with the dense
finite-difference Jacobian (their assertions are on the solution, which does not depend on how J is differenced)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_equal

from ivp_amd.pyfront import Event, solve_ivp

pytestmark = pytest.mark.gpu
METHODS = ["RK23", "RK45", "DOP853", "BDF"]
FUN_RATIONAL = "dydx[0] = y[1] / x; dydx[1] = y[1] * (y[0] + 2 * y[1] - 1) / (x * (y[0] - 1));"   # test_helpers.py:23-25
FUN_LINEAR = "dydx[0] = -y[0] - 5 * y[1]; dydx[1] = y[0] + y[1];"                                  # test_helpers.py:11-12
Y0 = [1 / 3, 2 / 9]


def sol_rational(t):
    return np.asarray((t / (t + 10), 10 * t / (t + 10) ** 2))


def sol_linear(t):
    return np.vstack((-5 * np.sin(2 * t), 2 * np.cos(2 * t) + np.sin(2 * t)))


def compute_error(y, y_true, rtol, atol):
    e = (y - y_true) / (atol + rtol * np.abs(y_true))
    return np.linalg.norm(e, axis=0) / np.sqrt(e.shape[0])


# ---- tests/test_events.py -------------------------------------------------------------------------------------------
EVENT_1 = "y[0] - pow(y[1], 0.7)"
EVENT_2 = "pow(y[1], 0.6) - y[0]"


@pytest.mark.parametrize("method", METHODS)
def test_events_per_method(method):   # test_events.py:9-97
    res = solve_ivp(FUN_RATIONAL, [5, 8], Y0, method=method, events=(Event(EVENT_1), Event(EVENT_2)))
    assert_equal(res.status, 0)
    assert_equal(len(res.t_events[0]), 1)
    assert_equal(len(res.t_events[1]), 1)
    assert 5.3 < res.t_events[0][0] < 5.7
    assert 7.3 < res.t_events[1][0] < 7.7


def test_terminal_event():   # test_events.py:100-112
    res = solve_ivp(FUN_RATIONAL, [5, 8], Y0, method="RK45", events=Event("x - 7.4", terminal=True), dense_output=True)
    assert_equal(res.status, 1)
    assert_equal(len(res.t_events[0]), 1)
    assert 7.3 < res.t_events[0][0] < 7.5


def test_event_direction():   # test_events.py:115-142
    res = solve_ivp(FUN_RATIONAL, [5, 8], Y0, method="RK45", events=Event(EVENT_1, direction=1))
    assert_equal(res.status, 0)
    assert_equal(len(res.t_events[0]), 1)
    assert 5.3 < res.t_events[0][0] < 5.7
    res = solve_ivp(FUN_RATIONAL, [5, 8], Y0, method="RK45", events=Event(EVENT_1, direction=-1))
    assert_equal(res.status, 0)
    assert_equal(len(res.t_events[0]), 0)


# ---- tests/test_step_control.py -------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("t_span", [[5, 9], [5, 1]])
def test_max_step(method, t_span):   # test_step_control.py:9-49
    rtol, atol = 1e-3, 1e-6
    res = solve_ivp(FUN_RATIONAL, t_span, Y0, rtol=rtol, max_step=0.5, atol=atol, method=method, dense_output=True)
    assert_equal(res.t[0], t_span[0])
    assert_equal(res.t[-1], t_span[-1])
    assert np.all(np.abs(np.diff(res.t)) <= 0.5 + 1e-15)
    assert res.success
    assert_equal(res.status, 0)
    if t_span[1] > t_span[0]:
        assert np.all(compute_error(res.y, sol_rational(res.t), rtol, atol) < 5)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("t_span", [[5, 9], [5, 1]])
def test_first_step(method, t_span):   # test_step_control.py:52-93
    first_step = 0.1
    res = solve_ivp(FUN_RATIONAL, t_span, Y0, rtol=1e-3, max_step=0.5, atol=1e-6, method=method, dense_output=True,
                    first_step=first_step)
    assert_equal(res.t[0], t_span[0])
    assert_equal(res.t[-1], t_span[-1])
    assert_allclose(first_step, np.abs(res.t[1] - 5))
    assert res.success
    assert_equal(res.status, 0)


@pytest.mark.parametrize("method", METHODS)
def test_max_steps(method):   # test_step_control.py:96-127
    res = solve_ivp(FUN_RATIONAL, [5, 9], Y0, rtol=1e-3, atol=1e-6, method=method, max_steps=1)
    assert not res.success
    assert_equal(res.status, -1)
    assert "NeedLargerNMax" in res.message or "max" in res.message.lower()
    res = solve_ivp(FUN_RATIONAL, [5, 9], Y0, rtol=1e-3, atol=1e-6, method=method, max_steps=1_000_000)
    assert res.success
    assert_equal(res.status, 0)


@pytest.mark.parametrize("method", METHODS)
def test_default_max_steps_is_unlimited(method):   # test_step_control.py:130-151
    res = solve_ivp("dydx[0] = -0.001 * y[0];", [0, 100000], [1.0], method=method, rtol=1e-8, atol=1e-10)
    assert res.success, f"Method {method} failed with message: {res.message}"
    assert_equal(res.status, 0)


def test_min_step_parameter():   # test_step_control.py:154-166 (BDF leg)
    res = solve_ivp(FUN_RATIONAL, [5, 9], Y0, rtol=1e-3, atol=1e-6, method="BDF", min_step=1e-10)
    assert res.success, res.message
    assert_equal(res.status, 0)


# ---- tests/test_edge_cases.py: the right-hand side must never be evaluated outside the interval -----------------------
# (the reference's functions raise there; a device function cannot, so it returns NaN, which no solver survives)
@pytest.mark.parametrize("method", METHODS)
def test_tbound_respected_small_interval(method):   # test_edge_cases.py:55-66 (gh-17341)
    res = solve_ivp("dydx[0] = x > 1e-4 ? nan(\"\") : 2 * y[0];", (0.0, 1e-4), np.array([1]), method=method)
    assert res.success
    assert np.isfinite(res.y).all()


@pytest.mark.parametrize("method", METHODS)
def test_tbound_respected_larger_interval(method):   # test_edge_cases.py:69-95 (gh-8848)
    src = ("const double r = exp(x); const double V = -11 / r + 10 * r / (0.05 + r * r);"
           "const double out = (x < -17 || x > 2) ? nan(\"\") : 1.0;"
           "dydx[0] = out * (r * y[1]); dydx[1] = out * (-2.0 * r * ((-0.2 - V) * y[0] + 1 / r * y[1]));")
    result = solve_ivp(src, (-17, 2), y0=np.array([1, -11]), max_step=0.03, vectorized=False, t_eval=None, atol=1e-8, rtol=1e-5,
                       method=method)
    assert result.success
    assert np.isfinite(result.y).all()


@pytest.mark.parametrize("method", METHODS)
def test_tbound_respected_oscillator(method):   # test_edge_cases.py:98-121 (gh-9198)
    src = ("const double out = x > 205 ? nan(\"\") : 1.0;"
           "dydx[0] = out * 1.73307544e-02; dydx[1] = out * 6.49376470e-06; dydx[2] = out * 0.0; dydx[3] = out * 0.0;")
    result = solve_ivp(src, (100.0, 200.0), np.array([134.08298555, 138.82348612, 100., 0.]), dense_output=True, max_step=100.0,
                       method=method)
    assert result.success
    assert np.isfinite(result.y).all()


# ---- tests/test_basic_integration.py --------------------------------------------------------------------------------
@pytest.mark.parametrize("method", METHODS)
def test_integration_forward(method):   # test_basic_integration.py:12-104
    rtol, atol = 1e-3, 1e-6
    res = solve_ivp(FUN_RATIONAL, [5, 9], Y0, rtol=rtol, atol=atol, method=method, dense_output=True)
    assert_equal(res.t[0], 5)
    assert res.success
    assert_equal(res.status, 0)
    assert np.all(compute_error(res.y, sol_rational(res.t), rtol, atol) < 5)


@pytest.mark.parametrize("method", ["RK23", "RK45"])
def test_integration_backward(method):   # test_basic_integration.py:107-135
    res = solve_ivp(FUN_RATIONAL, [5, 1], Y0, rtol=1e-3, atol=1e-6, method=method, dense_output=True)
    assert_equal(res.t[0], 5)
    assert res.success
    assert_equal(res.status, 0)


# ---- tests/test_stiff.py ----------------------------------------------------------------------------------------------
def test_integration_const_jac_BDF():   # test_stiff.py:35-53, 77-95 (dense and "sparse" constant Jacobian: the same matrix)
    rtol, atol = 1e-3, 1e-6
    res = solve_ivp(FUN_LINEAR, [0, 2], [0, 2], rtol=rtol, atol=atol, method="BDF", dense_output=True,
                    jac=np.array([[-1, -5], [1, 1]]))
    assert_equal(res.t[0], 0)
    assert res.success
    assert_equal(res.status, 0)
    assert res.nfev < 100
    assert np.all(compute_error(res.y, sol_linear(res.t), rtol, atol) < 10)


def test_integration_stiff_BDF():   # test_stiff.py:122-145
    res = solve_ivp("dydx[0] = -0.04 * y[0] + 1e4 * y[1] * y[2];"
                    "dydx[1] = 0.04 * y[0] - 1e4 * y[1] * y[2] - 3e7 * y[1] * y[1];"
                    "dydx[2] = 3e7 * y[1] * y[1];", [0, 1e8], [1e4, 0, 0], rtol=1e-6, atol=1e-6, method="BDF")
    assert res.nfev < 5000
    assert res.njev < 600


MEDAZKO = r"""
// fun_medazko (tests/test_helpers.py:54-79), component form: y_ext = [phi, 0, y..., y[-2]]
__device__ double ode_comp(int i, double t, const double* y, const double* p)
{
    const int n = 200;
    const double k = 100.0, c = 4.0, d = 1.0 / n;
    const double phi = t <= 5 ? 2.0 : 0.0;
    auto ext = [&](int m) { return m == 0 ? phi : (m == 1 ? 0.0 : (m == 2 * n + 2 ? y[2 * n - 2] : y[m - 2])); };
    const int j = i / 2 + 1;
    if (i & 1) return -k * ext(2 * j + 1) * ext(2 * j);
    const double s = j * d - 1.0;
    const double alpha = 2 * s * s * s / (c * c), beta = s * s * s * s / (c * c);
    return alpha * (ext(2 * j + 2) - ext(2 * j - 2)) / (2 * d) + beta * (ext(2 * j - 2) - 2 * ext(2 * j) + ext(2 * j + 2)) / (d * d)
           - k * ext(2 * j) * ext(2 * j + 1);
}
"""


def test_integration_sparse_difference_BDF():   # test_stiff.py:148-165: the 400-state Medazko problem, reference's golden values
    n = 200
    y0 = np.zeros(2 * n)
    y0[1::2] = 1
    res = solve_ivp(MEDAZKO, [0, 20], y0, method="BDF")
    assert_equal(res.t[0], 0)
    assert res.success
    assert_equal(res.status, 0)
    assert_allclose(res.y[78, -1], 0.233994e-3, rtol=1e-2)
    assert_allclose(res.y[79, -1], 0, atol=1e-3)


def test_medazko_bdf_equals_the_oracle_bit_for_bit():
    """The same 400-state run against the CPU oracle driven by a numpy restatement of fun_medazko with the device
    function's operation order: one wavefront per trajectory, 400 x 400 LU by wavefront, identical bits and counters."""
    from oracle import oracle as O
    n, k, c = 200, 100.0, 4.0
    d = 1.0 / n
    i = np.arange(2 * n)
    j = i // 2 + 1
    s = j * d - 1.0
    alpha, beta = 2 * s * s * s / (c * c), s * s * s * s / (c * c)

    def medazko(t, y, p):
        ext = np.concatenate(([2.0 if t <= 5 else 0.0, 0.0], y, [y[2 * n - 2]]))
        even = (alpha * (ext[2 * j + 2] - ext[2 * j - 2]) / (2 * d) + beta * (ext[2 * j - 2] - 2 * ext[2 * j] + ext[2 * j + 2]) / (d * d)
                - k * ext[2 * j] * ext[2 * j + 1])
        odd = -k * ext[2 * j + 1] * ext[2 * j]
        return np.where(i & 1, odd, even)

    y0 = np.zeros(2 * n)
    y0[1::2] = 1
    res = solve_ivp(MEDAZKO, [0, 20], y0, method="BDF")
    ref = O.solve_ivp(medazko, 0.0, 20.0, y0, detpow=True, method="BDF")
    assert (res.nfev, res.njev, res.nlu) == (ref.nfev, ref.njev, ref.nlu)
    assert res.t.size == ref.t.size and np.array_equal(res.t, ref.t)
    assert np.array_equal(res.y.T.view(np.uint64), np.ascontiguousarray(ref.y).view(np.uint64))


# ---- tests/test_args.py: test_args_with_events / test_args_single_value / test_array_rtol are tests/test_ivp.py:731-841,
# restated in tests/test_pyfront.py ------------------------------------------------------------------------------------
