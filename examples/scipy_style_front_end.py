"""The reference's Python entry point (``ivp.solve_ivp``, SciPy's signature) over the GPU path.

Two of the reference's own test problems, written the way its tests call them (tests/test_ivp.py:152-170 and
tests/test_stiff.py:148-165); the Python callables become device code: the body of ``ode`` for small systems, the
component form ``ode_comp(i, t, y, p)`` for more than 8 states (one wavefront per trajectory).
"""
import numpy as np

from ivp_amd.pyfront import Event, solve_ivp

# 1. an upward cannon shot with a terminal, downward-only event and dense output
hit_ground = Event("y[0]", terminal=True, direction=-1)
sol = solve_ivp("dydx[0] = y[1]; dydx[1] = -9.80665;", [0, np.inf], [0, 0.01], max_step=0.05 * 0.001 / 9.80665,
                events=hit_ground, dense_output=True)
print(f"cannon: status {sol.status} ({sol.message}), hit the ground at t = {sol.t_events[0][0]:.8f} (reference: 0.00203943), "
      f"sol(0.01) = {sol.sol(0.01)}")

# 2. the Medazko problem: 400 stiff states from a method-of-lines discretisation, BDF, finite-difference Jacobian
MEDAZKO = r"""
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
y0 = np.zeros(400)
y0[1::2] = 1
res = solve_ivp(MEDAZKO, [0, 20], y0, method="BDF", dense_output=True)
print(f"medazko: {res.t.size - 1} steps, nfev {res.nfev}, njev {res.njev}, nlu {res.nlu}; y[78](20) = {res.y[78, -1]:.6e} "
      f"(reference: 2.33994e-04), y[79](20) = {res.y[79, -1]:.2e}; sol(10.0)[78] = {res.sol(10.0)[78]:.6e}")
assert res.success and abs(res.y[78, -1] / 0.233994e-3 - 1) < 1e-2 and abs(res.y[79, -1]) < 1e-3
