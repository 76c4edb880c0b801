"""Robertson's stiff kinetics through BDF, with the trait's default forward-difference Jacobian and with an `IVP::jac`
override (the reference's tests/test_ivp.py:327-342 problem; src/ivp.rs:67-107 is the hook).  The override is device
code next to `ode`: a built-in functor here, or a `jac()` in a hiprtc snippet (second half)."""
import numpy as np

import ivp_amd
from ivp_amd import Options, solve_ivp

y0 = [1e4, 0.0, 0.0]
opts = Options(method="BDF", rtol=1e-6, atol=1e-6)
fd = solve_ivp(ivp_amd.Robertson(), 0.0, 1e8, y0, opts)
an = solve_ivp(ivp_amd.RobertsonJac(), 0.0, 1e8, y0, opts)
print(f"forward differences: status {fd.status.name}, nfev {fd.nfev}, njev {fd.njev}, nlu {fd.nlu}, steps {len(fd.t)}")
print(f"analytic Jacobian  : status {an.status.name}, nfev {an.nfev}, njev {an.njev}, nlu {an.nlu}, steps {len(an.t)}")
print(f"  y(1e8) = {an.y[-1]},  mass {an.y[-1].sum():.6f}")
assert fd.status.is_success() and an.status.is_success() and an.nfev < 5000 and an.njev < 200
np.testing.assert_allclose(an.y[-1], fd.y[-1], rtol=1e-3)

user = ivp_amd.DeviceIVP(r"""
__device__ void ode(double t, const double* s, double* d, const double* p)
{ const double x = s[0], y = s[1], z = s[2];
  d[0] = -p[0] * x + p[1] * y * z; d[1] = p[0] * x - p[1] * y * z - p[2] * y * y; d[2] = p[2] * y * y; }
__device__ void jac(double t, const double* s, double* j, const double* p)      // j[row * 3 + col]
{ const double y = s[1], z = s[2];
  j[0] = -p[0]; j[1] = p[1] * z;                     j[2] = p[1] * y;
  j[3] = p[0];  j[4] = -p[1] * z - 2.0 * p[2] * y;   j[5] = -p[1] * y;
  j[6] = 0.0;   j[7] = 2.0 * p[2] * y;               j[8] = 0.0; }
""", n=3, params=(0.04, 1e4, 3e7), jac=True)
# a sweep over the slow rate constant: 64 trajectories at once, each with its own parameters
B = 64
k1 = np.linspace(0.02, 0.08, B)
params = np.stack([k1, np.full(B, 1e4), np.full(B, 3e7)])
r = ivp_amd.solve_ivp_batch(user, 0.0, 1e6, np.repeat(np.array(y0)[:, None], B, axis=1), params, opts)
print(f"sweep of {B} rate constants: all success {bool((r.status == 0).all())}, njev {int(r.njev.min())}..{int(r.njev.max())}, "
      f"x(1e6) from {r.y_end[0].min():.2f} to {r.y_end[0].max():.2f}")
assert (r.status == 0).all()
