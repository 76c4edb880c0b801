"""A user-defined system: the device-side `impl IVP for T` is a HIP snippet compiled at run time with hiprtc."""
import numpy as np

import ivp_amd
from ivp_amd import Options, solve_ivp, solve_ivp_batch

pendulum = ivp_amd.DeviceIVP(r'''
__device__ void ode(double t, const double* y, double* dydt, const double* p)
{ dydt[0] = y[1]; dydt[1] = -p[0] * y[1] - sin(y[0]) + p[1] * cos(p[2] * t); }''', n=2, params=(0.2, 0.7, 1.3))
sol = solve_ivp(pendulum, 0.0, 20.0, [0.3, 0.0], Options(method="DOP853", rtol=1e-10, atol=1e-12, dense_output=True))
print(f"driven pendulum: status {sol.status.name}, {sol.naccpt} steps, theta(20) = {sol.y[-1][0]:.9f}, theta(7.5) = {sol.sol(7.5)[0]:.9f}")
# a parameter sweep over the driving amplitude: the struct's fields become per-trajectory arrays
B = 2048
params = np.stack([np.full(B, 0.2), np.linspace(0.1, 1.5, B), np.full(B, 1.3)])
r = solve_ivp_batch(pendulum, 0.0, 20.0, np.repeat([[0.3], [0.0]], B, axis=1), params, Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
print(f"sweep over {B} amplitudes: theta(20) from {r.y_end[0].min():.3f} to {r.y_end[0].max():.3f}, {int(r.naccpt.sum())} steps in total")
assert (r.status == 0).all()
