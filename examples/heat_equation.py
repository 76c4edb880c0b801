"""A large system (n = 256): method-of-lines heat equation, one wavefront per trajectory, temperature snapshots with
t_eval, and a user-defined 40-component system compiled at run time (component form)."""
import numpy as np

import ivp_amd
from ivp_amd import Heat1D256, Options, solve_ivp

x = np.arange(1, 257) / 257.0
y0 = np.sin(np.pi * x) + 0.5 * np.sin(3 * np.pi * x)
sol = solve_ivp(Heat1D256(kappa=100.0), 0.0, 0.5, y0, Options(method="DOP853", rtol=1e-8, atol=1e-10, t_eval=[0.0, 0.1, 0.25, 0.5]))
lam = lambda m: 100.0 * (2 - 2 * np.cos(np.pi * m / 257))
for t, y in zip(sol.t, sol.y):
    exact = np.exp(-lam(1) * t) * np.sin(np.pi * x) + 0.5 * np.exp(-lam(3) * t) * np.sin(3 * np.pi * x)
    print(f"t = {t:4.2f}: max temperature {y.max():.6f}, error vs eigenmode solution {np.abs(y - exact).max():.1e}")
print(f"status {sol.status.name}, {sol.naccpt} steps, nfev {sol.nfev}")

ring = ivp_amd.DeviceIVP(r'''
__device__ double ode_comp(int i, double t, const double* y, const double* p)
{   // 20 masses on a ring: positions y[0..20), velocities y[20..40)
    if (i < 20) return y[20 + i];
    const int k = i - 20;
    return p[0] * (y[(k + 19) % 20] - 2.0 * y[k] + y[(k + 1) % 20]);
}''', n=40, params=(3.0,))
q0 = np.zeros(40); q0[0] = 1.0
s = solve_ivp(ring, 0.0, 10.0, q0, Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
energy = lambda q: 0.5 * (q[20:] ** 2).sum() + 0.5 * 3.0 * ((q[:20] - np.roll(q[:20], 1)) ** 2).sum()
print(f"ring of oscillators: energy drift {abs(energy(s.y[-1]) - energy(q0)):.2e} over {s.naccpt} steps")
assert abs(energy(s.y[-1]) - energy(q0)) < 1e-6
