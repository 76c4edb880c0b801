"""Stiff Van der Pol oscillator with the variable-order BDF method (the reference's examples/van_der_pol.rs; BASELINE
config C5 is 10 000 of these with mu ~ 1000)."""
import numpy as np

from ivp_amd import Options, StiffVanDerPol, VanDerPol, solve_ivp, solve_ivp_batch

sol = solve_ivp(StiffVanDerPol(eps=1e-3), 0.0, 2.0, [2.0, 0.0], Options(method="BDF", rtol=1e-6, atol=1e-8))
print(f"eps = 1e-3: status {sol.status.name}, {sol.naccpt} steps, nfev {sol.nfev}, njev {sol.njev}, nlu {sol.nlu}, y(2) = {sol.y[-1]}")
mu = np.array([[10.0, 100.0, 1000.0]])
r = solve_ivp_batch(VanDerPol(), 0.0, 3000.0, np.repeat([[2.0], [0.0]], 3, axis=1), mu, Options(method="BDF", rtol=1e-4, atol=1e-6))
for m, y, na in zip(mu[0], r.y_end.T, r.naccpt):
    print(f"mu = {m:6.0f}: y(3000) = {y}, {int(na)} accepted steps")
assert (r.status == 0).all()
