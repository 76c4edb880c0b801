"""BASELINE config C1 / the reference's examples/exponential_decay.rs: y' = -k y sampled at t = 0..10."""
import numpy as np

from ivp_amd import ExponentialDecay, Options, solve_ivp

sol = solve_ivp(ExponentialDecay(k=0.5), 0.0, 10.0, [10.0], Options(method="DOPRI5", rtol=1e-8, atol=1e-10, t_eval=list(range(11))))
for t, y in zip(sol.t, sol.y[:, 0]):
    print(f"t = {t:4.1f}  y = {y:.8f}  exact = {10 * np.exp(-0.5 * t):.8f}")
print(f"status {sol.status.name}, nfev {sol.nfev}, accepted {sol.naccpt}, rejected {sol.nrejct}")
assert np.abs(sol.y[:, 0] - 10 * np.exp(-0.5 * sol.t)).max() < 1e-6
