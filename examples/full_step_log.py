"""Every accepted step of every trajectory of a batch -- the reference's `Solution.t` / `Solution.y`, which are Vecs that
grow with each step (src/solve/solout.rs:387-428) -- in CSR form: two passes on the device (count, fill), memory equal to
the number of records.  Also: more event occurrences than the default buffers hold."""
import numpy as np
import torch

import ivp_amd
from ivp_amd import CR3BP, EventConfig, Options, SHOZeroEvent, solve_ivp, solve_ivp_batch_logged, workloads

B = 5000
y0, mu, t0, t1 = workloads.cr3bp_batch(B)
dev = torch.device("cuda:0")
log = solve_ivp_batch_logged(CR3BP(), t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(mu, device=dev),
                             Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
total = int(log.log_offsets[-1])
print(f"{B} trajectories: {total} records ({total * 56 / 1e6:.1f} MB); a dense [max, n, B] log would hold "
      f"{int(log.n_log.max()) * B} ({int(log.n_log.max()) * B * 56 / 1e6:.1f} MB)")
t, y = log.log_of(0)                                   # trajectory 0 = the unperturbed Arenstorf orbit
one = solve_ivp(CR3BP(mu=float(mu[0, 0])), t0, t1, y0[:, 0], Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
assert np.array_equal(t.cpu().numpy(), one.t) and np.array_equal(y.cpu().numpy(), one.y)
print(f"trajectory 0: {len(one.t)} steps, identical to solve_ivp() of that orbit; closest lunar approach "
      f"{np.sqrt((one.y[:, 0] - 1 + mu[0, 0]) ** 2 + one.y[:, 1] ** 2).min():.4f}")

# 120 zero crossings of a harmonic oscillator through event buffers sized for 64: solve_ivp reruns until all of them fit
s = solve_ivp(SHOZeroEvent(EventConfig()), 0.0, 120 * np.pi, [1.0, 0.0], Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
print(f"SHO over 60 periods: {len(s.t_events[0])} zero crossings recorded, spacing error {np.abs(np.diff(s.t_events[0]) - np.pi).max():.1e}")
assert len(s.t_events[0]) == 120
