"""10 000 perturbed Arenstorf orbits of the Earth-Moon restricted three-body problem, integrated at once on the GPU
(the reference's examples/cr3bp.rs solves ONE with DOP853; BASELINE config C2 is this batch at 100k with DOPRI5)."""
import time

import numpy as np
import torch

import ivp_amd
from ivp_amd import CR3BP, Method, Options, solve_ivp, solve_ivp_batch, workloads


def jacobi(s, mu):
    x, y, z, vx, vy, vz = s
    r1 = np.sqrt((x + mu) ** 2 + y ** 2 + z ** 2)
    r2 = np.sqrt((x - 1 + mu) ** 2 + y ** 2 + z ** 2)
    return x ** 2 + y ** 2 + 2 * (1 - mu) / r1 + 2 * mu / r2 - (vx ** 2 + vy ** 2 + vz ** 2)


# one orbit, the crate's call shape: solve_ivp(&f, x0, xend, &y0, options)
period = workloads.ARENSTORF_PERIOD
y0 = [0.994, 0.0, 0.0, 0.0, -2.00158510637908252240537862224, 0.0]
sol = solve_ivp(CR3BP(mu=workloads.ARENSTORF_MU), 0.0, period, y0, Options(method=Method.DOP853, rtol=1e-12, atol=1e-14, dense_output=True))
print(f"single orbit: status {sol.status.name}, nfev {sol.nfev}, {len(sol.t)} steps, closure error {np.abs(sol.y[-1] - y0).max():.2e}")
print(f"  dense output at T/2: {sol.sol(period / 2)[:2]}")

# the batch
B = 10_000
y0b, mu, t0, t1 = workloads.cr3bp_batch(B)
dev = torch.device("cuda:0")
y0d, mud = torch.as_tensor(y0b, device=dev), torch.as_tensor(mu, device=dev)
opts = Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
r = solve_ivp_batch(CR3BP(), t0, t1, y0d, mud, opts)      # warm-up (allocations)
torch.cuda.synchronize()
t = time.perf_counter()
r = solve_ivp_batch(CR3BP(), t0, t1, y0d, mud, opts, None, r)
torch.cuda.synchronize()
dt = time.perf_counter() - t
acc = int(r.naccpt.sum())
drift = np.abs(jacobi(r.y_end.cpu().numpy(), mu[0]) - jacobi(y0b, mu[0]))
print(f"batch of {B}: {dt * 1e3:.2f} ms, {acc} accepted steps = {acc / dt:.3e} steps/s, all success: {bool((r.status == 0).all())}")
print(f"  Jacobi-constant drift: median {np.median(drift):.2e}, max {drift.max():.2e}")
assert (r.status == 0).all() and np.median(drift) < 1e-3
