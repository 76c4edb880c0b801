#!/usr/bin/env python3
"""Helper of tests/test_gpu_deferred_events.py: solves a hiprtc CR3BP batch with two event functions (x-axis and y-axis crossings,
none terminal) and writes every output to an .npz.  The event-refinement mode comes from the environment
(IVP_TUNE_DEFER_EVENTS, read once per process by the library), hence a process of its own.
  python tests/helpers/events_dump.py OUT.npz METHOD B"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ivp_amd  # noqa: E402
from ivp_amd import workloads as W  # noqa: E402

SRC = r"""
__device__ void ode(double t, const double* s, double* d, const double* p)
{
    const double mu = p[0];
    const double x = s[0], y = s[1], z = s[2], vx = s[3], vy = s[4], vz = s[5];
    const double a = x + mu, b = x - 1.0 + mu;
    const double r1 = sqrt(a * a + y * y + z * z), r2 = sqrt(b * b + y * y + z * z);
    const double r13 = r1 * r1 * r1, r23 = r2 * r2 * r2;
    d[0] = vx; d[1] = vy; d[2] = vz;
    d[3] = x + 2.0 * vy - (1.0 - mu) * (x + mu) / r13 - mu * (x - 1.0 + mu) / r23;
    d[4] = y - 2.0 * vx - (1.0 - mu) * y / r13 - mu * y / r23;
    d[5] = -(1.0 - mu) * z / r13 - mu * z / r23;
}
__device__ void events(double t, const double* s, double* g, const double* p) { g[0] = s[1]; g[1] = s[0]; }
"""

out, method, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
y0, p, t0, t1 = W.cr3bp_batch(100_000)   # BASELINE C2's batch (other sizes draw other perturbations: 20 000 holds a collision orbit of 1e7 steps)
y0, p = np.ascontiguousarray(y0[:, :B]), np.ascontiguousarray(p[:, :B])
dev = torch.device("cuda:0")
f = ivp_amd.DeviceIVP(SRC, n=6, params=(W.ARENSTORF_MU,), events=[ivp_amd.EventConfig(), ivp_amd.EventConfig().negative()])
tol = dict(rtol=1e-6, atol=1e-9) if method == "DOPRI5" else dict(rtol=1e-8, atol=1e-10)
res = {}
for name, extra in (("end", {}), ("teval", dict(t_eval=np.linspace(t0, t1, 9))), ("log", dict(max_log=1024))):
    o = ivp_amd.Options(method=method, max_events=6, **tol, **extra)   # 6 slots: the x-axis event overflows them on purpose
    r = ivp_amd.solve_ivp_batch(f, t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev), o)
    torch.cuda.synchronize()
    for k in ("y_end", "t_end", "h_next", "status", "nfev", "naccpt", "nrejct", "t_events", "y_events", "n_event_hits", "y_eval", "n_filled", "t_log", "y_log", "n_log"):
        v = getattr(r, k, None)
        if v is not None:
            res[f"{name}.{k}"] = v.cpu().numpy()
np.savez(out, **res)
print("ok", {k: v.shape for k, v in res.items() if k.endswith("n_event_hits")}, int(res["end.n_event_hits"].sum()))
