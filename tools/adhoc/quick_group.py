"""ad hoc: wave-per-trajectory throughput (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd

dev = torch.device("cuda:0")
for name, f, n, B, t1, par in (("decay100", ivp_amd.LinearDecay100(), 100, 20000, 10.0, None),
                               ("heat256", ivp_amd.Heat1D256(100.0), 256, 20000, 0.5, 100.0)):
    rng = np.random.default_rng(0)
    y0 = torch.as_tensor(rng.uniform(0, 1, (n, B)), device=dev)
    p = None if par is None else torch.full((1, B), par, dtype=torch.float64, device=dev)
    for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FAST):
        o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp)
        out = ivp_amd.solve_ivp_batch(f, 0.0, t1, y0, p, o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        K = 5
        for _ in range(K):
            out = ivp_amd.solve_ivp_batch(f, 0.0, t1, y0, p, o, None, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / K
        att = int(out.nstep.sum())
        print(f"{name} B={B} fp={fp.name}: {dt*1e3:.2f} ms/solve, attempts {att}, {att/dt:.3e} attempts/s, "
              f"{att*n/dt:.3e} component-steps/s, status {np.bincount(out.status.cpu().numpy())}", flush=True)
