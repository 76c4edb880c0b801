"""ad hoc: where does the cooperative kernel pay off? (not a test)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
def bench(name, f, y0, p, t0, t1, method, tol):
    y0d = torch.as_tensor(np.ascontiguousarray(y0), device=dev); pd = None if p is None else torch.as_tensor(np.ascontiguousarray(p), device=dev)
    res = []
    for variant in (2, 3):
        o = ivp_amd.Options(method=method, rtol=tol[0], atol=tol[1], variant=variant)
        out = ivp_amd.solve_ivp_batch(f, t0, t1, y0d, pd, o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t = time.perf_counter()
            out = ivp_amd.solve_ivp_batch(f, t0, t1, y0d, pd, o, None, out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        res.append(min(ts) * 1e3)
    print(f"{name} {method} B={y0.shape[1]}: resident {res[0]:.3f} ms, cooperative {res[1]:.3f} ms  ratio {res[0]/res[1]:.2f}", flush=True)
for B in (8, 512):
    y0, p, t0, t1 = W.cr3bp_batch(100000)
    bench("cr3bp", ivp_amd.CR3BP(), y0[:, :B], p[:, :B], t0, t1, "DOP853", (1e-8, 1e-10))
    bench("cr3bp", ivp_amd.CR3BP(), y0[:, :B], p[:, :B], t0, t1, "DOPRI5", (1e-6, 1e-9))
    rng = np.random.default_rng(1)
    yl = 1.0 + 0.1 * rng.standard_normal((3, B)); pl = np.repeat(np.array([[10.0], [28.0], [8 / 3]]), B, axis=1)
    bench("lorenz", ivp_amd.Lorenz(), yl, pl, 0.0, 10.0, "DOPRI5", (1e-6, 1e-9))
    bench("lorenz", ivp_amd.Lorenz(), yl, pl, 0.0, 10.0, "DOP853", (1e-8, 1e-10))
    ys = rng.standard_normal((2, B))
    bench("sho", ivp_amd.SHO(), ys, None, 0.0, 50.0, "DOPRI5", (1e-6, 1e-9))
