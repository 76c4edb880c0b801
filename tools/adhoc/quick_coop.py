"""ad hoc: C2 timing by kernel variant (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
for B in (100000, 1000, 8):
    y0, p, t0, t1 = W.cr3bp_batch(B)
    y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
    for variant in (0, 1, 2, 3):
        o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, variant=variant, profile=1)
        out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t = time.perf_counter()
            out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, None, out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        print(f"B={B} variant={variant}: min {min(ts)*1e3:.3f} ms median {np.median(ts)*1e3:.3f} ms launches {out.stats.get('launches')} kernel_ms {out.stats.get('step_kernel_ms'):.3f}", flush=True)
