"""ad hoc: C3 (DOP853) timing by kernel variant (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
for B in (1000000, 10000, 8):
    y0, p, t0, t1 = W.vdp_batch(B)
    y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev); t1d = torch.as_tensor(t1, device=dev)
    for variant in (0, 1, 2, 3):
        if B == 1000000 and variant == 3: continue
        o = ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10, variant=variant, profile=1)
        out = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), t0, t1d, y0d, pd, o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            out = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), t0, t1d, y0d, pd, o, None, out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        st = out.stats
        print(f"B={B} variant={variant}: min {min(ts)*1e3:.3f} ms launches {st['launches']} kernel_ms {st['step_kernel_ms']:.3f} coop {st['coop_launches']} {st['coop_kernel_ms']:.3f}", flush=True)
