"""ad hoc: compare lean-kernel occupancy hints (not a test).  usage: python tests/quick_minwaves.py <lib.so>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ivp_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ivp_amd
from ivp_amd import workloads as W

dev = torch.device("cuda:0")
def run(name, f, gen, opts, K):
    y0, p, t0, t1 = gen
    y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
    t1d = t1 if np.isscalar(t1) else torch.as_tensor(t1, device=dev)
    for variant in (1, 0):
        o = ivp_amd.Options(variant=variant, **opts)
        out = ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(K):
            t = time.perf_counter()
            out = ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, o, None, out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        print(f"{name} variant={variant}: min {min(ts)*1e3:.3f} ms  median {np.median(ts)*1e3:.3f} ms", flush=True)

run("C2", ivp_amd.CR3BP(), W.cr3bp_batch(100000), dict(method="DOPRI5", rtol=1e-6, atol=1e-9), 15)
run("C3", ivp_amd.VanDerPol(), W.vdp_batch(1000000), dict(method="DOP853", rtol=1e-8, atol=1e-10), 5)
