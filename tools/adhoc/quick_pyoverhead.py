"""ad hoc: where the host time of one solve_ivp_batch call goes (not a test)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
y0, p, t0, t1 = W.cr3bp_batch(8)
y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
f = ivp_amd.CR3BP()
out = ivp_amd.solve_ivp_batch(f, t0, t1, y0d, pd, o)
torch.cuda.synchronize()
def run(n):
    global out
    for _ in range(n):
        out = ivp_amd.solve_ivp_batch(f, t0, t1, y0d, pd, o, None, out)
run(50)
t = time.perf_counter(); run(200); dt = (time.perf_counter() - t) / 200
print(f"B=8 wall per call {dt*1e6:.1f} us")
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
