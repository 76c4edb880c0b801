"""ad hoc: C2 wall time (not a test); policy knobs come from IVP_EXP_* environment variables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
y0, p, t0, t1 = W.cr3bp_batch(100000)
y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, profile=1)
out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o)
torch.cuda.synchronize()
ts = []
for _ in range(60):
    t = time.perf_counter()
    out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, None, out)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
st = out.stats
print(f"bulk={os.environ.get('IVP_EXP_BULK_LAUNCHES')} cap={os.environ.get('IVP_EXP_COOP_CAP')}: min {min(ts)*1e3:.3f} median {np.median(ts)*1e3:.3f} ms launches {st['launches']} coop {st['coop_launches']} kernel {st['step_kernel_ms']:.3f} coop_ms {st['coop_kernel_ms']:.3f}", flush=True)
