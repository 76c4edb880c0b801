"""ad hoc: C2 timing + result hash with an alternative build of the library (not a test): quick_lib_ab.py [lib.so]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ivp_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
y0, p, t0, t1 = W.cr3bp_batch(100000)
y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, profile=1)
out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    t = time.perf_counter()
    out = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, None, out)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
h = hashlib.sha256(out.y_end.cpu().numpy().tobytes() + out.naccpt.cpu().numpy().tobytes()).hexdigest()[:16]
st = out.stats
print(f"{os.path.basename(_lib.LIB_PATH)}: min {min(ts)*1e3:.3f} median {np.median(ts)*1e3:.3f} ms kernel {st['step_kernel_ms']:.3f} coop {st['coop_kernel_ms']:.3f} hash {h}", flush=True)
