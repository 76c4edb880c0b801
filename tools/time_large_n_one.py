#!/usr/bin/env python3
"""One N = 100 linear-decay system (BDF, rtol 1e-5) or one dense 64-state system (rtol 1e-6): the per-step cost of the large-n BDF
path, for rocprofv3 --kernel-trace and the phase-clock build.
  python tools/time_large_n_one.py [B] [variant] [decay100|dense64]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ivp_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
which = sys.argv[3] if len(sys.argv) > 3 else "decay100"
rng = np.random.default_rng(5)
if which == "dense64":
    prob, t1 = ivp_amd.Dense64(), 0.6
    y0 = torch.as_tensor(1.0 + 0.5 * rng.standard_normal((64, B)), device="cuda:0")
    par = torch.as_tensor(np.full((1, B), 3.0) * (1.0 + 0.2 * rng.uniform(-1, 1, (1, B))), device="cuda:0")
    o = ivp_amd.Options(method="BDF", rtol=1e-6, atol=1e-9, variant=variant)
else:
    prob, t1, par = ivp_amd.LinearDecay100(), 3.0, None
    y0 = torch.as_tensor(1.0 + 0.3 * rng.standard_normal((100, B)), device="cuda:0")
    o = ivp_amd.Options(method="BDF", rtol=1e-5, atol=1e-8, variant=variant)
r = ivp_amd.solve_ivp_batch(prob, 0.0, t1, y0, par, o)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter()
    r = ivp_amd.solve_ivp_batch(prob, 0.0, t1, y0, par, o, out=r)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(dict(ms=float(np.median(ts)), nstep=int(r.nstep.max()), naccpt=int(r.naccpt.max()), nfev=int(r.nfev.max()), njev=int(r.njev.max()), nlu=int(r.nlu.max())))
