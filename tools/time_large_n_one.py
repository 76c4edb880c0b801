#!/usr/bin/env python3
"""One N = 100 linear-decay system (BDF, rtol 1e-5): the per-step cost of the large-n BDF path, for rocprofv3 --kernel-trace.
  python tools/time_large_n_one.py [B] [variant]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ivp_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(5)
y0 = torch.as_tensor(1.0 + 0.3 * rng.standard_normal((100, B)), device="cuda:0")
o = ivp_amd.Options(method="BDF", rtol=1e-5, atol=1e-8, variant=variant)
r = ivp_amd.solve_ivp_batch(ivp_amd.LinearDecay100(), 0.0, 3.0, y0, None, o)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter()
    r = ivp_amd.solve_ivp_batch(ivp_amd.LinearDecay100(), 0.0, 3.0, y0, None, o, out=r)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(dict(ms=float(np.median(ts)), nstep=int(r.nstep.max()), naccpt=int(r.naccpt.max()), nfev=int(r.nfev.max()), njev=int(r.njev.max()), nlu=int(r.nlu.max())))
