#!/usr/bin/env python3
"""Batched stiff 256-cell heat equations (kappa = 4000, BDF, rtol 1e-5): ms per solve for 1 / 256 / 1024 systems; one JSON line.
  python tools/time_large_n_heat.py [t1]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ivp_amd  # noqa: E402

t1 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
rng = np.random.default_rng(5)
x = np.linspace(0.0, 1.0, 258)[1:-1]
out = {"problem": "heat1d256, kappa 4000 (+-10 %), BDF, rtol 1e-5, atol 1e-8", "t1": t1}
for B in (1, 256, 1024):
    y0 = np.sin(np.pi * x)[:, None] + 0.05 * rng.standard_normal((256, B))
    kap = 4000.0 * (1.0 + 0.1 * rng.uniform(-1, 1, (1, B)))
    yd, pd = torch.as_tensor(y0, device="cuda:0"), torch.as_tensor(kap, device="cuda:0")
    o = ivp_amd.Options(method="BDF", rtol=1e-5, atol=1e-8)
    r = ivp_amd.solve_ivp_batch(ivp_amd.Heat1D256(4000.0), 0.0, t1, yd, pd, o)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        r = ivp_amd.solve_ivp_batch(ivp_amd.Heat1D256(4000.0), 0.0, t1, yd, pd, o, out=r)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    out[str(B)] = {"ms": float(np.median(ts)), "accepted": int(r.naccpt.sum().item()), "nlu": int(r.nlu.sum().item()), "ok": bool((r.status == 0).all().item()),
                   "y_sum": float(r.y_end.sum().item())}
print(json.dumps(out))
