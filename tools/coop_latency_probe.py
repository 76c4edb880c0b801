"""How long does ONE step attempt take in each stepping kernel when the machine is nearly empty?

The tail of a solve is bound by the longest trajectory: time = attempts x (time of one attempt of one wave).  This probe
runs small C2 batches (so every wave has a SIMD to itself) through the thread-per-trajectory kernel (variant 2) and the
lane-cooperative kernel (variant 3) and prints kernel time / (longest trajectory's attempts).
Run on the MI355X:  python tools/coop_latency_probe.py
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W

dev = torch.device("cuda", 0)
ctx = ivp_amd.Context(0)
prob = ivp_amd.CR3BP()
rows = []
for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FMA):
    for B in (64, 512, 4096, 8192):
        y0, p, t0, t1 = W.cr3bp_batch(B)
        y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
        for variant in (2, 3):
            opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp, variant=variant, profile=1)
            best = None
            for rep in range(4):
                out = ivp_amd.solve_ivp_batch(prob, t0, t1, y0d, pd, opts, ctx)
                st = out.stats
                if best is None or st["step_kernel_ms"] < best["step_kernel_ms"]:
                    best = dict(st)
            att = (out.nstep).to(torch.int64)
            mx, mean = int(att.max()), float(att.double().mean())
            row = dict(fp=fp.name, B=B, variant=variant, kernel_ms=best["step_kernel_ms"], launches=best["launches"],
                       coop_launches=best.get("coop_launches"), max_attempts=mx, mean_attempts=mean,
                       us_per_attempt_of_longest=1e3 * best["step_kernel_ms"] / mx)
            rows.append(row)
            print(json.dumps(row), flush=True)
