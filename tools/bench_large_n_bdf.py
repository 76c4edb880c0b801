#!/usr/bin/env python3
"""Batched large-n BDF: factors of (I - cJ) resident in LDS (variant 2) vs in global memory (variant 1) vs the automatic
per-launch choice (variant 0: LDS while the active set fits two wavefronts per CU), 1 ... 20k systems per batch.
Prints one JSON line per (problem, batch size).
  python tools/bench_large_n_bdf.py > gpurun_out/large_n_bdf.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ivp_amd  # noqa: E402


def run(prob, name, y0, p, t1, opts, reps):
    dev = torch.device("cuda:0")
    y0d = torch.as_tensor(y0, device=dev)
    pd = None if p is None else torch.as_tensor(p, device=dev)
    out = {}
    ref = None
    for label, variant in (("lds", 2), ("global", 1), ("auto", 0)):
        o = ivp_amd.Options(variant=variant, **opts)
        r = ivp_amd.solve_ivp_batch(prob, 0.0, t1, y0d, pd, o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = ivp_amd.solve_ivp_batch(prob, 0.0, t1, y0d, pd, o, out=r)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        out[label + "_ms"] = ms
        if ref is None:
            ref = r.y_end.clone()
            out.update(accepted=int(r.naccpt.sum().item()), nlu=int(r.nlu.sum().item()), all_success=bool((r.status == 0).all().item()))
        else:
            out["same_bits"] = bool(torch.equal(ref, r.y_end))
    out.update(problem=name, B=int(y0.shape[1]), speedup=out["global_ms"] / out["lds_ms"])
    print(json.dumps(out), flush=True)


def main():
    rng = np.random.default_rng(5)
    sizes = [1, 64, 256, 1024, 4096, 20000]
    for B in sizes:
        reps = 3 if B >= 4096 else 5
        y0 = 1.0 + 0.3 * rng.standard_normal((100, B))
        run(ivp_amd.LinearDecay100(), "linear_decay100 (diagonal Jacobian)", y0, None, 3.0, dict(method="BDF", rtol=1e-5, atol=1e-8), reps)
        y0 = 1.0 + 0.5 * rng.standard_normal((64, B))
        k = np.full((1, B), 3.0) * (1.0 + 0.2 * rng.uniform(-1, 1, (1, B)))
        run(ivp_amd.Dense64(), "dense64 (full Jacobian)", y0, k, 0.6, dict(method="BDF", rtol=1e-6, atol=1e-9), reps)


if __name__ == "__main__":
    main()
