#!/usr/bin/env python3
"""Single-GPU prediction of the C4 strong-scaling curve (BASELINE config C4: ONE batch of perturbed Arenstorf orbits, fixed
permutation, contiguous shards, RCCL gather of the end states): every rank's shard of the permuted batch is timed ALONE on
one MI355X; the maximum over the ranks is what the job takes (gather excluded: 1.2 MB per rank at 100k).

  python tools/strong_scaling_prediction.py [--batches 100000 400000 800000 1600000] > profiles/rNN_strong_scaling_prediction.json

The 100k batch is BASELINE's; the larger ones show where ">= 6x at 8 GPUs" becomes reachable: the floor of a shard is the
sequential attempt chain of its slowest trajectory (~700 attempts x 1.7 us), which does not shrink with the shard, so the
speed-up follows the share of the wall time that is throughput-bound."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivp_amd  # noqa: E402
from ivp_amd import workloads as W  # noqa: E402
from ivp_amd.distributed import shard_bounds  # noqa: E402
from kernel_sha import kernel_sources_sha256  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[100_000, 400_000, 800_000, 1_600_000])
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--max-steps", type=int, default=5000,
                    help="Options.max_steps of every solve.  BASELINE's 100k batch needs at most 702 attempts per trajectory; larger draws of "
                         "the same perturbed orbits contain a few COLLISION orbits (400k: one trajectory with > 200 000 steps, two that end in "
                         "StepSizeTooSmall) whose sequential time loops would set the wall time of any batch that contains them -- they end "
                         "with NeedLargerNMax here, as they would in a production sweep with a step budget")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, max_steps=a.max_steps)
    prob, ctx = ivp_amd.CR3BP(), ivp_amd.Context(0)

    def timeit(ys, ps, t1):
        o = [None]

        def f():
            o[0] = ivp_amd.solve_ivp_batch(prob, 0.0, t1, ys, ps, opts, ctx, o[0])
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts)) * 1e3, int(o[0].nstep.max().item())

    out = {"kernel_sources_sha256": kernel_sources_sha256(), "what": __doc__.split("\n\n")[0], "fp_mode": "strict", "max_steps": a.max_steps, "batches": {}}
    for B in a.batches:
        y0, p, t0, t1 = W.cr3bp_batch(B)
        perm = W.shard_permutation(B)
        res = {}
        full = timeit(torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev), t1)
        res["1"] = {"ms": full[0], "max_attempts": full[1]}
        for N in (2, 4, 8):
            per = []
            for r in range(N):
                lo, hi = shard_bounds(B, N, r)
                idx = perm[lo:hi]
                per.append(timeit(torch.as_tensor(np.ascontiguousarray(y0[:, idx]), device=dev),
                                  torch.as_tensor(np.ascontiguousarray(p[:, idx]), device=dev), t1))
            res[str(N)] = {"per_rank_ms": [round(x, 3) for x, _ in per], "per_rank_max_attempts": [b for _, b in per],
                           "max_ms": max(x for x, _ in per), "speedup": full[0] / max(x for x, _ in per)}
        out["batches"][str(B)] = res
        print(f"B = {B}: N = 1 {full[0]:.2f} ms; " + ", ".join(f"N = {N}: {res[str(N)]['max_ms']:.2f} ms ({res[str(N)]['speedup']:.2f}x)" for N in (2, 4, 8)),
              file=sys.stderr, flush=True)
    ok = [int(B) for B, r in out["batches"].items() if r["8"]["speedup"] >= 6.0]
    out["smallest_batch_with_6x_at_8_gpus"] = min(ok) if ok else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
