#!/usr/bin/env python3
"""Times the one-pass accepted-step log (ivp_batch_solve_logged_device) on a BASELINE workload, beside the end-state solve
and the counted two-pass log; meant to run under `rocprofv3 --kernel-trace --stats` as well (tools/profile_logged.sh).

  python tools/time_logged.py [c2|c3] [--fp strict|fma] [--solves K] [--only one|two|end]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ivp_amd  # noqa: E402
from ivp_amd import workloads as W  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_sha import kernel_sources_sha256  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="c2", choices=["c2", "c3"])
    ap.add_argument("--fp", default="strict", choices=["strict", "fma"])
    ap.add_argument("--solves", type=int, default=10)
    ap.add_argument("--only", default="all", choices=["all", "one", "two", "end", "none"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--fetch-reps", type=int, default=0, help="time scan + gather alone this many times (the pool of one logged solve)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.workload == "c2":
        f = ivp_amd.CR3BP()
        y0, p, t0, t1 = W.cr3bp_batch(a.batch or 100_000)
        o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    else:
        f = ivp_amd.VanDerPol()
        y0, p, t0, t1 = W.vdp_batch(a.batch or 1_000_000)
        o = dict(method="DOP853", rtol=1e-8, atol=1e-10)
    o["fp_mode"] = ivp_amd.FpMode.FMA if a.fp == "fma" else ivp_amd.FpMode.STRICT
    opts = ivp_amd.Options(**o)
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    t1d = torch.as_tensor(t1, device=dev) if np.ndim(t1) else t1
    ctx = ivp_amd.Context(0)

    def timed(fn, k):
        out = fn(None)
        out = fn(out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(k):
            t = time.perf_counter()
            out = fn(out)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        return float(np.median(ts)), float(np.min(ts)), out

    res = {"workload": a.workload, "fp": a.fp, "B": int(y0.shape[1]), "kernel_sources_sha256": kernel_sources_sha256()}
    if a.only in ("all", "end"):
        med, mn, out = timed(lambda prev: ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, opts, ctx, prev), a.solves)
        res["end_state_ms"] = {"median": med, "min": mn}
    if a.only in ("all", "one"):
        med, mn, out = timed(lambda prev: ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, opts, ctx, out=prev), a.solves)
        res["one_pass_ms"] = {"median": med, "min": mn, "log_info": out.log_info}
        del out
    if a.fetch_reps:
        # the second half alone (scan + gather kernel): the records of the solve above are fetched again and again from the pool
        import ctypes as C
        from ivp_amd import _lib
        r = ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, opts, ctx)
        total = int(r.log_offsets[-1])
        tb, yb = torch.empty(total, dtype=torch.float64, device=dev), torch.empty((total, y0.shape[0]), dtype=torch.float64, device=dev)
        sl = _lib.StepLogT()
        sl.t, sl.y, sl.capacity = tb.data_ptr(), yb.data_ptr(), total
        ts = []
        for _ in range(a.fetch_reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            rc = ctx.lib.ivp_step_log_fetch_device(ctx.handle, C.byref(sl), None)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
            assert rc == 0, ctx.last_error()
        assert torch.equal(tb, r.t_log) and torch.equal(yb, r.y_log)
        byts = total * (y0.shape[0] + 1) * 8
        res["fetch_ms"] = {"median": float(np.median(ts)), "min": float(np.min(ts)), "records": total, "payload_GB": byts / 1e9,
                           "pool_used_GB": r.log_info["pool_used_bytes"] / 1e9,
                           "GBs_read_plus_written": (byts + r.log_info["pool_used_bytes"]) / (float(np.min(ts)) * 1e-3) / 1e9}
    if a.only in ("all", "two"):
        med, mn, out = timed(lambda prev: ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, opts, ctx, two_pass=True), max(3, a.solves // 2))
        res["two_pass_ms"] = {"median": med, "min": mn}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
