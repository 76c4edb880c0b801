"""Per-launch timeline of one solve of a bench workload (IVP_TRACE_LAUNCHES=1 makes the library print it).
Run on the MI355X:  IVP_TRACE_LAUNCHES=1 python tools/trace_launches.py [c2|c3|c5] [strict|fma]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ivp_amd
from ivp_amd import workloads as W
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
fp = ivp_amd.FpMode.FMA if (len(sys.argv) > 2 and sys.argv[2] == "fma") else ivp_amd.FpMode.STRICT
dev = torch.device("cuda", 0)
ctx = ivp_amd.Context(0)
prob = getattr(ivp_amd, wl["problem"])()
y0, p, t0, t1 = getattr(W, wl["gen"])(wl["B"])
y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
t1d = torch.as_tensor(t1, device=dev) if hasattr(t1, "__len__") else t1
for profile in (0, 0, 1):
    opts = ivp_amd.Options(method=wl["method"], rtol=wl["rtol"], atol=wl["atol"], fp_mode=fp, profile=profile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = ivp_amd.solve_ivp_batch(prob, t0, t1d, y0d, pd, opts, ctx)
    e1.record(); torch.cuda.synchronize()
    print("profile", profile, "solve ms", e0.elapsed_time(e1), file=sys.stderr)
print({k: v for k, v in out.stats.items()}, file=sys.stderr)
