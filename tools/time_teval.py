import sys, time, os, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device('cuda:0')
wl = sys.argv[1]
if wl == 'c3':
    f = ivp_amd.VanDerPol(); y0, p, t0, t1 = W.vdp_batch(1_000_000); o = dict(method="DOP853", rtol=1e-8, atol=1e-10)
else:
    f = ivp_amd.CR3BP(); y0, p, t0, t1 = W.cr3bp_batch(100_000); o = dict(method="DOP853", rtol=1e-10, atol=1e-12)
t_hi = float(np.max(t1))
te = np.linspace(0.0, t_hi, 128)
y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
t1d = torch.as_tensor(t1, device=dev) if np.ndim(t1) else t1
ctx = ivp_amd.Context(0)
def timed(opts, k=5):
    out = ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, opts, ctx)
    out = ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, opts, ctx, out)
    torch.cuda.synchronize(); ts = []
    for _ in range(k):
        t = time.perf_counter(); out = ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, opts, ctx, out); torch.cuda.synchronize(); ts.append((time.perf_counter()-t)*1e3)
    return float(np.median(ts)), out
ms_end, _ = timed(ivp_amd.Options(**o))
ms, out = timed(ivp_amd.Options(t_eval=te, **o))
print(json.dumps({"workload": wl, "defer": os.environ.get("IVP_TUNE_DEFER_EVAL", "1"), "end_ms": ms_end, "t_eval_ms": ms, "records": int(out.n_filled.sum().item()),
                  "checksum": float(out.y_eval.double().abs().sum().item())}))
