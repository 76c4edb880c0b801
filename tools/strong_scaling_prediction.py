"""Single-GPU prediction of the C4 strong-scaling curve: time every rank's shard of the permuted 100k batch alone."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
import ivp_amd
from ivp_amd import workloads as W
from ivp_amd.distributed import shard_bounds
dev = torch.device('cuda:0')
B = 100000
y0, p, t0, t1 = W.cr3bp_batch(B)
perm = W.shard_permutation(B)
opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
prob = ivp_amd.CR3BP(); ctx = ivp_amd.Context(0)
def timeit(ys, ps, n=15):
    o = [None]
    def f(): o[0] = ivp_amd.solve_ivp_batch(prob, t0, t1, ys, ps, opts, ctx, o[0])
    for _ in range(3): f()
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        t=time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t)
    return float(np.median(ts))*1e3, int(o[0].nstep.max().item())
res = {}
full = timeit(torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev))
res["1"] = {"ms": full[0], "max_attempts": full[1]}
for N in (2, 4, 8):
    per = []
    for r in range(N):
        lo, hi = shard_bounds(B, N, r); idx = perm[lo:hi]
        per.append(timeit(torch.as_tensor(np.ascontiguousarray(y0[:, idx]), device=dev), torch.as_tensor(np.ascontiguousarray(p[:, idx]), device=dev)))
    res[str(N)] = {"per_rank_ms": [round(a, 3) for a, _ in per], "per_rank_max_attempts": [b for _, b in per], "max_ms": max(a for a, _ in per)}
print(json.dumps(res))
