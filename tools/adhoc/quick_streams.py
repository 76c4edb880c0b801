"""Ad-hoc GPU experiment: K complete C2 solves issued from S host threads on S streams/contexts."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, '.')
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
y0, p, t0, t1 = W.cr3bp_batch(100_000)
y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
prob = ivp_amd.CR3BP()
for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FAST):
    for S in (1, 2, 3, 4):
        K = 48
        opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp)
        ctxs = [ivp_amd.Context(0) for _ in range(S)]
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        outs = [None] * S
        def work(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    outs[i] = ivp_amd.solve_ivp_batch(prob, t0, t1, y0d, pd, opts, ctxs[i], outs[i])
        for i in range(S): work(i, 2)
        torch.cuda.synchronize()
        t = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, K // S)) for i in range(S)]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        acc = int(outs[0].naccpt.sum().item())
        print(f"{fp.name} streams={S}: {dt/K*1e3:.3f} ms/solve  {acc*K/dt:.3e} steps/s", flush=True)
