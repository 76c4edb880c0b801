import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
for B in (100_000,):
    y0, p, t0, t1 = W.cr3bp_batch(B)
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FAST):
        for chunk in (32, 64, 128, 512):
            opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp, chunk_attempts=chunk, profile=2)
            for it in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                r = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, opts)
                torch.cuda.synchronize(); dt = time.perf_counter() - t
            acc = int(r.naccpt.sum().item())
            s = r.stats
            print(f"B={B} {fp.name} chunk={chunk}: wall {dt*1e3:.3f} ms  acc {acc}  steps/s {acc/dt:.3e}  kernel_ms {s['step_kernel_ms']:.3f} init_ms {s['init_kernel_ms']:.3f} launches {s['launches']} util {s['total_attempts']/max(s['lane_attempt_slots'],1):.3f}", flush=True)
