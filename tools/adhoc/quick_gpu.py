"""Ad-hoc GPU exploration (not a test): timings of the BASELINE configs under a few launch policies."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device("cuda:0")

def run(name, prob, y0, p, t0, t1, reps=5, **kw):
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    t1d = torch.as_tensor(np.atleast_1d(t1), device=dev) if np.ndim(t1) else t1
    opts = ivp_amd.Options(profile=2, **kw)
    best = 1e9
    for it in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = ivp_amd.solve_ivp_batch(prob, t0, t1d, y0d, pd, opts)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    acc = int(r.naccpt.sum().item()); s = r.stats
    att = s['total_attempts']
    print(f"{name}: wall {best*1e3:.3f} ms acc {acc} att {att} steps/s {acc/best:.3e} kern_ms {s['step_kernel_ms']:.3f} launches {s['launches']} util {att/max(s['lane_attempt_slots'],1):.3f} maxatt {int((r.nstep if kw.get('method')!='RK23' else r.naccpt+r.nrejct).max().item())}", flush=True)
    return r

if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3"]
    if "c2" in which:
        y0, p, t0, t1 = W.cr3bp_batch(100_000)
        for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FAST):
            for variant in (1, 2, 0):
                r = run(f"C2 {fp.name} variant={variant}", ivp_amd.CR3BP(), y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp, variant=variant)
        ns = r.nstep.cpu().numpy()
        print("attempt percentiles", np.percentile(ns, [0, 50, 90, 99, 99.9, 99.99, 100]), "count>256", (ns > 256).sum(), ">384", (ns > 384).sum())
    if "c5" in which:
        y0, p, t0, t1 = W.vdp_stiff_batch(10_000)
        for variant in (1, 2):
            r = run(f"C5 BDF variant={variant}", ivp_amd.VanDerPol(1000.0), y0, p, t0, t1, reps=3, method="BDF", rtol=1e-4, atol=1e-6, variant=variant)
        print("njev", int(r.njev.sum().item()), "nlu", int(r.nlu.sum().item()), "nfev", int(r.nfev.sum().item()))
    if "c3" in which:
        y0, p, t0, t1 = W.vdp_batch(1_000_000)
        for fp in (ivp_amd.FpMode.STRICT, ivp_amd.FpMode.FAST):
            for variant in (1, 2, 0):
                r = run(f"C3 {fp.name} variant={variant}", ivp_amd.VanDerPol(), y0, p, t0, t1, reps=3, method="DOP853", rtol=1e-8, atol=1e-10, fp_mode=fp, variant=variant)
        ns = r.nstep.cpu().numpy()
        print("attempt percentiles", np.percentile(ns, [0, 50, 90, 99, 100]))
