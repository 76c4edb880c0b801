"""The problem list of the reference's own benchmark script (benches/benchmark.py:100-148) on the GPU path.

The reference times ONE trajectory per problem against SciPy; a GPU is the wrong tool for one small trajectory (its time
loop is sequential: microseconds per step attempt, however many lanes idle beside it), so every problem is timed twice:
one trajectory, and a batch of perturbed copies integrated together -- the regime this package exists for.
"""
import time

import numpy as np
import torch

import ivp_amd

PROBLEMS = [   # (name, problem, t_span, y0, methods, rtol, atol)                       benches/benchmark.py
    ("Van der Pol (non-stiff, mu=1)", ivp_amd.VanDerPol(1.0), (0.0, 100.0), [2.0, 0.0], ["RK45", "DOP853"], 1e-6, 1e-8),       # :104-112
    ("Van der Pol (stiff, mu=1000)", ivp_amd.VanDerPol(1000.0), (0.0, 3000.0), [2.0, 0.0], ["BDF"], 1e-4, 1e-6),               # :115-123 (Radau: outside the path)
    ("Lorenz System (chaotic)", ivp_amd.Lorenz(10.0, 28.0, 8.0 / 3.0), (0.0, 100.0), [1.0, 1.0, 1.0], ["RK45", "DOP853"], 1e-8, 1e-10),   # :126-134
    ("Large Linear System (N=100)", ivp_amd.LinearDecay100(), (0.0, 10.0), list(np.ones(100)), ["RK45"], 1e-6, 1e-8),          # :137-145
]
BATCH = 4096
dev = torch.device("cuda:0")
ctx = ivp_amd.default_context()
rng = np.random.default_rng(7)
print(f"{'problem':34s} {'method':7s} {'1 trajectory':>14s} {'steps':>7s} {BATCH:>6d} trajectories {'steps/s':>12s}")
for name, prob, (t0, t1), y0, methods, rtol, atol in PROBLEMS:
    for method in methods:
        opts = ivp_amd.Options(method=method, rtol=rtol, atol=atol)
        y1 = torch.as_tensor(np.asarray(y0, dtype=np.float64).reshape(-1, 1), device=dev)
        p1 = torch.as_tensor(np.asarray(prob.params(), dtype=np.float64).reshape(-1, 1), device=dev) if prob.n_params else None
        yb = np.asarray(y0, dtype=np.float64)[:, None] * (1.0 + 1e-3 * rng.standard_normal((len(y0), BATCH)))
        ybd = torch.as_tensor(yb, device=dev)
        pbd = p1.repeat(1, BATCH).contiguous() if p1 is not None else None
        times = []
        for yy, pp in ((y1, p1), (ybd, pbd)):
            out = ivp_amd.solve_ivp_batch(prob, t0, t1, yy, pp, opts, ctx)   # warm-up (and JIT-free: built-in problems)
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = ivp_amd.solve_ivp_batch(prob, t0, t1, yy, pp, opts, ctx, out)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t, int(out.naccpt.sum().item()), bool((out.status == 0).all().item())))
        (ts, ns, ok1), (tb, nb, okb) = times
        print(f"{name:34s} {method:7s} {ts * 1e3:11.2f} ms {ns:7d} {tb * 1e3:16.2f} ms {nb / tb:12.3e}" + ("" if ok1 and okb else "  (not all Success)"))
