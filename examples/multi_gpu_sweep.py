"""One batch over several GPUs from ONE process and ONE host thread (`ivp_batch_solve_multi` in the C ABI): the batch is
cut into contiguous shards after a fixed permutation, every shard runs on its own context / device, and the end states
are gathered with peer copies.  On a one-GPU box the shards share device 0 (still independent contexts and streams).
The one-process-per-GPU form over RCCL is `ivp_amd.distributed.solve_ivp_sharded` (see bench.py --gpus N)."""
import time

import numpy as np
import torch

import ivp_amd
from ivp_amd import CR3BP, Options, workloads
from ivp_amd.distributed import solve_ivp_batch_multi

B = 40_000
y0, mu, t0, t1 = workloads.cr3bp_batch(B)
perm = workloads.shard_permutation(B)
ngpu = torch.cuda.device_count()
devices = list(range(ngpu)) if ngpu > 1 else [0, 0]
opts = Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
ctxs = [ivp_amd.Context(d) for d in devices]
r = solve_ivp_batch_multi(CR3BP(), t0, t1, y0, mu, opts, devices=devices, contexts=ctxs, permutation=perm)   # warm-up
t = time.perf_counter()
r = solve_ivp_batch_multi(CR3BP(), t0, t1, y0, mu, opts, devices=devices, contexts=ctxs, permutation=perm)
dt = time.perf_counter() - t
print(f"{B} trajectories over {len(devices)} contexts on {ngpu} GPU(s): {dt * 1e3:.2f} ms incl. shard upload and gather, "
      f"{int(r.naccpt.sum())} accepted steps, status histogram {torch.bincount(r.status.to(torch.int64)).tolist()} "
      f"(index = the crate's Status enum; a perturbed orbit that hits the Moon ends with StepSizeTooSmall)")
dev = torch.device("cuda:0")
one = ivp_amd.solve_ivp_batch(CR3BP(), t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(mu, device=dev), opts)
assert torch.equal(one.y_end, r.y_end.to(dev)) and torch.equal(one.naccpt, r.naccpt.to(dev))
print("identical to the single-context solve, bit for bit")
