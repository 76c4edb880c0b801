"""Multi-GPU driver: one process per GPU, trajectories sharded across ranks, ONE gather at the end.

The reference has no parallelism of any kind; a "batch" there is B back-to-back solve_ivp()
calls.  Trajectories are independent, so the batch shards trivially: after a fixed permutation (which
equalises step-count skew, BASELINE config C4) rank r owns the contiguous slice [lo_r, hi_r) and runs the
stepping kernels on its own GPU with no communication at all.  The only collective is the final
all-gather of the end states / statistics (RCCL over xGMI when the backend is "nccl"; at C4 sizes it is a
600 kB-per-rank, latency-bound message, so one fused all-gather of a packed buffer beats several small ones).

`solve_fn` is the per-shard integrator; it defaults to the HIP path (`ivp_amd.solve_ivp_batch`).  The CPU
test-suite injects the oracle there to exercise the sharding + gather logic under gloo without a GPU.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import api

FIELDS_F64 = ("t_end", "h_next")
FIELDS_INT = ("status", "nfev", "nstep", "naccpt", "nrejct")


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous, balanced shards: the first B % world ranks get one extra trajectory."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def solve_ivp_sharded(f: api.IVP, t0, t1, y0: np.ndarray, params: Optional[np.ndarray], options: api.Options,
                      *, permutation: Optional[np.ndarray] = None, group=None, device=None,
                      solve_fn: Optional[Callable] = None, gather: bool = True) -> dict:
    """Integrate a batch held (replicated) as host arrays on every rank; returns the gathered result on every
    rank as numpy arrays in the ORIGINAL trajectory order: y_end[n,B], t_end, h_next, status, nfev, nstep,
    naccpt, nrejct.  With ``gather=False`` only this rank's shard is returned (plus its index list)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    n, B = y0.shape
    perm = np.arange(B) if permutation is None else np.asarray(permutation)
    lo, hi = shard_bounds(B, world, rank)
    idx = perm[lo:hi]
    t0a = np.atleast_1d(np.asarray(t0, dtype=np.float64))
    t1a = np.atleast_1d(np.asarray(t1, dtype=np.float64))
    sh_t0 = t0a if t0a.size == 1 else t0a[idx]
    sh_t1 = t1a if t1a.size == 1 else t1a[idx]
    sh_y0 = np.ascontiguousarray(y0[:, idx])
    sh_p = None if params is None else np.ascontiguousarray(np.asarray(params, dtype=np.float64)[:, idx])

    backend = dist.get_backend(group) if dist.is_initialized() else None
    use_cuda = backend == "nccl"   # the packed gather buffer follows the backend (RCCL: device memory; gloo: host memory)
    if solve_fn is None:
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        r = api.solve_ivp_batch(f, torch.as_tensor(sh_t0, device=dev) if sh_t0.size > 1 else float(sh_t0[0]),
                                torch.as_tensor(sh_t1, device=dev) if sh_t1.size > 1 else float(sh_t1[0]),
                                torch.as_tensor(sh_y0, device=dev),
                                None if sh_p is None else torch.as_tensor(sh_p, device=dev), options)
        shard = {k: getattr(r, k) for k in ("y_end",) + FIELDS_F64 + FIELDS_INT}
    else:
        shard = solve_fn(f, sh_t0, sh_t1, sh_y0, sh_p, options)
    if not gather or world == 1:
        out = {k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in shard.items()}
        if world == 1:
            return _unpermute(out, perm, n, B)
        out["index"] = idx
        return out

    # pack everything into one f64 buffer [n + 2 + 5, max_shard] so that a single collective moves it
    max_sh = shard_bounds(B, world, 0)[1] - shard_bounds(B, world, 0)[0]
    rows = n + len(FIELDS_F64) + len(FIELDS_INT)
    tdev = torch.device(f"cuda:{torch.cuda.current_device()}") if use_cuda else torch.device("cpu")
    pack = torch.zeros((rows, max_sh), dtype=torch.float64, device=tdev)
    m = hi - lo

    def as_t(v):
        t = v if hasattr(v, "device") and not isinstance(v, np.ndarray) else torch.as_tensor(np.ascontiguousarray(v))
        return t.to(tdev)

    pack[:n, :m] = as_t(shard["y_end"])
    row = n
    for k in FIELDS_F64:
        pack[row, :m] = as_t(shard[k])
        row += 1
    for k in FIELDS_INT:
        # counters are exact in f64 up to 2^53 steps; status is a small enum
        pack[row, :m] = as_t(shard[k]).to(torch.float64)
        row += 1
    parts = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(parts, pack, group=group)
    allp = torch.stack(parts).cpu().numpy()

    out = {"y_end": np.empty((n, B)), **{k: np.empty(B) for k in FIELDS_F64},
           "status": np.empty(B, dtype=np.int32), **{k: np.empty(B, dtype=np.uint64) for k in FIELDS_INT[1:]}}
    for r_ in range(world):
        a, b = shard_bounds(B, world, r_)
        mm = b - a
        out["y_end"][:, a:b] = allp[r_, :n, :mm]
        row = n
        for k in FIELDS_F64:
            out[k][a:b] = allp[r_, row, :mm]
            row += 1
        for k in FIELDS_INT:
            out[k][a:b] = allp[r_, row, :mm].astype(out[k].dtype)
            row += 1
    return _unpermute(out, perm, n, B)


def _unpermute(out: dict, perm: np.ndarray, n: int, B: int) -> dict:
    """Shard position q holds trajectory perm[q]; scatter back to original order."""
    res = {}
    for k, v in out.items():
        v = np.asarray(v)
        r = np.empty_like(v)
        r[..., perm] = v
        res[k] = r
    return res


class OverlappedGather:
    """Double-buffered, asynchronous all-gather of one result array per step, used by bench.py --gpus N.

    Step i's gather (RCCL over xGMI under backend "nccl": the C4 "gather of sol.y") runs while step i+1 integrates
    into the other buffer; ``slot()`` returns which of the two buffers the next step may write (after making sure the
    gather that last read it has completed), ``launch(k, tensor)`` starts the gather of that step's result, and
    ``drain()`` waits for everything outstanding.  Backend-agnostic (gloo on CPU tensors in the tests)."""

    def __init__(self, shape, dtype, device, group=None):
        import torch
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.gathered = [torch.empty((self.world,) + tuple(shape), dtype=dtype, device=device) for _ in range(2)]
        self.works = [None, None]
        self.steps = 0

    def slot(self) -> int:
        k = self.steps & 1
        self.steps += 1
        if self.works[k] is not None:   # the gather that last read this slot's result must have completed
            self.works[k].wait()
            self.works[k] = None
        return k

    def launch(self, k: int, tensor) -> None:
        dist = self._dist
        if dist.get_backend(self.group) == "gloo":   # gloo has no all_gather_into_tensor
            self.works[k] = dist.all_gather(list(self.gathered[k].unbind(0)), tensor, group=self.group, async_op=True)
        else:
            self.works[k] = dist.all_gather_into_tensor(self.gathered[k], tensor, group=self.group, async_op=True)

    def drain(self) -> None:
        for k in range(2):
            if self.works[k] is not None:
                self.works[k].wait()
                self.works[k] = None
