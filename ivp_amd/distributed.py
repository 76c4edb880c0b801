"""Multi-GPU drivers: trajectories sharded across GPUs, ONE gather at the end.

The reference has no parallelism of any kind; a "batch" there is B back-to-back solve_ivp()
calls (src/solve/solve_ivp.rs:99-313) with no coupling between them.  The batch therefore shards trivially: after a
fixed permutation (which equalises step-count skew, BASELINE config C4) shard r is the contiguous slice
[lo_r, hi_r) and is integrated by its own GPU with no communication at all.  The only data movement is the final
gather of the end states / statistics.

Two drivers, the same partitioning:

What is gathered is the reference's ``Solution`` (src/solve/solution.rs:7-20), not only the end states: with
``Options.t_eval`` the sampled trajectory (``y_eval`` / ``eval_idx`` / ``n_filled``) are members of the same arena (still
ONE collective), and with ``log=True`` every accepted step of every trajectory (``Solution.t`` / ``Solution.y``, the
"RCCL gather of sol.y" of BASELINE config C4) travels as a CSR log: the counts ride in the arena, the records in ONE
second collective of the padded per-rank record buffers, and the offsets are re-based so that trajectory b's records are
``t_log[log_offsets[b]:log_offsets[b+1]]`` in the ORIGINAL trajectory order on every rank.

* ``solve_ivp_sharded``  one PROCESS per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI, "gloo" in the
  CPU tests).  Every rank's result arrays are views into ONE contiguous byte arena (``ResultArena``): the kernels write
  their results straight into it and the gather is a single ``all_gather_into_tensor`` of raw bytes -- no packing
  kernels, no dtype punning, no host bounce.  At C4 sizes (100k trajectories over 8 GPUs) the arena is 1.2 MB per
  rank, i.e. a latency-bound message, which is why it is ONE collective.
* ``solve_ivp_batch_multi``  one process, one host THREAD, N contexts (``ivp_batch_solve_multi`` in the C ABI): the
  shards are driven through the resumable submit/poll entry points and gathered with peer copies
  (``hipMemcpyPeerAsync`` / peer-enabled 2-D copies over xGMI).  Both gathers move the same bytes
  (tests/test_gpu_multi.py compares them).

``solve_fn`` is the per-shard integrator; it defaults to the HIP path (``ivp_amd.solve_ivp_batch``).  The CPU
test-suite injects the oracle there to exercise the sharding + gather logic under gloo without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib, api

# end-state members of a shard's result, in arena order: (name, numpy dtype, rows per trajectory or None = n)
ARENA_FIELDS = (("y_end", np.float64, None), ("t_end", np.float64, 1), ("h_next", np.float64, 1),
                ("nfev", np.int64, 1), ("nstep", np.int64, 1), ("naccpt", np.int64, 1), ("nrejct", np.int64, 1),
                ("status", np.int32, 1))


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous, balanced shards: the first B % world ranks get one extra trajectory."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


class ResultArena:
    """Result arrays of ONE shard (capacity ``m`` trajectories, state dimension ``n``) as views into one contiguous byte
    buffer: the end-state members, with ``eval_rows`` > 0 also the t_eval samples (``y_eval [eval_rows, n, m]``,
    ``eval_idx [eval_rows, m]``, ``n_filled [m]``) and with ``log_counts`` the accepted-step counts ``n_log [m]``.
    ``solution()`` hands the views to ``solve_ivp_batch(out=...)`` so the kernels write into the arena;
    ``split(gathered, counts)`` turns the gathered ``[world, nbytes]`` buffer back into arrays."""

    def __init__(self, n: int, m: int, device, eval_rows: int = 0, log_counts: bool = False):
        import torch
        self.n, self.m, self.eval_rows = int(n), int(m), int(eval_rows)
        self.layout = []   # (name, torch dtype, rows, byte offset)
        off = 0
        tdt = {np.float64: torch.float64, np.int64: torch.int64, np.int32: torch.int32}
        fields = list(ARENA_FIELDS)
        if self.eval_rows > 0:
            fields += [("y_eval", np.float64, self.eval_rows * self.n), ("eval_idx", np.int32, self.eval_rows), ("n_filled", np.int32, 1)]
        if log_counts:
            fields += [("n_log", np.int32, 1)]
        self.fields = tuple(fields)
        for name, dt, rows in self.fields:
            rows = self.n if rows is None else rows
            self.layout.append((name, tdt[dt], rows, off))
            off += np.dtype(dt).itemsize * rows * self.m
            off = (off + 255) & ~255       # every member starts on a 256-byte boundary
        self.nbytes = off
        self.buf = torch.zeros(max(self.nbytes, 256), dtype=torch.uint8, device=device)
        self.views = self.views_of(self.buf)

    def views_of(self, buf) -> dict:
        """Typed views of a byte buffer with this arena's layout (``buf``: 1-D uint8, at least ``nbytes`` long)."""
        out = {}
        for name, tdt, rows, off in self.layout:
            nb = tdt.itemsize * rows * self.m
            v = buf[off:off + nb].view(tdt)
            if name == "y_eval":
                out[name] = v.view(self.eval_rows, self.n, self.m)
            else:
                out[name] = v.view(rows, self.m) if rows > 1 or name in ("y_end", "eval_idx") else v
        return out

    def solution(self, count: Optional[int] = None) -> api.BatchSolution:
        """A BatchSolution whose members are the arena's views.  Only a full arena (count == m) can be written in
        place: a shard with fewer trajectories has a different SoA stride and goes through ``store``."""
        assert count is None or count == self.m
        v = self.views
        return api.BatchSolution(y_end=v["y_end"], t_end=v["t_end"], status=v["status"], nfev=v["nfev"], nstep=v["nstep"],
                                 naccpt=v["naccpt"], nrejct=v["nrejct"], h_next=v["h_next"], y_eval=v.get("y_eval"),
                                 eval_idx=v.get("eval_idx"), n_filled=v.get("n_filled"), n_log=v.get("n_log"))

    def store(self, shard: dict, count: int) -> None:
        """Copy a shard result (``count`` <= m trajectories; tensors or numpy arrays) into the arena's first columns.
        Members the shard result does not carry (an integrator without eval_idx, say) are left as they are."""
        import torch
        for name, tdt, rows, _ in self.layout:
            if name not in shard or shard[name] is None:
                continue
            src = shard[name]
            t = src if api._is_torch(src) else torch.as_tensor(np.ascontiguousarray(src))
            t = t.to(device=self.buf.device, dtype=tdt)
            self.views[name][..., :count] = t.reshape(self.views[name][..., :count].shape)

    def split(self, gathered, counts: Sequence[int]) -> dict:
        """``gathered``: ``[world, nbytes]`` uint8.  Returns name -> array with the shards' columns concatenated in
        rank order (``sum(counts)`` columns)."""
        import torch
        parts = {name: [] for name, *_ in self.layout}
        for r, cnt in enumerate(counts):
            v = self.views_of(gathered[r])
            for name in parts:
                parts[name].append(v[name][..., :cnt])
        return {name: torch.cat(p, dim=-1) for name, p in parts.items()}


def _as_numpy(d: dict) -> dict:
    out = {}
    for k, v in d.items():
        a = v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)
        if k in ("nfev", "nstep", "naccpt", "nrejct"):
            a = a.astype(np.uint64)
        if k == "n_log":
            a = a.astype(np.uint32)
        out[k] = a
    return out


def _unpermute(out: dict, perm, B: int) -> dict:
    """Shard position q holds trajectory perm[q]; scatter back to original order (numpy or torch arrays)."""
    res = {}
    for k, v in out.items():
        if api._is_torch(v):
            import torch
            r = torch.empty_like(v)
            r.index_copy_(v.dim() - 1, torch.as_tensor(np.asarray(perm), device=v.device, dtype=torch.int64), v)
        else:
            v = np.asarray(v)
            r = np.empty_like(v)
            r[..., np.asarray(perm)] = v
        res[k] = r
    return res


def solve_ivp_sharded(f: api.IVP, t0, t1, y0: np.ndarray, params: Optional[np.ndarray], options: api.Options,
                      *, permutation: Optional[np.ndarray] = None, group=None, device=None,
                      solve_fn: Optional[Callable] = None, gather: bool = True, as_numpy: bool = True,
                      log: bool = False) -> dict:
    """Integrate a batch held (replicated) as host arrays on every rank; returns the gathered result on every
    rank in the ORIGINAL trajectory order: y_end[n,B], t_end, h_next, status, nfev, nstep, naccpt, nrejct
    (numpy arrays, or tensors on the gather device with ``as_numpy=False``).  With ``options.t_eval`` also the sampled
    trajectories ``y_eval [rows, n, B]``, ``eval_idx [rows, B]``, ``n_filled [B]`` (same collective); with ``log=True``
    (t_eval must be None) also ``Solution.t`` / ``Solution.y`` of every trajectory as a CSR log -- ``n_log [B]``,
    ``log_offsets [B + 1]``, ``t_log [total]``, ``y_log [total, n]`` -- moved by one more collective.
    With ``gather=False`` only this rank's shard is returned (plus its index list).  A rank whose shard is empty
    (world > B) still takes part in the collectives.
    ``solve_fn(f, t0, t1, y0, params, options)`` (tests) returns a dict of the end-state members, plus ``y_eval`` /
    ``n_filled`` (/ ``eval_idx``) when options.t_eval is set; it is called with ``log=True`` for the CSR log and then
    also returns ``n_log``, ``t_log [total]``, ``y_log [total, n]`` of its shard."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    n, B = y0.shape
    perm = np.arange(B) if permutation is None else np.asarray(permutation)
    lo, hi = shard_bounds(B, world, rank)
    m = hi - lo
    idx = perm[lo:hi]
    t0a = np.atleast_1d(np.asarray(t0, dtype=np.float64))
    t1a = np.atleast_1d(np.asarray(t1, dtype=np.float64))
    sh_t0 = t0a if t0a.size == 1 else t0a[idx]
    sh_t1 = t1a if t1a.size == 1 else t1a[idx]
    sh_y0 = np.ascontiguousarray(y0[:, idx])
    sh_p = None if params is None else np.ascontiguousarray(np.asarray(params, dtype=np.float64)[:, idx])

    backend = dist.get_backend(group) if dist.is_initialized() else None
    use_cuda = backend == "nccl" or (solve_fn is None and backend is None)   # arena follows the backend (RCCL: device; gloo: host)
    if solve_fn is None:
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    else:
        dev = torch.device("cpu")
    tdev = dev if use_cuda else torch.device("cpu")
    m_max = shard_bounds(B, world, 0)[1] - shard_bounds(B, world, 0)[0]
    if log and (options.t_eval is not None or options.t_eval_per_trajectory is not None):
        raise ValueError("the accepted-step log is what solve_ivp records when t_eval is None")
    eval_rows = 0 if options.t_eval is None else len(options.t_eval) + (1 if f.n_events() else 0)
    arena = ResultArena(n, m_max, tdev, eval_rows=max(eval_rows, 1) if options.t_eval is not None else 0, log_counts=log)
    if eval_rows:
        arena.views["eval_idx"].fill_(-1)

    dev_in = None
    if m > 0:
        if solve_fn is None:
            in_place = m == m_max and use_cuda and tdev == dev   # the kernels write straight into the arena
            dev_in = (torch.as_tensor(sh_t0, device=dev) if sh_t0.size > 1 else float(sh_t0[0]),
                      torch.as_tensor(sh_t1, device=dev) if sh_t1.size > 1 else float(sh_t1[0]),
                      torch.as_tensor(sh_y0, device=dev), None if sh_p is None else torch.as_tensor(sh_p, device=dev))
            if log:
                # ONE integration: end states and statistics into the arena, every accepted step into the context's page pool
                # and from there into this shard's CSR run (api.solve_ivp_batch_logged -> ivp_batch_solve_logged_device)
                r = api.solve_ivp_batch_logged(f, *dev_in, options, out=arena.solution() if in_place else None)
                local_log = {"t_log": r.t_log.to(tdev), "y_log": r.y_log.to(tdev)}
            else:
                r = api.solve_ivp_batch(f, *dev_in, options, out=arena.solution() if in_place else None)
            if not in_place:
                arena.store({k: getattr(r, k, None) for k, *_ in arena.fields}, m)
        else:
            shard_res = solve_fn(f, sh_t0, sh_t1, sh_y0, sh_p, options, log=True) if log else solve_fn(f, sh_t0, sh_t1, sh_y0, sh_p, options)
            arena.store(shard_res, m)
    if not gather or world == 1:
        out = {k: v[..., :m] for k, v in arena.views.items()}
        if log:
            out.update(_local_csr_log(solve_fn, locals().get("shard_res"), locals().get("local_log"), n, tdev))
        if world == 1:
            out = _unpermute_solution(out, perm, B)
            return _as_numpy(out) if as_numpy else out
        out = _as_numpy(out) if as_numpy else out
        out["index"] = idx
        return out

    # ---- collective 1: the arena (end states, statistics, t_eval samples, step counts) ----
    gathered = torch.empty((world, arena.buf.numel()), dtype=torch.uint8, device=tdev)
    if backend == "gloo":   # gloo has no all_gather_into_tensor
        dist.all_gather(list(gathered.unbind(0)), arena.buf, group=group)
    else:
        dist.all_gather_into_tensor(gathered, arena.buf, group=group)
    counts = [shard_bounds(B, world, r_)[1] - shard_bounds(B, world, r_)[0] for r_ in range(world)]
    out = arena.split(gathered, counts)
    if log:
        # ---- collective 2: the records.  Every rank knows every shard's counts now, hence the padded size ----
        nl = out["n_log"].to(torch.int64)
        bounds = np.cumsum([0] + counts)
        totals = [int(nl[bounds[r_]:bounds[r_ + 1]].sum().item()) for r_ in range(world)]
        cap = max(max(totals), 1)
        mine = _local_csr_log(solve_fn, locals().get("shard_res"), locals().get("local_log"), n, tdev) if m > 0 else None
        rec = torch.zeros(cap * (n + 1), dtype=torch.float64, device=tdev)     # [t_log | y_log], padded to the largest shard
        if mine is not None:
            tot = int(mine["t_log"].shape[0])
            rec[:tot] = mine["t_log"]
            rec[cap:cap + tot * n] = mine["y_log"].reshape(-1)
        grec = torch.empty((world, rec.numel()), dtype=torch.float64, device=tdev)
        if backend == "gloo":
            dist.all_gather(list(grec.unbind(0)), rec, group=group)
        else:
            dist.all_gather_into_tensor(grec, rec, group=group)
        out["t_log"] = torch.cat([grec[r_, :totals[r_]] for r_ in range(world)])
        out["y_log"] = torch.cat([grec[r_, cap:cap + totals[r_] * n].view(totals[r_], n) for r_ in range(world)])
        off = torch.zeros(B + 1, dtype=torch.int64, device=tdev)
        torch.cumsum(nl, 0, out=off[1:])
        out["log_offsets"] = off
    out = _unpermute_solution(out, perm, B)
    return _as_numpy(out) if as_numpy else out


def _local_csr_log(solve_fn, shard_res, local_log, n, tdev) -> dict:
    """This rank's records in its own (shard) order: what the one-pass logged solve of the HIP path returned, or the arrays
    the injected integrator returned."""
    import torch
    if solve_fn is not None:
        return {"t_log": torch.as_tensor(np.ascontiguousarray(shard_res["t_log"], dtype=np.float64)).to(tdev),
                "y_log": torch.as_tensor(np.ascontiguousarray(shard_res["y_log"], dtype=np.float64)).reshape(-1, n).to(tdev)}
    return local_log


def _unpermute_solution(out: dict, perm, B: int) -> dict:
    """Original trajectory order for every member: the SoA members by column, a CSR log record by record (shard
    position q holds trajectory perm[q]; its records move to the slot its count earns in the original order)."""
    csr = {k: out.pop(k) for k in ("t_log", "y_log", "log_offsets") if k in out}
    res = _unpermute(out, perm, B)
    if "t_log" in csr:
        import torch
        tl, yl = csr["t_log"], csr["y_log"]
        dev = tl.device
        permt = torch.as_tensor(np.asarray(perm), device=dev, dtype=torch.int64)
        nl_orig = res["n_log"].to(torch.int64)                   # counts in original order
        nl_shard = nl_orig[permt]                                # counts in shard order
        off_shard = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nl_shard, 0, out=off_shard[1:])
        off_orig = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nl_orig, 0, out=off_orig[1:])
        total = int(off_shard[-1].item())
        q = torch.repeat_interleave(torch.arange(B, device=dev), nl_shard)        # shard position of every record
        k = torch.arange(total, device=dev) - off_shard[q]                       # its index within the trajectory
        dest = off_orig[permt[q]] + k
        t2, y2 = torch.empty_like(tl), torch.empty_like(yl)
        t2[dest] = tl
        y2[dest] = yl
        res["t_log"], res["y_log"], res["log_offsets"] = t2, y2, off_orig
    return res


def solve_ivp_batch_multi(f: api.IVP, t0, t1, y0, params=None, options: api.Options = None, *,
                          devices: Optional[Sequence[int]] = None, contexts: Optional[Sequence[api.Context]] = None,
                          permutation: Optional[np.ndarray] = None, gather_device: Optional[int] = None,
                          log: bool = False) -> api.BatchSolution:
    """One process, one host thread, one context per entry of ``devices`` (``ivp_batch_solve_multi``): the batch is cut
    into ``len(devices)`` contiguous balanced shards (after ``permutation``), each integrated on its device, and the
    results are gathered with peer copies onto ``gather_device`` (default: the first device).  ``devices`` may name
    the same GPU more than once (the degenerate single-GPU case).  Gathered: the end states and statistics; with
    ``options.t_eval`` the sampled trajectories (y_eval / eval_idx / n_filled); with ``log=True`` every accepted step
    of every trajectory as a CSR log (n_log, log_offsets, t_log, y_log: two passes over the shards, count then fill)."""
    import torch
    options = options or api.Options()
    if options.max_log or options.dense_output:
        raise ValueError("solve_ivp_batch_multi gathers end states, t_eval samples and the CSR step log (log=True), not dense [max_log] buffers")
    if log and options.t_eval is not None:
        raise ValueError("the accepted-step log is what solve_ivp records when t_eval is None")
    devices = list(devices) if devices is not None else list(range(torch.cuda.device_count()))
    if not devices:
        raise RuntimeError("no HIP device")
    contexts = list(contexts) if contexts is not None else [api.Context(d) for d in devices]
    gdev = devices[0] if gather_device is None else int(gather_device)
    home = torch.device("cuda", gdev)
    y0 = torch.as_tensor(y0, dtype=torch.float64).to(home)
    n, B = int(y0.shape[0]), int(y0.shape[1])
    if n != f.n:
        raise ValueError(f"y0 must have shape [n={f.n}, B]")
    if f.n_params:
        if params is None:
            params = np.repeat(np.asarray(f.params(), dtype=np.float64)[:, None], B, axis=1)
        params = torch.as_tensor(params, dtype=torch.float64).to(home)
    else:
        params = None
    perm_t = None if permutation is None else torch.as_tensor(np.asarray(permutation), device=home, dtype=torch.int64)

    def vec(v):
        if np.ndim(v) == 0 and not api._is_torch(v):
            return float(v)
        t = torch.as_tensor(v, dtype=torch.float64).to(home)
        return float(t.reshape(-1)[0]) if t.numel() == 1 else t
    t0v, t1v = vec(t0), vec(t1)

    world = len(devices)
    keep: list = []
    base_opts = {k: v for k, v in options.__dict__.items() if k not in ("max_log", "count_log")}
    copt = (api.Options(**base_opts) if log else options)._c(n, keep)
    prob = api._problem_c(f)
    shards = (_lib.ShardT * world)()
    ptr = lambda a: None if a is None else C.c_void_p(a.data_ptr())
    tdt = {np.float64: torch.float64, np.int64: torch.int64, np.int32: torch.int32}
    eval_rows = 0 if options.t_eval is None else max(len(options.t_eval) + (1 if f.n_events() else 0), 1)
    fields = list(ARENA_FIELDS)
    if eval_rows:
        fields += [("y_eval", np.float64, (eval_rows, n)), ("eval_idx", np.int32, (eval_rows,)), ("n_filled", np.int32, 1)]
    # per-trajectory grids (every reference solve_ivp() call has its own Options.t_eval): the samples are time-major CSR
    # records; a shard's records are one contiguous run of the batch-wide arrays, which the library places in the gather
    rec_off = None
    if options.t_eval_per_trajectory is not None:
        if permutation is not None:
            raise ValueError("t_eval_per_trajectory with a permutation: permute the grids and the batch yourself")
        if len(options.t_eval_per_trajectory) != B:
            raise ValueError(f"t_eval_per_trajectory needs one grid per trajectory ({B})")
        sizes = np.array([len(g) for g in options.t_eval_per_trajectory], dtype=np.int64) + (1 if f.n_events() else 0)
        rec_off = np.zeros(B + 1, dtype=np.int64)
        rec_off[1:] = np.cumsum(sizes)
        fields += [("n_filled", np.int32, 1)]
    if log:
        fields += [("n_log", np.int32, 1)]
    shape_of = lambda rows, cols: (n, cols) if rows is None else ((cols,) if rows == 1 else tuple(rows) + (cols,))
    shard_res: list = [None] * world
    for k, d in enumerate(devices):
        lo, hi = shard_bounds(B, world, k)
        m = hi - lo
        dev = torch.device("cuda", d)
        S = shards[k]
        S.ctx = contexts[k].handle
        S.first, S.count = lo, m
        if m == 0:
            continue
        sel = slice(lo, hi) if perm_t is None else perm_t[lo:hi]
        cut = lambda a: (a[..., sel] if perm_t is None else a.index_select(a.dim() - 1, sel)).to(dev).contiguous()
        ys, ps = cut(y0), None if params is None else cut(params)
        a0 = cut(t0v) if api._is_torch(t0v) else torch.as_tensor([t0v], dtype=torch.float64, device=dev)
        a1 = cut(t1v) if api._is_torch(t1v) else torch.as_tensor([t1v], dtype=torch.float64, device=dev)
        res = {name: torch.zeros(shape_of(rows, m), dtype=tdt[dt], device=dev) for name, dt, rows in fields}
        if rec_off is not None:
            recs = max(int(rec_off[hi] - rec_off[lo]), 1)
            res["y_eval"] = torch.zeros((recs, n), dtype=torch.float64, device=dev)
            res["eval_idx"] = torch.zeros(recs, dtype=torch.int32, device=dev)
        shard_res[k] = res
        keep += [ys, ps, a0, a1, res]
        S.y0, S.params, S.t0, S.t0_len, S.t1, S.t1_len = ptr(ys), ptr(ps), ptr(a0), a0.numel(), ptr(a1), a1.numel()
        for name in res:
            setattr(S.out, name, ptr(res[name]))
        S.hip_stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    g = {name: torch.zeros(shape_of(rows, B), dtype=tdt[dt], device=home) for name, dt, rows in fields}
    if rec_off is not None:
        g["y_eval"] = torch.zeros((max(int(rec_off[-1]), 1), n), dtype=torch.float64, device=home)
        g["eval_idx"] = torch.zeros(max(int(rec_off[-1]), 1), dtype=torch.int32, device=home)
    gathered = _lib.BatchResultT()
    for name in g:
        setattr(gathered, name, ptr(g[name]))
    for d in set(devices) | {gdev}:
        torch.cuda.synchronize(d)    # inputs were produced on torch streams; the shard streams may differ
    log_info = {}
    if not log:
        rc = contexts[0].lib.ivp_batch_solve_multi(shards, world, C.byref(prob), B, C.byref(copt), gdev, C.byref(gathered))
        if rc != 0:
            raise api.ConfigError(rc, contexts[0].last_error())
    else:
        # ONE integration per shard (ivp_batch_solve_logged_multi): every shard records into its own context's page pool; the
        # call returns the batch-wide offsets and the total, the records are then fetched into buffers of exactly that size --
        # each shard's chains become one contiguous run of the batch-wide CSR log on the gather device
        g["log_offsets"] = torch.zeros(B + 1, dtype=torch.int64, device=home)
        sl = _lib.StepLogT()
        sl.offsets = ptr(g["log_offsets"])
        sl.defer = 1
        rc = contexts[0].lib.ivp_batch_solve_logged_multi(shards, world, C.byref(prob), B, C.byref(copt), gdev, C.byref(gathered), C.byref(sl))
        if rc != 0:
            raise api.ConfigError(rc, contexts[0].last_error())
        total_all = int(sl.total)
        g["t_log"] = torch.empty(max(total_all, 1), dtype=torch.float64, device=home)
        g["y_log"] = torch.empty((max(total_all, 1), n), dtype=torch.float64, device=home)
        sl.t, sl.y, sl.capacity, sl.defer = ptr(g["t_log"]), ptr(g["y_log"]), max(total_all, 1), 0
        torch.cuda.synchronize(gdev)
        rc = contexts[0].lib.ivp_step_log_fetch_multi(shards, world, C.byref(prob), B, C.byref(copt), gdev, C.byref(sl))
        if rc != 0:
            raise api.ConfigError(rc, contexts[0].last_error())
        g["t_log"], g["y_log"] = g["t_log"][:total_all], g["y_log"][:total_all]
        log_info = {"passes": int(sl.passes), "records": total_all, "page_slots": int(sl.page_slots), "pool_bytes": int(sl.pool_bytes),
                    "pool_used_bytes": int(sl.pool_used_bytes)}
    if perm_t is not None:
        g = _unpermute_solution(g, permutation, B)
    return api.BatchSolution(y_end=g["y_end"], t_end=g["t_end"], status=g["status"], nfev=g["nfev"], nstep=g["nstep"],
                             naccpt=g["naccpt"], nrejct=g["nrejct"], h_next=g["h_next"], y_eval=g.get("y_eval"),
                             eval_idx=g.get("eval_idx"), n_filled=g.get("n_filled"), n_log=g.get("n_log"),
                             log_offsets=g.get("log_offsets"), t_log=g.get("t_log"), y_log=g.get("y_log"),
                             eval_offsets=None if rec_off is None else torch.as_tensor(rec_off, device=home), log_info=log_info)


class OverlappedGather:
    """Double-buffered, asynchronous all-gather of one byte arena per step, used by bench.py --gpus N.

    Step i's gather (RCCL over xGMI under backend "nccl": the C4 "gather of sol.y") runs while step i+1 integrates
    into the other arena; ``slot()`` returns which of the two buffers the next step may write (after making sure the
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


def run_steps(solve_into: Callable, n: int, m: int, device, steps: int, warmup: int, *, gather: bool, d2h: bool = False,
              on_step: Optional[Callable] = None):
    """The step loop of bench.py (kept here so that the CPU suite can run the very same code under gloo).

    ``solve_into(sol)`` integrates this rank's shard of ``m`` trajectories and leaves the end states in ``sol`` (a
    BatchSolution whose members are views of a ResultArena); it returns the solution object to report (normally
    ``sol``).  One step = one complete solve (+ the all-gather of the shard's byte arena when ``gather``: step i's
    gather overlaps step i+1's integration through double buffering, and every gather is waited for inside the timed
    region).  ``d2h``: every solve is followed by a copy of the arena into pinned host memory.
    Returns (elapsed seconds on this rank, last solution, OverlappedGather or None, the two arenas)."""
    import time
    import torch
    import torch.distributed as dist
    is_cuda = torch.device(device).type == "cuda"
    arenas = [ResultArena(n, m, device) for _ in range(2)]
    sols = [a.solution() for a in arenas]
    og = OverlappedGather((arenas[0].buf.numel(),), torch.uint8, device) if gather else None
    host = None
    if d2h:
        host = torch.empty(arenas[0].buf.numel(), dtype=torch.uint8)
        host = host.pin_memory() if is_cuda else host
    cnt = [0]

    def step():
        if og is not None:
            k = og.slot()
        else:
            k = cnt[0] & 1
            cnt[0] += 1
        out = solve_into(sols[k])
        if og is not None:
            og.launch(k, arenas[k].buf)
        if host is not None:
            host.copy_(arenas[k].buf, non_blocking=True)
            if is_cuda:
                torch.cuda.current_stream().synchronize()     # the caller owns the results on the host now
        return out

    def drain():
        if og is not None:
            og.drain()
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        if is_cuda:
            torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    drain()
    t_begin = time.perf_counter()
    for _ in range(steps):
        out = step()
        if on_step is not None:
            on_step(out)
    drain()
    return time.perf_counter() - t_begin, out, og, arenas
