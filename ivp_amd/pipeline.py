"""Throughput mode: several independent batches in flight on one GPU.

A single batch is latency-bound by its slowest trajectory (at C2 the last 0.5 % of the
trajectories keep a handful of wavefronts busy for half of the wall time while the other 1000 SIMDs idle).
Independent batches do not have to wait for each other: with one ``Context`` + HIP stream + host thread per
in-flight batch, the tail of batch i overlaps the throughput-bound head of batch i+1.  Measured on MI355X (C2,
strict FP): 3.7 ms per solve with one batch in flight, 1.4 ms with four.

A solve is a sequence of rounds (a few kernel launches, then the count of still-running trajectories travels to the
host); only the hand-over between rounds needs the host.  ``BatchPipeline.map`` therefore drives all streams from ONE
host thread through the resumable entry points (``ivp_batch_submit_device`` / ``ivp_batch_poll``): whenever a
context's round has finished it enqueues the next one, and a finished solve hands its context to the next batch.
``map_threads`` is the older one-blocking-thread-per-stream form (ctypes releases the GIL during the call).
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence

from . import api


class BatchPipeline:
    """``streams`` contexts/streams on one device; ``map`` integrates a sequence of batches through them."""

    def __init__(self, streams: int = 4, device: int = 0):
        import torch
        self.device = torch.device("cuda", device)
        self.ctxs = [api.Context(device) for _ in range(streams)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(streams)]

    def map(self, f: api.IVP, batches: Sequence[dict], options: api.Options,
            on_done: Optional[Callable[[int, api.BatchSolution], None]] = None,
            out_per_context: Optional[Sequence[api.BatchSolution]] = None) -> List[api.BatchSolution]:
        """``batches``: dicts with keys t0, t1, y0, params (CUDA tensors / scalars as for solve_ivp_batch).  Results
        come back in input order.  ``out_per_context``: one reusable BatchSolution per stream (results of different
        batches then alias: consume them in ``on_done``)."""
        import time
        import torch
        results: List[Optional[api.BatchSolution]] = [None] * len(batches)
        nxt = 0
        inflight: List[Optional[tuple]] = [None] * len(self.ctxs)   # (batch index, PendingBatch) per context
        remaining = len(batches)
        # the inputs may have been produced by work still queued on the caller's stream: every side stream starts
        # behind it, and the caller's stream is made to wait for the side streams at the end (results are consumed there)
        caller = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(caller)
        try:
            while remaining:
                progressed = False
                for w in range(len(self.ctxs)):
                    if inflight[w] is not None:
                        k, pend = inflight[w]
                        if pend.done():
                            results[k] = pend.result()
                            inflight[w] = None
                            remaining -= 1
                            progressed = True
                            if on_done is not None:
                                on_done(k, results[k])
                    if inflight[w] is None and nxt < len(batches):
                        b = batches[nxt]
                        with torch.cuda.stream(self.streams[w]):
                            out = out_per_context[w] if out_per_context is not None else b.get("out")
                            inflight[w] = (nxt, api.solve_ivp_batch(f, b["t0"], b["t1"], b["y0"], b.get("params"), options,
                                                                    self.ctxs[w], out, wait=False))
                        nxt += 1
                        progressed = True
                # nothing ready: poll again (a round lasts 0.3-2 ms; each poll is one hipEventQuery per context),
                # yielding the core between sweeps instead of spinning flat out
                if not progressed:
                    time.sleep(0)
        finally:
            for slot in inflight:   # an exception must not leave solves in flight on the pipeline's contexts
                if slot is not None:
                    try:
                        slot[1].result()
                    except Exception:
                        pass
        for st in self.streams:   # results were allocated and written on the side streams: order the caller's stream after them
            caller.wait_stream(st)
        for r in results:
            if r is not None:
                for t in (r.y_end, r.t_end, r.status, r.nfev, r.nstep, r.naccpt, r.nrejct, r.h_next):
                    if t is not None and hasattr(t, "record_stream"):
                        t.record_stream(caller)
        torch.cuda.synchronize(self.device)
        return results  # type: ignore[return-value]

    def map_threads(self, f: api.IVP, batches: Sequence[dict], options: api.Options,
                    on_done: Optional[Callable[[int, api.BatchSolution], None]] = None) -> List[api.BatchSolution]:
        """One blocking host thread per stream; batch k is integrated by worker k % streams."""
        import torch
        results: List[Optional[api.BatchSolution]] = [None] * len(batches)
        errors: List[BaseException] = []

        def work(w: int):
            try:
                with torch.cuda.stream(self.streams[w]):
                    for k in range(w, len(batches), len(self.streams)):
                        b = batches[k]
                        results[k] = api.solve_ivp_batch(f, b["t0"], b["t1"], b["y0"], b.get("params"), options,
                                                         self.ctxs[w], b.get("out"))
                        if on_done is not None:
                            on_done(k, results[k])
            except BaseException as e:  # surfaced to the caller below
                errors.append(e)

        threads = [threading.Thread(target=work, args=(w,)) for w in range(len(self.streams))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize(self.device)
        if errors:
            raise errors[0]
        return results  # type: ignore[return-value]
