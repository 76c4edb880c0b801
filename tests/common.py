"""Shared helpers for the parity tests: three interchangeable back ends with one result shape.

  oracle  -- oracle/ (CPU restatement of the reference; the checker)
  emul    -- tests/host_emul (the kernel bodies of rk_core.h run lane by lane on the CPU; test-only)
  gpu     -- libivp_hip.so through the C ABI (the product)
"""
import numpy as np

from oracle import oracle as O

KEYS_INT = ("status", "nfev", "nstep", "naccpt", "nrejct", "njev", "nlu")


def oracle_batch(rhs, y0, params, t0, t1, detpow=True, **opts):
    opts = dict(opts)
    opts.pop("max_log", None)
    opts.pop("max_events", None)
    opts.pop("chunk", None)
    opts.pop("variant", None)
    return O.solve_batch(rhs, y0, params, t0, t1, detpow=detpow, **opts)


def emul_batch(rhs, y0, params, t0, t1, **opts):
    from tests.host_emul import emul as E
    return E.solve_batch(rhs, y0, params, t0, t1, **opts)


def gpu_batch(rhs, y0, params, t0, t1, *, fast=False, chunk=0, device_arrays=False, event_direction=None,
              event_terminal=None, **opts):
    import ivp_amd
    if event_direction is not None or event_terminal is not None:
        ne = len(event_direction or event_terminal)
        cfgs = [ivp_amd.EventConfig(ivp_amd.Direction(int(np.sign((event_direction or [0] * ne)[i]))),
                                    (event_terminal or [0] * ne)[i] or None) for i in range(ne)]
        f = ivp_amd.BUILTIN[rhs](*cfgs) if rhs != "ball" else ivp_amd.BUILTIN[rhs](9.81, 0.02, *cfgs)
    else:
        f = ivp_amd.BUILTIN[rhs]()
    o = ivp_amd.Options(fp_mode=ivp_amd.FpMode.FAST if fast else ivp_amd.FpMode.STRICT, chunk_attempts=chunk, **opts)
    if device_arrays:
        import torch
        dev = torch.device("cuda:0")
        y0 = torch.as_tensor(np.ascontiguousarray(y0), device=dev)
        params = None if params is None else torch.as_tensor(np.ascontiguousarray(params), device=dev)
    r = ivp_amd.solve_ivp_batch(f, t0, t1, y0, params if f.n_params else None, o)
    out = {}
    for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct", "y_eval", "eval_idx",
              "n_filled", "t_log", "y_log", "n_log", "seg_cont", "seg_xold", "seg_h", "n_seg", "njev", "nlu",
              "t_events", "y_events", "n_event_hits", "t_term"):
        v = getattr(r, k)
        if v is None:
            continue
        if device_arrays:
            v = v.cpu().numpy()
        out[k] = v
    if "n_event_hits" in out:
        out["n_ev"] = out["n_event_hits"]
    out["stats"] = r.stats
    return out


def assert_bitexact(got, ref, what=""):
    """Bit-for-bit equality of end state, end time, next step and every counter."""
    for k in ("y_end", "t_end", "h_next"):
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"{what}{k}: {np.count_nonzero(~same)} of {same.size} values differ; max |d| = {np.nanmax(np.abs(a - b))}"
    for k in KEYS_INT:
        if k not in got or k not in ref:
            continue
        a, b = np.asarray(got[k]).astype(np.int64), np.asarray(ref[k]).astype(np.int64)
        assert np.array_equal(a, b), f"{what}{k}: {np.count_nonzero(a != b)} of {a.size} differ"
