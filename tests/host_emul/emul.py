"""TEST-ONLY driver of tests/host_emul/libemul_{strict,fast}.so (see emul.cpp): runs the per-lane
kernel bodies of ivp_amd/csrc/rk_core.h on the CPU with the GPU launch loop's chunk schedule."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
VP = C.c_void_p


class KArgs(C.Structure):  # must match ivp_amd/csrc/ivp_kargs.h
    _fields_ = [
        ("B", C.c_uint32),
        ("y0", VP), ("params", VP), ("t0", VP), ("t1", VP),
        ("t0_stride", C.c_uint32), ("t1_stride", C.c_uint32),
        ("rtol", C.c_double * 8), ("atol", C.c_double * 8), ("rtol_dev", VP), ("atol_dev", VP),
        ("first_step", C.c_double), ("max_step", C.c_double),
        ("nmax", C.c_uint64),
        ("has_first_step", C.c_int32), ("has_max_step", C.c_int32),
        ("ctl_uround", C.c_double), ("ctl_safety", C.c_double), ("ctl_facc1", C.c_double), ("ctl_facc2", C.c_double),
        ("ctl_beta", C.c_double), ("ctl_expo1", C.c_double), ("ctl_scale_min", C.c_double), ("ctl_scale_max", C.c_double),
        ("ctl_nstiff", C.c_uint64), ("has_ctl", C.c_int32),
        ("y", VP), ("k1", VP), ("x", VP), ("h", VP), ("facold", VP), ("hlamb", VP),
        ("flags", VP), ("status", VP), ("nfev", VP), ("nstep", VP), ("naccpt", VP), ("nrejct", VP),
        ("perm_in", VP), ("count_in", VP), ("perm_out", VP), ("count_out", VP),
        ("chunk", C.c_uint32),
        ("t_eval", VP), ("n_eval", C.c_int32),
        ("y_eval", VP), ("eval_idx", VP), ("n_filled", VP), ("next_idx", VP), ("teval_off", VP), ("teval_extra", C.c_uint32),
        ("max_log", C.c_uint32),
        ("t_log", VP), ("y_log", VP), ("n_log", VP), ("t_last", VP), ("log_off", VP),
        ("collect_dense", C.c_int32),
        ("seg_cont", VP), ("seg_xold", VP), ("seg_h", VP), ("n_seg", VP),
        ("ev_direction", C.c_int32 * 4), ("ev_terminal", C.c_uint32 * 4), ("ev_direction_dev", VP), ("ev_terminal_dev", VP), ("max_events", C.c_uint32),
        ("t_events", VP), ("y_events", VP), ("n_ev", VP), ("prev_event", VP), ("t_term", VP),
        ("min_step", C.c_double), ("has_min_step", C.c_int32),
        ("bdf_d", VP), ("bdf_jac", VP), ("bdf_lu", VP), ("bdf_piv", VP), ("njev", VP), ("nlu", VP),
        ("err_flag", VP),
        ("slot_counter", VP),
        ("spec_cap", C.c_uint32), ("spec_min", C.c_uint32), ("ran_out", VP), ("lds_lu", C.c_uint32), ("lpw", C.c_uint32),
        ("count_next", VP),
        ("window", C.c_uint32),
        ("def_rec", VP), ("def_cap", C.c_uint32),
        ("log_pool", VP), ("log_region", C.c_uint64), ("log_sub_mask", C.c_uint32), ("log_alloc", VP),
        ("evd_rec", VP), ("evd_cnt", VP), ("evd_cap", C.c_uint32),
    ]


LOG_SLOTS = 32
LOG_SUBPOOLS = 64
LOG_ALLOC_STRIDE = 16

_libs = {}
RHS = {"decay": 0, "sho": 1, "vdp": 2, "cr3bp": 3, "lorenz": 4, "zero": 5, "rational": 6, "exp2": 7,
       "linear": 8, "robertson": 9, "vdp_eps": 10, "sho_ev": 11, "ball": 12, "cannon": 13, "rational_ev": 14, "robertson_jac": 15}
RHS_DIMS = {0: (1, 1), 1: (2, 0), 2: (2, 1), 3: (6, 1), 4: (3, 3), 5: (3, 0), 6: (2, 0), 7: (2, 0),
            8: (2, 0), 9: (3, 0), 10: (2, 1), 11: (2, 0), 12: (2, 2), 13: (2, 0), 14: (2, 0)}
RHS_NE = {11: 1, 12: 1, 13: 1, 14: 3}
METHODS = {"RK23": 0, "DOPRI5": 1, "RK45": 1, "DOP853": 2, "RK4": 3, "BDF": 5}
NCOEF = {0: 4, 1: 5, 2: 8, 3: 4, 5: 7}


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib(fast=False):
    key = bool(fast)
    if key not in _libs:
        build()
        L = C.CDLL(os.path.join(_HERE, "libemul_fast.so" if fast else "libemul_strict.so"))
        L.emul_solve.restype = C.c_int
        L.emul_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(KArgs), C.POINTER(C.c_uint64)]
        L.emul_kargs_size.restype = C.c_size_t
        assert L.emul_kargs_size() == C.sizeof(KArgs), (L.emul_kargs_size(), C.sizeof(KArgs))
        assert L.emul_is_fast() == int(fast)
        _libs[key] = L
    return _libs[key]


def solve_batch(rhs, y0, params, t0, t1, *, method="DOPRI5", rtol=1e-3, atol=1e-6, max_steps=None, t_eval=None,
                first_step=None, max_step=None, min_step=None, dense_output=False, max_log=0, chunk=64, fast=False,
                event_direction=None, event_terminal=None, max_events=16, settings=None, paged_log=None, flavour_log_only=True, defer_eval=True, defer_events=True):
    """``paged_log=pool_doubles``: the one-pass step log -- records go to wave pages in a pool of that many doubles, chained
    per trajectory (ivp_kargs.h; on the host a "wave" is one lane, so every page has one column); ``res['log_pool']``,
    ``res['log_cur']``, ``res['log_used']`` (doubles) and ``res['log_overflow']`` come back next to ``n_log``
    (``gather_pages`` below lays them out as the CSR log)."""
    L = lib(fast)
    rid = RHS[rhs]
    n, npar = RHS_DIMS[rid]
    m = METHODS[method.upper()]
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    B = y0.shape[1]
    params = np.ascontiguousarray(params, dtype=np.float64) if npar else np.zeros((1, B))
    t0 = np.atleast_1d(np.asarray(t0, dtype=np.float64)).copy()
    t1 = np.atleast_1d(np.asarray(t1, dtype=np.float64)).copy()
    a = KArgs()
    a.B = B
    p = lambda arr: arr.ctypes.data_as(VP)
    a.y0, a.params, a.t0, a.t1 = p(y0), p(params), p(t0), p(t1)
    a.t0_stride, a.t1_stride = int(t0.size > 1), int(t1.size > 1)
    rt = np.broadcast_to(np.asarray(rtol, dtype=np.float64), (n,))
    at = np.broadcast_to(np.asarray(atol, dtype=np.float64), (n,))
    for i in range(8):
        a.rtol[i] = rt[i] if i < n else rt[-1]
        a.atol[i] = at[i] if i < n else at[-1]
    a.first_step = float(first_step or 0.0)
    a.has_first_step = int(first_step is not None)
    a.max_step = float(max_step or 0.0)
    a.has_max_step = int(max_step is not None)
    a.nmax = int(max_steps) if max_steps else 2 ** 64 - 1
    a.min_step = float(min_step or 0.0)
    a.has_min_step = int(min_step is not None)
    # controller settings, derived like ivp_capi.cpp does (method struct defaults unless overridden)
    from oracle.oracle import SETTINGS_DEFAULTS
    st = {**SETTINGS_DEFAULTS.get(m, SETTINGS_DEFAULTS[0]), **(settings or {})}
    a.ctl_uround, a.ctl_safety, a.ctl_beta = st["uround"], st["safety_factor"], st["beta"]
    a.ctl_facc1, a.ctl_facc2 = 1.0 / st["scale_min"], 1.0 / st["scale_max"]
    a.ctl_expo1 = (1.0 / 8.0 - st["beta"] * 0.2) if m == 2 else (0.2 - st["beta"] * 0.75)
    a.ctl_scale_min, a.ctl_scale_max, a.ctl_nstiff = st["scale_min"], st["scale_max"], int(st["stiff_test"])
    a.has_ctl = int(settings is not None)
    if settings is not None and max_steps is None:
        a.nmax = 10_000 if m == 0 else 100_000
    res = {
        "y_end": np.zeros((n, B)), "t_end": np.zeros(B), "h_next": np.zeros(B), "status": np.zeros(B, dtype=np.int32),
        "nfev": np.zeros(B, dtype=np.uint64), "nstep": np.zeros(B, dtype=np.uint64),
        "naccpt": np.zeros(B, dtype=np.uint64), "nrejct": np.zeros(B, dtype=np.uint64),
        "njev": np.zeros(B, dtype=np.uint64), "nlu": np.zeros(B, dtype=np.uint64),
    }
    a.y, a.x, a.h, a.status = p(res["y_end"]), p(res["t_end"]), p(res["h_next"]), p(res["status"])
    a.nfev, a.nstep, a.naccpt, a.nrejct = p(res["nfev"]), p(res["nstep"]), p(res["naccpt"]), p(res["nrejct"])
    a.njev, a.nlu = p(res["njev"]), p(res["nlu"])
    a.chunk = chunk
    a.n_eval = -1
    ne_ev = RHS_NE.get(rid, 0)
    full = t_eval is not None or max_log > 0 or ne_ev > 0 or paged_log is not None
    keep = []
    if full:
        res["n_filled"] = np.zeros(B, dtype=np.int32)
        res["n_log"] = np.zeros(B, dtype=np.uint32)
        res["n_seg"] = np.zeros(B, dtype=np.uint32)
        a.n_filled, a.n_log, a.n_seg = p(res["n_filled"]), p(res["n_log"]), p(res["n_seg"])
        if t_eval is not None:
            te = np.ascontiguousarray(t_eval, dtype=np.float64)
            keep.append(te)
            ne = te.size
            a.n_eval = ne
            a.t_eval = p(te) if ne else None
            res["y_eval"] = np.full((max(ne, 1), n, B), np.nan)
            res["eval_idx"] = np.full((max(ne, 1), B), -1, dtype=np.int32)
            a.y_eval, a.eval_idx = p(res["y_eval"]), p(res["eval_idx"])
        elif paged_log is not None:
            region = int(paged_log) // LOG_SUBPOOLS
            res["log_region"] = region
            res["log_pool"] = np.full(region * LOG_SUBPOOLS + 256, np.nan)   # 256 guard doubles the bodies are not told about
            res["log_alloc"] = np.zeros(LOG_SUBPOOLS * LOG_ALLOC_STRIDE, dtype=np.uint64)
            a.log_pool, a.log_region, a.log_sub_mask, a.log_alloc = p(res["log_pool"]), region, LOG_SUBPOOLS - 1, p(res["log_alloc"])
            a.t_log = a.y_log = p(res["log_pool"])   # the "mode 2" marker, like the library sets it
        elif max_log > 0:
            res["t_log"] = np.full((max_log, B), np.nan)
            res["y_log"] = np.full((max_log, n, B), np.nan)
            a.t_log, a.y_log = p(res["t_log"]), p(res["y_log"])
        a.max_log = max_log
        if dense_output and max_log > 0:
            nc = NCOEF[m] * n
            res["seg_cont"] = np.full((max_log, nc, B), np.nan)
            res["seg_xold"] = np.full((max_log, B), np.nan)
            res["seg_h"] = np.full((max_log, B), np.nan)
            a.seg_cont, a.seg_xold, a.seg_h = p(res["seg_cont"]), p(res["seg_xold"]), p(res["seg_h"])
            a.collect_dense = 1
    if ne_ev:
        for i, d in enumerate((event_direction or [])[:4]):
            a.ev_direction[i] = int(d)
        for i, t in enumerate((event_terminal or [])[:4]):
            a.ev_terminal[i] = int(t or 0)
        a.max_events = max_events
        res["t_events"] = np.full((ne_ev, max_events, B), np.nan)
        res["y_events"] = np.full((ne_ev, max_events, n, B), np.nan)
        res["n_ev"] = np.zeros((ne_ev, B), dtype=np.uint32)
        res["t_term"] = np.full(B, np.nan)
        a.t_events, a.y_events, a.n_ev, a.t_term = p(res["t_events"]), p(res["y_events"]), p(res["n_ev"]), p(res["t_term"])
        # the library's rule (ivp_capi.cpp): no terminal event, explicit method -> the roots are found after the stepping
        if defer_events and method != "BDF" and not any(int(t or 0) for t in (event_terminal or [])):
            fields = 4 + 3 * ne_ev + n + NCOEF[m] * n
            cap = max(ne_ev * max_events, 1)
            res["evd_rec"] = np.full((cap, fields, B), np.nan)
            res["evd_cnt"] = np.zeros(B, dtype=np.uint32)
            a.evd_rec, a.evd_cnt, a.evd_cap = p(res["evd_rec"]), p(res["evd_cnt"]), cap
    chunks = C.c_uint64(0)
    # the library's choice of kernel flavour (ivp_capi.cpp): log-only when every accepted step is recorded and nothing else is
    # asked of the device DefaultSolOut
    log_only = full and t_eval is None and not dense_output and ne_ev == 0 and first_step is None and (max_log > 0 or paged_log is not None)
    flavour = 2 if (log_only and flavour_log_only) else int(full)
    # ... deferred t_eval sampling (flavour 3): DOP853 with t_eval and nothing else asked of the device DefaultSolOut
    if m == 2 and t_eval is not None and len(np.atleast_1d(t_eval)) > 0 and not dense_output and ne_ev == 0 and defer_eval:
        flavour = 3
        cap = max(len(np.atleast_1d(t_eval)), 1)
        res["def_rec"] = np.full((cap, n + 4, B), np.nan)
        a.def_rec, a.def_cap = p(res["def_rec"]), cap
    rc = L.emul_solve(m, rid, flavour, C.byref(a), C.byref(chunks))
    if rc == -5:
        raise ValueError("IVP_ERR_INVALID_STEP_SIZE")
    assert rc == 0
    res["chunks"] = chunks.value
    if paged_log is not None:
        cnt = res["log_alloc"][::LOG_ALLOC_STRIDE]
        used = (cnt & np.uint64((1 << 40) - 1)).astype(np.int64) + (cnt >> np.uint64(40)).astype(np.int64)
        res["log_used"] = int(used.sum())
        res["log_overflow"] = bool((used > res["log_region"]).any())
    return res


LOG_SLOTS = 32
LOG_SUBPOOLS = 64
LOG_ALLOC_STRIDE = 16


def gather_pages(res, n):
    """What log_gather.hip does, in numpy: enumerate every sub-pool's directory, read each page's column headers
    (trajectory, first record index, slot bits) and lay the records out as the CSR log.
    Returns (offsets [B+1], t [total], y [total, n])."""
    cnt = res["n_log"].astype(np.int64)
    off = np.zeros(cnt.size + 1, dtype=np.int64)
    off[1:] = np.cumsum(cnt)
    pool, region = res["log_pool"], res["log_region"]
    words = pool.view(np.uint64)
    np1 = n + 1
    W = 8 if np1 <= 9 else 1
    hdr = lambda cols: (1 + 2 * cols + 15) & ~15   # IVP_LOG_HDR
    page_doubles = lambda cols, slots: hdr(cols) + ((((cols + W - 1) // W) * slots * W * np1 + 15) & ~15)
    t = np.full(int(off[-1]), np.nan)
    y = np.full((int(off[-1]), n), np.nan)
    filled = np.zeros(int(off[-1]), dtype=bool)
    for sub in range(LOG_SUBPOOLS):
        arenas = int(res["log_alloc"][sub * LOG_ALLOC_STRIDE]) >> 40
        for e in range(arenas):
            entry = int(words[(sub + 1) * region - 1 - e])
            base0, acols, k = entry >> 8, ((entry >> 2) & 0x3F) + 1, (entry & 3) + 1
            for p in range(k):
                page = base0 + p * page_doubles(acols, LOG_SLOTS)
                cols, slots = (int(v) for v in pool[page:page + 1].view(np.uint32))
                if cols == 0:
                    continue
                body = page + hdr(cols)
                # whole 128-byte lines: pages from the region's start, bodies within their page (IVP_LOG_HDR): a slot row of a group
                # then fills whole 64-byte sectors
                assert (page - sub * region) % 16 == 0 and (body - page) % 16 == 0 and (W * np1 * 8) % 64 == 0 or W == 1
                for col in range(cols):
                    j, k0, bits, _ = (int(v) for v in pool[page + 1 + 2 * col:page + 3 + 2 * col].view(np.uint32))
                    r = 0
                    for sl in range(LOG_SLOTS):
                        if (bits >> sl) & 1:
                            at = body + (((col // W) * slots + sl) * W + col % W) * np1
                            rec = pool[at:at + np1]
                            q = int(off[j]) + k0 + r
                            assert sl < slots and q < off[j + 1] and not filled[q]
                            t[q], y[q], filled[q] = rec[0], rec[1:], True
                            r += 1
    assert filled.all(), "every record of every trajectory must be in exactly one slot of one page"
    return off, t, y
