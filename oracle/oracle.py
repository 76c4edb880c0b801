"""ctypes front end of the CPU ORACLE (test infrastructure, NOT product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  See ``ivp_oracle.h`` for what the oracle restates and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

METHODS = {"RK23": 0, "DOPRI5": 1, "RK45": 1, "DOP853": 2, "RK4": 3, "BDF": 5}
RHS = {"decay": 0, "sho": 1, "vdp": 2, "cr3bp": 3, "lorenz": 4, "zero": 5, "rational": 6, "exp2": 7,
       "linear": 8, "robertson": 9, "vdp_eps": 10, "sho_ev": 11, "ball": 12, "cannon": 13, "rational_ev": 14,
       "robertson_jac": 15, "linear_decay100": 100, "heat1d256": 101, "dense64": 102}
RHS_DIMS = {0: (1, 1), 1: (2, 0), 2: (2, 1), 3: (6, 1), 4: (3, 3), 5: (3, 0), 6: (2, 0), 7: (2, 0),
            8: (2, 0), 9: (3, 0), 10: (2, 1), 11: (2, 0), 12: (2, 2), 13: (2, 0), 14: (2, 0), 15: (3, 0), 100: (100, 0), 101: (256, 1), 102: (64, 1)}
STATUS = ["Success", "UserInterrupt", "NeedLargerNMax", "StepSizeTooSmall", "ProbablyStiff",
          "SingularMatrix", "PoorConvergence"]


class _Options(C.Structure):
    _fields_ = [
        ("method", C.c_int),
        ("rtol", C.POINTER(C.c_double)), ("rtol_len", C.c_int),
        ("atol", C.POINTER(C.c_double)), ("atol_len", C.c_int),
        ("has_max_steps", C.c_int), ("max_steps", C.c_uint64),
        ("t_eval", C.POINTER(C.c_double)), ("n_eval", C.c_int),
        ("has_first_step", C.c_int), ("first_step", C.c_double),
        ("has_max_step", C.c_int), ("max_step", C.c_double),
        ("dense_output", C.c_int),
        ("has_min_step", C.c_int), ("min_step", C.c_double),
        ("events", C.c_void_p), ("n_events", C.c_int), ("ev_direction", C.c_int * 16), ("ev_terminal", C.c_uint64 * 16),
        ("attempt_guard", C.c_uint64),
        ("has_settings", C.c_int), ("uround", C.c_double), ("safety_factor", C.c_double), ("scale_min", C.c_double),
        ("scale_max", C.c_double), ("beta", C.c_double), ("stiff_test", C.c_uint64),
        ("jac", C.c_void_p),
    ]


# per-method struct defaults (dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37); keys a direct method call may override
SETTINGS_DEFAULTS = {
    1: dict(uround=2.3e-16, safety_factor=0.9, scale_min=0.2, scale_max=10.0, beta=0.04, stiff_test=1000),
    2: dict(uround=2.3e-16, safety_factor=0.9, scale_min=0.333, scale_max=6.0, beta=0.0, stiff_test=1000),
    0: dict(uround=2.3e-16, safety_factor=0.9, scale_min=0.2, scale_max=10.0, beta=0.0, stiff_test=1000),
}


class _Solution(C.Structure):
    _fields_ = [
        ("len", C.c_size_t), ("t", C.POINTER(C.c_double)), ("y", C.POINTER(C.c_double)),
        ("nfev", C.c_uint64), ("njev", C.c_uint64), ("nlu", C.c_uint64),
        ("nstep", C.c_uint64), ("naccpt", C.c_uint64), ("nrejct", C.c_uint64),
        ("status", C.c_int), ("h_next", C.c_double),
        ("has_dense", C.c_int), ("ncoef", C.c_int), ("n", C.c_int),
        ("nseg", C.c_size_t), ("seg_cont", C.POINTER(C.c_double)),
        ("seg_xold", C.POINTER(C.c_double)), ("seg_h", C.POINTER(C.c_double)),
        ("n_events", C.c_int), ("ev_len", C.c_size_t * 16),
        ("t_events", C.POINTER(C.c_double) * 16), ("y_events", C.POINTER(C.c_double) * 16),
    ]


_ODE_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

_libs = {}


_LIBS = ("liboracle.so", "liboracle_detpow.so", "liboracle_fma.so")


def build(force: bool = False) -> None:
    """Compile the oracle's three shared objects (gcc only)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in _LIBS)
    if not need:
        src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ivp_oracle.c", "ivp_oracle.h", "Makefile"))
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src for f in _LIBS)
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def lib(detpow: bool = False, fma: bool = False):
    """liboracle.so (libm pow: the faithful restatement), liboracle_detpow.so (portable pow: bit-comparable with the
    kernels' strict mode) or liboracle_fma.so (portable pow + the fused multiply-add sites of the kernels' FMA mode)."""
    detpow = bool(detpow or fma)
    key = (detpow, bool(fma))
    if key not in _libs:
        path = os.path.join(_HERE, "liboracle_fma.so" if fma else ("liboracle_detpow.so" if detpow else "liboracle.so"))
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_builtin_rhs.restype = C.c_void_p
        L.orc_builtin_rhs.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_builtin_events.restype = C.c_void_p
        L.orc_builtin_events.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.orc_builtin_jac.restype = C.c_void_p
        L.orc_builtin_jac.argtypes = [C.c_int]
        L.orc_solve_ivp.restype = C.c_int
        L.orc_solve_ivp.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double,
                                    C.POINTER(C.c_double), C.POINTER(_Options), C.POINTER(_Solution)]
        L.orc_solution_free.argtypes = [C.POINTER(_Solution)]
        L.orc_solution_eval.restype = C.c_int
        L.orc_solution_eval.argtypes = [C.POINTER(_Solution), C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.orc_solution_eval_extrapolate.restype = C.c_int
        L.orc_solution_eval_extrapolate.argtypes = L.orc_solution_eval.argtypes
        L.orc_detpow.restype = C.c_double
        L.orc_detpow.argtypes = [C.c_double, C.c_double]
        L.orc_uses_detpow.restype = C.c_int
        L.orc_batch_solve.restype = C.c_int64
        L.orc_batch_solve.argtypes = [
            C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
            C.POINTER(_Options), C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_uses_fma.restype = C.c_int
        assert L.orc_uses_detpow() == int(detpow) and L.orc_uses_fma() == int(bool(fma))
        _libs[key] = L
    return _libs[key]


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _OptHolder:
    """Builds the C options struct and keeps the numpy buffers it points at alive."""

    def __init__(self, method="DOPRI5", rtol=1e-3, atol=1e-6, max_steps=None, t_eval=None,
                 first_step=None, max_step=None, min_step=None, dense_output=False, attempt_guard=0,
                 event_direction=None, event_terminal=None, settings=None):
        m = METHODS[method.upper()] if isinstance(method, str) else int(method)
        self.rtol = np.atleast_1d(np.asarray(rtol, dtype=np.float64)).copy()
        self.atol = np.atleast_1d(np.asarray(atol, dtype=np.float64)).copy()
        self.t_eval = None if t_eval is None else np.ascontiguousarray(t_eval, dtype=np.float64)
        o = _Options()
        o.method = m
        o.rtol, o.rtol_len = _dptr(self.rtol), self.rtol.size
        o.atol, o.atol_len = _dptr(self.atol), self.atol.size
        o.has_max_steps = int(max_steps is not None)
        o.max_steps = int(max_steps or 0)
        if self.t_eval is None:
            o.t_eval, o.n_eval = None, -1
        else:
            o.t_eval, o.n_eval = _dptr(self.t_eval), self.t_eval.size
        o.has_first_step = int(first_step is not None)
        o.first_step = float(first_step or 0.0)
        o.has_max_step = int(max_step is not None)
        o.max_step = float(max_step or 0.0)
        o.dense_output = int(bool(dense_output))
        o.has_min_step = int(min_step is not None)
        o.min_step = float(min_step or 0.0)
        o.attempt_guard = int(attempt_guard)
        if settings is not None:   # a direct `DOPRI5 {..}.solve()`-style call with struct fields overridden
            if m not in SETTINGS_DEFAULTS:
                raise ValueError("settings apply to RK23 / DOPRI5 / DOP853")
            unknown = set(settings) - set(SETTINGS_DEFAULTS[m])
            if unknown:
                raise ValueError(f"unknown settings {sorted(unknown)}")
            full = {**SETTINGS_DEFAULTS[m], **settings}
            o.has_settings = 1
            o.uround, o.safety_factor = float(full["uround"]), float(full["safety_factor"])
            o.scale_min, o.scale_max, o.beta = float(full["scale_min"]), float(full["scale_max"]), float(full["beta"])
            o.stiff_test = int(full["stiff_test"])
            if max_steps is None:   # a direct method call has no `None`: the struct default applies
                o.has_max_steps, o.max_steps = 1, (10_000 if m == 0 else 100_000)
        self.event_direction = list(event_direction or [])
        self.event_terminal = list(event_terminal or [])
        for i, d in enumerate(self.event_direction[:16]):
            o.ev_direction[i] = int(d)
        for i, t in enumerate(self.event_terminal[:16]):
            o.ev_terminal[i] = int(t or 0)
        self.c = o
        self.method = m


@dataclass
class OracleSolution:
    t: np.ndarray
    y: np.ndarray            # [len, n] time-major like the reference's Vec<Vec<f64>>
    nfev: int
    njev: int
    nlu: int
    nstep: int
    naccpt: int
    nrejct: int
    status: int
    h_next: float
    method: int
    _c: Optional[_Solution] = field(default=None, repr=False)
    _lib: object = field(default=None, repr=False)
    seg_xold: Optional[np.ndarray] = None
    seg_h: Optional[np.ndarray] = None
    seg_cont: Optional[np.ndarray] = None
    t_events: Optional[list] = None
    y_events: Optional[list] = None

    @property
    def status_name(self) -> str:
        return STATUS[self.status]

    def sol(self, t: float) -> np.ndarray:
        out = np.zeros(self._c.n)
        rc = self._lib.orc_solution_eval(C.byref(self._c), self.method, float(t), _dptr(out))
        if rc == -1:
            raise ValueError("InterpolationError::NotEnabled")
        if rc == -2:
            raise ValueError("InterpolationError::OutOfRange")
        return out

    def sol_extrapolate(self, t: float) -> np.ndarray:
        out = np.zeros(self._c.n)
        rc = self._lib.orc_solution_eval_extrapolate(C.byref(self._c), self.method, float(t), _dptr(out))
        if rc != 0:
            raise ValueError("no dense output")
        return out

    def sol_span(self):
        if self.seg_xold is None or len(self.seg_xold) == 0:
            return None
        return float(self.seg_xold[0]), float(self.seg_xold[-1] + self.seg_h[-1])

    def __del__(self):
        if self._c is not None and self._lib is not None and C is not None:   # (module globals are gone at interpreter shutdown)
            self._lib.orc_solution_free(C.byref(self._c))
            self._c = None


def solve_ivp(fun, x0: float, xend: float, y0: Sequence[float], *, params: Sequence[float] = (),
              detpow: bool = False, fma: bool = False, events=None, n_events: int = 0, jac=None, **options) -> OracleSolution:
    """One reference-style ``solve_ivp`` call.  ``fun`` is a built-in RHS name (see ``RHS``) or a
    Python callable ``f(x, y, p) -> dydx`` (slow; small cases only)."""
    L = lib(detpow, fma)
    y0a = np.ascontiguousarray(y0, dtype=np.float64)
    pa = np.ascontiguousarray(params if len(params) else [0.0], dtype=np.float64)
    n = y0a.size
    keep = None
    if isinstance(fun, str):
        nn, npp = C.c_int(), C.c_int()
        fptr = L.orc_builtin_rhs(RHS[fun], C.byref(nn), C.byref(npp))
        if n and nn.value != n:
            raise ValueError(f"rhs {fun} has n={nn.value}, got y0 of length {n}")
    else:
        def _tramp(x, yp, dp, pp):
            yy = np.ctypeslib.as_array(yp, shape=(n,))
            pv = np.ctypeslib.as_array(pp, shape=(pa.size,))
            out = np.asarray(fun(x, yy, pv), dtype=np.float64)
            for i in range(n):
                dp[i] = out[i]
        keep = _ODE_FN(_tramp)
        fptr = C.cast(keep, C.c_void_p).value
    oh = _OptHolder(**options)
    keep_ev = None
    if events is not None:   # Python callable g = events(x, y, p) -> n_events values (trait IVP::events, src/ivp.rs:31-40)
        def _ev_tramp(x, yp, gp, pp):
            yy = np.ctypeslib.as_array(yp, shape=(n,))
            pv = np.ctypeslib.as_array(pp, shape=(pa.size,))
            g = np.asarray(events(x, yy, pv), dtype=np.float64).ravel()
            for i in range(n_events):
                gp[i] = g[i]
        keep_ev = _ODE_FN(_ev_tramp)
        oh.c.events, oh.c.n_events = C.cast(keep_ev, C.c_void_p).value, int(n_events)
    elif isinstance(fun, str):
        ne = C.c_int()
        evp = L.orc_builtin_events(RHS[fun], C.byref(ne))
        if evp:
            oh.c.events, oh.c.n_events = evp, ne.value
    keep_jac = None
    if jac is not None:   # Python callable J = jac(x, y, p) -> n x n (the trait's jac override, src/ivp.rs:67-107)
        def _jac_tramp(x, yp, jp, pp):
            yy = np.ctypeslib.as_array(yp, shape=(n,))
            pv = np.ctypeslib.as_array(pp, shape=(pa.size,))
            J = np.asarray(jac(x, yy, pv), dtype=np.float64).reshape(n * n)
            for i in range(n * n):
                jp[i] = J[i]
        keep_jac = _ODE_FN(_jac_tramp)
        oh.c.jac = C.cast(keep_jac, C.c_void_p).value
    elif isinstance(fun, str):
        oh.c.jac = L.orc_builtin_jac(RHS[fun])
    s = _Solution()
    rc = L.orc_solve_ivp(fptr, _dptr(pa), n, float(x0), float(xend), _dptr(y0a), C.byref(oh.c), C.byref(s))
    if rc != 0:
        raise ValueError(f"oracle config error {rc}")
    t = np.ctypeslib.as_array(s.t, shape=(s.len,)).copy() if s.len else np.zeros(0)
    y = (np.ctypeslib.as_array(s.y, shape=(s.len, n)).copy() if (s.len and n) else np.zeros((s.len, n)))
    out = OracleSolution(t=t, y=y, nfev=s.nfev, njev=s.njev, nlu=s.nlu, nstep=s.nstep, naccpt=s.naccpt,
                         nrejct=s.nrejct, status=s.status, h_next=s.h_next, method=oh.method, _c=s, _lib=L)
    if s.has_dense and s.nseg:
        out.seg_xold = np.ctypeslib.as_array(s.seg_xold, shape=(s.nseg,)).copy()
        out.seg_h = np.ctypeslib.as_array(s.seg_h, shape=(s.nseg,)).copy()
        if n:
            out.seg_cont = np.ctypeslib.as_array(s.seg_cont, shape=(s.nseg, s.ncoef * n)).copy()
    out.t_events, out.y_events = [], []
    for i in range(s.n_events):
        m = s.ev_len[i]
        out.t_events.append(np.ctypeslib.as_array(s.t_events[i], shape=(m,)).copy() if m else np.zeros(0))
        out.y_events.append(np.ctypeslib.as_array(s.y_events[i], shape=(m, n)).copy() if m else np.zeros((0, n)))
    del keep, keep_jac
    return out


def solve_batch(rhs: str, y0: np.ndarray, params: Optional[np.ndarray], t0, t1, *, threads: int = 1,
                detpow: bool = False, fma: bool = False, **options) -> dict:
    """B back-to-back reference-style solves. ``y0`` is SoA ``[n, B]``, ``params`` ``[p, B]``."""
    L = lib(detpow, fma)
    rid = RHS[rhs]
    n, npar = RHS_DIMS[rid]
    y0 = np.ascontiguousarray(y0, dtype=np.float64)
    assert y0.shape[0] == n
    B = y0.shape[1]
    if npar:
        params = np.ascontiguousarray(params, dtype=np.float64)
        assert params.shape == (npar, B)
    else:
        params = np.zeros((1, B))
    t0 = np.atleast_1d(np.asarray(t0, dtype=np.float64)).copy()
    t1 = np.atleast_1d(np.asarray(t1, dtype=np.float64)).copy()
    assert t0.size in (1, B) and t1.size in (1, B)
    oh = _OptHolder(**options)
    ne = C.c_int()
    evp = L.orc_builtin_events(rid, C.byref(ne))
    if evp:
        oh.c.events, oh.c.n_events = evp, ne.value
    res = {
        "y_end": np.zeros((n, B)), "t_end": np.zeros(B), "status": np.zeros(B, dtype=np.int32),
        "nfev": np.zeros(B, dtype=np.uint64), "nstep": np.zeros(B, dtype=np.uint64),
        "naccpt": np.zeros(B, dtype=np.uint64), "nrejct": np.zeros(B, dtype=np.uint64),
        "h_next": np.zeros(B), "njev": np.zeros(B, dtype=np.uint64), "nlu": np.zeros(B, dtype=np.uint64),
    }
    y_eval = n_filled = None
    if oh.t_eval is not None and oh.t_eval.size:
        y_eval = np.full((oh.t_eval.size, n, B), np.nan)
        n_filled = np.zeros(B, dtype=np.int32)
        res["y_eval"], res["n_filled"] = y_eval, n_filled
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    total = L.orc_batch_solve(rid, B, vp(y0), vp(params), vp(t0), t0.size, vp(t1), t1.size,
                              C.byref(oh.c), int(threads),
                              vp(res["y_end"]), vp(res["t_end"]), vp(res["status"]), vp(res["nfev"]),
                              vp(res["nstep"]), vp(res["naccpt"]), vp(res["nrejct"]), vp(res["h_next"]),
                              vp(y_eval), vp(n_filled), vp(res["njev"]), vp(res["nlu"]))
    if total < 0:
        raise ValueError(f"oracle config error {total}")
    res["total_accepted"] = int(total)
    return res


def detpow(x: float, e: float) -> float:
    return lib(False).orc_detpow(float(x), float(e))
