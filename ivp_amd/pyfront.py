"""SciPy-style front end: the reference's Python entry point over the GPU path.

``solve_ivp(fun, t_span, y0, method=None, t_eval=None, dense_output=False, events=None, vectorized=False, args=None,
jac=None, jac_sparsity=None, **options)`` has the signature, the option names, the ``(n_states, n_points)`` result
layout and the status / message mapping of the reference's ``ivp.solve_ivp`` (src/python/solve.rs:150-222, 342-432);
``OdeResult`` and ``OdeSolution`` are its result classes (src/python/result.rs:14-99, src/python/solution.rs:13-140).

The one thing that cannot carry over is a Python callable as right-hand side: the stepping kernels run on the GPU, so
``fun`` is device code.  Accepted forms:

* an ``ivp_amd.IVP`` instance (one of the built-in problems, or a ``DeviceIVP``); ``args`` must then be None;
* a string: either the statements of ``ode``'s body, written in terms of ``x``, ``y[i]``, ``dydx[i]`` and the
  parameters ``p[k]`` (``args`` become ``p``), e.g. ``"dydx[0] = -p[0] * y[0];"``, or complete HIP definitions
  (anything containing ``__device__``) exactly as ``DeviceIVP`` takes them.

The statement form works for any state dimension up to 512: above 8 states (one wavefront per trajectory, component
form) it is wrapped so that the body is evaluated into a local vector and the requested component returned -- n times
the arithmetic; for a large system whose components are cheap on their own define
``__device__ double ode_comp(int i, double x, const double* y, const double* p)`` (and ``jac_col``) yourself.

``events`` (string form only; a built-in problem carries its own): one ``Event`` or a list of them, each a C
expression of ``x``, ``y``, ``p`` whose sign changes are located, with the ``terminal`` / ``direction`` attributes the
reference reads off the Python event functions (solve.rs:246-289).  ``jac``: a string with the statements filling
``j[row * n + col]``, or a constant matrix (njev is then reported as 0, solve.rs:216-218).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import numpy as np

from . import api

__all__ = ["solve_ivp", "OdeResult", "OdeSolution", "Event"]


class Event:
    """One event function: ``expr`` is its value as a C expression of ``x``, ``y[i]``, ``p[k]``."""

    def __init__(self, expr: str, terminal: bool = False, direction: float = 0.0):
        self.expr = expr
        self.terminal = terminal
        self.direction = direction


class OdeSolution:
    """``class OdeSolution`` (src/python/solution.rs:13-140): callable dense output; a float gives ``(n,)``, a 1-D
    array or list gives ``(n, n_points)``.  Outside the covered range the nearest segment extrapolates."""

    def __init__(self, inner: api.ContinuousOutput):
        self.inner = inner

    @property
    def t_min(self) -> Optional[float]:
        span = self.inner.t_span()
        return None if span is None else span[0]

    @property
    def t_max(self) -> Optional[float]:
        span = self.inner.t_span()
        return None if span is None else span[1]

    def __call__(self, t):
        if np.isscalar(t):   # solution.rs:98-108
            y = self.inner.evaluate_extrapolate(float(t))
            if y is None:
                raise ValueError("t is outside the solution range")
            return y
        ts = np.asarray(t, dtype=np.float64)
        if ts.ndim != 1:
            raise TypeError("t must be float or 1D array")
        if ts.size == 0:      # solution.rs:32-34
            return np.zeros(0)
        cols = []
        for ti in ts:
            yi = self.inner.evaluate_extrapolate(float(ti))
            if yi is None:
                raise ValueError(f"t={ti} is outside the solution range")
            cols.append(yi)
        return np.stack(cols, axis=1) if self.inner.n_states else np.zeros((0, ts.size))

    def __repr__(self) -> str:
        span = self.inner.t_span()
        return "<OdeSolution: empty>" if span is None else f"<OdeSolution: t_min={span[0]:.4f}, t_max={span[1]:.4f}>"


class OdeResult:
    """``class OdeResult`` (src/python/result.rs:14-99): attribute and ``res["key"]`` access."""

    _fields = ("t", "y", "t_events", "y_events", "nfev", "njev", "nlu", "status", "message", "success", "sol")

    def __init__(self, **kw):
        for k in self._fields:
            setattr(self, k, kw[k])

    def __getitem__(self, key: str):
        if key not in self._fields:
            raise KeyError(key)
        return getattr(self, key)

    def __repr__(self) -> str:
        return (f"  message: {self.message}\n  success: {self.success}\n   status: {self.status}\n"
                f"     nfev: {self.nfev}\n     njev: {self.njev}\n      nlu: {self.nlu}")


class _Degenerate(api.IVP):
    """Stands in for the device problem when the state is empty or the interval has zero length."""
    rhs_id = -1
    n_params = 0

    def __init__(self, n: int, n_events: int):
        self.n = n
        self._ne = n_events

    def n_events(self) -> int:
        return self._ne


def _event_list(events) -> List:
    if events is None:
        return []
    if isinstance(events, (list, tuple)):
        return list(events)
    return [events]


def _event_config(ev) -> api.EventConfig:
    """solve.rs:266-285: ``terminal`` counts only when it is a bool; ``direction`` is read as a float and truncated."""
    cfg = api.EventConfig()
    if getattr(ev, "terminal", None) is True:
        cfg.terminal()
    d = getattr(ev, "direction", None)
    if isinstance(d, (int, float)) and not isinstance(d, bool):
        d = int(d)
        if d > 0:
            cfg.positive()
        elif d < 0:
            cfg.negative()
    return cfg


def _device_problem(fun: str, n: int, args, events: list, jac, ctx) -> api.DeviceIVP:
    if "__device__" in fun:
        src = fun
    elif n <= api.MAX_LANE_N:
        src = "__device__ void ode(double x, const double* y, double* dydx, const double* p) {\n" + fun + "\n}\n"
    else:
        # more than 8 states: one wavefront per trajectory, the kernels ask for ONE component at a time.  The statement
        # form still describes the whole right-hand side, so it is wrapped: evaluate the body into a local vector and
        # return the requested entry (n times the arithmetic; write `__device__ double ode_comp(int i, ...)` yourself
        # for a large system whose components are cheap to evaluate one by one)
        src = ("__device__ double ode_comp(int ivp_i, double x, const double* y, const double* p) {\n"
               f"  double dydx[{n}];\n" + fun + "\n  return dydx[ivp_i];\n}\n")
    if events:
        body = "\n".join(f"  g[{i}] = ({getattr(e, 'expr', e)});" for i, e in enumerate(events))
        src += "__device__ void events(double x, const double* y, double* g, const double* p) {\n" + body + "\n}\n"
    has_jac = jac is not None
    if has_jac:
        jmat = None
        if isinstance(jac, str):
            jbody = jac
        else:
            jmat = np.asarray(jac, dtype=np.float64)
            if jmat.shape != (n, n):
                raise ValueError(f"jac must be a string or an ({n}, {n}) matrix")
            jbody = "\n".join(f"  j[{r * n + c}] = {float(jmat[r, c])!r};" for r in range(n) for c in range(n))
        if "__device__" in jbody:
            src += jbody
        elif n <= api.MAX_LANE_N:
            src += "__device__ void jac(double x, const double* y, double* j, const double* p) {\n" + jbody + "\n}\n"
        elif jmat is not None:
            # a constant matrix on the wave-per-trajectory path (Jacobian column by column): one switch case per column with
            # that column's non-zero entries -- O(nnz) code, O(column) work per call, no private storage
            cases = []
            for c in range(n):
                rows = [r for r in range(n) if jmat[r, c] != 0.0]
                if rows:
                    cases.append(f"  case {c}: " + " ".join(f"column[{r}] = {float(jmat[r, c])!r};" for r in rows) + " break;")
            src += ("__device__ void jac_col(int ivp_c, double x, const double* y, double* column, const double* p) {\n"
                    f"  for (int r = 0; r < {n}; ++r) column[r] = 0.0;\n  switch (ivp_c) {{\n" + "\n".join(cases) + "\n  default: break;\n  }\n}\n")
        else:
            # Statement form (`j[row * n + col] = ...;`) on the wave-per-trajectory path, which takes the Jacobian column by
            # column: `j` is an O(1) PROXY, not an n x n private array (80 KB per lane at n = 100, 2 MB at n = 512) -- a write
            # to an entry of the requested column lands in `column`, any other write goes to a dummy.  With constant indices
            # the test folds at compile time, so only the requested column's statements survive per case of ivp_c.  The
            # column starts zeroed (entries the body never writes are 0, like the reference's Matrix, bdf.rs:152).
            # Limitation: a body that READS entries of j sees only the requested column (other entries read as 0).
            src += ("struct ivp_jac_proxy {\n  double* column; int col; double sink;\n"
                    f"  __device__ double& operator[](int idx) {{ sink = 0.0; return (idx % {n} == col) ? column[idx / {n}] : sink; }}\n}};\n"
                    "__device__ void jac_col(int ivp_c, double x, const double* y, double* column, const double* p) {\n"
                    f"  for (int r = 0; r < {n}; ++r) column[r] = 0.0;\n  ivp_jac_proxy j{{column, ivp_c, 0.0}};\n" + jbody + "\n}\n")
    params = () if args is None else tuple(float(a) for a in (args if isinstance(args, (tuple, list)) else (args,)))
    return api.DeviceIVP(src, n, params, ctx=ctx, events=[_event_config(e) for e in events], jac=has_jac)


def solve_ivp(fun: Union[api.IVP, str], t_span, y0, method=None, t_eval=None, dense_output: bool = False, events=None,
              vectorized: bool = False, args=None, jac=None, jac_sparsity=None, ctx: api.Context = None,
              **options) -> OdeResult:
    """``ivp.solve_ivp`` (src/python/solve.rs:150-222).  Recognised ``options``: rtol, atol (scalar or per-component),
    max_step, min_step, first_step, max_steps (solve.rs:292-340); like the reference, other keys are ignored."""
    del vectorized   # accepted and unused, as in the reference (solve.rs:166)
    if jac_sparsity is not None:
        # the reference groups finite-difference columns by the sparsity pattern (src/python/sparsity.rs); the
        # device BDF differences a dense n <= 8 Jacobian column by column -- refusing beats silently different nfev
        raise NotImplementedError("jac_sparsity is not supported on the GPU path (dense Jacobians only)")
    t0, tf = (float(v) for v in t_span)
    y0v = np.atleast_1d(np.asarray(y0, dtype=np.float64))
    ev = _event_list(events)
    if isinstance(fun, api.IVP):
        if args is not None:
            raise ValueError("args: a built-in problem carries its own parameters")
        if ev or jac is not None:
            raise ValueError("events / jac: a built-in problem carries its own")
        problem = fun
        has_events = problem.n_events() > 0
        constant_jac = False
    elif isinstance(fun, str):
        if y0v.size == 0 or abs(tf - t0) < 1e-15:
            # never reaches an integrator (solve_ivp.rs:110-176): nothing to compile, and no GPU is touched
            problem = _Degenerate(y0v.size, len(ev))
        else:
            # the explicit methods never call the Jacobian (the reference accepts and ignores it there)
            uses_jac = isinstance(method, str) and api.Method.from_str(method) == api.Method.BDF
            try:
                problem = _device_problem(fun, y0v.size, args, ev, jac if uses_jac else None, ctx)
            except api.IvpError as e:   # a snippet that does not compile surfaces like any other solver failure (solve.rs:216-221)
                raise RuntimeError(f"Solver failed: {e}") from e
        has_events = events is not None
        constant_jac = jac is not None and not isinstance(jac, str)
    else:
        raise TypeError("fun must be an ivp_amd.IVP instance or device source text (see the module docstring): "
                        "the stepping kernels cannot call back into Python")
    opts = api.Options(
        method=api.Method.from_str(method) if isinstance(method, str) else api.Method.DOPRI5,   # solve.rs:224-232
        rtol=options.get("rtol", 1e-3), atol=options.get("atol", 1e-6),
        max_steps=options.get("max_steps"), first_step=options.get("first_step"),
        max_step=options.get("max_step"), min_step=options.get("min_step"),
        t_eval=None if t_eval is None else np.asarray(t_eval, dtype=np.float64), dense_output=bool(dense_output))
    try:
        s = api.solve_ivp(problem, t0, tf, y0v, opts, ctx)
    except api.IvpError as e:   # solve.rs:216-221
        raise RuntimeError(f"Solver failed: {e}") from e
    status = 0 if s.status == api.Status.Success else (1 if s.status == api.Status.UserInterrupt else -1)
    t_events = y_events = None
    if has_events:   # solve.rs:371-403: arrays per event; an event without hits gives an empty list for y
        t_events = [np.asarray(te, dtype=np.float64) for te in s.t_events]
        y_events = [np.asarray(ye, dtype=np.float64) if len(ye) else [] for ye in s.y_events]
    return OdeResult(
        t=np.asarray(s.t, dtype=np.float64),
        y=np.ascontiguousarray(np.asarray(s.y, dtype=np.float64).T) if len(s.t) else np.zeros((0, 0)),
        t_events=t_events, y_events=y_events, nfev=s.nfev, njev=0 if constant_jac else s.njev, nlu=s.nlu,
        status=status, message=s.status.name, success=status >= 0,
        sol=OdeSolution(s.continuous_sol) if s.continuous_sol is not None else None)
