"""Host-side mirror of the reference's solve_ivp()/IVP/Options surface for the GPU path.

Names, argument meaning, defaults and error behaviour follow the Rust crate Ryan-D-Gast/ivp 0.5.1
(file:line citations are relative to the reference tree):

* ``Method``   -- ``enum Method`` and ``From<&str>``            src/solve/options.rs:13-73
* ``Options``  -- ``struct Options`` builder defaults           src/solve/options.rs:75-123
* ``Status``   -- ``enum Status``                               src/status.rs:4-26
* ``IVP``      -- ``trait IVP`` (the RHS becomes device code)   src/ivp.rs:27-121
* ``Solution`` -- ``struct Solution`` + sol/sol_many/sol_span   src/solve/solution.rs:7-97
* ``ContinuousOutput``                                          src/solve/cont.rs:9-153
* ``solve_ivp``                                                 src/solve/solve_ivp.rs:99-313

Everything numeric runs in libivp_hip.so on the GPU; this module only marshals buffers, maps error
codes to exceptions and post-processes logged steps/segments on the host (the reference's
ContinuousOutput is host-side post-processing too).  ``solve_ivp_batch`` is the batched entry point
the reference does not have: B independent solve_ivp() calls advanced in lock-step.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib


# ------------------------------------------------------------------------------------------------
# enums / errors
# ------------------------------------------------------------------------------------------------
class Method(enum.IntEnum):
    """src/solve/options.rs:13-27 (same discriminant order)."""
    RK23 = 0
    DOPRI5 = 1
    DOP853 = 2
    RK4 = 3
    RADAU = 4
    BDF = 5

    @staticmethod
    def from_str(s: str) -> "Method":
        """``impl From<&str> for Method`` (options.rs:61-73): RK45 == DOPRI5, unknown => DOPRI5."""
        return {
            "RK23": Method.RK23, "DOPRI5": Method.DOPRI5, "RK45": Method.DOPRI5, "DOP853": Method.DOP853,
            "RK4": Method.RK4, "RADAU": Method.RADAU, "RADAU5": Method.RADAU, "BDF": Method.BDF, "BDF15": Method.BDF,
        }.get(str(s).upper(), Method.DOPRI5)

    def coeffs_per_state(self) -> int:
        """options.rs:34-43."""
        return {Method.RK4: 4, Method.RK23: 4, Method.DOPRI5: 5, Method.DOP853: 8, Method.RADAU: 4, Method.BDF: 7}[self]


class Status(enum.IntEnum):
    """src/status.rs:4-19."""
    Success = 0
    UserInterrupt = 1
    NeedLargerNMax = 2
    StepSizeTooSmall = 3
    ProbablyStiff = 4
    SingularMatrix = 5
    PoorConvergence = 6

    def is_success(self) -> bool:
        return self in (Status.Success, Status.UserInterrupt)


class FpMode(enum.IntEnum):
    """``ivp_fp_mode_t``: STRICT = the reference's IEEE operation sequence; FMA = the same sequence with the marked
    multiply-add sites fused (a defined arithmetic with its own oracle build; identical bits in every kernel variant)."""
    STRICT = 0
    FMA = 1
    FAST = 1   # older name of FMA


class IvpError(Exception):
    """``enum Error`` (src/error.rs:7-14)."""


class ConfigError(IvpError):
    """``Error::Config(ConfigError)`` (src/error.rs:18-60); ``code`` is the C ABI value."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"{_lib.ERRORS.get(code, code)}: {msg}")
        self.code = code


class InterpolationError(IvpError):
    """``Error::Interpolation`` (src/error.rs:72-80)."""


class Direction(enum.IntEnum):
    """``enum Direction`` (src/solve/event.rs:59-77); the values are the C ABI's ev_direction encoding."""
    All = 0
    Positive = 1
    Negative = -1


@dataclass
class EventConfig:
    """``struct EventConfig`` (src/solve/event.rs:5-57)."""
    direction: Direction = Direction.All
    terminal_count: Optional[int] = None

    def terminal(self):
        self.terminal_count = 1
        return self

    def positive(self):
        self.direction = Direction.Positive
        return self

    def negative(self):
        self.direction = Direction.Negative
        return self

    def all(self):
        self.direction = Direction.All
        return self


# ------------------------------------------------------------------------------------------------
# IVP: device right-hand sides
# ------------------------------------------------------------------------------------------------
class IVP:
    """Device-side analogue of ``trait IVP`` (src/ivp.rs:27-121).

    A problem is a device functor id (or a JIT handle) plus the values of the user's struct fields,
    passed per trajectory as ``params``.
    """
    rhs_id: int = -1
    n: int = 0
    n_params: int = 0

    def params(self) -> Sequence[float]:
        return ()

    def n_events(self) -> int:  # trait IVP::n_events (src/ivp.rs:42-46); the event functions are device code
        return 0

    def event_config(self, index: int) -> EventConfig:  # trait IVP::event_config (src/ivp.rs:48-52)
        return EventConfig()


@dataclass
class ExponentialDecay(IVP):  # examples/exponential_decay.rs:5-14
    k: float = 0.5
    rhs_id = 0; n = 1; n_params = 1
    def params(self): return (self.k,)


@dataclass
class SHO(IVP):  # tests/common.rs:3-9
    rhs_id = 1; n = 2; n_params = 0


@dataclass
class VanDerPol(IVP):  # benches/benchmark.py:22-27
    mu: float = 1.0
    rhs_id = 2; n = 2; n_params = 1
    def params(self): return (self.mu,)


@dataclass
class CR3BP(IVP):  # examples/cr3bp.rs:9-37
    mu: float = 0.012277471
    rhs_id = 3; n = 6; n_params = 1
    def params(self): return (self.mu,)

    def jacobi_constant(self, s):  # examples/cr3bp.rs:14-20
        x, y, z, vx, vy, vz = s
        r1 = np.sqrt((x + self.mu) ** 2 + y * y + z * z)
        r2 = np.sqrt((x - 1.0 + self.mu) ** 2 + y * y + z * z)
        u = 0.5 * (x * x + y * y) + (1.0 - self.mu) / r1 + self.mu / r2
        return 2.0 * u - (vx * vx + vy * vy + vz * vz)


@dataclass
class Lorenz(IVP):  # benches/benchmark.py:30-37
    sigma: float = 10.0
    rho: float = 28.0
    beta: float = 8.0 / 3.0
    rhs_id = 4; n = 3; n_params = 3
    def params(self): return (self.sigma, self.rho, self.beta)


@dataclass
class ZeroRhs(IVP):  # tests/ivp.rs:11-19
    rhs_id = 5; n = 3; n_params = 0


@dataclass
class Rational(IVP):  # tests/test_helpers.py:23-25
    rhs_id = 6; n = 2; n_params = 0


@dataclass
class Exp2(IVP):  # tests/ivp.rs:291-298
    rhs_id = 7; n = 2; n_params = 0


@dataclass
class LinearSystem(IVP):  # tests/test_helpers.py:11-12  (y' = [[-1,-5],[1,1]] y)
    rhs_id = 8; n = 2; n_params = 0


@dataclass
class Robertson(IVP):  # tests/test_ivp.py:327-333
    rhs_id = 9; n = 3; n_params = 0


@dataclass
class RobertsonJac(IVP):  # Robertson + `fn jac` override: the analytic Jacobian (trait IVP::jac, src/ivp.rs:67-107)
    rhs_id = 15; n = 3; n_params = 0


@dataclass
class StiffVanDerPol(IVP):  # examples/van_der_pol.rs:5-15
    eps: float = 1e-3
    rhs_id = 10; n = 2; n_params = 1
    def params(self): return (self.eps,)


class _EventProblem(IVP):
    """Built-in problems that carry event functions; ``configs`` are the per-event EventConfig values."""
    _ne = 1

    def __init__(self, *configs: EventConfig):
        self.configs = list(configs) + [EventConfig() for _ in range(self._ne - len(configs))]

    def n_events(self) -> int:
        return self._ne

    def event_config(self, index: int) -> EventConfig:
        return self.configs[index]


class SHOZeroEvent(_EventProblem):  # tests/ivp.rs:151-221: SHO with event y[0]
    rhs_id = 11; n = 2; n_params = 0


class BouncingBall(_EventProblem):  # examples/bouncing_ball.rs:5-31
    rhs_id = 12; n = 2; n_params = 2

    def __init__(self, gravity: float = 9.81, drag: float = 0.02, *configs: EventConfig):
        super().__init__(*(configs or (EventConfig().terminal().negative(),)))
        self.gravity, self.drag = gravity, drag

    def params(self):
        return (self.gravity, self.drag)


class Cannon(_EventProblem):  # tests/test_ivp.py:152-160
    rhs_id = 13; n = 2; n_params = 0


class RationalEvents(_EventProblem):  # tests/test_ivp.py:345-353
    rhs_id = 14; n = 2; n_params = 0
    _ne = 3


@dataclass
class LinearDecay100(IVP):  # benches/benchmark.py:40-42,139-148 "Large Linear System (N=100)"
    """y' = -y with 100 components: a large-n problem (one wavefront per trajectory; explicit RK methods)."""
    rhs_id = 100; n = 100; n_params = 0


@dataclass
class Heat1D256(IVP):
    """Method-of-lines heat equation y_i' = kappa (y_{i-1} - 2 y_i + y_{i+1}), 256 interior nodes, zero ends."""
    kappa: float = 1.0
    rhs_id = 101; n = 256; n_params = 1
    def params(self): return (self.kappa,)


@dataclass
class Dense64(IVP):
    """y' = A y with a dense, diagonally dominant 64 x 64 matrix (a_ii = -k (4 + i mod 5), a_ij = (((5 i + 3 j) & 15) - 8) / 256):
    a full Jacobian for the per-trajectory LU of BDF on the wave-per-trajectory path."""
    k: float = 1.0
    rhs_id = 102; n = 64; n_params = 1
    def params(self): return (self.k,)


MAX_LANE_N = 8   # largest n of the thread-per-trajectory kernels; above it one wavefront owns a trajectory

BUILTIN = {"linear_decay100": LinearDecay100, "heat1d256": Heat1D256, "dense64": Dense64, "sho_ev": SHOZeroEvent, "ball": BouncingBall, "cannon": Cannon, "rational_ev": RationalEvents,
           "linear": LinearSystem, "robertson": Robertson, "robertson_jac": RobertsonJac, "vdp_eps": StiffVanDerPol, "decay": ExponentialDecay, "sho": SHO, "vdp": VanDerPol, "cr3bp": CR3BP, "lorenz": Lorenz,
           "zero": ZeroRhs, "rational": Rational, "exp2": Exp2}


class DeviceIVP(IVP):
    """User-defined system: the device-side ``impl IVP for T { fn ode(&self, x, y, dydx) }``.

    ``source`` is HIP device code defining
    ``__device__ void ode(double x, const double* y, double* dydx, const double* p)``;
    ``params`` are the values of the struct's fields (``p[...]`` inside ``ode``).
    For ``8 < n <= 512`` the snippet defines the component form
    ``__device__ double ode_comp(int i, double x, const double* y, const double* p)`` instead (one wavefront per
    trajectory; RK23 / DOPRI5 / DOP853 / RK4; ``events`` keeps the whole-state signature).
    """
    rhs_id = 1000

    def __init__(self, source: str, n: int, params: Sequence[float] = (), ctx: "Context" = None,
                 events: Sequence[EventConfig] = (), jac: bool = False):
        """``events``: one EventConfig per event function; ``source`` must then also define
        ``__device__ void events(double x, const double* y, double* g, const double* p)``.
        ``jac=True``: ``source`` also overrides the trait's Jacobian (src/ivp.rs:67-107), used by BDF in place of the
        default forward differences: ``__device__ void jac(double x, const double* y, double* j, const double* p)``
        with ``j[row * n + col]`` for n <= 8; for larger systems the column form
        ``__device__ void jac_col(int col, double x, const double* y, double* column, const double* p)``.  Entries the
        override never writes are zero (the reference's Matrix starts zeroed)."""
        self.source = source
        self.n = int(n)
        self._params = tuple(float(v) for v in params)
        self.n_params = len(self._params)
        self._events = list(events)
        self._ctx = ctx or default_context()
        # One compiled record per distinct (source, dimensions): the record owns the hiprtc modules (one per device,
        # method and mode, built on first use), none of which depends on the parameter VALUES or the event settings, so
        # a second DeviceIVP with the same text -- pyfront.solve_ivp builds one per call, as the reference's Python
        # front end wraps its callable per call -- reuses them instead of compiling again.  Records live as long as the
        # process (a few hundred bytes of host state plus the loaded code objects).
        key = (source, self.n, self.n_params, len(self._events), bool(jac))
        h = _rhs_records.get(key)
        if h is None:
            h = C.c_void_p()
            rc = self._ctx.lib.ivp_rhs_compile_ex(self._ctx.handle, source.encode(), self.n, self.n_params,
                                                  len(self._events), 1 if jac else 0, C.byref(h))
            if rc != 0:
                raise ConfigError(rc, self._ctx.last_error())
            _rhs_records[key] = h
        self.handle = h

    def params(self):
        return self._params

    def n_events(self):
        return len(self._events)

    def event_config(self, index):
        return self._events[index]


_rhs_records: dict = {}   # (source, n, n_params, n_events, jac) -> ivp_rhs_compile_ex handle


# ------------------------------------------------------------------------------------------------
# Options
# ------------------------------------------------------------------------------------------------
@dataclass
class Options:
    """``Options::builder()...build()`` (src/solve/options.rs:75-123) for the explicit-RK fields, plus
    the knobs that exist only on the GPU path."""
    method: Union[Method, str] = Method.DOPRI5
    rtol: Union[float, Sequence[float]] = 1e-3
    atol: Union[float, Sequence[float]] = 1e-6
    max_steps: Optional[int] = None
    t_eval: Optional[Sequence[float]] = None
    # Batch calls only: one output grid PER TRAJECTORY -- every reference solve_ivp() call has its own Options.t_eval
    # (options.rs:75-123).  A sequence of B sequences (ragged); mutually exclusive with t_eval.  The samples come back as
    # time-major CSR records (BatchSolution.eval_offsets / eval_of(b)).
    t_eval_per_trajectory: Optional[Sequence[Sequence[float]]] = None
    first_step: Optional[float] = None
    max_step: Optional[float] = None
    min_step: Optional[float] = None      # read by BDF only (src/solve/solve_ivp.rs:271)
    dense_output: bool = False
    # A direct per-method call -- ``DOPRI5::builder().safety_factor(0.8).build().solve(..)`` (dopri5.rs:34-72,
    # dop853.rs:34-63, rk23.rs:17-37): a dict with any of uround / safety_factor / scale_min / scale_max / beta /
    # stiff_test; missing keys take the struct's defaults, and max_steps then defaults to the struct's 100_000
    # (RK23: 10_000) instead of solve_ivp's unlimited.  None = what solve_ivp() runs.
    settings: Optional[dict] = None
    # GPU-only
    fp_mode: FpMode = FpMode.STRICT
    chunk_attempts: int = 0
    max_log: int = 0
    max_events: int = 64               # capacity of t_events / y_events per event and trajectory
    variant: int = 0                   # stepping-kernel variant: 0 auto, 1 lean registers, 2 coefficients resident,
                                       # 3 lane-cooperative (8 lanes per trajectory; DOPRI5 / DOP853) -- strict results never depend on it
    profile: int = 0                   # 1: HIP-event kernel timing, 2: + batch totals (see ivp_run_stats_t)
    count_log: bool = False            # counting pass of the CSR step log: only n_log is produced (solve_ivp_batch_logged)

    def _c(self, n: int, keep: list) -> _lib.OptionsT:
        o = _lib.OptionsT()
        _lib.load().ivp_options_default(C.byref(o))
        m = self.method if isinstance(self.method, Method) else Method.from_str(self.method)
        o.method = int(m)
        for name in ("rtol", "atol"):
            v = getattr(self, name)
            if np.isscalar(v):
                setattr(o, name, float(v))
            else:
                arr = np.ascontiguousarray(v, dtype=np.float64)
                keep.append(arr)
                setattr(o, name + "_vec", arr.ctypes.data_as(_lib.c_double_p))
                setattr(o, name + "_vec_len", arr.size)
        if self.max_steps is not None:
            if self.max_steps <= 0:
                # XXX::solve(): nmax == 0 => Err(Config(MustBePositive)) (dopri5.rs:183-189)
                raise ConfigError(-1, "invalid max_steps: 0 (must be > 0)")
            o.max_steps = int(self.max_steps)
        if self.t_eval is not None:
            te = np.ascontiguousarray(self.t_eval, dtype=np.float64)
            keep.append(te)
            if not te.size:   # Some(vec![]): a non-NULL pointer to a buffer that stays alive as long as the options struct
                te = np.zeros(1)
                keep.append(te)
            o.t_eval = te.ctypes.data_as(_lib.c_double_p)
            o.n_eval = len(self.t_eval)
        if self.t_eval_per_trajectory is not None:
            if self.t_eval is not None:
                raise ValueError("give t_eval (one shared grid) or t_eval_per_trajectory, not both")
            grids = [np.ascontiguousarray(g, dtype=np.float64).reshape(-1) for g in self.t_eval_per_trajectory]
            off = np.zeros(len(grids) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([g.size for g in grids])
            te = np.concatenate(grids) if grids and off[-1] else np.zeros(1)
            keep += [te, off]
            o.t_eval = te.ctypes.data_as(_lib.c_double_p)
            o.n_eval = int(off[-1])
            o.t_eval_offsets = off.ctypes.data_as(C.POINTER(C.c_uint64))
        if self.first_step is not None:
            o.has_first_step, o.first_step = 1, float(self.first_step)
        if self.max_step is not None:
            o.has_max_step, o.max_step = 1, float(self.max_step)
        o.dense_output = int(bool(self.dense_output))
        if self.min_step is not None:
            o.has_min_step, o.min_step = 1, float(self.min_step)
        if self.settings is not None:
            if m not in (Method.RK23, Method.DOPRI5, Method.DOP853):
                raise ConfigError(-100, "settings apply to RK23 / DOPRI5 / DOP853")
            rc = _lib.load().ivp_options_method_defaults(C.byref(o), int(m))
            if rc != 0:
                raise ConfigError(rc, "ivp_options_method_defaults failed")
            for key, val in self.settings.items():
                if key not in ("uround", "safety_factor", "scale_min", "scale_max", "beta", "stiff_test"):
                    raise ValueError(f"unknown method setting {key!r}")
                setattr(o, key, int(val) if key == "stiff_test" else float(val))
            o.has_settings = 1
            if self.max_steps is None:
                o.max_steps = 10_000 if m == Method.RK23 else 100_000
        o.fp_mode = int(self.fp_mode)
        o.chunk_attempts = int(self.chunk_attempts)
        o.max_log = int(self.max_log)
        o.max_events = int(self.max_events)
        o.variant = int(self.variant)
        o.profile = int(self.profile)
        o.count_log = int(bool(self.count_log))
        return o

    @property
    def method_enum(self) -> Method:
        return self.method if isinstance(self.method, Method) else Method.from_str(self.method)


# ------------------------------------------------------------------------------------------------
# Context
# ------------------------------------------------------------------------------------------------
class Context:
    """One ``ivp_ctx_t``: device scratch for one (device, stream) pair. Not thread-safe."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.ivp_ctx_create(C.byref(h), int(device))
        if rc != 0:
            raise IvpError(f"ivp_ctx_create(device={device}) failed: {_lib.ERRORS.get(rc, rc)} "
                           "(libivp_hip needs a HIP device; there is no CPU fallback)")
        self.handle = h
        self.device = device
        self._scalars = {}   # device copies of scalar t0 / t1 values (saves two tiny host-to-device copies per call)

    def last_error(self) -> str:
        return self.lib.ivp_last_error_string(self.handle).decode(errors="replace")

    def stats(self) -> dict:
        s = _lib.RunStatsT()
        self.lib.ivp_ctx_get_stats(self.handle, C.byref(s))
        return {k: getattr(s, k) for k, _ in _lib.RunStatsT._fields_}

    def close(self):
        if self.handle:
            self.lib.ivp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ------------------------------------------------------------------------------------------------
# Dense output (host post-processing of the logged segments)
# ------------------------------------------------------------------------------------------------
def _interpolate(method: Method, xi: float, cont: np.ndarray, n: int, xold: float, h: float) -> np.ndarray:
    """dopri5.rs:467-478, dop853.rs:659-670, rk23.rs:313-321 (same association)."""
    if n == 0:
        return np.zeros(0)
    c = cont.reshape(-1, n)
    if method == Method.DOPRI5:
        th = (xi - xold) / h
        th1 = 1.0 - th
        return c[0] + th * (c[1] + th1 * (c[2] + th * (c[3] + th1 * c[4])))
    if method == Method.DOP853:
        s = (xi - xold) / h
        s1 = 1.0 - s
        conpar = c[4] + s * (c[5] + s1 * (c[6] + s * c[7]))
        return c[0] + s * (c[1] + s1 * (c[2] + s * (c[3] + s1 * conpar)))
    if method == Method.BDF:  # bdf.rs:618-656; per-state blocks [D0, D1..D5, order]
        if h == 0.0:
            return np.zeros(n)
        blk = cont.reshape(n, 7)
        order = int(min(max(round(blk[0, 6]), 1), 5))
        x_new = xold + h
        p = np.zeros(5)
        for k in range(order):
            xf = (xi - (x_new - h * k)) / (h * (k + 1.0))
            p[k] = xf if k == 0 else p[k - 1] * xf
        out = blk[:, 0].copy()
        for k in range(order):
            out = out + blk[:, 1 + k] * p[k]
        return out
    if method == Method.RK4:  # rk4.rs:229-244
        t = (xi - xold) / h
        t2 = t * t
        t3 = t2 * t
        h00 = 2.0 * t3 - 3.0 * t2 + 1.0
        h10 = t3 - 2.0 * t2 + t
        h01 = -2.0 * t3 + 3.0 * t2
        h11 = t3 - t2
        return h00 * c[0] + h10 * h * c[1] + h01 * c[3] + h11 * h * c[2]
    xc = (xi - xold) / h
    x2 = xc * xc
    x3 = x2 * xc
    return c[0] + h * (c[1] * xc + c[2] * x2 + c[3] * x3)


class ContinuousOutput:
    """``struct ContinuousOutput`` (src/solve/cont.rs:9-153)."""

    def __init__(self, method: Method, n_states: int, seg_cont: np.ndarray, seg_xold: np.ndarray, seg_h: np.ndarray):
        keep = seg_h != 0.0  # from_segments filters h == 0 (cont.rs:23)
        self.method = method
        self.n_states = n_states
        self.cont = seg_cont[keep]
        self.xold = seg_xold[keep]
        self.h = seg_h[keep]

    @staticmethod
    def constant(method: Method, x0: float, y0: np.ndarray) -> "ContinuousOutput":
        n = len(y0)
        nc = method.coeffs_per_state()
        cont = np.zeros((1, nc * n))
        if method == Method.BDF:   # cont.rs:44-51
            cont[0, 0::nc] = y0
            cont[0, nc - 1::nc] = 1.0
        else:
            cont[0, :n] = y0
        return ContinuousOutput(method, n, cont, np.array([x0]), np.array([1e-15]))

    def t_span(self):
        if len(self.h) == 0:
            return None
        return float(self.xold[0]), float(self.xold[-1] + self.h[-1])

    def _find(self, t: float) -> Optional[int]:
        tol = 1e-12
        a, b = self.xold, self.xold + self.h
        left, right = np.minimum(a, b), np.maximum(a, b)
        hit = np.nonzero((t >= left - tol) & (t <= right + tol))[0]
        return int(hit[0]) if hit.size else None

    def evaluate(self, t: float) -> Optional[np.ndarray]:
        s = self._find(t)
        if s is None:
            return None
        return _interpolate(self.method, t, self.cont[s], self.n_states, self.xold[s], self.h[s])

    def evaluate_many(self, ts) -> List[Optional[np.ndarray]]:
        return [self.evaluate(t) for t in ts]

    def evaluate_extrapolate(self, t: float) -> Optional[np.ndarray]:
        if len(self.h) == 0:
            return None
        s = self._find(t)
        if s is None:
            first_left = min(self.xold[0], self.xold[0] + self.h[0])
            last_right = max(self.xold[-1], self.xold[-1] + self.h[-1])
            if t < first_left:
                s = 0
            elif t > last_right:
                s = len(self.h) - 1
            else:
                return None
        return _interpolate(self.method, t, self.cont[s], self.n_states, self.xold[s], self.h[s])


# ------------------------------------------------------------------------------------------------
# Solution
# ------------------------------------------------------------------------------------------------
@dataclass
class Solution:
    """``struct Solution`` (src/solve/solution.rs:7-20); ``y`` is time-major ``[len(t), n]``."""
    t: np.ndarray
    y: np.ndarray
    t_events: list
    y_events: list
    nfev: int
    njev: int
    nlu: int
    nstep: int
    naccpt: int
    nrejct: int
    status: Status
    continuous_sol: Optional[ContinuousOutput] = None
    h_next: float = 0.0

    def sol(self, t: float) -> np.ndarray:  # solution.rs:25-47
        dense = self.continuous_sol
        if dense is None or dense.t_span() is None:
            raise InterpolationError("NotEnabled")
        start, end = dense.t_span()
        lo, hi = min(start, end), max(start, end)
        if t < lo or t > hi:
            raise InterpolationError(f"OutOfRange t={t} [{start}, {end}]")
        out = dense.evaluate(t)
        if out is None:
            raise InterpolationError(f"OutOfRange t={t} [{start}, {end}]")
        return out

    def sol_many(self, ts) -> np.ndarray:  # solution.rs:51-69
        return np.array([self.sol(t) for t in ts])

    def sol_span(self):  # solution.rs:72-74
        return None if self.continuous_sol is None else self.continuous_sol.t_span()

    def iter(self):  # solution.rs:77-79
        return zip(self.t, self.y)

    def __iter__(self):
        return self.iter()


@dataclass
class BatchSolution:
    """Struct-of-arrays result of B independent solve_ivp() calls (host numpy or device torch arrays)."""
    y_end: object
    t_end: object
    status: object
    nfev: object
    nstep: object
    naccpt: object
    nrejct: object
    h_next: object
    y_eval: object = None
    eval_idx: object = None
    n_filled: object = None
    t_log: object = None
    y_log: object = None
    n_log: object = None
    seg_cont: object = None
    seg_xold: object = None
    seg_h: object = None
    n_seg: object = None
    t_events: object = None
    y_events: object = None
    n_event_hits: object = None
    t_term: object = None
    njev: object = None
    nlu: object = None
    eval_offsets: object = None  # per-trajectory t_eval grids: [B+1] record offsets into y_eval [total, n] / eval_idx [total]
    log_offsets: object = None   # CSR step log (solve_ivp_batch_logged): [B+1] record offsets; t_log [total], y_log [total, n]
    stats: dict = field(default_factory=dict)
    event_overflow: bool = False  # some trajectory detected more occurrences of an event than max_events could store
    log_info: dict = field(default_factory=dict)   # solve_ivp_batch_logged: passes (1 = page pool, 2 = counted fill pass), pages, ...
    _log_buffers: object = None   # the full-capacity t / y buffers behind t_log / y_log (reused through `out=`)

    def eval_of(self, b: int):
        """(index into trajectory b's own t_eval grid, y) of its emitted samples (per-trajectory grids); index -1 marks the
        sample a terminal event appended."""
        lo = int(self.eval_offsets[b])
        m = int(self.n_filled[b])
        return self.eval_idx[lo:lo + m], self.y_eval[lo:lo + m]

    def log_of(self, b: int):
        """(t, y) of trajectory b from a CSR step log: Solution.t / Solution.y of that solve_ivp() call."""
        lo, hi = int(self.log_offsets[b]), int(self.log_offsets[b + 1])
        return self.t_log[lo:hi], self.y_log[lo:hi]


# ------------------------------------------------------------------------------------------------
# solve
# ------------------------------------------------------------------------------------------------
def _problem_c(f: IVP) -> _lib.ProblemT:
    p = _lib.ProblemT()
    p.rhs_id, p.n, p.n_params = int(f.rhs_id), int(f.n), int(f.n_params)
    p.jit = getattr(f, "handle", None)
    return p


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _flag_event_overflow(res: "BatchSolution") -> None:
    """The reference's t_events / y_events are Vecs that grow with every hit (src/solve/solout.rs:158-331); the batch
    buffers hold max_events per event and trajectory.  n_event_hits keeps counting past the capacity: report it."""
    if res.n_event_hits is None or res.t_events is None:
        return
    cap = int(res.t_events.shape[1])
    most = int(res.n_event_hits.max()) if res.n_event_hits.shape[0] and res.n_event_hits.shape[1] else 0
    if most > cap:
        import warnings
        res.event_overflow = True
        warnings.warn(f"event buffers overflowed: up to {most} occurrences of one event in a trajectory, capacity "
                      f"max_events = {cap}; t_events / y_events hold the first {cap} -- rerun with Options(max_events >= {most})",
                      RuntimeWarning, stacklevel=3)


class PendingBatch:
    """A solve in flight (``solve_ivp_batch(..., wait=False)``): ``done()`` advances it without blocking,
    ``result()`` blocks until the results are final.  One per Context at a time."""

    def __init__(self, ctx: "Context", res: "BatchSolution", profile: bool, keep: list):
        self._ctx, self._res, self._profile, self._keep = ctx, res, profile, keep
        self._done = False

    def _finish(self):
        self._done = True
        self._keep = None
        if self._profile:
            self._res.stats = self._ctx.stats()
        _flag_event_overflow(self._res)

    def done(self) -> bool:
        if self._done:
            return True
        flag = C.c_int(0)
        rc = self._ctx.lib.ivp_batch_poll(self._ctx.handle, C.byref(flag))
        if rc != 0:
            self._done = True
            raise ConfigError(rc, self._ctx.last_error())
        if flag.value:
            self._finish()
        return self._done

    def result(self) -> "BatchSolution":
        if not self._done:
            rc = self._ctx.lib.ivp_batch_wait(self._ctx.handle)
            if rc != 0:
                self._done = True
                raise ConfigError(rc, self._ctx.last_error())
            self._finish()
        return self._res


def solve_ivp_batch(f: IVP, t0, t1, y0, params=None, options: Options = None, ctx: Context = None,
                    out: BatchSolution = None, wait: bool = True, _steplog=None):
    """B independent ``solve_ivp(f, t0[b], t1[b], y0[:, b], options)`` calls on the GPU.

    ``y0``: ``[n, B]`` float64, numpy (host path: staged through the library) or a CUDA torch tensor
    (zero-copy device path on torch's current stream).  ``params``: ``[n_params, B]`` per-trajectory
    values of the problem struct's fields; defaults to ``f.params()`` broadcast over the batch.
    ``t0`` / ``t1``: scalars or ``[B]`` arrays of the same kind as ``y0``.
    ``wait=False`` (device arrays only): enqueue the solve and return a ``PendingBatch``; several contexts can then be
    driven from one host thread (``ivp_batch_submit_device`` / ``ivp_batch_poll``).
    """
    options = options or Options()
    on_device = _is_torch(y0)
    if on_device:
        import torch
        xp_zeros = lambda shape, dt: torch.zeros(shape, dtype=dt, device=y0.device)
        f64, i32, u64, u32 = torch.float64, torch.int32, torch.int64, torch.int32
        if y0.dtype != torch.float64 or not y0.is_contiguous():
            raise ValueError("y0 must be a contiguous float64 tensor [n, B]")
        ptr = lambda a: None if a is None else C.c_void_p(a.data_ptr())
        dev_index = y0.device.index or 0
        ctx = ctx or default_context(dev_index)

        def as_arr(v):
            if _is_torch(v):
                return v
            if np.ndim(v) == 0:   # a scalar t0 / t1: one 8-byte device tensor per distinct value, kept on the context
                key = (str(y0.device), float(v))
                t = ctx._scalars.get(key)
                if t is None:
                    if len(ctx._scalars) > 256:
                        ctx._scalars.clear()
                    t = ctx._scalars[key] = torch.as_tensor(np.array([float(v)]), device=y0.device)
                return t
            return torch.as_tensor(np.atleast_1d(np.asarray(v, dtype=np.float64)), device=y0.device)
    else:
        y0 = np.ascontiguousarray(y0, dtype=np.float64)
        xp_zeros = lambda shape, dt: np.zeros(shape, dtype=dt)
        f64, i32, u64, u32 = np.float64, np.int32, np.uint64, np.uint32
        ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        as_arr = lambda v: np.atleast_1d(np.ascontiguousarray(v, dtype=np.float64))
        ctx = ctx or default_context(0)
    if y0.ndim != 2 or y0.shape[0] != f.n:
        raise ValueError(f"y0 must have shape [n={f.n}, B], got {tuple(y0.shape)}")
    B = int(y0.shape[1])
    n = f.n
    if f.n_params:
        if params is None:
            pv = np.repeat(np.asarray(f.params(), dtype=np.float64)[:, None], B, axis=1)
            params = as_arr(pv) if not on_device else __import__("torch").as_tensor(pv, device=y0.device)
        elif not on_device:
            params = np.ascontiguousarray(params, dtype=np.float64)
        if tuple(params.shape) != (f.n_params, B):
            raise ValueError(f"params must have shape [{f.n_params}, {B}]")
        if on_device and not params.is_contiguous():
            params = params.contiguous()
    else:
        params = None
    t0a, t1a = as_arr(t0), as_arr(t1)
    t0_len, t1_len = int(t0a.shape[0]), int(t1a.shape[0])

    keep: list = []
    copt = options._c(n, keep)
    method = options.method_enum
    per_traj = options.t_eval_per_trajectory is not None
    if per_traj:
        if len(options.t_eval_per_trajectory) != B:
            raise ValueError(f"t_eval_per_trajectory needs one grid per trajectory ({B}), got {len(options.t_eval_per_trajectory)}")
    ne = 0 if options.t_eval is None else len(options.t_eval)
    ml = int(options.max_log)
    nc = method.coeffs_per_state() * n if method != Method.RADAU else 0

    res = out or BatchSolution(
        y_end=xp_zeros((n, B), f64), t_end=xp_zeros((B,), f64), status=xp_zeros((B,), i32),
        nfev=xp_zeros((B,), u64), nstep=xp_zeros((B,), u64), naccpt=xp_zeros((B,), u64),
        nrejct=xp_zeros((B,), u64), h_next=xp_zeros((B,), f64))
    if out is not None:
        # a reused result set goes to the kernels as raw pointers: every buffer it carries must have the shape, dtype
        # and placement this call needs (a BatchSolution from another batch size / option set would be written out of bounds)
        nev_ = f.n_events()
        rows_e = max(ne + (1 if nev_ else 0), 1)
        mev_ = max(int(options.max_events), 1)
        want = dict(y_end=((n, B), f64), t_end=((B,), f64), status=((B,), i32), nfev=((B,), u64), nstep=((B,), u64),
                    naccpt=((B,), u64), nrejct=((B,), u64), h_next=((B,), f64), njev=((B,), u64), nlu=((B,), u64),
                    y_eval=((rows_e, n, B), f64), eval_idx=((rows_e, B), i32), n_filled=((B,), i32),
                    t_log=((ml, B), f64), y_log=((ml, n, B), f64), n_log=((B,), u32),
                    seg_cont=((ml, nc, B), f64), seg_xold=((ml, B), f64), seg_h=((ml, B), f64), n_seg=((B,), u32),
                    t_events=((nev_, mev_, B), f64), y_events=((nev_, mev_, n, B), f64), n_event_hits=((nev_, B), u32), t_term=((B,), f64))
        if out.log_offsets is not None:   # CSR step log: t_log [total], y_log [total, n], offsets [B+1]
            if out.t_log is None or out.y_log is None:
                raise ValueError("out.log_offsets needs out.t_log and out.y_log")
            total = int(out.t_log.shape[0])
            want["t_log"], want["y_log"] = ((total,), f64), ((total, n), f64)
            want["log_offsets"] = ((B + 1,), u64)
        if options.t_eval_per_trajectory is not None:
            # CSR sample records (time-major, one run per trajectory): y_eval [total, n], eval_idx [total] -- the same shape /
            # dtype / placement / contiguity checks as every other member (they reach the kernels as raw pointers too)
            tot_ = int(sum(len(g_) for g_ in options.t_eval_per_trajectory)) + (B if nev_ else 0)
            want["y_eval"], want["eval_idx"] = ((max(tot_, 1), n), f64), ((max(tot_, 1),), i32)
        for name, (shape, dt) in want.items():
            v = getattr(out, name)
            if v is None:
                continue
            if _is_torch(v) != on_device:
                raise ValueError(f"out.{name}: {'device' if on_device else 'host'} array expected")
            if tuple(v.shape) != tuple(shape) or v.dtype != dt:
                raise ValueError(f"out.{name}: expected shape {tuple(shape)} dtype {dt}, got {tuple(v.shape)} {v.dtype}")
            if on_device and (v.device != y0.device or not v.is_contiguous()):
                raise ValueError(f"out.{name}: must be a contiguous tensor on {y0.device}")
            if not on_device and not v.flags["C_CONTIGUOUS"]:
                raise ValueError(f"out.{name}: must be C-contiguous")
    if res.njev is None:
        res.njev = xp_zeros((B,), u64)
        res.nlu = xp_zeros((B,), u64)
    if options.t_eval is not None and res.y_eval is None:
        rows = max(ne + (1 if f.n_events() else 0), 1)     # a terminal event appends its own sample
        res.y_eval = xp_zeros((rows, n, B), f64)
        res.eval_idx = xp_zeros((rows, B), i32)
        res.n_filled = xp_zeros((B,), i32)
    if per_traj:
        extra = 1 if f.n_events() else 0                    # a terminal event appends its own sample
        sizes = np.array([len(g) for g in options.t_eval_per_trajectory], dtype=np.int64) + extra
        offs = np.zeros(B + 1, dtype=np.int64)
        offs[1:] = np.cumsum(sizes)
        total = int(offs[-1])
        if res.y_eval is None:
            res.y_eval = xp_zeros((max(total, 1), n), f64)
            res.eval_idx = xp_zeros((max(total, 1),), i32)
            res.n_filled = xp_zeros((B,), i32)
        elif tuple(res.y_eval.shape) != (max(total, 1), n):
            raise ValueError(f"out.y_eval: expected shape {(max(total, 1), n)} for these per-trajectory grids")
        res.eval_offsets = __import__("torch").as_tensor(offs, device=y0.device) if on_device else offs
    if options.count_log and res.n_log is None:
        res.n_log = xp_zeros((B,), u32)
    if options.t_eval is None and ml > 0 and res.t_log is None and res.log_offsets is None and _steplog is None:
        res.t_log = xp_zeros((ml, B), f64)
        res.y_log = xp_zeros((ml, n, B), f64)
        res.n_log = xp_zeros((B,), u32)
    if options.dense_output and ml > 0 and res.seg_cont is None:
        res.seg_cont = xp_zeros((ml, nc, B), f64)
        res.seg_xold = xp_zeros((ml, B), f64)
        res.seg_h = xp_zeros((ml, B), f64)
        res.n_seg = xp_zeros((B,), u32)

    ne_ev = f.n_events()
    if ne_ev:
        for i in range(min(ne_ev, 4)):
            cfg = f.event_config(i)
            copt.ev_direction[i] = int(cfg.direction)
            copt.ev_terminal[i] = int(cfg.terminal_count or 0)
        if ne_ev > 4:   # trait IVP::n_events is unbounded (src/ivp.rs:31-52): the configurations travel as arrays
            dirs = np.array([int(f.event_config(i).direction) for i in range(ne_ev)], dtype=np.int32)
            terms = np.array([int(f.event_config(i).terminal_count or 0) for i in range(ne_ev)], dtype=np.uint32)
            keep += [dirs, terms]
            copt.ev_direction_vec = dirs.ctypes.data_as(C.POINTER(C.c_int32))
            copt.ev_terminal_vec = terms.ctypes.data_as(C.POINTER(C.c_uint32))
            copt.n_event_cfg = ne_ev
        mev = max(int(options.max_events), 1)
        if res.t_events is None:
            res.t_events = xp_zeros((ne_ev, mev, B), f64)
            res.y_events = xp_zeros((ne_ev, mev, n, B), f64)
            res.n_event_hits = xp_zeros((ne_ev, B), u32)
            res.t_term = xp_zeros((B,), f64)
    r = _lib.BatchResultT()
    for name, _ in _lib.BatchResultT._fields_:
        setattr(r, name, ptr(getattr(res, name)))
    prob = _problem_c(f)
    if not wait:
        if not on_device:
            raise ValueError("wait=False needs device arrays (the host-pointer entry point copies results back at the end)")
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(y0.device).cuda_stream)
        rc = ctx.lib.ivp_batch_submit_device(ctx.handle, C.byref(prob), B, ptr(y0), ptr(params), ptr(t0a), t0_len,
                                             ptr(t1a), t1_len, C.byref(copt), C.byref(r), stream)
        if rc != 0:
            raise ConfigError(rc, ctx.last_error())
        return PendingBatch(ctx, res, bool(options.profile), keep + [y0, params, t0a, t1a, copt, r])
    if _steplog is not None:   # solve_ivp_batch_logged: ONE integration that also records every accepted step (page pool + gather)
        import torch
        if not on_device:
            raise ValueError("the one-pass step log takes device arrays")
        stream = C.c_void_p(torch.cuda.current_stream(y0.device).cuda_stream)
        rc = ctx.lib.ivp_batch_solve_logged_device(ctx.handle, C.byref(prob), B, ptr(y0), ptr(params), ptr(t0a), t0_len,
                                                   ptr(t1a), t1_len, C.byref(copt), C.byref(r), C.byref(_steplog), stream)
        if rc == -105:   # IVP_ERR_LOG_CAPACITY: the integration is complete, the records wait in the pool for larger buffers
            rc = 0
    elif on_device:
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(y0.device).cuda_stream)
        rc = ctx.lib.ivp_batch_solve_device(ctx.handle, C.byref(prob), B, ptr(y0), ptr(params), ptr(t0a), t0_len,
                                            ptr(t1a), t1_len, C.byref(copt), C.byref(r), stream)
    else:
        rc = ctx.lib.ivp_batch_solve(ctx.handle, C.byref(prob), B, ptr(y0), ptr(params), ptr(t0a), t0_len,
                                     ptr(t1a), t1_len, C.byref(copt), C.byref(r))
    if rc != 0:
        raise ConfigError(rc, ctx.last_error())
    if options.profile:
        res.stats = ctx.stats()
    _flag_event_overflow(res)
    return res


def solve_ivp_batch_logged(f: IVP, t0, t1, y0, params=None, options: Options = None, ctx: Context = None,
                           out: BatchSolution = None, reserve: int = 0, two_pass: bool = False) -> BatchSolution:
    """``Solution.t`` / ``Solution.y`` of B independent solves -- every accepted step of every trajectory -- in CSR form,
    from ONE integration (``ivp_batch_solve_logged_device``).

    The reference pushes one record per accepted step into growing Vecs while it integrates (src/solve/solout.rs:387-428)
    and returns them (src/solve/solve_ivp.rs:288-312).  Here the stepping kernels append the records to per-trajectory
    chains of pages drawn from a device pool; once every count is known a gather kernel lays them out in trajectory order.
    ``reserve``: expected total number of records (sizes the pool; default: what the context learnt from its last logged
    solve of this batch size, else 1024 per trajectory and at least 256 MB).  A pool that runs dry costs a second integration, never records.
    ``out``: a previous result of this function whose buffers are reused when they are large enough (one library call, no
    allocation).  ``two_pass=True`` runs the older counted form (counting solve + scan + filling solve) instead.

    Returns a BatchSolution with ``log_offsets`` [B+1], ``t_log`` [total], ``y_log`` [total, n] (time-major like the
    reference's ``Vec<Vec<f64>>``), ``n_log``, the end-state members and ``log_info``; ``log_of(b)`` slices one
    trajectory.  ``y0`` may be a numpy array (moved to the context's device) or a CUDA tensor."""
    import torch
    options = options or Options()
    if options.t_eval is not None or options.t_eval_per_trajectory is not None:
        raise ValueError("the accepted-step log is what solve_ivp records when t_eval is None")
    ctx = ctx or default_context(y0.device.index or 0 if _is_torch(y0) else 0)
    dev = y0.device if _is_torch(y0) else torch.device("cuda", ctx.device)
    to_dev = lambda a: None if a is None else (a if _is_torch(a) else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev))
    y0d, pd = to_dev(y0), to_dev(params)
    tt = lambda v: v if (np.ndim(v) == 0 and not _is_torch(v)) else to_dev(v)
    t0d, t1d = tt(t0), tt(t1)
    n, B = int(y0d.shape[0]), int(y0d.shape[1])
    base = {k: v for k, v in options.__dict__.items() if k not in ("max_log", "count_log", "profile")}
    ml = options.max_log if options.dense_output else 0
    if two_pass:
        cnt = solve_ivp_batch(f, t0d, t1d, y0d, pd, Options(**base, max_log=ml, count_log=True), ctx)
        offsets = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(cnt.n_log.to(torch.int64), 0, out=offsets[1:])
        total = int(offsets[-1].item())
        o2 = BatchSolution(y_end=cnt.y_end, t_end=cnt.t_end, status=cnt.status, nfev=cnt.nfev, nstep=cnt.nstep, naccpt=cnt.naccpt,
                           nrejct=cnt.nrejct, h_next=cnt.h_next, njev=cnt.njev, nlu=cnt.nlu, n_log=cnt.n_log,
                           t_log=torch.empty(max(total, 1), dtype=torch.float64, device=dev),
                           y_log=torch.empty((max(total, 1), n), dtype=torch.float64, device=dev), log_offsets=offsets)
        res = solve_ivp_batch(f, t0d, t1d, y0d, pd, Options(**base, max_log=ml, profile=options.profile), ctx, o2)
        res.t_log, res.y_log = res.t_log[:total], res.y_log[:total]
        res.log_info = {"passes": 2, "form": "counting solve + scan + filling solve"}
        return res

    sl = _lib.StepLogT()
    offsets = out.log_offsets if (out is not None and out.log_offsets is not None and tuple(out.log_offsets.shape) == (B + 1,)) \
        else torch.zeros(B + 1, dtype=torch.int64, device=dev)
    sl.offsets = C.c_void_p(offsets.data_ptr())
    sl.reserve = int(reserve)
    bufs = getattr(out, "_log_buffers", None) if out is not None else None
    if bufs is not None and (bufs[0].device != dev or bufs[1].shape[1] != n):
        bufs = None
    if bufs is not None:
        sl.t, sl.y, sl.capacity = C.c_void_p(bufs[0].data_ptr()), C.c_void_p(bufs[1].data_ptr()), int(bufs[0].shape[0])
    else:
        sl.defer = 1     # integrate and count; the records are fetched into buffers of exactly `total` records below
    if out is not None:
        keep = ("y_end", "t_end", "status", "nfev", "nstep", "naccpt", "nrejct", "h_next", "njev", "nlu", "n_log", "seg_cont", "seg_xold",
                "seg_h", "n_seg", "t_events", "y_events", "n_event_hits", "t_term")
        o1 = BatchSolution(**{k: getattr(out, k) for k in keep})
    else:
        o1 = BatchSolution(y_end=torch.zeros((n, B), dtype=torch.float64, device=dev), t_end=torch.zeros(B, dtype=torch.float64, device=dev),
                           status=torch.zeros(B, dtype=torch.int32, device=dev), nfev=torch.zeros(B, dtype=torch.int64, device=dev),
                           nstep=torch.zeros(B, dtype=torch.int64, device=dev), naccpt=torch.zeros(B, dtype=torch.int64, device=dev),
                           nrejct=torch.zeros(B, dtype=torch.int64, device=dev), h_next=torch.zeros(B, dtype=torch.float64, device=dev))
    if o1.n_log is None:
        o1.n_log = torch.zeros(B, dtype=torch.int32, device=dev)
    opts1 = Options(**base, max_log=ml, profile=options.profile)
    res = solve_ivp_batch(f, t0d, t1d, y0d, pd, opts1, ctx, o1, _steplog=sl)
    total = int(sl.total)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if bufs is None or sl.capacity < total:
        # first call / grown log: buffers of exactly `total` records, filled from the pool (no second integration)
        bufs = (torch.empty(max(total, 1), dtype=torch.float64, device=dev), torch.empty((max(total, 1), n), dtype=torch.float64, device=dev))
        sl.t, sl.y, sl.capacity, sl.defer = C.c_void_p(bufs[0].data_ptr()), C.c_void_p(bufs[1].data_ptr()), max(total, 1), 0
        rc = ctx.lib.ivp_step_log_fetch_device(ctx.handle, C.byref(sl), stream)
        if rc != 0:   # the pool had run dry: integrate again -- its size now follows the counted total
            sl.reserve = 0   # automatic: what the first attempt asked of its fullest sub-pool
            res = solve_ivp_batch(f, t0d, t1d, y0d, pd, opts1, ctx, o1, _steplog=sl)
            sl.passes += 1
    res.log_offsets = offsets
    res._log_buffers = bufs
    res.t_log, res.y_log = bufs[0][:total], bufs[1][:total]
    res.log_info = {"passes": int(sl.passes), "records": total, "page_slots": int(sl.page_slots), "pool_bytes": int(sl.pool_bytes),
                    "pool_used_bytes": int(sl.pool_used_bytes), "form": "page pool + gather (one integration)" if sl.passes == 1 else "the pool ran dry: second integration"}
    return res


def solve_ivp(f: IVP, x0: float, xend: float, y0: Sequence[float], options: Options = None,
              ctx: Context = None) -> Solution:
    """``solve_ivp(&f, x0, xend, &y0, options) -> Result<Solution, Error>`` (solve_ivp.rs:99-108) for
    one trajectory, executed by the GPU kernels (a batch of one).  ``Err(..)`` becomes an exception."""
    options = options or Options()
    method = options.method_enum
    y0 = np.asarray(y0, dtype=np.float64)
    n = y0.size
    n_events = f.n_events()
    if n == 0 or abs(xend - x0) < 1e-15:
        # Degenerate cases never reach an integrator in the reference either (solve_ivp.rs:110-176):
        # pure host bookkeeping, no arithmetic of the path.
        if abs(xend - x0) < 1e-15:
            if options.t_eval is not None:
                t = np.array([te for te in options.t_eval if abs(te - x0) < 1e-12], dtype=np.float64)
            else:
                t = np.array([x0])
        else:
            t = np.asarray(options.t_eval, dtype=np.float64) if options.t_eval is not None else np.array([x0, xend])
        y = np.repeat(y0[None, :], len(t), axis=0) if n else np.zeros((len(t), 0))
        cs = ContinuousOutput.constant(method, x0, y0) if options.dense_output else None
        return Solution(t=t, y=y, t_events=[[] for _ in range(n_events)], y_events=[[] for _ in range(n_events)],
                        nfev=0, njev=0, nlu=0, nstep=0, naccpt=0, nrejct=0, status=Status.Success, continuous_sol=cs)
    if n != f.n:
        raise ValueError(f"y0 has {n} components, problem has {f.n}")
    need_log = options.t_eval is None or options.dense_output
    cap = options.max_log or (4096 if need_log else 0)
    ev_cap = max(int(options.max_events), 1)
    import warnings
    while True:
        # Solution.t / .y / .t_events / .y_events are unbounded Vecs in the reference (solout.rs:158-331,387-428):
        # a buffer that turned out too small means a rerun with room for everything, never a truncated result
        o = Options(**{**options.__dict__, "max_log": cap, "max_events": ev_cap})
        pr = np.asarray(f.params(), dtype=np.float64).reshape(f.n_params, 1) if f.n_params else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            r = solve_ivp_batch(f, x0, xend, y0.reshape(n, 1), pr, o, ctx)
        used = 0
        if r.n_log is not None:
            used = max(used, int(r.n_log[0]))
        if r.n_seg is not None:
            used = max(used, int(r.n_seg[0]))
        hits = int(np.asarray(r.n_event_hits).max()) if (n_events and r.n_event_hits is not None) else 0
        log_ok = used <= cap or not need_log
        if log_ok and hits <= ev_cap:
            break
        if not log_ok:
            cap = int(used * 1.25) + 16  # the log overflowed: rerun with room for every accepted step
        if hits > ev_cap:
            ev_cap = int(hits * 1.25) + 8
    if options.t_eval is not None:
        m = int(r.n_filled[0])
        te = np.asarray(options.t_eval, dtype=np.float64)
        idx = r.eval_idx[:m, 0]
        t = np.where(idx >= 0, te[np.maximum(idx, 0)], r.t_term[0] if r.t_term is not None else np.nan) if m else np.zeros(0)
        y = r.y_eval[:m, :, 0].copy()
    else:
        m = int(r.n_log[0])
        t = r.t_log[:m, 0].copy()
        y = r.y_log[:m, :, 0].copy()
    cs = None
    if options.dense_output:
        ns = int(r.n_seg[0])
        cs = ContinuousOutput(method, n, r.seg_cont[:ns, :, 0].copy(), r.seg_xold[:ns, 0].copy(), r.seg_h[:ns, 0].copy())
    t_events, y_events = [], []
    for i in range(n_events):
        k = min(int(r.n_event_hits[i, 0]), r.t_events.shape[1])
        t_events.append(r.t_events[i, :k, 0].copy())
        y_events.append(r.y_events[i, :k, :, 0].copy())
    return Solution(t=t, y=y, t_events=t_events, y_events=y_events,
                    nfev=int(r.nfev[0]), njev=int(r.njev[0]), nlu=int(r.nlu[0]), nstep=int(r.nstep[0]), naccpt=int(r.naccpt[0]),
                    nrejct=int(r.nrejct[0]), status=Status(int(r.status[0])), continuous_sol=cs,
                    h_next=float(r.h_next[0]))
