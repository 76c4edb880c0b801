"""ivp_amd -- MI355X-native batched explicit Runge-Kutta IVP integrator.

Drop-in for the explicit-RK stepping path of the Rust crate Ryan-D-Gast/ivp (DOPRI5 / DOP853 / RK23
behind ``solve_ivp``): hand-written HIP kernels for gfx950 behind a C ABI (include/ivp_hip.h), this
package being the host-side mirror of the reference's ``solve_ivp`` / ``IVP`` / ``Options`` surface.
"""
from .api import (  # noqa: F401
    BUILTIN, BouncingBall, Cannon, CR3BP, Direction, EventConfig, RationalEvents, SHOZeroEvent, BatchSolution, ConfigError, Context, ContinuousOutput, DeviceIVP, Exp2, ExponentialDecay,
    Dense64, FpMode, Heat1D256, LinearDecay100, InterpolationError, IVP, IvpError, LinearSystem, Lorenz, Method, Options, PendingBatch, Rational, Robertson, RobertsonJac, SHO, Solution, Status, StiffVanDerPol,
    VanDerPol, ZeroRhs, default_context, solve_ivp, solve_ivp_batch, solve_ivp_batch_logged,
)
from . import pyfront, workloads  # noqa: F401

__all__ = [
    "BUILTIN", "BouncingBall", "Cannon", "CR3BP", "Direction", "EventConfig", "RationalEvents", "SHOZeroEvent", "BatchSolution", "ConfigError", "Context", "ContinuousOutput", "DeviceIVP", "Exp2",
    "ExponentialDecay", "Dense64", "FpMode", "Heat1D256", "LinearDecay100", "InterpolationError", "IVP", "IvpError", "LinearSystem", "Lorenz", "Method", "Options", "PendingBatch",
    "Rational", "Robertson", "RobertsonJac", "SHO", "Solution", "Status", "StiffVanDerPol", "VanDerPol", "ZeroRhs", "default_context", "solve_ivp",
    "solve_ivp_batch", "solve_ivp_batch_logged", "pyfront", "workloads",
]
