"""GPU parity tests proper: libivp_hip.so (through the C ABI) against the CPU oracle.  Run with -m gpu."""
import numpy as np
import pytest

from ivp_amd import workloads as W
from tests.common import assert_bitexact, gpu_batch, oracle_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method,rtol,atol", [("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)])
def test_cr3bp_strict_bitexact_vs_oracle(method, rtol, atol):
    """Strict-FP kernels reproduce the CPU restatement bit for bit (y, t, h, every counter) when both use
    the same portable step-controller pow: proves IEEE-correct f64 div/sqrt on gfx950 and that the
    kernel is the same operation sequence as the reference algorithm."""
    y0, p, t0, t1 = W.cr3bp_batch(2048)
    g = gpu_batch("cr3bp", y0, p, t0, t1, method=method, rtol=rtol, atol=atol)
    o = oracle_batch("cr3bp", y0, p, t0, t1, method=method, rtol=rtol, atol=atol, threads=8)
    assert (g["status"] == 0).all()
    assert_bitexact(g, o, f"{method}: ")
