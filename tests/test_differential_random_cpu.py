"""Seeded randomised differential test: the kernel bodies of rk_core.h / bdf_core.h (run on the CPU by tests/host_emul
with the GPU launch schedule) against the oracle, over random combinations of method, problem, interval direction,
tolerances (scalar or per component), first_step / max_step / max_steps, controller settings, output mode (end state,
t_eval incl. points outside the span and repeated points, step log + dense segments) and chunk length.
Everything must agree bit for bit.  The same generator drives a GPU run in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.common import assert_bitexact, emul_batch, oracle_batch

PROBLEMS = {   # name -> (n, params or None, y0 sampler, (t_lo, t_hi) window the problem is well behaved in)
    "sho": (2, None, lambda r: r.standard_normal(2), (-3.0, 3.0)),
    "vdp": (2, lambda r: [r.uniform(0.5, 3.0)], lambda r: r.uniform(-2, 2, 2), (0.0, 6.0)),
    "lorenz": (3, lambda r: [10.0, 28.0 * r.uniform(0.8, 1.1), 8.0 / 3.0], lambda r: 1.0 + r.standard_normal(3), (0.0, 2.0)),
    "linear": (2, None, lambda r: r.standard_normal(2), (-2.0, 2.0)),
    "rational": (2, None, lambda r: np.array([1 / 3, 2 / 9]) * (1 + 1e-2 * r.standard_normal(2)), (4.0, 9.0)),
    "decay": (1, lambda r: [r.uniform(0.1, 3.0)], lambda r: r.uniform(0.5, 2.0, 1), (0.0, 4.0)),
}
METHODS = ["RK23", "DOPRI5", "DOP853", "RK4", "BDF"]


def random_case(seed):
    r = np.random.default_rng(seed)
    method = METHODS[r.integers(len(METHODS))]
    rhs = list(PROBLEMS)[r.integers(len(PROBLEMS))]
    n, par, y0f, (lo, hi) = PROBLEMS[rhs]
    a, b = sorted(r.uniform(lo, hi, 2))
    if b - a < 0.05:
        b = a + 0.05
    t0, t1 = (a, b) if r.random() < 0.6 else (b, a)
    if rhs == "lorenz" or (rhs == "vdp" and method in ("RK23", "BDF")):
        # backward these are violently unstable: the reference's RK23 never returns from the finite-time blow-up
        # (deviation 1), BDF needs ~1e8 steps and backward Lorenz 1e5-1e6 steps with any method (the explicit
        # blow-up cases of tests/cases.py cover that regime); backward Van der Pol with DOPRI5 / DOP853 stays in
        t0, t1 = a, b
    if r.random() < 0.05:
        t1 = t0                                   # zero-length interval
    o = dict(method=method)
    if method != "RK4":
        e = r.integers(3, 9)
        if r.random() < 0.25:                     # Tolerance::Vector
            o["rtol"] = list(10.0 ** -r.uniform(e - 1, e + 1, n))
            o["atol"] = float(10.0 ** -(e + 3)) if r.random() < 0.5 else list(10.0 ** -r.uniform(e + 2, e + 4, n))
        else:
            o["rtol"], o["atol"] = float(10.0 ** -e), float(10.0 ** -(e + 3))
    span = abs(t1 - t0)
    if method == "RK4":
        if r.random() < 0.5 and span > 0:
            o["first_step"] = float(np.sign(t1 - t0) * span / r.integers(20, 200))
    else:
        if r.random() < 0.3:
            o["first_step"] = float(span * r.uniform(0.001, 0.3)) if span else 0.1
        if r.random() < 0.3:
            o["max_step"] = float((span if span else 1.0) * r.uniform(0.02, 0.5) * (1 if r.random() < 0.8 else -1))
    if r.random() < 0.15:
        o["max_steps"] = int(r.integers(1, 60))
    if method in ("RK23", "DOPRI5", "DOP853") and r.random() < 0.25:
        st = {}
        if r.random() < 0.7:
            st["safety_factor"] = float(r.uniform(0.5, 0.95))
        if method != "RK23" and r.random() < 0.5:
            st["beta"] = float(r.uniform(0.0, 0.15))
        if r.random() < 0.5:
            st["scale_max"] = float(r.uniform(3.0, 12.0))
        if method != "RK23" and r.random() < 0.3:
            st["stiff_test"] = int(r.integers(1, 40))
        o["settings"] = st
    mode = r.integers(3)
    extra = {}
    if mode == 1:                                 # t_eval: sorted along the integration direction, some outside, some repeated
        k = int(r.integers(1, 12))
        te = r.uniform(min(t0, t1) - 0.2 * span - 0.01, max(t0, t1) + 0.2 * span + 0.01, k)
        if r.random() < 0.5:
            te = np.concatenate([te, [t0, t1]])
        if r.random() < 0.3:
            te = np.concatenate([te, te[:2]])
        te = np.sort(te)
        o["t_eval"] = te if t1 >= t0 else te[::-1].copy()
    elif mode == 2:
        o["dense_output"] = bool(r.random() < 0.6)
        extra["max_log"] = int(r.integers(4, 400))
    B = int(r.integers(1, 9))
    y0 = np.stack([y0f(r) for _ in range(B)], axis=1)
    p = None if par is None else np.array([par(r) for _ in range(B)]).T.copy()
    chunk = int(r.choice([1, 3, 17, 64, 4096]))
    return rhs, y0, p, float(t0), float(t1), o, extra, chunk


def _same(a, b):
    """Bit-for-bit equality of two float arrays (NaN == NaN)."""
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all())


def compare(solve, seed, fma=False):
    """`fma`: the kernels' FMA arithmetic mode (solve must then run it: fast=True) against liboracle_fma.so."""
    rhs, y0, p, t0, t1, o, extra, chunk = random_case(seed)
    tag = f"seed {seed}{' [fma]' if fma else ''}: {rhs} {o} chunk {chunk}: "
    g = solve(rhs, y0, p, t0, t1, chunk=chunk, **o, **extra)
    ref = oracle_batch(rhs, y0, p, t0, t1, fma=fma, **o)       # end states and counters of the whole batch
    assert_bitexact(g, ref, tag)
    B = y0.shape[1]
    for b in range(B):
        s = O.solve_ivp(rhs, t0, t1, y0[:, b], params=() if p is None else tuple(p[:, b]), detpow=True, fma=fma, **o)
        if "t_eval" in o:
            m = int(g["n_filled"][b])
            assert m == len(s.t), tag
            assert _same(np.asarray(o["t_eval"])[g["eval_idx"][:m, b]], s.t), tag
            assert _same(g["y_eval"][:m, :, b], s.y), tag
        elif "max_log" in extra:
            cap = extra["max_log"]
            m = int(g["n_log"][b])
            assert m == len(s.t), tag
            k = min(m, cap)
            assert _same(g["t_log"][:k, b], s.t[:k]) and _same(g["y_log"][:k, :, b], s.y[:k]), tag
            if o.get("dense_output"):
                ns = int(g["n_seg"][b])
                nref = 0 if s.seg_h is None else len(s.seg_h)     # no accepted step => no segment
                assert ns == nref, tag
                k = min(ns, cap)
                if k == 0:
                    continue
                assert _same(g["seg_xold"][:k, b], s.seg_xold[:k]) and _same(g["seg_h"][:k, b], s.seg_h[:k]), tag
                assert _same(g["seg_cont"][:k, :, b], s.seg_cont[:k]), tag


@pytest.mark.parametrize("block", range(12))
def test_random_configurations_bitexact(block):
    for seed in range(1000 + 25 * block, 1000 + 25 * (block + 1)):
        compare(emul_batch, seed)


@pytest.mark.parametrize("block", range(6))
def test_random_configurations_bitexact_in_fma_mode(block):
    """The same random option combinations in the FMA arithmetic mode: host-compiled kernel bodies (-DIVP_FAST=1) vs
    liboracle_fma.so, bit for bit."""
    for seed in range(5000 + 25 * block, 5000 + 25 * (block + 1)):
        compare(lambda *a, **kw: emul_batch(*a, fast=True, **kw), seed, fma=True)
