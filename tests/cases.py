"""Parity cases shared by the CPU (emulated kernel bodies) and GPU (C ABI) test files.

Each case is (name, rhs, builder) where builder() -> (y0[n,B], params[p,B] or None, t0, t1, options).
Options use the reference's field names (src/solve/options.rs:75-123).
"""
import numpy as np

from ivp_amd import workloads as W


def _rep(v, B):
    return np.repeat(np.asarray(v, dtype=np.float64)[:, None], B, axis=1)


def c2_cr3bp(B=256, method="DOPRI5", rtol=1e-6, atol=1e-9):
    def b():
        y0, p, t0, t1 = W.cr3bp_batch(B)
        return y0, p, t0, t1, dict(method=method, rtol=rtol, atol=atol)
    return b


def c3_vdp(B=256, method="DOP853", rtol=1e-8, atol=1e-10):
    def b():
        y0, p, t0, t1 = W.vdp_batch(B)
        return y0, p, t0, t1, dict(method=method, rtol=rtol, atol=atol)
    return b


def c1_decay():
    return np.array([[1.0]]), np.array([[0.5]]), 0.0, 10.0, dict(method="DOPRI5", rtol=1e-6, atol=1e-9)


def sho(method, t0=0.0, t1=2 * np.pi, B=70, **kw):
    def b():
        rng = np.random.default_rng(5)
        y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
        y0[:, 0] = [1.0, 0.0]
        o = dict(method=method, rtol=1e-9, atol=1e-9)
        o.update(kw)
        return y0, None, t0, t1, o
    return b


def lorenz(method):
    def b():
        B = 65
        rng = np.random.default_rng(11)
        y0 = 1.0 + 0.1 * rng.standard_normal((3, B))
        p = _rep([10.0, 28.0, 8.0 / 3.0], B) * (1.0 + 0.01 * rng.standard_normal((3, B)))
        return y0, p, 0.0, 3.0, dict(method=method, rtol=1e-8, atol=1e-10)
    return b


def rational(method, t1):
    def b():
        B = 33
        y0 = _rep([1 / 3, 2 / 9], B)
        y0 *= 1.0 + 1e-3 * np.arange(B) / B
        return y0, None, 5.0, t1, dict(method=method, rtol=1e-3, atol=1e-6)
    return b


def zero_rhs(method):
    def b():
        te = np.array([10.0 * i / 20.0 for i in range(21)])
        return np.ones((3, 5)), None, 0.0, 10.0, dict(method=method, rtol=1e-9, atol=1e-12, t_eval=te)
    return b


def exp2_vector_rtol():
    return np.ones((2, 9)), None, 0.0, 1.0, dict(method="DOPRI5", rtol=[1e-2, 1e-10], atol=1e-10)


def mixed_intervals(method):
    """Per-trajectory t0/t1 including zero-length and backward intervals in one batch."""
    def b():
        B = 96
        rng = np.random.default_rng(3)
        y0 = rng.standard_normal((2, B))
        t0 = rng.uniform(-1, 1, B)
        t1 = t0 + rng.uniform(-3, 3, B)
        t1[::7] = t0[::7]                # zero interval: solve_ivp.rs:110-145
        t1[3::11] = t0[3::11] + 1e-16    # |xend-x0| < 1e-15
        return y0, None, t0, t1, dict(method=method, rtol=1e-6, atol=1e-8)
    return b


def blowup_vdp_backward(method):
    """Van der Pol integrated backwards blows up in finite time: DOPRI5/DOP853 must end with StepSizeTooSmall
    (dopri5.rs:274-277).  (RK23 has no such test in the reference and never terminates there.)"""
    def b():
        B = 24
        rng = np.random.default_rng(3)
        y0 = rng.standard_normal((2, B))
        t0 = rng.uniform(-1, 1, B)
        return y0, np.full((1, B), 1.0), t0, t0 - 3.0, dict(method=method, rtol=1e-6, atol=1e-8)
    return b


def stiff_vdp(method):
    """Large-mu Van der Pol with an explicit method: exercises the stiffness detector (ProbablyStiff)."""
    def b():
        B = 8
        y0 = _rep([2.0, 0.0], B)
        p = np.array([[1.0, 5.0, 50.0, 200.0, 500.0, 1000.0, 2000.0, 5000.0]])
        return y0, p, 0.0, 40.0, dict(method=method, rtol=1e-4, atol=1e-6)
    return b


def long_sho(method):
    """> 1000 accepted steps per trajectory: crosses the `accepted % 1000 == 0` stiffness-test cadence."""
    def b():
        B = 6
        y0 = _rep([1.0, 0.0], B)
        y0[0] += 0.01 * np.arange(B)
        rt = 1e-10 if method != "RK23" else 1e-6
        return y0, None, 0.0, 60.0, dict(method=method, rtol=rt, atol=rt)
    return b


CASES = []
for m, rt, at in (("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)):
    CASES.append((f"C2-cr3bp-{m}", "cr3bp", c2_cr3bp(256, m, rt, at)))
    CASES.append((f"C3-vdp-{m}", "vdp", c3_vdp(256, m, rt, at)))
CASES.append(("C1-decay", "decay", c1_decay))
for m in ("RK23", "DOPRI5", "DOP853"):
    CASES.append((f"sho-fwd-{m}", "sho", sho(m)))
    CASES.append((f"sho-bwd-{m}", "sho", sho(m, 2 * np.pi, 0.0)))
    CASES.append((f"sho-maxstep-{m}", "sho", sho(m, 0.0, 3.0, rtol=1e-6, atol=1e-9, max_step=0.05)))
    CASES.append((f"sho-negmaxstep-{m}", "sho", sho(m, 0.0, 3.0, rtol=1e-6, atol=1e-9, max_step=-0.05)))
    CASES.append((f"sho-firststep-{m}", "sho", sho(m, 0.0, 3.0, rtol=1e-3, atol=1e-6, first_step=0.1)))
    CASES.append((f"sho-negfirststep-bwd-{m}", "sho", sho(m, 3.0, 0.0, rtol=1e-3, atol=1e-6, first_step=0.1)))
    CASES.append((f"sho-maxsteps1-{m}", "sho", sho(m, 0.0, 3.0, max_steps=1)))
    CASES.append((f"sho-maxsteps40-{m}", "sho", sho(m, 0.0, 30.0, max_steps=40)))
    CASES.append((f"lorenz-{m}", "lorenz", lorenz(m)))
    CASES.append((f"rational-fwd-{m}", "rational", rational(m, 9.0)))
    CASES.append((f"rational-bwd-{m}", "rational", rational(m, 1.0)))
    CASES.append((f"zero-{m}", "zero", zero_rhs(m)))
    CASES.append((f"mixed-intervals-{m}", "sho", mixed_intervals(m)))
    CASES.append((f"long-sho-{m}", "sho", long_sho(m)))
for m in ("DOPRI5", "DOP853"):
    CASES.append((f"stiff-vdp-{m}", "vdp", stiff_vdp(m)))
    CASES.append((f"blowup-vdp-bwd-{m}", "vdp", blowup_vdp_backward(m)))
CASES.append(("exp2-vector-rtol", "exp2", exp2_vector_rtol))
# fixed-step RK4 (rk4.rs): default h = (xend - x0)/100, explicit first_step, step cap, backward
CASES.append(("rk4-cr3bp-default-h", "cr3bp", lambda: (*W.cr3bp_batch(64)[:3], 2.0, dict(method="RK4"))))
CASES.append(("rk4-sho-h", "sho", sho("RK4", 0.0, 2 * np.pi, first_step=2 * np.pi / 2000)))
CASES.append(("rk4-sho-bwd", "sho", sho("RK4", 3.0, 0.0, first_step=-0.01)))
CASES.append(("rk4-sho-maxsteps", "sho", sho("RK4", 0.0, 3.0, max_steps=30)))
CASES.append(("rk4-mixed-intervals", "sho", mixed_intervals("RK4")))
CASES.append(("rk4-zero-teval", "zero", zero_rhs("RK4")))

# BDF 1..5 (bdf.rs): the "next" row, incl. BASELINE C5's stiff Van der Pol (mu = 1000)
def bdf_vdp_stiff(B=24):
    def b():
        rng = np.random.default_rng(8)
        y0 = _rep([2.0, 0.0], B) * (1.0 + 0.05 * rng.standard_normal((2, B)))
        y0[:, 0] = [2.0, 0.0]
        p = np.full((1, B), 1000.0) * (1.0 + 0.1 * rng.uniform(-1, 1, (1, B)))
        p[0, 0] = 1000.0
        return y0, p, 0.0, 3000.0, dict(method="BDF", rtol=1e-4, atol=1e-6)   # benches/benchmark.py:118-126
    return b


def bdf_robertson():
    B = 6
    y0 = _rep([1e4, 0.0, 0.0], B)
    y0[0] *= 1.0 + 0.01 * np.arange(B)
    return y0, None, 0.0, 1e8, dict(method="BDF", rtol=1e-6, atol=1e-6)


CASES.append(("bdf-C5-vdp-mu1000", "vdp", bdf_vdp_stiff()))
CASES.append(("bdf-robertson", "robertson", bdf_robertson))
CASES.append(("bdf-vdp-eps", "vdp_eps", lambda: (_rep([2.0, 0.0], 5), np.array([[1e-3, 2e-3, 5e-3, 1e-2, 1e-1]]), 0.0, 2.0,
                                                dict(method="BDF", rtol=1e-6, atol=1e-8))))
CASES.append(("bdf-linear", "linear", lambda: (_rep([0.0, 2.0], 3), None, 0.0, 2.0, dict(method="BDF", rtol=1e-3, atol=1e-6))))
CASES.append(("bdf-sho-fwd", "sho", sho("BDF", B=20)))
CASES.append(("bdf-sho-bwd", "sho", sho("BDF", 2 * np.pi, 0.0, B=20, rtol=1e-6, atol=1e-9)))
CASES.append(("bdf-sho-maxstep-minstep", "sho", sho("BDF", 0.0, 3.0, B=20, rtol=1e-6, atol=1e-9, max_step=0.05, min_step=1e-4)))
CASES.append(("bdf-rational-firststep", "rational", lambda: (*rational("BDF", 9.0)()[:4], dict(method="BDF", rtol=1e-3, atol=1e-6, max_step=0.5, first_step=0.1))))
CASES.append(("bdf-sho-maxsteps", "sho", sho("BDF", 0.0, 30.0, B=20, max_steps=40)))
CASES.append(("bdf-mixed-intervals", "sho", mixed_intervals("BDF")))
CASES.append(("bdf-cr3bp-short", "cr3bp", lambda: (*W.cr3bp_batch(16)[:3], 2.0, dict(method="BDF", rtol=1e-6, atol=1e-9))))
CASES.append(("bdf-zero-rhs", "zero", lambda: (np.ones((3, 2)), None, 0.0, 10.0, dict(method="BDF", rtol=1e-6, atol=1e-9, max_steps=5000))))

# direct per-method calls with non-default struct fields (dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37)
CASES.append(("settings-dopri5-sho", "sho", sho("DOPRI5", 0.0, 20.0, rtol=1e-6, atol=1e-9,
                                                 settings=dict(safety_factor=0.8, beta=0.0, scale_min=0.5, scale_max=4.0))))
CASES.append(("settings-dopri5-beta-cr3bp", "cr3bp", lambda: (*W.cr3bp_batch(64), dict(method="DOPRI5", rtol=1e-6, atol=1e-9,
                                                                                   settings=dict(beta=0.1, safety_factor=0.95)))))
CASES.append(("settings-dop853-vdp", "vdp", lambda: (*W.vdp_batch(64), dict(method="DOP853", rtol=1e-8, atol=1e-10,
                                                                         settings=dict(beta=0.08, safety_factor=0.7, scale_max=3.0)))))
CASES.append(("settings-rk23-sho", "sho", sho("RK23", 0.0, 10.0, rtol=1e-5, atol=1e-8,
                                               settings=dict(safety_factor=0.5, scale_min=0.3, scale_max=3.0))))
for _m in ("DOPRI5", "DOP853"):
    CASES.append((f"settings-stifftest7-{_m}", "sho", (lambda mm: lambda: (*long_sho(mm)()[:4], dict(method=mm, rtol=1e-10, atol=1e-10, settings=dict(stiff_test=7))))(_m)))
    CASES.append((f"settings-stifftest1-stiff-vdp-{_m}", "vdp", (lambda mm: lambda: (*stiff_vdp(mm)()[:4], dict(method=mm, rtol=1e-4, atol=1e-6, settings=dict(stiff_test=1))))(_m)))
    CASES.append((f"settings-stifftest-huge-{_m}", "sho", (lambda mm: lambda: (*long_sho(mm)()[:4], dict(method=mm, rtol=1e-10, atol=1e-10, settings=dict(stiff_test=2 ** 20 + 3))))(_m)))
    CASES.append((f"settings-uround-{_m}", "sho", sho(_m, 50.0, 80.0, rtol=1e-12, atol=1e-14, settings=dict(uround=1e-4))))
CASES.append(("settings-maxsteps-default-rk23", "sho", sho("RK23", 0.0, 2000.0, B=3, rtol=1e-7, atol=1e-9, settings={})))

CASE_IDS = [c[0] for c in CASES]


EVENT_CASES = []
for _m in ("RK23", "DOPRI5", "DOP853", "RK4", "BDF"):
    _kw = dict(method=_m) if _m == "RK4" else dict(method=_m, rtol=1e-9 if _m != "BDF" else 1e-6, atol=1e-9)
    EVENT_CASES += [
        (f"sho-all-term2-{_m}", "sho_ev", 0.0, 6.0, [1.0, 0.0], (), dict(event_direction=[0], event_terminal=[2], **_kw)),
        (f"sho-pos-{_m}", "sho_ev", 0.0, 6.0, [1.0, 0.0], (), dict(event_direction=[1], event_terminal=[0], **_kw)),
        (f"sho-neg-term-bwd-{_m}", "sho_ev", 6.0, 0.0, [1.0, 0.0], (), dict(event_direction=[-1], event_terminal=[1], **_kw)),
        (f"sho-teval-term-{_m}", "sho_ev", 0.0, 6.0, [1.0, 0.0], (),
         dict(event_direction=[0], event_terminal=[2], t_eval=np.linspace(0, 6, 13), **_kw)),
    ]
    if _m != "RK4":
        EVENT_CASES += [
            (f"rational-3ev-term-{_m}", "rational_ev", 5.0, 8.0, [1 / 3, 2 / 9], (),
             dict(method=_m, event_direction=[0, 0, 0], event_terminal=[0, 0, 1], dense_output=True)),
            (f"rational-3ev-bwd-{_m}", "rational_ev", 8.0, 5.0, [4 / 9, 20 / 81], (),
             dict(method=_m, event_direction=[0, 0, 0], event_terminal=[0, 0, 0])),
        ]
EVENT_CASES += [
    ("ball", "ball", 0.0, 10.0, [10.0, 5.0], (9.81, 0.02),
     dict(method="DOPRI5", rtol=1e-8, atol=1e-10, event_direction=[-1], event_terminal=[1])),
    ("cannon", "cannon", 0.0, np.inf, [0.0, 0.01], (),
     dict(method="DOPRI5", max_step=0.05 * 0.001 / 9.80665, event_direction=[-1], event_terminal=[1], dense_output=True)),
]


def check_events_against_oracle(solve, case, exact=True, fma=False):
    """Events, outputs and statistics of a 3-trajectory batch vs one oracle solve_ivp call per trajectory."""
    from oracle import oracle as O
    name, rhs, t0, t1, y0, params, kw = case
    y0a = np.asarray(y0, float).reshape(-1, 1).repeat(3, axis=1)
    y0a[:, 1] *= 1.001
    y0a[:, 2] *= 0.999
    par = np.asarray(params, float).reshape(-1, 1).repeat(3, axis=1) if len(params) else None
    gkw = dict(kw)
    if "t_eval" not in gkw:
        gkw["max_log"] = 4096
    g = solve(rhs, y0a, par, t0, t1, **gkw)
    eq = np.array_equal if exact else (lambda a, b: np.allclose(a, b, rtol=1e-9, atol=1e-11))
    for b in range(3):
        s = O.solve_ivp(rhs, t0, t1, y0a[:, b], params=params, detpow=True, fma=fma, **kw)
        for i in range(len(s.t_events)):
            m = int(g["n_ev"][i, b])
            assert m == len(s.t_events[i]), (name, b, i)
            assert eq(g["t_events"][i, :m, b], s.t_events[i]) and eq(g["y_events"][i, :m, :, b], s.y_events[i]), (name, b, i)
        assert int(g["status"][b]) == s.status and int(g["nfev"][b]) == s.nfev and g["h_next"][b] == s.h_next or not exact
        if "t_eval" in kw:
            m = g["n_filled"][b]
            assert m == len(s.t) and eq(g["y_eval"][:m, :, b], s.y)
            if s.status == 1:
                assert eq(g["t_term"][b], s.t[-1]) and g["eval_idx"][m - 1, b] == -1
        else:
            m = g["n_log"][b]
            assert m == len(s.t) and eq(g["t_log"][:m, b], s.t) and eq(g["y_log"][:m, :, b], s.y)
