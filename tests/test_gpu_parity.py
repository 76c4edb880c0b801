"""GPU parity tests proper: libivp_hip.so, called through the C ABI, against the CPU oracle.  Run with -m gpu.

Bars (BASELINE.json north_star: "results match the reference CPU solve_ivp ... within a stated floating-point
tolerance"):
  * strict FP mode: BIT-EXACT equality with the oracle's portable-pow build for y_end, t_end, h_next and every
    counter (the GPU evaluates the same IEEE-754 operation sequence);
  * against the oracle's libm-pow build (what the Rust crate calls): differences come only from a few-ulp step
    factor -- tolerance 1e-9 absolute on O(1) states over non-chaotic horizons, stated in each test;
  * fast FP mode: same tolerance, and end-state error vs an independent truth within 10x of the oracle's.
"""
import json
import os

import numpy as np
import pytest

from ivp_amd import workloads as W
from oracle import oracle as O
from tests.cases import CASES, CASE_IDS, EVENT_CASES, c2_cr3bp, c3_vdp, check_events_against_oracle
from tests.common import assert_bitexact, gpu_batch, oracle_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_strict_gpu_bitexact_vs_oracle(case):
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    g = gpu_batch(rhs, y0, p, t0, t1, chunk=23, **o)
    r = oracle_batch(rhs, y0, p, t0, t1, threads=8, **o)
    assert_bitexact(g, r, name + ": ")
    if "t_eval" in o:
        assert np.array_equal(g["n_filled"], r["n_filled"])
        m = g["n_filled"]
        for b in range(y0.shape[1]):
            assert np.array_equal(g["y_eval"][: m[b], :, b], r["y_eval"][: m[b], :, b])


@pytest.mark.parametrize("chunk", [1, 7, 64, 4096])
@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853"])
def test_chunk_size_and_compaction_do_not_change_results(method, chunk):
    y0, p, t0, t1, o = c3_vdp(300, method)()
    a = gpu_batch("vdp", y0, p, t0, t1, chunk=chunk, **o)
    b = oracle_batch("vdp", y0, p, t0, t1, threads=8, **o)
    assert_bitexact(a, b)


def test_device_pointer_entry_point_matches_host_entry_point():
    y0, p, t0, t1, o = c2_cr3bp(1000)()
    a = gpu_batch("cr3bp", y0, p, t0, t1, **o)
    b = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, **o)
    assert_bitexact(a, b)


def test_nan_and_inf_lanes_retire_without_poisoning_neighbours():
    B = 200
    y0 = np.tile(np.array([[1.0], [0.0]]), (1, B))
    y0[0, 3] = np.nan
    y0[1, 77] = np.inf
    t0 = np.full(B, 0.5)
    for method in ("DOPRI5", "DOP853"):
        g = gpu_batch("sho", y0, None, t0, 3.0, method=method, rtol=1e-6, atol=1e-9)
        r = oracle_batch("sho", y0, None, t0, 3.0, method=method, rtol=1e-6, atol=1e-9)
        assert g["status"][3] == 3 and g["status"][77] == 3
        assert_bitexact(g, r, method + ": ")
    g = gpu_batch("sho", y0, None, t0, 3.0, method="RK23", rtol=1e-6, atol=1e-9)
    assert g["status"][3] == 3 and g["status"][77] == 3      # the reference would spin forever here
    good = np.ones(B, bool); good[[3, 77]] = False
    assert (g["status"][good] == 0).all()


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "RK4", "BDF"])
@pytest.mark.parametrize("direction", ["fwd", "bwd"])
def test_t_eval_sampling_matches_oracle(method, direction):
    B = 130
    rng = np.random.default_rng(2)
    y0 = np.stack([np.full(B, 1 / 3), np.full(B, 2 / 9)]) * (1 + 1e-3 * rng.standard_normal((2, B)))
    if direction == "fwd":
        t0, t1 = 5.0, 9.0
        te = np.array([4.0, 5.0, 5.01, 5.5, 7.0, 8.0, 8.01, 9.0, 9.5])
    else:
        t0, t1 = 5.0, 1.0
        te = np.array([5.0, 4.99, 3.0, 1.5, 1.1, 1.0, 0.5])
    o = dict(method=method, rtol=1e-3, atol=1e-6, t_eval=te)
    g = gpu_batch("rational", y0, None, t0, t1, chunk=5, **o)
    for b in range(0, B, 13):
        s = O.solve_ivp("rational", t0, t1, y0[:, b], detpow=True, **o)
        m = g["n_filled"][b]
        assert m == len(s.t)
        assert np.array_equal(te[g["eval_idx"][:m, b]], s.t)
        assert np.array_equal(g["y_eval"][:m, :, b], s.y)


@pytest.mark.parametrize("method", ["RK23", "DOPRI5", "DOP853", "BDF"])
@pytest.mark.parametrize("first_step", [None, 0.1])
def test_step_log_and_dense_segments_match_oracle(method, first_step):
    B = 70
    rng = np.random.default_rng(4)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    for (t0, t1) in ((0.0, 3.0), (3.0, 0.0)):
        o = dict(method=method, rtol=1e-5, atol=1e-8, dense_output=True)
        if first_step is not None:
            o["first_step"] = first_step
        g = gpu_batch("sho", y0, None, t0, t1, max_log=512, chunk=9, **o)
        for b in range(0, B, 9):
            s = O.solve_ivp("sho", t0, t1, y0[:, b], detpow=True, **o)
            m = g["n_log"][b]
            assert m == len(s.t)
            assert np.array_equal(g["t_log"][:m, b], s.t)
            assert np.array_equal(g["y_log"][:m, :, b], s.y)
            ns = g["n_seg"][b]
            assert ns == len(s.seg_h)
            assert np.array_equal(g["seg_xold"][:ns, b], s.seg_xold)
            assert np.array_equal(g["seg_h"][:ns, b], s.seg_h)
            assert np.array_equal(g["seg_cont"][:ns, :, b], s.seg_cont)


@pytest.mark.parametrize("case", EVENT_CASES, ids=[c[0] for c in EVENT_CASES])
def test_event_detection_matches_oracle(case):
    """Bit-exact except where the event functions themselves call pow (rational problem: ocml vs glibc pow)."""
    exact = not case[1].startswith("rational")
    check_events_against_oracle(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, chunk=7, **kw), case, exact=exact)


# ---- against the libm-pow oracle (the faithful restatement) and against independent truth --------------

@pytest.mark.parametrize("method,rtol,atol", [("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)])
@pytest.mark.parametrize("fast", [False, True])
def test_short_horizon_agreement_with_libm_pow_oracle(method, rtol, atol, fast):
    """short-horizon (t1 = 2.0, before the close lunar approach) GPU-vs-CPU agreement.
    Tolerance: 1e-9 absolute on O(1) states; step counts identical for >= 95 % of the trajectories."""
    y0, p, t0, _ = W.cr3bp_batch(1024)
    g = gpu_batch("cr3bp", y0, p, t0, 2.0, method=method, rtol=rtol, atol=atol, fast=fast)
    r = oracle_batch("cr3bp", y0, p, t0, 2.0, detpow=False, threads=8, method=method, rtol=rtol, atol=atol)
    assert (g["status"] == 0).all()
    assert np.abs(g["y_end"] - r["y_end"]).max() < 1e-9
    assert np.mean(g["naccpt"] == r["naccpt"]) > 0.95


@pytest.mark.parametrize("fast", [False, True])
def test_c2_end_state_error_within_10x_of_cpu_reference(fast):
    """BASELINE target: end-state error (vs SciPy DOP853@1e-13 truth) within 10x of the CPU reference's."""
    truth = np.asarray(json.load(open(os.path.join(GOLD, "scipy_truth.json")))["truth"]["cr3bp"]["y_end"])
    n = len(truth)
    y0, p, t0, t1 = W.cr3bp_batch(256)
    g = gpu_batch("cr3bp", y0[:, :n], p[:, :n], t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, fast=fast)
    r = oracle_batch("cr3bp", y0[:, :n], p[:, :n], t0, t1, detpow=False, method="DOPRI5", rtol=1e-6, atol=1e-9)
    eg = np.abs(g["y_end"].T - truth).max(axis=1)
    er = np.abs(r["y_end"].T - truth).max(axis=1)
    assert np.median(eg) <= 10.0 * np.median(er)
    assert eg.max() <= 10.0 * er.max()


# ---- BASELINE full sizes: size-independent properties -----------------------------------------------------

def _jacobi(s, mu):
    x, y, z, vx, vy, vz = s
    r1 = np.sqrt((x + mu) ** 2 + y * y + z * z)
    r2 = np.sqrt((x - 1.0 + mu) ** 2 + y * y + z * z)
    return 2.0 * (0.5 * (x * x + y * y) + (1.0 - mu) / r1 + mu / r2) - (vx * vx + vy * vy + vz * vz)


def test_c2_full_size_100k_cr3bp_dopri5():
    B = 100_000
    y0, p, t0, t1 = W.cr3bp_batch(B)
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    g = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, **o)
    assert (g["status"] == 0).all() and (g["t_end"] == t1).all()
    assert (g["nfev"] == 2 + 6 * g["nstep"]).all()                       # dopri5.rs:231,235,325
    assert (g["nstep"] >= g["naccpt"] + g["nrejct"]).all()
    # integral of motion (examples/cr3bp.rs:14-20): drift at the tolerance scale for the bulk of the sweep (a few
    # perturbed orbits graze the Moon, where rtol=1e-6 loses the constant -- the CPU oracle loses it identically)
    drift = np.abs(_jacobi(g["y_end"], p[0]) - _jacobi(y0, p[0]))
    assert np.median(drift) < 1e-4 and np.mean(drift > 1e-2) < 0.01
    # determinism + independence from batch composition: any sub-batch reproduces the same bits
    idx = np.random.default_rng(0).choice(B, 300, replace=False)
    sub = gpu_batch("cr3bp", y0[:, idx], p[:, idx], t0, t1, **o)
    for k in ("y_end", "t_end", "h_next", "nfev", "nstep", "naccpt", "nrejct"):
        assert np.array_equal(np.asarray(g[k])[..., idx], sub[k]), k
    # ALL 100 000 trajectories against the CPU oracle, bit for bit (end state, end time, next step, every counter)
    r = oracle_batch("cr3bp", y0, p, t0, t1, threads=16, **o)
    assert_bitexact(g, r, "C2 100k: ")


def test_c3_full_size_1m_vdp_dop853():
    B = 1_000_000
    y0, p, t0, t1 = W.vdp_batch(B)
    # odd symmetry of Van der Pol: f(-y) = -f(y) and the error norm is even, so the second half of the batch
    # (negated initial states, same end times) must come out as the exact negative of the first half
    h = B // 2
    y0[:, h:] = -y0[:, :h]
    t1[h:] = t1[:h]
    o = dict(method="DOP853", rtol=1e-8, atol=1e-10)
    g = gpu_batch("vdp", y0, p, t0, t1, device_arrays=True, **o)
    assert (g["status"] == 0).all() and (g["t_end"] == t1).all()
    assert np.array_equal(g["y_end"][:, h:], -g["y_end"][:, :h])
    assert np.array_equal(g["naccpt"][h:], g["naccpt"][:h]) and np.array_equal(g["nrejct"][h:], g["nrejct"][:h])
    assert (g["nfev"] == 2 + 11 * g["nstep"] + 4 * g["naccpt"]).all()     # dop853.rs:390,444,560
    # ALL 1 000 000 trajectories against the CPU oracle, bit for bit
    r = oracle_batch("vdp", y0, p, t0, t1, threads=16, **o)
    assert_bitexact(g, r, "C3 1M: ")


def test_maximum_size_batch_16m_trajectories_linearity():
    """16.7M one-component trajectories in one call (index arithmetic past 2^24, 262144 waves).  y' = -k y is linear in
    y0 only up to the controller: scaling y0 AND atol by a power of two scales every intermediate exactly, so the
    second half of the batch must be exactly 4x the first half; a sample is checked against the oracle."""
    import torch
    import ivp_amd
    B = 1 << 24
    h = B // 2
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    y0 = np.empty((1, B))
    y0[0, :h] = rng.uniform(0.5, 2.0, h)
    y0[0, h:] = y0[0, :h]
    k = np.empty((1, B))
    k[0, :h] = rng.uniform(0.1, 3.0, h)
    k[0, h:] = k[0, :h]
    t1 = np.empty(B)
    t1[:h] = rng.uniform(0.5, 4.0, h)
    t1[h:] = t1[:h]
    f = ivp_amd.ExponentialDecay()
    o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=0.0)      # atol = 0: the error norm is scale-free
    y0d = torch.as_tensor(y0, device=dev)
    y0d[0, h:] *= 4.0
    r = ivp_amd.solve_ivp_batch(f, 0.0, torch.as_tensor(t1, device=dev), y0d, torch.as_tensor(k, device=dev), o)
    ye = r.y_end.cpu().numpy()
    assert (r.status == 0).all().item()
    # the last step is h = xend - x and the reference stores x + h (dopri5.rs:280-283,409): equal to xend up to one rounding
    te = r.t_end.cpu().numpy()
    assert np.abs(te - t1).max() <= 4.5e-16 * 4.0 and np.mean(te != t1) < 0.05
    assert np.array_equal(te[h:], te[:h])
    assert np.array_equal(ye[0, h:], 4.0 * ye[0, :h])
    assert torch.equal(r.naccpt[h:], r.naccpt[:h]) and torch.equal(r.nfev[h:], r.nfev[:h])
    np.testing.assert_allclose(ye[0, :h], y0[0, :h] * np.exp(-k[0, :h] * t1[:h]), rtol=2e-5)
    idx = np.concatenate([rng.choice(h, 128, replace=False), [0, h - 1, h, B - 1]])
    yy = y0[:, idx].copy()
    yy[0, idx >= h] *= 4.0
    ref = oracle_batch("decay", yy, k[:, idx], 0.0, t1[idx], method="DOPRI5", rtol=1e-6, atol=0.0)
    assert np.array_equal(ye[:, idx], ref["y_end"]) and np.array_equal(r.naccpt.cpu().numpy()[idx].astype(np.uint64), ref["naccpt"])
    assert np.array_equal(te[idx], ref["t_end"])


def test_c5_full_size_10k_stiff_vdp_bdf():
    """BASELINE config C5: 10k stiff Van der Pol (mu ~ 1000), BDF order 1-5, per-trajectory LU in registers."""
    B = 10_000
    y0, p, t0, t1 = W.vdp_stiff_batch(B)
    o = dict(method="BDF", rtol=1e-4, atol=1e-6)
    g = gpu_batch("vdp", y0, p, t0, t1, device_arrays=True, **o)
    # a handful of trajectories stop one rounding error short of t1 and trip the stagnation guard
    # `x + 0.1|h| == x` (bdf.rs:325-328) with StepSizeTooSmall -- the reference does the same (the oracle comparison
    # below includes trajectory-for-trajectory status equality)
    assert np.isin(g["status"], (0, 3)).all() and np.mean(g["status"] != 0) < 0.01
    assert np.abs(g["t_end"] - t1).max() < 1e-9
    assert (g["njev"] > 0).all() and (g["nlu"] > 0).all() and (g["nstep"] == g["naccpt"] + g["nrejct"]).all()
    truth = json.load(open(os.path.join(GOLD, "scipy_stiff_truth.json")))["truth"]["vdp_mu1000_t3000"]
    assert np.abs(g["y_end"][:, 0] - truth).max() < 1e-2         # SciPy Radau @1e-10
    # ALL 10 000 trajectories against the CPU oracle, bit for bit (incl. status, njev, nlu)
    r = oracle_batch("vdp", y0, p, t0, t1, threads=16, **o)
    assert_bitexact(g, r, "C5 10k: ")


# ---- user-defined right-hand side (hiprtc): the device-side `impl IVP` -----------------------------------

CR3BP_SRC = r"""
__device__ void ode(double t, const double* s, double* d, const double* p)
{
    const double mu = p[0];
    const double x = s[0], y = s[1], z = s[2], vx = s[3], vy = s[4], vz = s[5];
    const double a = x + mu, b = x - 1.0 + mu;
    const double r1 = sqrt(a * a + y * y + z * z), r2 = sqrt(b * b + y * y + z * z);
    const double r13 = r1 * r1 * r1, r23 = r2 * r2 * r2;
    d[0] = vx; d[1] = vy; d[2] = vz;
    d[3] = x + 2.0 * vy - (1.0 - mu) * (x + mu) / r13 - mu * (x - 1.0 + mu) / r23;
    d[4] = y - 2.0 * vx - (1.0 - mu) * y / r13 - mu * y / r23;
    d[5] = -(1.0 - mu) * z / r13 - mu * z / r23;
}
"""


def test_user_rhs_compiled_with_hiprtc_matches_builtin_and_oracle():
    import ivp_amd
    y0, p, t0, t1 = W.cr3bp_batch(500)
    f = ivp_amd.DeviceIVP(CR3BP_SRC, n=6, params=(W.ARENSTORF_MU,))
    for method, rt, at in (("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("RK23", 1e-4, 1e-7)):
        o = ivp_amd.Options(method=method, rtol=rt, atol=at)
        r = ivp_amd.solve_ivp_batch(f, t0, t1, y0, p, o)
        ref = oracle_batch("cr3bp", y0, p, t0, t1, threads=8, method=method, rtol=rt, atol=at)
        got = {k: getattr(r, k) for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct")}
        assert_bitexact(got, ref, f"jit {method}: ")
        if method != "RK23":   # the hiprtc module's lane-cooperative kernel (generic shuffle-gather form), whole run
            r = ivp_amd.solve_ivp_batch(f, t0, t1, y0, p, ivp_amd.Options(method=method, rtol=rt, atol=at, variant=3, profile=1))
            got = {k: getattr(r, k) for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct")}
            assert_bitexact(got, ref, f"jit coop {method}: ")
            assert r.stats["coop_launches"] == r.stats["launches"] > 0


def test_user_rhs_python_callable_cross_check():
    """A system that is NOT among the built-ins: damped driven pendulum, checked against the oracle driven by the
    same formula as a Python callable."""
    import ivp_amd
    src = r"""
    __device__ void ode(double t, const double* y, double* d, const double* p)
    {
        d[0] = y[1];
        d[1] = -p[0] * y[1] - sin(y[0]) + p[1] * cos(p[2] * t);
    }
    """
    par = (0.2, 0.7, 1.3)
    f = ivp_amd.DeviceIVP(src, n=2, params=par)
    s = ivp_amd.solve_ivp(f, 0.0, 10.0, [0.3, 0.0], ivp_amd.Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
    o = O.solve_ivp(lambda t, y, p: [y[1], -par[0] * y[1] - np.sin(y[0]) + par[1] * np.cos(par[2] * t)],
                    0.0, 10.0, [0.3, 0.0], method="DOPRI5", rtol=1e-8, atol=1e-10)
    assert s.status == 0 and len(s.t) == len(o.t)
    # sin/cos come from different libms (ocml vs glibc): tolerance 1e-10 on an O(1) state
    np.testing.assert_allclose(s.y[-1], o.y[-1], rtol=0, atol=1e-10)


def test_jit_syntax_error_is_reported_not_fatal():
    import ivp_amd
    with pytest.raises(ivp_amd.ConfigError) as e:
        ivp_amd.DeviceIVP("__device__ void ode(double t, const double* y, double* d, const double* p) { d[0] = ; }", n=1)
    assert e.value.code == -104


def test_pipelined_batches_give_the_same_bits_as_sequential_solves():
    """Throughput mode (ivp_amd/pipeline.py): batches in flight on separate streams do not interact."""
    import torch
    import ivp_amd
    from ivp_amd.pipeline import BatchPipeline
    dev = torch.device("cuda:0")
    opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
    batches, refs = [], []
    for k in range(6):
        y0, p, t0, t1 = W.cr3bp_batch(3000 + 500 * k, seed=100 + k)
        b = dict(t0=t0, t1=t1, y0=torch.as_tensor(y0, device=dev), params=torch.as_tensor(p, device=dev))
        batches.append(b)
        refs.append(ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, b["y0"], b["params"], opts))
    pipe = BatchPipeline(3)
    for res in (pipe.map(ivp_amd.CR3BP(), batches, opts), pipe.map_threads(ivp_amd.CR3BP(), batches, opts)):
        for r, ref in zip(res, refs):
            for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
                assert torch.equal(getattr(r, k), getattr(ref, k)), k


def test_submit_poll_wait_entry_points():
    """ivp_batch_submit_device / ivp_batch_poll / ivp_batch_wait: a solve as a resumable operation."""
    import torch
    import ivp_amd
    dev = torch.device("cuda:0")
    y0, p, t0, t1 = W.cr3bp_batch(5000, seed=5)
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, profile=1)
    ref = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, opts)
    ctx = ivp_amd.Context(0)
    pend = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, opts, ctx, wait=False)
    with pytest.raises(ivp_amd.ConfigError):          # one solve in flight per context
        ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, opts, ctx, wait=False)
    polls = 0
    while not pend.done():
        polls += 1
    r = pend.result()
    assert torch.equal(r.y_end, ref.y_end) and torch.equal(r.naccpt, ref.naccpt) and r.stats["launches"] == ref.stats["launches"]
    # wait() without polling, then the context is free again; t_eval (host array consumed at submit time) works too
    pend = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10, t_eval=[1.0, 5.0, 17.0]), ctx, wait=False)
    r2 = pend.result()
    ref2 = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10, t_eval=[1.0, 5.0, 17.0]))
    assert torch.equal(r2.y_eval, ref2.y_eval) and torch.equal(r2.y_end, ref2.y_end)
    with pytest.raises(ValueError):
        ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0, p, opts, ctx, wait=False)      # host arrays cannot be asynchronous


def test_jit_disk_cache_round_trip(tmp_path, monkeypatch):
    """IVP_JIT_CACHE_DIR: compiled code objects are stored and reused; a cached module gives the same bits."""
    import time
    import ivp_amd
    monkeypatch.setenv("IVP_JIT_CACHE_DIR", str(tmp_path))
    src = CR3BP_SRC + "\n// cache-test variant\n"
    y0, p, t0, t1 = W.cr3bp_batch(300)
    opts = ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10)
    t = time.perf_counter()
    f1 = ivp_amd.DeviceIVP(src, n=6, params=(W.ARENSTORF_MU,))
    r1 = ivp_amd.solve_ivp_batch(f1, t0, t1, y0, p, opts)
    cold = time.perf_counter() - t
    files = sorted(os.listdir(tmp_path))
    assert len(files) >= 2 and all(f.endswith(".hsaco") for f in files), files     # the syntax-check module + DOP853
    t = time.perf_counter()
    f2 = ivp_amd.DeviceIVP(src, n=6, params=(W.ARENSTORF_MU,))
    r2 = ivp_amd.solve_ivp_batch(f2, t0, t1, y0, p, opts)
    warm = time.perf_counter() - t
    assert np.array_equal(r1.y_end, r2.y_end) and np.array_equal(r1.nfev, r2.nfev)
    assert sorted(os.listdir(tmp_path)) == files and warm < 0.5 * cold, (cold, warm)
    # a corrupted entry is ignored (recompiled), not fatal
    for f in files:
        with open(os.path.join(tmp_path, f), "wb") as fh:
            fh.write(b"not a code object")
    f3 = ivp_amd.DeviceIVP(src, n=6, params=(W.ARENSTORF_MU,))
    r3 = ivp_amd.solve_ivp_batch(f3, t0, t1, y0, p, opts)
    assert np.array_equal(r1.y_end, r3.y_end)


@pytest.mark.parametrize("block", range(6))
def test_random_configurations_bitexact_on_the_gpu(block):
    """tests/test_differential_random_cpu.py's generator against the product: random method / problem / options /
    output-mode combinations through the C ABI, default kernel choice and the cooperative kernels."""
    from tests.test_differential_random_cpu import compare, random_case

    for seed in range(3000 + 20 * block, 3000 + 20 * (block + 1)):
        compare(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, **kw), seed)
        if random_case(seed)[5]["method"] in ("DOPRI5", "DOP853"):
            compare(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, variant=3, **kw), seed)


_SHARD_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
import ivp_amd
from oracle import oracle as O

rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
B = 4001                                       # odd: unequal shards
y0, p, t0, t1 = W.cr3bp_batch(B, seed=77)
perm = W.shard_permutation(B)
opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
got = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, t1, y0, p, opt, permutation=perm, device="cuda:0")   # HIP path per shard
ref = O.solve_batch("cr3bp", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True, threads=4)
for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
    assert np.array_equal(np.asarray(got[k]).astype(ref[k].dtype), ref[k]), k
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_c4_sharded_solve_two_ranks_sharing_the_gpu(tmp_path):
    """BASELINE config C4 end to end with the real kernels: two ranks (here both on cuda:0; the collective runs over
    gloo because one GPU cannot host two RCCL ranks), contiguous shards after the fixed permutation, one packed gather,
    original order restored -- the gathered batch equals the oracle bit for bit on every rank."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER.format(root=root, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]


def test_pipeline_recovers_after_a_failing_batch():
    """A batch that fails validation (RK4 with a wrong-signed step) raises, and the pipeline's contexts are usable again."""
    import torch
    import ivp_amd
    from ivp_amd.pipeline import BatchPipeline
    dev = torch.device("cuda:0")
    y0, p, t0, t1 = W.cr3bp_batch(2000, seed=9)
    b = dict(t0=t0, t1=t1, y0=torch.as_tensor(y0, device=dev), params=torch.as_tensor(p, device=dev))
    pipe = BatchPipeline(2)
    with pytest.raises(ivp_amd.ConfigError):
        pipe.map(ivp_amd.CR3BP(), [b, b, b], ivp_amd.Options(method="RK4", first_step=-0.01))
    good = pipe.map(ivp_amd.CR3BP(), [b, b, b], ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    ref = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, b["y0"], b["params"], ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    assert all(torch.equal(g.y_end, ref.y_end) for g in good)


def test_fast_mode_results_do_not_depend_on_the_batch():
    """Fast-mode FMA fusion differs between the kernel variants, so the variant is chosen from the problem alone (never
    from the batch size or from what else is in the batch): the same trajectories give the same bits in a batch of 300,
    inside a batch of 140 000 (> the two-waves-per-SIMD threshold the strict-mode policy switches on) and next to
    trajectories that finish at once."""
    import torch
    import ivp_amd
    dev = torch.device("cuda:0")
    y0, p, t0, t1 = W.cr3bp_batch(140_000)
    opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=ivp_amd.FpMode.FAST)
    big = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, 3.0, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev), opt)
    sel = np.arange(0, 140_000, 467)
    small = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, 3.0, torch.as_tensor(np.ascontiguousarray(y0[:, sel]), device=dev),
                                    torch.as_tensor(np.ascontiguousarray(p[:, sel]), device=dev), opt)
    idx = torch.as_tensor(sel, device=dev)
    assert torch.equal(small.y_end, big.y_end[:, idx]) and torch.equal(small.naccpt, big.naccpt[idx]) and torch.equal(small.h_next, big.h_next[idx])
    t1v = np.full(sel.size, 3.0); t1v[::2] = 1e-3                    # half of the batch retires after a step or two
    mixed = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, torch.as_tensor(t1v, device=dev), torch.as_tensor(np.ascontiguousarray(y0[:, sel]), device=dev),
                                    torch.as_tensor(np.ascontiguousarray(p[:, sel]), device=dev), opt)
    assert torch.equal(mixed.y_end[:, 1::2], small.y_end[:, 1::2])
    y2, p2, _, t2 = W.vdp_batch(150_000)                             # n = 2: the lean variant, whatever the batch size
    o2 = ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10, fp_mode=ivp_amd.FpMode.FAST)
    b2 = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), 0.0, 5.0, torch.as_tensor(y2, device=dev), torch.as_tensor(p2, device=dev), o2)
    s2 = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), 0.0, 5.0, torch.as_tensor(np.ascontiguousarray(y2[:, :100]), device=dev),
                                 torch.as_tensor(np.ascontiguousarray(p2[:, :100]), device=dev), o2)
    assert torch.equal(s2.y_end, b2.y_end[:, :100])


@pytest.mark.parametrize("fma", [False, True], ids=["strict", "fma"])
@pytest.mark.parametrize("method", ["DOPRI5", "DOP853"])
def test_windowed_bulk_launches_change_nothing(method, fma):
    """70 000 CR3BP trajectories are 1094 waves on 1024 SIMDs: the automatic launch loop then steps only the first 65 536
    entries of the active list per launch and passes the rest on (IvpKArgs.window).  With an explicit chunk length the loop
    is the plain one (no window, no launch pairs): both must give the same bits, t_eval samples included, and the oracle's."""
    B = 70_000
    y0, p, t0, t1 = W.cr3bp_batch(B)
    t1 = 6.0
    o = dict(method=method, rtol=1e-6, atol=1e-9, t_eval=np.linspace(0.0, t1, 9))
    a = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=fma, **o)
    b = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, fast=fma, chunk=64, **o)
    assert_bitexact(a, b, "window vs plain: ")
    assert np.array_equal(a["n_filled"], b["n_filled"]) and (a["n_filled"] == 9).all()
    assert np.array_equal(a["y_eval"].view(np.uint64), b["y_eval"].view(np.uint64))
    idx = np.random.default_rng(5).choice(B, 2000, replace=False)
    r = oracle_batch("cr3bp", y0[:, idx], p[:, idx], t0, t1, threads=16, fma=fma, **o)
    for k in ("y_end", "t_end", "h_next", "nfev", "nstep", "naccpt", "nrejct"):
        assert np.array_equal(np.asarray(a[k])[..., idx], r[k]), k
    assert np.array_equal(a["y_eval"][:, :, idx].view(np.uint64), r["y_eval"].view(np.uint64))


def test_windowed_bulk_launches_with_the_csr_step_log():
    """The two-pass CSR step log (count, then fill at exact offsets) under windowed launches: a trajectory that waits
    behind the window keeps its log cursor; records equal those of the plain launch loop, and the oracle's for a sample."""
    import ivp_amd
    B = 70_000
    y0, p, t0, t1 = W.cr3bp_batch(B)
    t1 = 2.5
    f = ivp_amd.CR3BP()
    a = ivp_amd.solve_ivp_batch_logged(f, t0, t1, y0, p, ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    b = ivp_amd.solve_ivp_batch_logged(f, t0, t1, y0, p, ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, chunk_attempts=64))
    import torch
    assert torch.equal(a.log_offsets, b.log_offsets) and torch.equal(a.t_log.view(torch.int64), b.t_log.view(torch.int64))
    assert torch.equal(a.y_log.view(torch.int64), b.y_log.view(torch.int64)) and torch.equal(a.y_end.view(torch.int64), b.y_end.view(torch.int64))
    for j in (0, 17, 65_535, 65_536, 69_999):
        s = O.solve_ivp("cr3bp", t0, t1, y0[:, j], params=p[:, j], detpow=True, method="DOPRI5", rtol=1e-6, atol=1e-9)
        t, y = a.log_of(j)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y), j
