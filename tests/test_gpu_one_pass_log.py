"""One-pass accepted-step log on the GPU (ivp_batch_solve_logged* in include/ivp_hip.h): the reference's default output
contract -- Solution.t / Solution.y filled WHILE integrating (/root/reference/src/solve/solout.rs:387-428, returned at
src/solve/solve_ivp.rs:288-312) -- from ONE integration: page chains in a device pool + a gather kernel.

Bars: the records are bit-identical to the counted two-pass CSR log (count solve + scan + fill solve), to the dense
[max_log] log and to the oracle's Solution.t / Solution.y; a pool that runs dry costs a second integration, never a record;
the C ABI's host form and its library-allocated ("Vec returned by the callee") form deliver the same bytes."""
import ctypes as C

import numpy as np
import pytest

import ivp_amd
from ivp_amd import _lib
from ivp_amd import workloads as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _same_log(a, b):
    import torch
    assert torch.equal(a.log_offsets, b.log_offsets)
    assert torch.equal(a.n_log, b.n_log)
    assert torch.equal(a.t_log.view(torch.int64), b.t_log.view(torch.int64))
    assert torch.equal(a.y_log.view(torch.int64), b.y_log.view(torch.int64))
    for k in ("y_end", "t_end", "h_next"):
        assert torch.equal(getattr(a, k).view(torch.int64), getattr(b, k).view(torch.int64)), k
    for k in ("status", "nfev", "nstep", "naccpt", "nrejct"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


PROBLEMS = [
    ("cr3bp/DOPRI5", lambda: (ivp_amd.CR3BP(),) + W.cr3bp_batch(5000), dict(method="DOPRI5", rtol=1e-6, atol=1e-9)),
    ("cr3bp/DOPRI5/coop", lambda: (ivp_amd.CR3BP(),) + W.cr3bp_batch(700), dict(method="DOPRI5", rtol=1e-6, atol=1e-9, variant=3)),
    ("cr3bp/DOP853/fma", lambda: (ivp_amd.CR3BP(),) + W.cr3bp_batch(900), dict(method="DOP853", rtol=1e-9, atol=1e-11, fp_mode=ivp_amd.FpMode.FMA)),
    ("vdp/DOP853", lambda: (ivp_amd.VanDerPol(),) + W.vdp_batch(20000), dict(method="DOP853", rtol=1e-8, atol=1e-10)),
    ("vdp/RK23", lambda: (ivp_amd.VanDerPol(),) + W.vdp_batch(3000)[:3] + (8.0,), dict(method="RK23", rtol=1e-4, atol=1e-7)),
    ("vdp/BDF", lambda: (ivp_amd.VanDerPol(),) + W.vdp_stiff_batch(300), dict(method="BDF", rtol=1e-4, atol=1e-6)),
    ("cr3bp/first_step", lambda: (ivp_amd.CR3BP(),) + W.cr3bp_batch(400), dict(method="DOPRI5", rtol=1e-6, atol=1e-9, first_step=1e-3)),
]


@pytest.mark.parametrize("name,make,opts", PROBLEMS, ids=[c[0] for c in PROBLEMS])
def test_one_pass_log_equals_the_two_pass_log(name, make, opts):
    import torch
    f, y0, p, t0, t1 = make()
    dev = torch.device("cuda:0")
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    t1d = torch.as_tensor(t1, device=dev) if np.ndim(t1) else t1
    o = ivp_amd.Options(**opts)
    ctx = ivp_amd.Context(0)                        # a fresh context: no learnt pool size
    one = ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, o, ctx)
    two = ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, o, ctx, two_pass=True)
    assert one.log_info["passes"] == 1 and 0 < one.log_info["pool_used_bytes"] <= one.log_info["pool_bytes"], one.log_info
    _same_log(one, two)
    # the pages hold a slot for every attempt of every trajectory a wave steps: more than the records, not absurdly more
    payload = int(one.log_offsets[-1]) * (y0.shape[0] + 1) * 8
    assert payload <= one.log_info["pool_used_bytes"], one.log_info
    # again on the same context (pool sized from the learnt total) and into the previous result's buffers (one library call)
    again = ivp_amd.solve_ivp_batch_logged(f, t0, t1d, y0d, pd, o, ctx, out=one)
    assert again.log_info["passes"] == 1 and again.t_log.data_ptr() == one.t_log.data_ptr()
    _same_log(again, two)
    ctx.close()


def test_a_pool_that_runs_dry_costs_an_integration_not_a_record():
    import torch
    y0, p, t0, t1 = W.cr3bp_batch(3000)
    dev = torch.device("cuda:0")
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
    ctx = ivp_amd.Context(0)
    two = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, ctx, two_pass=True)
    dry = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, ctx, reserve=1)     # ~1 page per trajectory: far too small
    assert dry.log_info["passes"] == 2, dry.log_info
    _same_log(dry, two)
    # the context has learnt the total: the next call needs one integration
    nxt = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, ctx)
    assert nxt.log_info["passes"] == 1
    _same_log(nxt, two)
    # a reused result whose buffers are too small (a longer interval): the records wait in the pool for larger ones
    longer = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, 1.5 * t1, y0d, pd, o, ctx, out=nxt)
    ref = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, 1.5 * t1, y0d, pd, o, ctx, two_pass=True)
    assert int(longer.log_offsets[-1]) > int(two.log_offsets[-1])
    _same_log(longer, ref)
    ctx.close()


def test_one_pass_log_against_the_oracle_and_with_events():
    """Solution.t / Solution.y of single reference calls, incl. a terminal event (its point is appended, solout.rs:316-319)
    and the wave-per-trajectory kernels (n = 100)."""
    import torch
    y0, p, t0, t1 = W.cr3bp_batch(64)
    opt = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    r = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(**opt))
    for b in (0, 31, 63):
        s = O.solve_ivp("cr3bp", t0, t1, y0[:, b], params=p[:, b], detpow=True, **opt)
        t, y = r.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)
    f = ivp_amd.SHOZeroEvent(ivp_amd.EventConfig(ivp_amd.Direction.All, 3))
    y0e = np.array([[1.0, 0.3, 2.0], [0.0, 1.0, -1.0]])
    e = ivp_amd.solve_ivp_batch_logged(f, 0.0, 40.0, y0e, None, ivp_amd.Options(method="DOP853", rtol=1e-9, atol=1e-12))
    assert (e.status.cpu().numpy() == 1).all()
    for b in range(3):
        s = O.solve_ivp("sho_ev", 0.0, 40.0, y0e[:, b], detpow=True, method="DOP853", rtol=1e-9, atol=1e-12, event_direction=[0], event_terminal=[3])
        t, y = e.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)
    rng = np.random.default_rng(3)
    yl = 1.0 + 0.1 * rng.standard_normal((100, 5))
    g = ivp_amd.solve_ivp_batch_logged(ivp_amd.LinearDecay100(), 0.0, 5.0, yl, None, ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    for b in range(5):
        s = O.solve_ivp("linear_decay100", 0.0, 5.0, yl[:, b], detpow=True, method="DOPRI5", rtol=1e-6, atol=1e-9)
        t, y = g.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y), b


def test_c_abi_host_form_returns_owned_vectors():
    """ivp_batch_solve_logged through plain host pointers: offsets are the caller's, t / y come back library-allocated
    (owned = 1) like the Vecs a Rust callee returns, and are released with ivp_step_log_free."""
    L = _lib.load()
    ctx = ivp_amd.Context(0)
    B = 257
    y0, p, t0, t1 = W.cr3bp_batch(B)
    keep = []
    o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)._c(6, keep)
    prob = _lib.ProblemT()
    prob.rhs_id, prob.n, prob.n_params = 3, 6, 1
    y_end = np.zeros((6, B)); n_log = np.zeros(B, dtype=np.uint32); status = np.full(B, -7, dtype=np.int32); naccpt = np.zeros(B, dtype=np.uint64)
    r = _lib.BatchResultT()
    r.y_end, r.n_log, r.status, r.naccpt = y_end.ctypes.data, n_log.ctypes.data, status.ctypes.data, naccpt.ctypes.data
    offsets = np.zeros(B + 1, dtype=np.uint64)
    sl = _lib.StepLogT()
    sl.offsets = offsets.ctypes.data
    t0a, t1a = np.array([t0]), np.array([t1])
    rc = L.ivp_batch_solve_logged(ctx.handle, C.byref(prob), B, y0.ctypes.data, p.ctypes.data, t0a.ctypes.data, 1, t1a.ctypes.data, 1,
                                  C.byref(o), C.byref(r), C.byref(sl))
    assert rc == 0, ctx.last_error()
    assert sl.owned == 1 and sl.device == -1 and sl.passes == 1 and sl.total == int(n_log.sum()) == int(offsets[-1])
    assert (status == 0).all() and np.array_equal(np.diff(offsets.astype(np.int64)), n_log.astype(np.int64))
    t = np.ctypeslib.as_array(C.cast(sl.t, C.POINTER(C.c_double)), shape=(sl.total,)).copy()
    y = np.ctypeslib.as_array(C.cast(sl.y, C.POINTER(C.c_double)), shape=(sl.total, 6)).copy()
    L.ivp_step_log_free(C.byref(sl))
    assert sl.owned == 0 and not sl.t and not sl.y
    ref = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9), two_pass=True)
    assert np.array_equal(t, ref.t_log.cpu().numpy()) and np.array_equal(y, ref.y_log.cpu().numpy())
    assert np.array_equal(y_end, ref.y_end.cpu().numpy())
    # caller-provided host buffers that are too small: IVP_ERR_LOG_CAPACITY with the total on the record, nothing written
    small_t, small_y = np.full(10, -1.0), np.full((10, 6), -1.0)
    sl2 = _lib.StepLogT()
    sl2.offsets, sl2.t, sl2.y, sl2.capacity = offsets.ctypes.data, small_t.ctypes.data, small_y.ctypes.data, 10
    rc = L.ivp_batch_solve_logged(ctx.handle, C.byref(prob), B, y0.ctypes.data, p.ctypes.data, t0a.ctypes.data, 1, t1a.ctypes.data, 1,
                                  C.byref(o), C.byref(r), C.byref(sl2))
    assert rc == -105 and sl2.total == sl.total and (small_t == -1.0).all()
    ctx.close()


def test_c2_full_size_one_pass_equals_two_pass():
    """BASELINE C2, all 100 000 trajectories, 16.9 M records: the one-pass log and the counted two-pass log are the same bytes."""
    import torch
    y0, p, t0, t1 = W.cr3bp_batch(100_000)
    dev = torch.device("cuda:0")
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
    one = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, o)
    two = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, o, two_pass=True)
    assert one.log_info["passes"] == 1
    _same_log(one, two)
